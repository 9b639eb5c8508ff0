#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (kernel stats / PMC counter collection) into the small files kept under profiles/.

  python tools/summarize_profile.py stats <kernel_stats.csv> <out.md> "<title>" "<command>"
  python tools/summarize_profile.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <batch> <out.json> <out.md>
  python tools/summarize_profile.py valu <sq_summary.md> <workload> <batch> <out.json>     (SQ_INSTS_VALU per launch per kernel)
"""
import collections, csv, json, os, sys


def short(name):
    """kernel name without namespace and template arguments - except k_describe's: a blur split by level launches k_describe<true> (levels below
    the split, blur per keypoint) AND k_describe<false> (the rest) once each per step; they are different kernels and stay two rows"""
    n = name.split("(")[0].replace("orbx::", "").replace("void ", "")
    base = n.split("<")[0]
    if base == "k_describe" and "<" in n:
        return "k_describe<%s>" % ("PB" if n.split("<")[1].split(">")[0].strip() in ("true", "1") else "plain")
    return base


def add_step_sums(per):
    """'k_describe' = the sum of its two launches of a step (what bench.py's event profile, which brackets both, calls k_describe)"""
    parts = [k for k in per if k.startswith("k_describe<")]
    if len(parts) > 1:
        if isinstance(per[parts[0]], dict):
            per["k_describe"] = {c: sum(per[k].get(c, 0.0) for k in parts) for c in per[parts[0]]}
        else:
            per["k_describe"] = int(sum(per[k] for k in parts))
    elif len(parts) == 1:
        per["k_describe"] = per[parts[0]]


def stats(src, out, title, cmd):
    rows = list(csv.DictReader(open(src)))
    with open(out, "w") as f:
        f.write("# %s\n\nCommand (GPU box, 1x MI355X): `%s`\n\n" % (title, cmd))
        f.write("| kernel | calls | total ms | avg us | % | min us | max us |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.1f | %.2f | %.1f | %.1f |\n" % (
                short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["Percentage"]), int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))


def traffic(fetch_csv, write_csv, workload, batch, out_json, out_md):
    def collect(path, counter):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in agg.items()}
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request for wide
    # coalesced reads, so the read side is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
    fe, wr = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
    data = json.load(open(out_json)) if os.path.exists(out_json) else {}
    data.setdefault(workload, {})[str(batch)] = {}      # (a fresh entry: kernels of an earlier round's path do not linger)
    per = data[workload][str(batch)]
    with open(out_md, "w") as f:
        f.write("# HBM traffic per launch from rocprofv3 --pmc (workload %s, %s frames per launch)\n\n" % (workload, batch))
        f.write("Separate passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass). bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024;\n"
                "the factor 2 is the gfx950 FETCH_SIZE correction for wide coalesced reads and over-states kernels whose reads are narrow.\n\n")
        f.write("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | corrected HBM bytes / launch | per frame |\n|---|---|---|---|---|\n")
        for k in sorted(set(fe) | set(wr)):
            if not k.startswith("k_"):
                continue
            b = (2 * fe.get(k, 0) + wr.get(k, 0)) * 1024
            per[k] = int(b)
            f.write("| %s | %.0f | %.0f | %d | %d |\n" % (k, fe.get(k, 0), wr.get(k, 0), b, b / int(batch)))
        add_step_sums(per)
        tot = sum(v for k, v in per.items() if k.startswith("k_") and not k.startswith("k_describe<") and k != "k_packedSelfTest")
        f.write("\nSum over the kernels of a step: %d bytes = %.2f MB per frame.\n" % (tot, tot / int(batch) / 1e6))
    json.dump(data, open(out_json, "w"), indent=1, sort_keys=True)


def valu(sq_md, workload, batch, out_json):
    lines = [l for l in open(sq_md) if l.startswith("|")]
    names = [c.strip() for c in lines[0].strip().strip("|").split("|")]
    data = json.load(open(out_json)) if os.path.exists(out_json) else {}
    data.setdefault(workload, {})[str(batch)] = {}
    per = data[workload][str(batch)]
    for l in lines[2:]:
        cells = [c.strip() for c in l.strip().strip("|").split("|")]
        row = dict(zip(names, cells))
        per[row["kernel"]] = {k: float(row[k]) for k in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_LDS", "SQ_INSTS_SALU") if k in row and row[k] != "-"}
    add_step_sums(per)
    json.dump(data, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:6])
    elif sys.argv[1] == "valu":
        valu(*sys.argv[2:6])
    else:
        traffic(*sys.argv[2:9])
