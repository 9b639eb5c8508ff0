#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the tracked files under profiles/.
usage: publish_profiles.py <tag> <workload> <batch> ["state description"]     (tag names the round and state: r02_final)"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload, batch = sys.argv[1], sys.argv[2], sys.argv[3]
desc = sys.argv[4] if len(sys.argv) > 4 else "state " + tag
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag + ("_" + workload if workload != "mono640" else ""))
stem = os.path.join(ROOT, "profiles", "%s_%s_b%s" % (tag, workload, batch))      # tag carries the round: r02_final ...
tool = os.path.join(ROOT, "tools", "summarize_profile.py")
cmd = "ORBX_SPLIT=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras" + ("" if workload == "mono640" else " --workload " + workload)
subprocess.check_call(["cp", os.path.join(src, "kernel_stats.csv"), stem + "_kernel_stats.csv"])
subprocess.check_call([sys.executable, tool, "stats", os.path.join(src, "kernel_stats.csv"), stem + "_kernel_stats.md",
                       "rocprofv3 --kernel-trace --stats - %s (%s, %s frames per launch)" % (desc, workload, batch),
                       "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd])
subprocess.check_call([sys.executable, tool, "traffic", os.path.join(src, "fetch.csv"), os.path.join(src, "write.csv"), workload, batch,
                       os.path.join(ROOT, "profiles", "traffic.json"), stem + "_hbm_traffic.md"])
subprocess.check_call([sys.executable, tool, "valu", os.path.join(src, "sq_summary.md"), workload, batch,
                       os.path.join(ROOT, "profiles", "valu.json")])
# stamp both counter files with the hash of the kernel sources they were measured on (written on the GPU box by collect_profiles.sh;
# bench.py compares it with the tree it runs in and reports null on a mismatch) and add the static issue-class census
src_hash = open(os.path.join(src, "source_hash.txt")).read().strip()
census = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "valu_census.py"), "--json"]))
for name in ("traffic.json", "valu.json"):
    path = os.path.join(ROOT, "profiles", name)
    data = json.load(open(path))
    data[workload][batch]["_source_hash"] = src_hash
    if name == "valu.json":
        for k, c in census.items():
            if k in data[workload][batch]:
                data[workload][batch][k]["full_rate_share"] = c["share"]
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
v = {k: x for k, x in json.load(open(os.path.join(ROOT, "profiles", "valu.json")))[workload][batch].items() if not k.startswith("_")}
stats = {}
for r in csv.DictReader(open(stem + "_kernel_stats.csv")):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from summarize_profile import short
    stats[short(r["Name"])] = float(r["AverageNs"]) / 1e6
peak = 1024 * 2.4 / 4.0
out = ["# SQ counters per launch - %s (%s, %s frames per launch); kernel sources %s\n\n" % (desc, workload, batch, src_hash),
       "Two `rocprofv3 --pmc` passes of `%s` (no trace options), averaged per kernel by `tools/pmc_summary.py`:\n" % cmd,
       "pass A `SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES`, pass B `SQ_LDS_BANK_CONFLICT "
       "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY`.\n\n",
       open(os.path.join(src, "sq_summary.md")).read(),
       "\n## Vector-instruction issue roofline\n\n",
       "gfx950 has two issue classes (`profiles/r03_valu_issue_rate.md`): ~2 cycles per wave64 instruction for v_fma_f32 / v_add_f32 / v_mul_f32 / v_add_u32 /\n"
       "v_sub_u32 / v_and_b32 / v_or_b32 / v_mov_b32, ~4 cycles for everything else these kernels use.  `upper` prices every instruction at 4 cycles\n"
       "(peak %.0f G wave-instr/s = 1024 SIMDs x 2.4 GHz / 4); `lower` prices the kernel's full-rate share (static census of its ISA, `tools/valu_census.py`)\n"
       "at 2 cycles.  The true issue-slot occupancy lies between the two.  Duration = average of the same kernel in the kernel stats of this state.\n\n" % peak,
       "| kernel | SQ_INSTS_VALU per launch | per wave | avg duration ms | G wave-instr/s | full-rate share (static) | issue occupancy: lower - upper |\n|---|---|---|---|---|---|---|\n"]
if "k_describe" in v and "k_describe" not in stats:      # (the sum row of a blur split by level: its duration is the sum of its two launches)
    stats["k_describe"] = sum(t for k, t in stats.items() if k.startswith("k_describe<"))
for k in sorted(v):
    if k in stats:
        g = v[k]["SQ_INSTS_VALU"] / (stats[k] * 1e-3) / 1e9
        fr = v[k].get("full_rate_share", 0.0)
        out.append("| %s | %.4g | %.0f | %.4f | %.1f | %.2f | %.2f - %.2f |\n" % (k, v[k]["SQ_INSTS_VALU"], v[k]["SQ_INSTS_VALU"] / v[k]["SQ_WAVES"], stats[k], g, fr,
                                                                              g * (1 - fr / 2) / peak, g / peak))
open(stem + "_sq_counters.md", "w").write("".join(out))
print("".join(out[-8:]))
