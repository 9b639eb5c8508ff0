#!/usr/bin/env python3
"""Soak of the emulated quad-tree kernel over the GPU fuzz corpus (tools/fuzz_parity.py's generator, the seeds of tests/test_gpu_fuzz.py):
every case through every compiled variant of k_octree_body.inc (256 / 512 / 1024 threads, queued and scratch-free "r" forms, which differ
in the short phase-2 pass; with and without the small-batch start from k_fast's leaf tables) under AddressSanitizer + UBSan, a subset under
ThreadSanitizer, several poison bytes; results vs the oracle.
usage: soak.py [cases_per_seed] [seed ...]        writes a markdown summary to stdout"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
import numpy as np                      # noqa: E402
import oracle_lib as O                  # noqa: E402
import emu_case as E                    # noqa: E402
import tsan_summary as S                # noqa: E402
from fuzz_parity import draw_case       # noqa: E402

BENIGN = {"write(4) k_octree_body.inc:%d <-> write(4) k_octree_body.inc:%d"}     # filled by benign_races()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seeds = [int(a) for a in sys.argv[2:]] or [101, 103]
    bins = E.build()
    tmp = tempfile.mkdtemp(prefix="octemu_")
    case, out = os.path.join(tmp, "c.bin"), os.path.join(tmp, "o.bin")
    rows = []
    races = {}
    t0 = time.time()
    for seed in seeds:
        rng = np.random.default_rng(seed)
        for t in range(n):
            c = draw_case(rng, t)
            o = O.Oracle(c["nf"], c["sf"], c["nlevels"], c["ini"], c["mn"])
            o.extract(c["img"], c["lap"])
            ncand = sum(len(o.candidates(l)) for l in range(c["nlevels"]))
            what = "seed %d case %d: %dx%d nf=%d levels=%d sf=%.1f %s, %d candidates" % (seed, t, c["cols"], c["rows"], c["nf"], c["nlevels"], c["sf"], c["variant"], ncand)
            res = []
            # (kind, threads, scratch-free variant, poison byte, start from k_fast's leaf tables = the small-batch form)
            variants = ([("asan", T, r, 0xA5, False) for T in (256, 512, 1024) for r in (0, 1)] + [("plain", 512, 0, 0x00, False), ("plain", 512, 1, 0xFF, False)]
                        + [("asan", 1024, 1, 0xA5, True), ("asan", 256, 1, 0x3C, True)]
                        + ([("tsan", 256, 1, 0xA5, False), ("tsan", 512, 0, 0xA5, False), ("tsan", 512, 1, 0xA5, True)] if ncand < 40000 else []))
            for kind, T, roomy, poison, leaf in variants:
                E.write_case(case, o, c["rows"], c["cols"], c["nf"], c["sf"], c["nlevels"], T, roomy, c["lap"], poison, leaf_tables=leaf)
                try:
                    rc, err = E.run_case(bins[kind], case, out, timeout=900, env={"TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0",
                                                                                    "ASAN_OPTIONS": "detect_leaks=0"})
                except Exception as e:      # a hang = a collective some lane never reached
                    res.append("%s T=%d r=%d: TIMEOUT %r" % (kind, T, roomy, e)); continue
                if rc == 3:
                    res.append("rejected geometry"); break
                if kind == "tsan":
                    for k, v in S.summarize(err).items():
                        races[k] = races.get(k, 0) + v
                    bad = E.check(o, c["nlevels"], c["lap"], E.read_result(out, c["nlevels"])) if rc in (0, 66) else ["rc %d" % rc]
                else:
                    bad = E.check(o, c["nlevels"], c["lap"], E.read_result(out, c["nlevels"])) if rc == 0 else ["rc %d: %s" % (rc, err.strip().splitlines()[:12])]
                if bad:
                    res.append("%s T=%d r=%d poison=%02x leaf=%d: %s" % (kind, T, roomy, poison, leaf, bad))
            rows.append((what, res))
            print("%-100s %s   [%.0f s]" % (what, "ok" if not res else res, time.time() - t0), file=sys.stderr, flush=True)
    print("| case | result |\n|---|---|")
    for what, res in rows:
        print("| %s | %s |" % (what, "bit-exact in every variant, no sanitizer report" if not res else "; ".join(res)))
    print("\nThreadSanitizer reports (racing source-line pairs, count):\n")
    for k, v in sorted(races.items(), key=lambda kv: -kv[1]):
        print("* %d x `%s`" % (v, k))
    if not races:
        print("* none")


if __name__ == "__main__":
    main()
