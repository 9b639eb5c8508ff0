#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the tracked files under profiles/.
usage: publish_profiles.py <tag> <workload> <batch> ["state description"]"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload, batch = sys.argv[1], sys.argv[2], sys.argv[3]
desc = sys.argv[4] if len(sys.argv) > 4 else "state " + tag
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag + ("_" + workload if workload != "mono640" else ""))
stem = os.path.join(ROOT, "profiles", "r01_%s_%s_b%s" % (tag, workload, batch))
tool = os.path.join(ROOT, "tools", "summarize_profile.py")
cmd = "python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline" + ("" if workload == "mono640" else " --workload " + workload)
subprocess.check_call(["cp", os.path.join(src, "kernel_stats.csv"), stem + "_kernel_stats.csv"])
subprocess.check_call([sys.executable, tool, "stats", os.path.join(src, "kernel_stats.csv"), stem + "_kernel_stats.md",
                       "rocprofv3 --kernel-trace --stats - round 1, %s (%s, %s frames per launch)" % (desc, workload, batch),
                       "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd])
subprocess.check_call([sys.executable, tool, "traffic", os.path.join(src, "fetch.csv"), os.path.join(src, "write.csv"), workload, batch,
                       os.path.join(ROOT, "profiles", "traffic.json"), stem + "_hbm_traffic.md"])
subprocess.check_call([sys.executable, tool, "valu", os.path.join(src, "sq_summary.md"), workload, batch,
                       os.path.join(ROOT, "profiles", "valu.json")])
v = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))[workload][batch]
stats = {}
for r in csv.DictReader(open(stem + "_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("orbx::", "").replace("void ", "").split("<")[0]
    stats[n] = float(r["AverageNs"]) / 1e6
peak = 1024 * 2.4 / 4.0
out = ["# SQ counters per launch - round 1, %s (%s, %s frames per launch)\n\n" % (desc, workload, batch),
       "Two `rocprofv3 --pmc` passes of `%s` (no trace options), averaged per kernel by `tools/pmc_summary.py`:\n" % cmd,
       "pass A `SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES`, pass B `SQ_LDS_BANK_CONFLICT "
       "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY`.\n\n",
       open(os.path.join(src, "sq_summary.md")).read(),
       "\n## Vector-instruction issue roofline\n\n",
       "Peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (16 lanes per clock per SIMD) = %.0f G wave-instr/s; the microbenchmark\n"
       "`profiles/r01_valu_issue_rate.md` reaches 4.11-4.2 cycles (592 G/s) for the integer / packed instructions these kernels use.  Duration = average of the same kernel in the kernel stats of this state.\n\n" % peak,
       "| kernel | SQ_INSTS_VALU per launch | per wave | avg duration ms | G wave-instr/s | fraction of VALU issue peak |\n|---|---|---|---|---|---|\n"]
for k in sorted(v):
    if k in stats:
        g = v[k]["SQ_INSTS_VALU"] / (stats[k] * 1e-3) / 1e9
        out.append("| %s | %.4g | %.0f | %.4f | %.1f | %.2f |\n" % (k, v[k]["SQ_INSTS_VALU"], v[k]["SQ_INSTS_VALU"] / v[k]["SQ_WAVES"], stats[k], g, g / peak))
open(stem + "_sq_counters.md", "w").write("".join(out))
print("".join(out[-8:]))
