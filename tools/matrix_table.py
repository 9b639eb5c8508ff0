#!/usr/bin/env python3
"""gpurun_out/matrix.jsonl (tools/bench_matrix.sh) -> markdown table.  usage: matrix_table.py [round tag] > profiles/r03_bench_matrix.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open(os.path.join(ROOT, "gpurun_out", "matrix.jsonl")).read().splitlines()
print("# bench.py matrix - %s, 1x MI355X, device-resident inputs and outputs\n" % (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].endswith(".jsonl") else "round-6 final"))
print("Collected by `tools/bench_matrix.sh` in one gpurun call (`--no-cpu-baseline --no-extras` on every row).  Default frames per call: 512 (640x480 workloads),\n"
      "128 (hd720, hd1080).  Timed steps run the default overlap where it applies (the blur of the coarse levels on its side stream, staggered tails at 512 x 640x480; section 4 of DESIGN.md); the per-kernel figures are bench.py's event-profiled, serial pass.\n")
print("| bench.py arguments | frames/call | frames/s | ms/step | keypoints/frame | extra | path_frac (HBM) | kernel us per step |\n|---|---|---|---|---|---|---|---|")
args = None
for l in lines:
    if l.startswith("#"):
        args = l[2:]
        continue
    if not l.strip():
        continue
    d = json.loads(l)
    c = d["config"]
    extra = ", ".join("%s %s" % (k.replace("mean_", "").replace("_", " "), v) for k, v in c.items() if k.startswith("mean_") and k != "mean_keypoints_per_frame")
    ks = " ".join("%s=%d" % (k.replace("k_", "").split("+")[0] + ("+" if "+" in k else ""), round(v * 1e3)) for k, v in sorted(d["roofline"]["kernel_ms_per_step"].items()))
    print("| `%s` | %d | %d | %.4f | %d | %s | %.4f | %s |" % (args, c["frames_per_gpu_per_step"], d["value"], d["ms_per_step"], round(c["mean_keypoints_per_frame"]), extra, d["roofline"]["path_frac"], ks))
print("\nDefault run (`python bench.py --steps 30`, no other flags), full JSON line:\n\n```\n%s\n```\n" % open(os.path.join(ROOT, "gpurun_out", "matrix_default.json")).read().strip())
hp = os.path.join(ROOT, "gpurun_out", "host_path.txt")
if os.path.exists(hp):
    print("PCIe-inclusive host path (`tools/host_path_rate.py`):\n\n```\n%s\n```" % open(hp).read().strip())
