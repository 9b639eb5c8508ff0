#!/usr/bin/env python3
"""Diagnostic: k_fast time of several builds of the library (e.g. -DORBX_FAST_SKIP=n variants).
usage: fast_breakdown.py lib1.so [lib2.so ...]   — each in its own subprocess (one library per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] == "all"):
    libs = sys.argv[1:] if sys.argv[1] != "all" else sorted(os.path.join(ROOT, "extractorb_amd", f) for f in os.listdir(os.path.join(ROOT, "extractorb_amd")) if f.startswith("liborbx") and f.endswith(".so"))
    for lib in libs:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), lib])
    sys.exit(0)
sys.path.insert(0, ROOT)
import extractorb_amd.orbextractor as M
M._LIB = os.path.abspath(sys.argv[1])
import numpy as np, torch
import extractorb_amd as X
from extractorb_amd import synth
B = int(os.environ.get("B", "256"))
variant = os.environ.get("VARIANT", "noise")
fr = synth.frames(variant, 0, B, 480, 640)
ex = X.ORBextractor(1000, max_batch=B)
cap = ex.capacity
ex.set_stream(torch.cuda.current_stream().cuda_stream)
d_img = torch.from_numpy(fr).cuda()
d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(3):
    ex.extract_batch_device(d_img, B, 480, 640, d_k, d_d, d_n, d_m, cap)
torch.cuda.synchronize()
ex.profile(True)
for _ in range(10):
    ex.extract_batch_device(d_img, B, 480, 640, d_k, d_d, d_n, d_m, cap)
pr = ex.profile_read()
print("%-40s %s" % (os.path.basename(sys.argv[1]), "  ".join("%s=%.3f" % (k, v[0] / 10) for k, v in sorted(pr.items()) if k.startswith("k_") and v[1])), flush=True)
