#!/usr/bin/env python3
"""The clock k_fast actually runs at in the benchmark's launch shape (512 x 640x480 noise frames per launch), from a -DORBX_FAST_CLOCK build:
sum of delta s_memtime / sum of delta s_memrealtime x 100 MHz over every cell-wave, after >= 2 s of back-to-back launches.
usage: fast_clock.py <liborbx_clock.so> [batch]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ORBX_LIBRARY"] = sys.argv[1]
os.environ["ORBX_SPLIT"] = "0"
import numpy as np, torch
import extractorb_amd as X
from extractorb_amd import sharding, synth
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
fr = torch.from_numpy(synth.frames("noise", 0, B, 480, 640)).cuda()
ex = X.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
ex.set_stream(torch.cuda.current_stream().cuda_stream)
cap = 1024
lay = sharding.slab_layout(B, cap)
slab = torch.zeros(lay["bytes"], dtype=torch.uint8, device="cuda"); b = slab.data_ptr()
run = lambda: ex.extract_batch_device(fr, B, 480, 640, b + lay["keypoints"], b + lay["descriptors"], b + lay["n"], b + lay["mono"], cap, lapping=(0, 1000))
L = X.load_library()
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(20): run()
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 8192)()
L.orbx_debug_fast_clock(buf, 1)
steps = 50
t1 = time.perf_counter()
for _ in range(steps): run()
torch.cuda.synchronize()
dt = time.perf_counter() - t1
L.orbx_debug_fast_clock(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
a = a[a[:, 1] > 0]
mt, rt, n = a[:, 0].sum(), a[:, 1].sum(), len(a)
print("k_fast in-kernel clock: %.3f GHz  (%d sampled cell-waves, mean wave lifetime %.1f us = %.0f shader cycles); step %.3f ms" % (mt / rt * 0.1, n, rt / n / 100.0, mt / n, dt / steps * 1e3))
