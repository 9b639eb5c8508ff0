#!/usr/bin/env python3
"""Times orbx_frame_finish_device (UndistortKeyPoints + AssignFeaturesToGrid) on 256 extracted frames."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import extractorb_amd as X
from extractorb_amd import synth

B = 256
frames = synth.frames("noise", 0, 64, 480, 640)
frames = np.concatenate([frames] * 4)
ex = X.ORBextractor(1000, max_batch=B)
cap = ex.capacity
ex.set_stream(torch.cuda.current_stream().cuda_stream)
d_img = torch.from_numpy(frames).cuda()
d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
ex.extract_batch_device(d_img, B, 480, 640, d_k, d_d, d_n, d_m, cap)
d_un = torch.zeros_like(d_k); d_off = torch.zeros((B, 3073), dtype=torch.int32, device="cuda")
d_idx = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_in = torch.zeros(B, dtype=torch.int32, device="cuda")
for name, cam in [("pinhole (k1 = 0)", X.camera(500, 500, 320, 240)), ("EuRoC distortion", X.camera(458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))]:
    b = X.compute_image_bounds(cam, 640, 480)
    for _ in range(3):
        ex.frame_finish_device(B, d_k, d_n, cap, cam, b, d_un, d_off, d_idx, d_in)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ex.frame_finish_device(B, d_k, d_n, cap, cam, b, d_un, d_off, d_idx, d_in)
    e1.record(); torch.cuda.synchronize()
    print("%-18s %.1f us per %d frames (%.3f us/frame)" % (name, e0.elapsed_time(e1) * 1e3 / 50, B, e0.elapsed_time(e1) * 1e3 / 50 / B))
