#!/usr/bin/env python3
"""Does a hipGraph replay of the single-frame launch sequence beat the plain stream launches?  (One 640x480 frame, device-resident.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import extractorb_amd as X
from extractorb_amd import sharding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fr = torch.from_numpy(synth.frames("noise", 0, B, 480, 640)).cuda()
ex = X.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
cap = 1024
lay = sharding.slab_layout(B, cap)
slab = torch.zeros(lay["bytes"], dtype=torch.uint8, device="cuda"); b = slab.data_ptr()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ex.set_stream(s.cuda_stream)
    run = lambda: ex.extract_batch_device(fr, B, 480, 640, b + lay["keypoints"], b + lay["descriptors"], b + lay["n"], b + lay["mono"], cap, lapping=(0, 1000))
    for _ in range(20): run()
    s.synchronize()
    t = time.perf_counter()
    for _ in range(500): run()
    s.synchronize()
    plain = (time.perf_counter() - t) / 500
    ref = slab.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run()
    for _ in range(20): g.replay()
    s.synchronize()
    t = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / 500
    print("B=%d: stream launches %.1f us per call, graph replay %.1f us per call; results equal: %s" % (B, plain * 1e6, graph * 1e6, bool(torch.equal(ref, slab))))
