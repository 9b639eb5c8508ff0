#!/usr/bin/env python3
"""One frame per call, device-resident, `calls` times at a given size: the workload of tools/trace_gaps.sh (launch boundaries of a single-frame call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import extractorb_amd as X
from extractorb_amd import synth
rows, cols, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 60
img = torch.from_numpy(synth.frames("noise", 0, 1, rows, cols)).cuda()
ex = X.ORBextractor(1000, max_width=cols, max_height=rows)
cap = ex.capacity
k = torch.zeros(cap * 28, dtype=torch.uint8, device="cuda"); d = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda"); n = torch.zeros(2, dtype=torch.int32, device="cuda")
for _ in range(calls):
    ex.extract_batch_device(img, 1, rows, cols, k.data_ptr(), d.data_ptr(), n.data_ptr(), n.data_ptr() + 4, cap)
torch.cuda.synchronize()
