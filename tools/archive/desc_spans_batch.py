"""Diagnostic: stage times of k_describe's waves of frame 0 inside a LARGE batch (needs a -DORBX_DESC_STAMPS build): the waves run beside seven
others per SIMD, as in the benchmark.  usage (GPU box): python tools/desc_spans_batch.py ./stamps.so [frames=512]"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ORBX_LIBRARY"] = sys.argv[1]
os.environ["ORBX_SPLIT"] = "0"
import torch
import extractorb_amd as X
from extractorb_amd import synth
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
fr = torch.from_numpy(np.concatenate([synth.frames("noise", 0, 16, 480, 640)] * (B // 16))).cuda()
ex = X.ORBextractor(1000, max_batch=B)
cap = ex.capacity
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"); n = torch.zeros(2 * B, dtype=torch.int32, device="cuda")
for _ in range(3):
    ex.extract_batch_device(fr, B, 480, 640, k.data_ptr(), d.data_ptr(), n.data_ptr(), n.data_ptr() + 4 * B, cap)
torch.cuda.synchronize()
L = X.load_library()
buf = np.zeros(6 * 1024, np.uint64)
L.orbx_debug_desc_stamps(buf.ctypes.data_as(C.c_void_p))
t = buf.astype(np.int64).reshape(-1, 6)
t = t[(t[:, 0] > 0) & (t[:, 4] > 0)]
names = ["weights + barrier, per-level counters", "level geometry + selection entry", "patch loads -> LDS", "IC_Angle + rBRIEF"]
print("%d waves of frame 0 in a batch of %d" % (len(t), B))
tot = (t[:, 4] - t[:, 0]) / 100.0
for i, nm in enumerate(names):
    dd = (t[:, i + 1] - t[:, i]) / 100.0
    print("%-45s mean %.2f max %.2f us (%.0f %%)" % (nm, dd.mean(), dd.max(), 100 * dd.mean() / tot.mean()))
print("wave lifetime (before the final stores): mean %.2f us" % tot.mean())
