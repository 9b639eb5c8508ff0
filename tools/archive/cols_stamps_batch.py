"""Diagnostic: level times of region ORBX_CHAIN_STAMP_T of the region-major pyramid inside a LARGE batch (needs a -DORBX_CHAIN_STAMPS build; every
frame's workgroup for that region stamps the same slots, the last one to run is what is read: a workgroup under full load).
usage (GPU box): python tools/cols_stamps_batch.py ./stamps.so [frames=512]"""
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
buf = np.zeros(32, np.uint64)
L.orbx_debug_chain_stamps(buf.ctypes.data_as(C.c_void_p))
t = buf.astype(np.int64)
names = ["start", "loads + coefficients + image rectangle in LDS"] + ["level %d derived, level %d written" % (j + 1, j) for j in range(0, 9)]
nn = max(i for i in range(20) if t[i] > 0)
for i in range(1, nn + 1):
    print("%-55s %8.2f us" % (names[i], (t[i] - t[i - 1]) / 100.0))
print("total %.2f us (forms %s)" % ((t[nn] - t[0]) / 100.0, ex.last_forms()))
