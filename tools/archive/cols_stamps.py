"""Diagnostic: level times of one workgroup of the region-major pyramid (k_pyr_cols) and the spans of all (needs a -DORBX_CHAIN_STAMPS build).
usage (GPU box): python tools/cols_stamps.py ./stamps.so"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ORBX_LIBRARY"] = sys.argv[1]
os.environ["ORBX_PYR_COLS"] = "1"
import extractorb_amd as X
from extractorb_amd import synth
fr = synth.frames("noise", 0, 1, 480, 640)
ex = X.ORBextractor(1000, max_batch=1)
for _ in range(3):
    ex.extract_batch(fr)
L = X.load_library()
buf = np.zeros(32, np.uint64)
L.orbx_debug_chain_stamps(buf.ctypes.data_as(C.c_void_p))
t = buf.astype(np.int64)
names = ["start", "loads + coefficients + image rectangle in LDS"] + ["level %d derived, level %d written" % (j + 1, j) for j in range(0, 9)]
n = max(i for i in range(20) if t[i] > 0)
for i in range(1, n + 1):
    print("%-55s %8.2f us" % (names[i], (t[i] - t[i - 1]) / 100.0))
print("total %.2f us" % ((t[n] - t[0]) / 100.0))
sp = np.zeros(3 * 2048, np.uint64)
L.orbx_debug_chain_spans(sp.ctypes.data_as(C.c_void_p))
s = sp.astype(np.int64).reshape(-1, 3)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
d = (s[:, 1] - s[:, 0]) / 100.0
print("%d workgroups: starts 0..%.2f us, duration mean %.2f max %.2f, last end %.2f us" % (len(s), (s[:, 0].max() - t0) / 100.0, d.mean(), d.max(), (s[:, 1].max() - t0) / 100.0))
if len(s) == 192:      # 640x480 at 40-px regions: durations as the 16 x 12 grid of regions (us)
    sp3 = sp.astype(np.int64).reshape(-1, 3)[:192]
    dd = ((sp3[:, 1] - sp3[:, 0]) / 100.0).reshape(12, 16)
    print("durations of the 16 x 12 regions (us):"); print(np.array2string(dd, precision=1, floatmode="fixed", max_line_width=200))
