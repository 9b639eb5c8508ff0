"""Diagnostic: where one level-0 quad-tree workgroup spends its time (needs a -DORBX_OCT_STAMPS build)."""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import extractorb_amd.orbextractor as M
M._LIB = sys.argv[1]
import extractorb_amd as X
from extractorb_amd import synth
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fr = synth.frames("noise", 0, B, 480, 640)
ex = X.ORBextractor(int(sys.argv[3]) if len(sys.argv) > 3 else 1000, max_batch=B)
for _ in range(3):
    ex.extract_batch(fr)
L = X.load_library()
buf = np.zeros(128, np.uint64)
L.orbx_debug_oct_stamps(buf.ctypes.data_as(C.c_void_p))
n = int(buf[0])
names = {0: "start", 1: "count pyramid + first size", 2: "node-level(ph1)", 3: "node-level(ph2)", 4: "build", 5: "sweep(if any)", 6: "final", 7: "scan+roots+tables", 8: "sweep0", 9: "pyramid", 10: "ph2: child look-ups + rank counting -> barrier", 11: "ph2: growth scan -> barrier", 12: "ph2: break rank -> barrier", 13: "ph2: creation offsets -> barrier", 14: "ph2: kept scan -> barrier", 15: "ph2: node creation -> barrier"}
prev = None
for i in range(n):
    t, sid = int(buf[1 + i]) >> 8, int(buf[1 + i]) & 0xff
    if prev is not None:
        print("%-48s %8.2f us" % (names[sid], (t - prev) / 100.0))   # s_memrealtime ticks at 100 MHz
    prev = t
sp = np.zeros(32, np.uint64)
if hasattr(L, "orbx_debug_oct_spans") and L.orbx_debug_oct_spans(sp.ctypes.data_as(C.c_void_p)) == 0:
    s = sp.astype(np.int64).reshape(-1, 2)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    print("frame 0's workgroups, (start, end) us after the first start: " + "  ".join("L%d %.2f-%.2f" % (i, (a - t0) / 100.0, (b - t0) / 100.0) for i, (a, b) in enumerate(s)))
