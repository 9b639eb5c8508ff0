"""Diagnostic: stage times of k_describe's waves for one 640x480 frame (needs a -DORBX_DESC_STAMPS build).  usage (GPU box): python tools/desc_spans.py ./stamps.so"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ORBX_LIBRARY"] = sys.argv[1]
import extractorb_amd as X
from extractorb_amd import synth
fr = synth.frames("noise", 0, 1, 480, 640)
ex = X.ORBextractor(1000, max_batch=1)
for _ in range(3):
    ex.extract_batch(fr)
L = X.load_library()
buf = np.zeros(6 * 1024, np.uint64)
L.orbx_debug_desc_stamps(buf.ctypes.data_as(C.c_void_p))
t = buf.astype(np.int64).reshape(-1, 6)
t = t[(t[:, 0] > 0) & (t[:, 4] > 0)]
t0 = t[:, 0].min()
names = ["weights + barrier, per-level counters", "level geometry + selection entry", "patch loads -> LDS", "IC_Angle + rBRIEF"]
print("%d waves: starts 0..%.2f us, ends (before the final stores) ..%.2f us" % (len(t), (t[:, 0].max() - t0) / 100.0, (t[:, 4].max() - t0) / 100.0))
for i, n in enumerate(names):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    print("%-45s mean %.2f max %.2f us" % (n, d.mean(), d.max()))
