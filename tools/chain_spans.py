"""Diagnostic: start / end of every tile of the one-launch pyramid (k_pyr_chain, one frame), per level (needs a -DORBX_CHAIN_STAMPS build).
usage (GPU box): python tools/chain_spans.py ./stamps.so"""
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
buf = np.zeros(3 * 2048, np.uint64)
L.orbx_debug_chain_spans(buf.ctypes.data_as(C.c_void_p))
t = buf.astype(np.int64).reshape(-1, 3)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("%d tiles; first start 0, last start %.2f us, last end %.2f us" % (len(t), (t[:, 0].max() - t0) / 100.0, (t[:, 1].max() - t0) / 100.0))
for lvl in sorted(set(t[:, 2])):
    m = t[t[:, 2] == lvl]
    d = (m[:, 1] - m[:, 0]) / 100.0
    print("level %d: %4d tiles  start %.2f..%.2f us  duration mean %.2f max %.2f  last end %.2f us" % (lvl, len(m), (m[:, 0].min() - t0) / 100.0, (m[:, 0].max() - t0) / 100.0, d.mean(), d.max(), (m[:, 1].max() - t0) / 100.0))
