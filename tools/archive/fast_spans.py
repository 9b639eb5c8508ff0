"""Diagnostic: start / end of every wave of k_fast (FAST cells and the blur lanes that ride in its launch) for one 640x480 frame
(needs a -DORBX_FAST_CLOCK build).   usage (GPU box): python tools/fast_spans.py ./clock.so"""
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
buf = np.zeros(2 * 4096, np.uint64)
L.orbx_debug_fast_spans(buf.ctypes.data_as(C.c_void_p))
s = buf.astype(np.int64).reshape(-1, 2)
idx = np.nonzero(s[:, 0] > 0)[0]
s = s[idx]
t0 = s[:, 0].min()
st, en = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0
d = en - st
print("%d waves: starts 0..%.2f us, lifetime mean %.2f max %.2f, last end %.2f us" % (len(s), st.max(), d[d > 0].mean(), d.max(), en.max()))
nb = 14
edges = np.linspace(0, idx.max() + 1, nb + 1).astype(int)
for lo, hi in zip(edges[:-1], edges[1:]):      # (FAST cells come first in the grid, level 0 first; the blur's lanes are the last workgroups)
    m = (idx >= lo) & (idx < hi) & (d > 0)
    if m.any():
        print("waves %4d..%4d: start %.2f..%.2f  lifetime mean %.2f max %.2f  last end %.2f" % (lo, hi - 1, st[m].min(), st[m].max(), d[m].mean(), d[m].max(), en[m].max()))
order = np.argsort(-en)[:8]
print("last to end:", ", ".join("wave %d (start %.2f, %.2f us)" % (idx[i], st[i], d[i]) for i in order))
if len(sys.argv) > 2:
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    for i in range(len(idx)):
        if lo <= idx[i] < hi:
            print("wave %4d start %.2f lifetime %.2f" % (idx[i], st[i], d[i]))
mid = np.zeros(4 * 4096, np.uint64)
if hasattr(L, "orbx_debug_fast_mid") and L.orbx_debug_fast_mid(mid.ctypes.data_as(C.c_void_p)) == 0:
    m = mid.astype(np.int64).reshape(-1, 4)
    wide = os.environ.get("ORBX_FAST_WIDE") == "1"
    for name, sel in ((("level-0 cells (waves 0..1199)", range(0, 1200)), ("deepest cells (waves 3120..3259)", range(3120, 3260))) if wide else (("level-0 cells (waves 0..299)", range(0, 300)), ("deepest cells (waves 780..814)", range(780, 815)))):
        rows = [(m[i, 0] - buf.astype(np.int64).reshape(-1, 2)[i, 0], m[i, 1] - m[i, 0], m[i, 2] - m[i, 1], buf.astype(np.int64).reshape(-1, 2)[i, 1] - m[i, 2]) for i in sel if m[i, 0] > 0 and m[i, 2] > 0]
        a = np.array(rows) / 100.0
        print("%s: start -> staged %.2f, score pass %.2f, NMS + count %.2f, emit %.2f us (means of %d waves)" % (name, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), len(a)))
