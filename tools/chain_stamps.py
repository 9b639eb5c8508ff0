"""Diagnostic: step times of the last (coarsest-level) tile of k_pyr_rest (needs a -DORBX_CHAIN_STAMPS build)."""
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
buf = np.zeros(32, np.uint64)
L.orbx_debug_chain_stamps(buf.ctypes.data_as(C.c_void_p))
t = buf.astype(np.int64)
names = ["start", "loads issued + first region + coefficients in LDS"] + ["step to level %d" % j for j in range(2, 8)] + ["tile written"]
n = max(i for i in range(20) if t[i] > 0)
for i in range(1, n + 1):
    print("%-55s %8.2f us" % (names[min(i, len(names) - 1)], (t[i] - t[i - 1]) / 100.0))
print("first stage: tile record read %.2f, region loads issued %.2f, coefficient loads issued %.2f, coefficients stored %.2f, region stored + barrier %.2f us" % tuple((a - b) / 100.0 for a, b in ((t[20], t[0]), (t[21], t[20]), (t[22], t[21]), (t[23], t[22]), (t[1], t[23]))))
print("total %.2f us" % ((t[n] - t[0]) / 100.0))
