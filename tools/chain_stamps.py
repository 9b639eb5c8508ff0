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
names = ["start", "loads issued + level-1 region + coefficients in LDS"] + ["level %d region" % j for j in range(2, 7)] + ["tile written"]
for i in range(1, 8):
    print("%-55s %8.2f us" % (names[i], (t[i] - t[i - 1]) / 100.0))
