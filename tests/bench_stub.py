"""The extractor that stands in for the HIP path when the CPU suite REHEARSES bench.py's N > 1 control flow
(`bench.py --backend gloo --extractor-factory bench_stub:make_extractor`, tests/test_bench_world2.py).

Test infrastructure: the result slabs are written by the oracle on host memory.  bench.py itself holds no such
code: it imports this module only when the flag names it, prints `value: null` and
`"stub": true`, and its timed region never reaches the oracle."""
import ctypes

import numpy as np

import oracle_lib as O


class OracleSlabWriter:
    """`.capacity` + `.extract_batch_device(...)` with the argument meaning of extractorb_amd.ORBextractor's, on HOST pointers."""

    def __init__(self, nfeatures, scale, levels, ini_th, min_th):
        self.capacity = nfeatures + 3 * levels
        self.o = O.Oracle(nfeatures, scale, levels, ini_th, min_th)

    def extract_batch_device(self, d_imgs, nB, rows, cols, pk, pd, pn, pm, cap, lapping=(0, 1000)):
        imgs = d_imgs.numpy()
        for f in range(nB):
            mono, k, d = self.o.extract(imgs[f], lapping)
            ctypes.memmove(pk + f * cap * 28, k.ctypes.data, len(k) * 28)
            ctypes.memmove(pd + f * cap * 32, np.ascontiguousarray(d).ctypes.data, len(k) * 32)
            ctypes.c_int32.from_address(pn + 4 * f).value = len(k)
            ctypes.c_int32.from_address(pm + 4 * f).value = mono


def make_extractor(nfeatures, scale, levels, ini_th, min_th):
    return OracleSlabWriter(nfeatures, scale, levels, ini_th, min_th)
