#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (orbx_extract_batch: H2D + path + D2H + sync per call).
Reported in DESIGN.md next to bench.py's device-resident `value`; never used as `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import extractorb_amd as X
from extractorb_amd import synth

def main():
    for B in (1, 8, 64, 256):
        fr = synth.frames("noise", 0, min(B, 64), 480, 640)
        fr = np.concatenate([fr] * ((B + len(fr) - 1) // len(fr)))[:B]
        ex = X.ORBextractor(1000, max_batch=B)
        L, h = ex._L, ex._h
        cap = ex.capacity
        kps = np.zeros((B, cap), X.KEYPOINT_DTYPE); desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
        import ctypes as C
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        call = lambda: L.orbx_extract_batch(h, B, p(fr), 480, 640, 640, 480 * 640, None, p(kps), p(desc), cap, p(n), p(mono), None, None)
        for _ in range(3):
            assert call() == 0
        reps = max(3, 256 // B)
        t = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t) / reps
        print("B=%4d  %8.3f ms/call  %9.1f frames/s (pageable host buffers, synchronous)" % (B, dt * 1e3, B / dt))

if __name__ == "__main__":
    main()
