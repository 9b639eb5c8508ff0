#!/usr/bin/env python3
"""Anatomy of the one-frame host call (orbx_extract_view, pageable image): the library's own per-phase wall clock (ORBX_HOST_TIMING=1) next to the
call's total as the caller sees it.  usage (GPU box): python tools/host_call_anatomy.py"""
import ctypes as C, os, sys, time
os.environ["ORBX_HOST_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import extractorb_amd as X
from extractorb_amd import synth

rows, cols = 480, 640
imgs = synth.frames("noise", 0, 8, rows, cols)      # eight different pageable images in turn (cold-ish caches, as a camera stream)
ex = X.ORBextractor(1000)
L, h = ex._L, ex._h
L.orbx_debug_host_timing.argtypes = [C.c_void_p, C.c_void_p]
pk, pd, plk, plc, n, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
kps = np.zeros(ex.capacity, X.KEYPOINT_DTYPE); desc = np.zeros((ex.capacity, 32), np.uint8)


def call(i):
    im = imgs[i & 7]
    assert L.orbx_extract_view(h, im.ctypes.data_as(C.c_void_p), rows, cols, cols, 0, 1000, 0, C.byref(pk), C.byref(pd), C.byref(n), C.byref(m), C.byref(plk), C.byref(plc)) == 0
    C.memmove(kps.ctypes.data, pk.value, 28 * n.value); C.memmove(desc.ctypes.data, pd.value, 32 * n.value)


for i in range(50):
    call(i)
out = (C.c_double * 8)(); calls = C.c_long()
L.orbx_debug_host_timing(out, C.byref(calls))
N = 400
t = time.perf_counter()
for i in range(N):
    call(i)
dt = (time.perf_counter() - t) / N
L.orbx_debug_host_timing(out, C.byref(calls))
us = [v / calls.value * 1e6 for v in out]
print("one 640x480 frame per call, pageable in, host out: %.1f us per call as the caller sees it (%d calls)" % (dt * 1e6, N))
print("  inside orbx_extract_view: enqueue %.1f us (pointer query %.1f, staging memcpy %.1f, staging + H2D enqueue %.1f, rest = table checks + 4 launches), wait %.1f us"
      % (us[0], us[2], us[3], us[4], us[1]))
print("  outside: %.1f us (ctypes + two result memmoves)" % (dt * 1e6 - us[0] - us[1]))
