#!/usr/bin/env python3
"""Where the host-in / host-out call of ONE frame spends its time: the same frame through
  v0  orbx_extract, pageable in / out (the maintainer's call shape, Frame.cc:419-427)
  v1  device-resident in / out + orbx_synchronize (the floor: kernels + one wait)
  v2  device-resident in, results written by the kernels straight into pinned host memory
  v3  image read by the pyramid kernel straight from pinned host memory, results as v2
  v4  v3 + the copy of a pageable image into the pinned buffer (the whole maintainer shape, zero-copy form)
  v5  orbx_extract_batch_begin / _end_view with a pinned image (H2D + D2H copies on the stream)
Medians of `reps` calls, microseconds."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import extractorb_amd as X
from extractorb_amd import synth


def med(f, reps):
    for _ in range(20):
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6


def main():
    import torch
    rows, cols = int(os.environ.get("ROWS", 480)), int(os.environ.get("COLS", 640))
    nf = int(os.environ.get("NFEAT", 1000))
    reps = int(os.environ.get("REPS", 400))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    img = synth.frames("noise", 0, 1, rows, cols)[0]
    ex = X.ORBextractor(nf, max_width=cols, max_height=rows, max_batch=1)
    L, h = ex._L, ex._h
    cap = ex.capacity
    kps = np.zeros(cap, X.KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    n, mono = C.c_int(), C.c_int()
    v0 = lambda: L.orbx_extract(h, p(img), rows, cols, cols, 0, 1000, p(kps), p(desc), cap, C.byref(n), C.byref(mono), None, None)
    assert v0() == 0
    want = (n.value, kps[:n.value].tobytes(), desc[:n.value].tobytes())
    out = {}
    out["v0 orbx_extract pageable"] = med(v0, reps)
    d_img = torch.from_numpy(img).cuda()
    d_k = torch.zeros(cap * 28, dtype=torch.uint8, device="cuda"); d_d = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    dev = lambda im, k, d, nn, mm: L.orbx_extract_batch_device(h, 1, C.c_void_p(im), rows, cols, cols, rows * cols, None, C.c_void_p(k), C.c_void_p(d), cap,
                                                              C.c_void_p(nn), C.c_void_p(mm), None, None)

    def v1():
        assert dev(d_img.data_ptr(), d_k.data_ptr(), d_d.data_ptr(), d_n.data_ptr(), d_n.data_ptr() + 4) == 0
        L.orbx_synchronize(h)
    out["v1 device in/out + sync"] = med(v1, reps)
    pk = X.pinned_empty(cap, X.KEYPOINT_DTYPE); pd = X.pinned_empty((cap, 32)); pn = X.pinned_empty(2, np.int32)
    pimg = X.pinned_empty((rows, cols))
    pimg[...] = img

    def v2():
        assert dev(d_img.data_ptr(), pk.ctypes.data, pd.ctypes.data, pn.ctypes.data, pn.ctypes.data + 4) == 0
        L.orbx_synchronize(h)
    out["v2 device in, pinned out (zero-copy)"] = med(v2, reps)
    assert (int(pn[0]), pk[:pn[0]].tobytes(), pd[:pn[0]].tobytes()) == want, "v2 differs"

    def v3():
        assert dev(pimg.ctypes.data, pk.ctypes.data, pd.ctypes.data, pn.ctypes.data, pn.ctypes.data + 4) == 0
        L.orbx_synchronize(h)
    pk[...] = 0; pd[...] = 0
    out["v3 pinned in (zero-copy), pinned out"] = med(v3, reps)
    assert (int(pn[0]), pk[:pn[0]].tobytes(), pd[:pn[0]].tobytes()) == want, "v3 differs"

    def v4():
        np.copyto(pimg, img)
        v3()
    out["v4 pageable -> pinned memcpy + v3"] = med(v4, reps)

    def v4b():
        np.copyto(pimg, img)
        v3()
        kps[:pn[0]] = pk[:pn[0]]; desc[:pn[0]] = pd[:pn[0]]
    out["v4b v4 + results copied to pageable arrays"] = med(v4b, reps)
    vk, vd, vn, vm, vc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()

    def v5():
        assert L.orbx_extract_batch_begin(h, 1, p(pimg), rows, cols, cols, rows * cols, None, 0) == 0
        assert L.orbx_extract_batch_end_view(h, C.byref(vk), C.byref(vd), C.byref(vc), C.byref(vn), C.byref(vm)) == 0
    out["v5 begin/end_view pinned in (H2D + D2H copies)"] = med(v5, reps)
    t = lambda f: med(f, reps)
    out["   (np.copyto pageable -> pinned alone)"] = t(lambda: np.copyto(pimg, img))
    print("%dx%d, %d features, one frame per call; median / p10 us" % (cols, rows, nf))
    for k, (m, lo) in out.items():
        print("  %-52s %8.1f %8.1f" % (k, m, lo))


if __name__ == "__main__":
    main()
