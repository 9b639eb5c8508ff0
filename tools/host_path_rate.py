#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (never used as bench.py's `value`):
  sync      orbx_extract_batch on pageable numpy buffers: H2D + path + D2H + wait per call
  pipelined orbx_extract_batch_begin/_end on pinned buffers (orbx_host_alloc), two handles used alternately so the
            transfers of one batch overlap the kernels of the other
`--json` prints one JSON line {"<B>": {"sync_fps": .., "pipelined_fps": .., "ms_per_batch": ..}, "hip_runtime": path} instead of the table: bench.py runs this
file as a CHILD process for its `host_to_host_fps` - a process that has imported torch runs liborbx.so on the HIP runtime torch bundles (ROCm 7.0 in this
image), where an input copy and another stream's kernels do not overlap (83 k frames/s at 64 frames per call); a C++ host links the system runtime (ROCm
7.2: 162 k).  HOST_RATE_TORCH=3 reproduces the in-torch figure here."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import extractorb_amd as X
from extractorb_amd import synth


def main():
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    extra = []
    mode = os.environ.get("HOST_RATE_TORCH", "")      # (round 6: what bench.py's process holds beside the two handles - torch's context and a 512-frame handle)
    if mode in ("1", "2", "3"):
        import torch
        torch.cuda.init()
        t = torch.zeros(1 << 20, device="cuda")
    if mode in ("1", "2", "4"):
        extra.append(X.ORBextractor(1000, max_batch=512))
        if mode == "2":
            extra[0].set_stream(torch.cuda.current_stream().cuda_stream)
    lap_mode = os.environ.get("HOST_RATE_LAP")
    as_json = "--json" in sys.argv
    result = {}
    for B in [int(b) for b in os.environ.get("HOST_RATE_BATCHES", "1,8,64,256").split(",")]:
        fr = synth.frames("noise", 0, min(B, 64), 480, 640)
        fr = np.concatenate([fr] * ((B + len(fr) - 1) // len(fr)))[:B]
        ex = X.ORBextractor(1000, max_batch=B)
        L, h = ex._L, ex._h
        cap = ex.capacity
        kps = np.zeros((B, cap), X.KEYPOINT_DTYPE); desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
        call = lambda: L.orbx_extract_batch(h, B, p(fr), 480, 640, 640, 480 * 640, None, p(kps), p(desc), cap, p(n), p(mono), None, None)
        for _ in range(3):
            assert call() == 0
        reps = max(3, 256 // B)
        t = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t) / reps
        # pipelined: two handles, pinned input
        exs = [ex, X.ORBextractor(1000, max_batch=B)]
        pin = [X.pinned_empty(fr.shape), X.pinned_empty(fr.shape)]
        for a in pin:
            a[...] = fr
        outs = [(np.zeros((B, cap), X.KEYPOINT_DTYPE), np.zeros((B, cap, 32), np.uint8), np.zeros(B, np.int32), np.zeros(B, np.int32)) for _ in range(2)]
        lap_arr = (C.c_int * (2 * B))(*([0, 1000] * B)) if lap_mode else None
        begin = lambda i: exs[i]._L.orbx_extract_batch_begin(exs[i]._h, B, p(pin[i]), 480, 640, 640, 480 * 640, lap_arr, 0)
        endc = lambda i: exs[i]._L.orbx_extract_batch_end(exs[i]._h, p(outs[i][0]), p(outs[i][1]), cap, p(outs[i][2]), p(outs[i][3]), None, None)
        assert begin(0) == 0 and begin(1) == 0 and endc(0) == 0 and endc(1) == 0
        assert outs[0][2].tolist() == n.tolist() and np.array_equal(outs[1][1], desc)
        vk, vd, vn, vm, vc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        end = lambda i: exs[i]._L.orbx_extract_batch_end_view(exs[i]._h, C.byref(vk), C.byref(vd), C.byref(vc), C.byref(vn), C.byref(vm))   # zero-copy
        for k in range(4):                      # the timed call shape once more, untimed (begin / end_view on both handles)
            assert begin(k & 1) == 0 and end(k & 1) == 0
        reps2 = max(16, 512 // B)
        t = time.perf_counter()
        assert begin(0) == 0
        for k in range(1, reps2):
            assert begin(k & 1) == 0            # next batch enqueued before the previous one is collected
            assert end((k - 1) & 1) == 0
        assert end((reps2 - 1) & 1) == 0
        dt2 = (time.perf_counter() - t) / reps2
        result[str(B)] = dict(sync_fps=round(B / dt, 1), pipelined_fps=round(B / dt2, 1), ms_per_batch=round(dt2 * 1e3, 4), timed_batches=reps2)
        if not as_json:
            print("B=%4d  sync/pageable %8.3f ms/call %9.1f frames/s   pipelined/pinned/zero-copy (2 handles) %8.3f ms/batch %9.1f frames/s"
                  % (B, dt * 1e3, B / dt, dt2 * 1e3, B / dt2))
        for a in pin:
            X.pinned_free(a)
    if as_json:
        import json
        result["hip_runtime"] = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
        print(json.dumps(result))


if __name__ == "__main__":
    main()
