#!/usr/bin/env python3
"""Randomised parity sweep over BATCH shapes on the GPU box: random frames per call (1 .. 160), image sizes and feature counts, so that the host's
launch policies (pyramid form and region size, FAST form, leaf tables, quad-tree workgroup size, blur form) are crossed at their thresholds;
three frames of every batch (first, a random one, last) against the oracle, and two calls on one handle against each other.
usage: fuzz_batches.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np


def run(n_cases, seed, progress=False):
    import extractorb_amd as X
    from extractorb_amd import synth
    from helpers import assert_same_result
    from test_gpu_parity import oracle_run
    rng = np.random.default_rng(seed)
    done = skipped = checked = 0
    t0 = time.time()
    for t in range(n_cases):
        B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 31, 32, 33, 48, 64, 65, 96, 127, 128, 129, 160]))
        rows, cols = [(480, 640), (240, 320), (333, 517), (376, 1241), (720, 1280), (480, 752), (1080, 1920)][int(rng.integers(0, 7))]
        cap = 140e6 if rows == 1080 else 40e6      # (1080p: up to 67 frames, so that its per-level pyramid launches and the 1024-thread queued quad-tree are drawn too)
        if rows * cols * B > cap:
            B = max(1, int(cap // (rows * cols)))
        nf = int(rng.choice([300, 500, 1000, 1200, 2000]))
        variant = ["noise", "textured", "sparse", "natural"][int(rng.integers(0, 4))]
        frames = synth.frames(variant, 9000 + t, min(B, 8), rows, cols)
        frames = np.concatenate([frames] * ((B + len(frames) - 1) // len(frames)))[:B].copy()
        if B > 1:
            frames[B - 1] = frames[B - 1][::-1, ::-1]          # (the last frame differs from the ones it repeats)
        try:
            ex = X.ORBextractor(nf, max_width=cols, max_height=rows, max_batch=B)
        except X.OrbxError:
            skipped += 1
            continue
        a = ex.extract_batch(frames)
        b = ex.extract_batch(frames)
        what = "case %d: B=%d %dx%d nf=%d %s" % (t, B, cols, rows, nf, variant)
        for f in range(B):
            assert a[f][0] == b[f][0] and np.array_equal(a[f][1], b[f][1]) and np.array_equal(a[f][2], b[f][2]), what + ": frame %d differs between two calls" % f
        for f in sorted({0, int(rng.integers(0, B)), B - 1}):
            o, want = oracle_run(frames[f], nf)
            assert_same_result(a[f][:3], want, what + " frame %d" % f)
            checked += 1
        done += 1
        if progress:
            print("%s ok  [%.0f s]" % (what, time.time() - t0), flush=True)
    return done, skipped, checked


if __name__ == "__main__":
    # test aids of the soak scripts: FUZZ_AIDS="poison=165,lds_pollute=77" (they cannot come from ORBX_* variables: include/orbx.h, orbx_debug_set_option)
    if os.environ.get("FUZZ_AIDS"):
        import extractorb_amd as _X
        for kv in os.environ["FUZZ_AIDS"].split(","):
            _X.debug_set_option(kv.split("=")[0], int(kv.split("=")[1]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d, s, c = run(n, seed, progress=True)
    print("batch fuzz: %d batches bit-exact on %d oracle-checked frames, %d geometries rejected" % (d, c, s))
