#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random image sizes, feature counts, pyramid parameters, thresholds, lapping areas
and content, every stage and the final arrays compared with the oracle (the same checks as tests/test_gpu_parity.py).
usage: fuzz_parity.py [n_cases] [seed]            (tests/test_gpu_fuzz.py runs a bounded sweep of the same generator under -m gpu)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np


def draw_case(rng, t):
    """One random geometry + content; pure function of the generator state."""
    from extractorb_amd import synth
    rows, cols = int(rng.integers(200, 900)), int(rng.integers(200, 1400))
    nf = int(rng.integers(30, 4000))
    nlevels = int(rng.integers(1, 10))
    if os.environ.get("FUZZ_BIG"):      # frames beyond 4096 px (two-dword candidates, the ..b quad-tree builds): a wide strip or a tall frame (height <= 2 x width)
        if rng.random() < 0.6:
            rows, cols = int(rng.integers(150, 1300)), int(rng.integers(4100, 7000))
        else:
            rows, cols = int(rng.integers(4100, 5200)), int(rng.integers(2100, 3000))
        nf = int(rng.integers(200, 6000))
        nlevels = int(rng.integers(1, 9))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0]))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
    variant = ["noise", "textured", "sparse", "natural"][int(rng.integers(0, 4))]
    lap = (int(rng.integers(-10, 400)), int(rng.integers(100, 1500)))
    img = synth.frames(variant, 5000 + t, 1, rows, cols)[0]
    if rng.random() < 0.3:       # blobs of saturated / flat content
        y, x = int(rng.integers(0, rows - 60)), int(rng.integers(0, cols - 60))
        img = img.copy(); img[y:y + 60, x:x + 60] = int(rng.choice([0, 255, 128]))
    return dict(rows=rows, cols=cols, nf=nf, nlevels=nlevels, sf=sf, ini=ini, mn=mn, variant=variant, lap=lap, img=img)


def run(n_cases, seed, progress=False):
    """Returns (bit-exact cases, geometries rejected by orbx_create, keypoints compared, descriptor bytes compared)."""
    import extractorb_amd as X
    from helpers import assert_same_result
    from test_gpu_parity import oracle_run, check_stages
    rng = np.random.default_rng(seed)
    done = skipped = nkp = 0
    t0 = time.time()
    for t in range(n_cases):
        c = draw_case(rng, t)
        try:
            ex = X.ORBextractor(c["nf"], c["sf"], c["nlevels"], c["ini"], c["mn"], max_width=c["cols"], max_height=c["rows"])
        except X.OrbxError:
            skipped += 1          # a pyramid level narrower than one FAST cell etc.: the reference has undefined behaviour there
            continue
        o, want = oracle_run(c["img"], c["nf"], c["lap"], c["nlevels"], c["sf"], c["ini"], c["mn"])
        mono, k, d, lvl = ex(c["img"], None, c["lap"])
        what = "seed %d case %d: %dx%d nf=%d levels=%d sf=%.1f th=%d/%d %s lap=%s" % (
            seed, t, c["cols"], c["rows"], c["nf"], c["nlevels"], c["sf"], c["ini"], c["mn"], c["variant"], c["lap"])
        check_stages(ex, o, lvl, c["nlevels"])
        assert_same_result((mono, k, d), want, what)
        done += 1
        nkp += len(k)
        if progress and done % 10 == 0:
            print("%d cases ok (%d rejected geometries), %.0f s" % (done, skipped, time.time() - t0), flush=True)
    return done, skipped, nkp, nkp * 32


if __name__ == "__main__":
    # test aids of the soak scripts: FUZZ_AIDS="poison=165,lds_pollute=77" (they cannot come from ORBX_* variables: include/orbx.h, orbx_debug_set_option)
    if os.environ.get("FUZZ_AIDS"):
        import extractorb_amd as _X
        for kv in os.environ["FUZZ_AIDS"].split(","):
            _X.debug_set_option(kv.split("=")[0], int(kv.split("=")[1]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    done, skipped, nkp, nbytes = run(n, seed, progress=True)
    print("fuzz parity: %d cases bit-exact (%d keypoints, %d descriptor bytes), %d geometries rejected by orbx_create, %.0f s"
          % (done, nkp, nbytes, skipped, time.time() - t0))
