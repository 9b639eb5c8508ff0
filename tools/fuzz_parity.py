#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random image sizes, feature counts, pyramid parameters, thresholds, lapping areas
and content, every stage and the final arrays compared with the oracle (the same checks as tests/test_gpu_parity.py).
usage: fuzz_parity.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import extractorb_amd as X
from extractorb_amd import synth
from helpers import assert_same_result
from test_gpu_parity import oracle_run, check_stages

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
done = skipped = 0
t0 = time.time()
for t in range(n_cases):
    rows, cols = int(rng.integers(200, 900)), int(rng.integers(200, 1400))
    nf = int(rng.integers(30, 4000))
    nlevels = int(rng.integers(1, 10))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0]))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
    variant = ["noise", "textured", "sparse", "natural"][int(rng.integers(0, 4))]
    lap = (int(rng.integers(-10, 400)), int(rng.integers(100, 1500)))
    img = synth.frames(variant, 5000 + t, 1, rows, cols)[0]
    if rng.random() < 0.3:       # blobs of saturated / flat content
        y, x = int(rng.integers(0, rows - 60)), int(rng.integers(0, cols - 60))
        img = img.copy(); img[y:y + 60, x:x + 60] = int(rng.choice([0, 255, 128]))
    try:
        ex = X.ORBextractor(nf, sf, nlevels, ini, mn, max_width=cols, max_height=rows)
    except X.OrbxError as e:
        skipped += 1
        continue
    o, want = oracle_run(img, nf, lap, nlevels, sf, ini, mn)
    mono, k, d, lvl = ex(img, None, lap)
    what = "case %d: %dx%d nf=%d levels=%d sf=%.1f th=%d/%d %s lap=%s" % (t, cols, rows, nf, nlevels, sf, ini, mn, variant, lap)
    check_stages(ex, o, lvl, nlevels)
    assert_same_result((mono, k, d), want, what)
    done += 1
    if done % 10 == 0:
        print("%d cases ok (%d rejected geometries), %.0f s" % (done, skipped, time.time() - t0), flush=True)
print("fuzz parity: %d cases bit-exact, %d geometries rejected by orbx_create, %.0f s" % (done, skipped, time.time() - t0))
