#!/usr/bin/env python3
"""Seeded soak of the matcher rows on the GPU box: the parity tests of tests/test_search_projection.py, tests/test_search_bow.py,
tests/test_search_init.py and tests/test_stereo.py bodies re-run with seeds outside the committed parametrisation.
usage: fuzz_matchers.py [n_seeds] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np

n, s0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 20), (int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
import test_search_projection as P
import test_search_bow as W
import test_search_init as I
from extractorb_amd import synth
t0 = time.time()
done = 0
for k in range(n):
    seed = s0 + k
    rng = np.random.default_rng(seed)
    ratio, stereo, check = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    P.test_gpu_search_equals_oracle_on_random_scenes(seed, ratio, stereo, check and not ratio, int(rng.choice([100, 64, 100])))
    big = rng.random() < 0.3          # the list-less kernel variant
    P.test_gpu_search_settles_crowded_requests_like_the_walk(seed, ratio, 2200 if big else 1024, 1500 if big else 768)
    case = dict(seed=seed, nnratio=float(rng.choice([0.6, 0.7, 0.9])), th_low=int(rng.choice([50, 100])), check=bool(rng.integers(0, 2)),
                n_nodes=int(rng.choice([3, 60, 400, 2000])), tie_heavy=bool(rng.random() < 0.3))
    W.test_gpu_search_by_bow_equals_oracle(case)
    W.test_gpu_keyframe_search_by_bow_equals_oracle(case)
    if k % 5 == 0:      # SearchForInitialization on a freshly extracted stream (heavier: six frames through the HIP path and the oracle)
        frames = synth.frames(["textured", "noise", "sparse", "natural"][int(rng.integers(0, 4))], seed, 6, 480, 640)
        I.run_gpu_pairs(frames, ((0, 1), (1, 1), 5), I.PINHOLE, int(rng.choice([300, 1000, 2000, 5000])), int(rng.choice([30, 100, 300])),
                        float(rng.choice([0.6, 0.9, 1.3])), bool(rng.integers(0, 2)), rounds=int(rng.integers(1, 3)))
    done += 1
print("matcher soak: %d seeds x 4 matcher parity bodies (+ SearchForInitialization every fifth seed) bit-exact, %.0f s" % (done, time.time() - t0))
