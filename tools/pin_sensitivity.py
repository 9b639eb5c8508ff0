#!/usr/bin/env python3
"""Prints the sensitivity table of the reference-held pin: the oracle's keypoint total on the Screenshot.png frame
(tests/golden/tum_room4_gray.png, nFeatures 1500) with ONE restated semantic swapped for a plausible alternative at a time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O                                   # noqa: E402
from helpers import load_gray                            # noqa: E402
from test_reference_pin import MUTATIONS, REFERENCE_PARAMS, REFERENCE_TOTAL   # noqa: E402

img = load_gray("tum_room4_gray.png")
print("| semantic swapped (reference lines) | per-level keypoints | total | 1420 moves? |")
print("|---|---|---|---|")
for m in [0] + sorted(MUTATIONS):
    o = O.Oracle(*REFERENCE_PARAMS)
    o.set_mutation(m)
    o.extract(img, (0, 1000))
    c = [len(o.level_keypoints(l)) for l in range(8)]
    name = "none (the restatement)" if m == 0 else MUTATIONS[m][0]
    print("| %s | %s | %d | %s |" % (name, " ".join(map(str, c)), sum(c), "—" if m == 0 else ("yes" if sum(c) != REFERENCE_TOTAL else "no")))
