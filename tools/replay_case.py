#!/usr/bin/env python3
"""Replays ONE draw of tools/fuzz_parity.py's generator (seed, case) several times on the GPU and reports where the bordered pyramid differs from the
oracle.   usage: replay_case.py <seed> <case> [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import extractorb_amd as X
from fuzz_parity import draw_case
from test_gpu_parity import oracle_run

seed, case, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(seed)
for t in range(case + 1):
    c = draw_case(rng, t)
print("case: %dx%d nf=%d levels=%d sf=%.1f th=%d/%d %s lap=%s" % (c["cols"], c["rows"], c["nf"], c["nlevels"], c["sf"], c["ini"], c["mn"], c["variant"], c["lap"]))
o, want = oracle_run(c["img"], c["nf"], c["lap"], c["nlevels"], c["sf"], c["ini"], c["mn"])
for rep in range(reps):
    ex = X.ORBextractor(c["nf"], c["sf"], c["nlevels"], c["ini"], c["mn"], max_width=c["cols"], max_height=c["rows"])
    for call in range(3):
        mono, k, d, lvl = ex(c["img"], None, c["lap"])
        print("rep %d call %d forms %s:" % (rep, call, ex.last_forms()), end=" ")
        for l in range(c["nlevels"]):
            g, w = ex.image_pyramid_level(l, 0, bordered=True), o.level(l, bordered=True)
            if g.shape != w.shape:
                print("level %d shape %s vs %s" % (l, g.shape, w.shape), end="; ")
                continue
            dif = np.argwhere(g != w)
            if len(dif):
                print("level %d: %d bytes differ, rows %d..%d cols %d..%d (of %s), first (%d,%d) gpu %d oracle %d" % (
                    l, len(dif), dif[:, 0].min(), dif[:, 0].max(), dif[:, 1].min(), dif[:, 1].max(), g.shape, dif[0][0], dif[0][1], g[tuple(dif[0])], w[tuple(dif[0])]), end="; ")
        print("ok" if True else "")
