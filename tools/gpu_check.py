#!/usr/bin/env python3
"""Stage-by-stage parity of the HIP path against the CPU oracle on one image (developer tool).

usage: python tools/gpu_check.py [luna|robot|tum|noise|noise1080] [nfeatures]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from extractorb_amd import ORBextractor  # noqa: E402


def load(name):
    from PIL import Image
    g = os.path.join(ROOT, "tests", "golden")
    if name == "luna":
        return np.array(Image.open(os.path.join(g, "luna_gray.png")))
    if name == "robot":
        return np.array(Image.open(os.path.join(g, "robot_865_gray.png")))
    if name == "tum":
        return np.array(Image.open(os.path.join(g, "tum_corridor_gray.png")))
    rng = np.random.default_rng(7)
    if name == "noise":
        return rng.integers(0, 256, (480, 640), dtype=np.uint8)
    if name == "noise1080":
        return rng.integers(0, 256, (1080, 1920), dtype=np.uint8)
    raise SystemExit("unknown image " + name)


def sort_kps(k):
    return k[np.lexsort((k["x"], k["y"]))]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "luna"
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lap = (0, 1000)
    img = load(name)
    rows, cols = img.shape
    o = O.Oracle(nf, 1.2, 8, 20, 7)
    t = time.time()
    mono_o, k_o, d_o = o.extract(img, lap)
    print("oracle: n=%d mono=%d (%.1f ms)" % (len(k_o), mono_o, (time.time() - t) * 1e3))
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=cols, max_height=rows, max_batch=2)
    t = time.time()
    mono_g, k_g, d_g, lvl_g = ex(img, None, lap)
    print("gpu   : n=%d mono=%d (%.1f ms incl. first-call set-up)" % (len(k_g), mono_g, (time.time() - t) * 1e3))
    ok = True
    for l in range(8):
        pg, po = ex.image_pyramid_level(l), o.level(l)
        same = pg.shape == po.shape and np.array_equal(pg, po)
        bg, bo = ex.image_pyramid_level(l, bordered=True), o.level(l, bordered=True)
        sameb = bg.shape == bo.shape and np.array_equal(bg, bo)
        blg, blo = ex.debug_blurred(l), o.blurred(l)
        has = len(o.level_keypoints(l)) > 0
        sameblur = (not has) or np.array_equal(blg, blo)
        cg, co = sort_kps(ex.debug_candidates(l)), sort_kps(o.candidates(l))
        samec = len(cg) == len(co) and np.array_equal(cg, co)
        lg, lo = lvl_g[l], o.level_keypoints(l)
        samel = len(lg) == len(lo) and np.array_equal(lg, lo)
        print("level %d: pyramid %s border %s blur %s candidates %s (%d vs %d) octree+angle %s (%d vs %d)" %
              (l, same, sameb, sameblur, samec, len(cg), len(co), samel, len(lg), len(lo)))
        if not samel and len(lg) == len(lo):
            bad = [i for i in range(len(lg)) if lg[i] != lo[i]]
            print("   first diffs:", [(i, lg[i], lo[i]) for i in bad[:3]], "n bad", len(bad))
        ok &= same and sameb and sameblur and samec and samel
    samek = len(k_g) == len(k_o) and np.array_equal(k_g, k_o)
    samed = d_g.shape == d_o.shape and np.array_equal(d_g, d_o)
    print("final keypoints %s descriptors %s mono %s" % (samek, samed, mono_g == mono_o))
    if not samed and d_g.shape == d_o.shape:
        bad = np.nonzero((d_g != d_o).any(axis=1))[0]
        print("   descriptor rows differing:", len(bad), bad[:10])
    ok &= samek and samed and mono_g == mono_o
    print("PARITY", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
