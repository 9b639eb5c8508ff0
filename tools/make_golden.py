#!/usr/bin/env python3
"""Generates tests/golden/ (run in the build container; the GPU box only reads the results).

Inputs: the reference's own demo images (data fixtures), decoded to 8-bit gray the
way cv::imread(..., IMREAD_GRAYSCALE) would and stored as lossless PNG so that the pixel values no longer
depend on a JPEG decoder:
  pic/luna.jpg                       -> luna_gray.png        (512x512, Y channel)
  pic/robot/865_im.jpg               -> robot_865_gray.png   (640x480)
  pic/TUM/dataset-corridor2_512_16/* -> tum_corridor_gray.png (512x512, 16-bit -> high byte)
Expected outputs: the CPU oracle's results on those inputs (oracle/orb_oracle.cpp), one .npz per case.
The reference ships no golden vectors and cannot be built here (no OpenCV), so these pin the oracle
against itself over time ("parity unpinned") and give the GPU box known-answer files.
"""
import glob
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

REF = "/root/reference/pic"
OUT = os.path.join(ROOT, "tests", "golden")


def gray8(path):
    im = Image.open(path)
    if im.mode in ("I;16", "I;16B", "I"):
        a = np.array(im).astype(np.uint32)
        return (a >> 8).astype(np.uint8)          # imread(GRAYSCALE) of a 16-bit PNG keeps the high byte
    if im.mode != "L":
        im.draft("L", im.size)                    # libjpeg's Y channel, as imread(GRAYSCALE) of a colour JPEG
        im = im.convert("L")
    return np.array(im, dtype=np.uint8)


CASES = [
    # name, image, nfeatures, lapping
    ("luna_1000", "luna_gray.png", 1000, (0, 1000)),       # BASELINE.json configs[0]
    ("luna_1000_lap00", "luna_gray.png", 1000, (0, 0)),    # rectified-stereo lapping (Frame.cc:109-110)
    ("luna_7500", "luna_gray.png", 7500, (0, 1000)),       # the demos' 5*1500 (main_orb_extractor.cpp:43)
    ("robot_865_1000", "robot_865_gray.png", 1000, (0, 1000)),
    ("robot_865_1200_lap", "robot_865_gray.png", 1200, (200, 400)),   # fisheye-style lapping band
    ("tum_corridor_1000", "tum_corridor_gray.png", 1000, (0, 1000)),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    srcs = {
        "luna_gray.png": os.path.join(REF, "luna.jpg"),
        "robot_865_gray.png": os.path.join(REF, "robot", "865_im.jpg"),
        "tum_corridor_gray.png": sorted(glob.glob(os.path.join(REF, "TUM", "dataset-corridor2_512_16", "**", "*.png"),
                                                  recursive=True))[0],
    }
    for name, src in srcs.items():
        g = gray8(src)
        Image.fromarray(g).save(os.path.join(OUT, name), optimize=True)
        print(name, g.shape, "mean %.4f" % g.mean())
    for case, img, nf, lap in CASES:
        g = np.array(Image.open(os.path.join(OUT, img)))
        o = O.Oracle(nf, 1.2, 8, 20, 7)
        mono, k, d = o.extract(g, lap)
        counts = np.array([len(o.level_keypoints(l)) for l in range(8)], np.int32)
        ncand = np.array([len(o.candidates(l)) for l in range(8)], np.int32)
        np.savez_compressed(os.path.join(OUT, case + ".npz"), image=img, nfeatures=nf, lapping=np.array(lap, np.int32),
                            mono_index=mono, keypoints=k, descriptors=d, level_counts=counts, candidate_counts=ncand)
        print(case, "n=%d mono=%d" % (len(k), mono), counts.tolist(), ncand.tolist())


if __name__ == "__main__":
    main()
