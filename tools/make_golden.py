#!/usr/bin/env python3
"""Generates tests/golden/ (run in the build container; the GPU box only reads the results).

Inputs: the reference's own demo images (data fixtures), decoded to 8-bit gray the
way cv::imread(..., IMREAD_GRAYSCALE) would and stored as lossless PNG so that the pixel values no longer
depend on a JPEG decoder:
  pic/luna.jpg                       -> luna_gray.png        (512x512, Y channel)
  pic/robot/865_im.jpg               -> robot_865_gray.png   (640x480)
  pic/TUM/dataset-corridor2_512_16/* -> tum_corridor_gray.png (512x512, 16-bit -> high byte)
  pic/TUM/dataset-room4_512_16/mav0/cam0/data/1520531124150444163.png -> tum_room4_gray.png (512x512, 16-bit -> high byte):
      the frame of the reference's ONE recorded result, img_folder/Screenshot.png ("ORB_SLAM3 has total 1420 keypoints",
      printed by src/orb_extractor/main_orb_extractor.cpp:34-53 with nFeatures = 1500): tests/test_reference_pin.py
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
    ("tum_room4_1500", "tum_room4_gray.png", 1500, (0, 1000)),         # the Screenshot.png frame and parameters: 1420 keypoints
]


def main():
    os.makedirs(OUT, exist_ok=True)
    srcs = {
        "luna_gray.png": os.path.join(REF, "luna.jpg"),
        "robot_865_gray.png": os.path.join(REF, "robot", "865_im.jpg"),
        "tum_corridor_gray.png": sorted(glob.glob(os.path.join(REF, "TUM", "dataset-corridor2_512_16", "**", "*.png"),
                                                  recursive=True))[0],
        "tum_room4_gray.png": os.path.join(REF, "TUM", "dataset-room4_512_16", "mav0", "cam0", "data", "1520531124150444163.png"),
    }
    for name, src in srcs.items():
        g = gray8(src)
        Image.fromarray(g).save(os.path.join(OUT, name), optimize=True)
        print(name, g.shape, "mean %.4f" % g.mean())
    # the reference's own rendering of its result: the "ORB_SLAM3 extract keypoints" window of img_folder/Screenshot.png (imshow of
    # drawKeypoints(image, keypoints of all levels scaled to level 0), 512 x 512 at native size), client area at (577, 315) of the screen capture
    shot = np.array(Image.open(os.path.join(os.path.dirname(REF), "img_folder", "Screenshot.png")).convert("RGB"))
    win = shot[315:315 + 512, 577:577 + 512]
    Image.fromarray(win).save(os.path.join(OUT, "screenshot_room4_window.png"), optimize=True)
    room4 = np.array(Image.open(os.path.join(OUT, "tum_room4_gray.png")))
    print("screenshot window: %.4f of its pixels equal the room4 frame exactly" % (win == room4[:, :, None]).all(axis=2).mean())
    for case, img, nf, lap in CASES:
        g = np.array(Image.open(os.path.join(OUT, img)))
        o = O.Oracle(nf, 1.2, 8, 20, 7)
        mono, k, d = o.extract(g, lap)
        counts = np.array([len(o.level_keypoints(l)) for l in range(8)], np.int32)
        ncand = np.array([len(o.candidates(l)) for l in range(8)], np.int32)
        np.savez_compressed(os.path.join(OUT, case + ".npz"), image=img, nfeatures=nf, lapping=np.array(lap, np.int32),
                            mono_index=mono, keypoints=k, descriptors=d, level_counts=counts, candidate_counts=ncand)
        print(case, "n=%d mono=%d" % (len(k), mono), counts.tolist(), ncand.tolist())


def from_opencv():
    """`make_golden.py --from-opencv`: for a maintainer on a machine WITH OpenCV 3.x (this image has none).

    Runs the seven OpenCV primitives the reference calls through the real cv2 on seeded inputs, compares
    each with the oracle's restatement, prints the first mismatch per primitive, and writes tests/golden/opencv_pins.npz
    (inputs + cv2's outputs + cv2.__version__).  tests/test_oracle_golden.py::test_oracle_equals_opencv_pins then checks the
    oracle against that file on every run: this is what turns "parity unpinned" into a pinned oracle.

    Check first, in this order (the three semantics a version change of OpenCV could move):
      1. 8-bit INTER_LINEAR resize: 11-bit coefficients, (b*(h>>4))>>16 vertical rounding            (A.1, all 3.x/4.x generic paths)
      2. 7x7 sigma-2 GaussianBlur taps {18,34,49,55,49,34,18} (sum 257): 3.0-3.4.x; builds from 4.1 on renormalise to 256  (A.2)
      3. undistortPoints: 5 fixed-point iterations in double                                            (3.x; 4.x iterates to a criterion)
    """
    import cv2
    rng = np.random.default_rng(20261004)
    pins, bad = {"opencv_version": np.array(cv2.__version__)}, []
    img = rng.integers(0, 256, (333, 517), dtype=np.uint8)
    smooth = cv2.GaussianBlur(rng.integers(0, 256, (480, 640), dtype=np.uint8), (9, 9), 3)
    # A.1 resize: the pyramid's own ratios (level l from level l-1, cvRound sizes)
    for i, (src, (dw, dh)) in enumerate([(img, (431, 278)), (smooth, (533, 400)), (smooth[:400, :533], (444, 333)), (img, (258, 166))]):
        want = cv2.resize(src, (dw, dh), interpolation=cv2.INTER_LINEAR)
        pins["resize_src_%d" % i], pins["resize_dst_%d" % i] = src, want
        if not np.array_equal(O.resize_linear(src, dw, dh), want):
            bad.append("resize case %d: %d pixels differ" % (i, int((O.resize_linear(src, dw, dh) != want).sum())))
    # A.2 blur
    for i, src in enumerate([img, smooth, np.full((64, 80), 255, np.uint8)]):
        want = cv2.GaussianBlur(src, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
        pins["blur_src_%d" % i], pins["blur_dst_%d" % i] = src, want
        if not np.array_equal(O.gaussian_blur7(src), want):
            bad.append("GaussianBlur case %d: %d pixels differ (taps renormalised to 256 in this OpenCV?)" % (i, int((O.gaussian_blur7(src) != want).sum())))
    # A.3 FAST 9/16 with NMS on cell-sized ROIs and on a whole image, thresholds 20 and 7
    for i, (src, th) in enumerate([(img[40:77, 100:137], 20), (img[40:77, 100:137], 7), (smooth, 20), (smooth, 7), (img, 20)]):
        det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        kps = det.detect(np.ascontiguousarray(src))
        want = np.array([(k.pt[0], k.pt[1], k.response) for k in kps], np.float32).reshape(-1, 3)
        got = O.fast(np.ascontiguousarray(src), th, True)
        g3 = np.stack([got["x"], got["y"], got["response"]], 1).astype(np.float32) if len(got) else np.zeros((0, 3), np.float32)
        pins["fast_src_%d" % i], pins["fast_th_%d" % i], pins["fast_out_%d" % i] = np.ascontiguousarray(src), np.array(th), want
        if g3.shape != want.shape or not np.array_equal(g3, want):
            bad.append("FAST case %d (threshold %d): %d vs %d keypoints or different order/response" % (i, th, len(g3), len(want)))
    # A.4 border
    want = cv2.copyMakeBorder(img, 19, 19, 19, 19, cv2.BORDER_REFLECT_101)
    pins["border_src"], pins["border_dst"] = img, want
    if not np.array_equal(np.pad(img, 19, mode="reflect"), want):
        bad.append("copyMakeBorder REFLECT_101 differs from np.pad(reflect)")
    # A.5 fastAtan2 on integer moments (every |m| < 2.9e6 is exact in float)
    y = rng.integers(-2900000, 2900000, 100000).astype(np.float32); x = rng.integers(-2900000, 2900000, 100000).astype(np.float32)
    want = np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    pins["atan_y"], pins["atan_x"], pins["atan_out"] = y, x, want
    if not np.array_equal(O.fast_atan2(y, x), want):
        bad.append("fastAtan2: %d of %d angles differ" % (int((O.fast_atan2(y, x) != want).sum()), len(y)))
    # undistortPoints (Frame::UndistortKeyPoints, src/Frame.cc:748-782) with EuRoC's camera
    cam = O.camera(458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    pts = (rng.random((2000, 1, 2)) * np.array([752, 480])).astype(np.float32)
    K = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float32)
    want = cv2.undistortPoints(pts, K, cam[4:8].astype(np.float32), None, K).reshape(-1, 2)
    kin = np.zeros(len(pts), O.KEYPOINT_DTYPE); kin["x"], kin["y"] = pts[:, 0, 0], pts[:, 0, 1]
    un, _, _ = O.frame_finish(cam, kin, O.image_bounds(cam, 752, 480))
    pins["undist_cam"], pins["undist_in"], pins["undist_out"] = cam, pts.reshape(-1, 2), want
    if not (np.array_equal(un["x"], want[:, 0]) and np.array_equal(un["y"], want[:, 1])):
        bad.append("undistortPoints: %d of %d points differ (max %.3g px)" % (
            int(((un["x"] != want[:, 0]) | (un["y"] != want[:, 1])).sum()), len(pts),
            float(max(np.abs(un["x"] - want[:, 0]).max(), np.abs(un["y"] - want[:, 1]).max()))))
    np.savez_compressed(os.path.join(OUT, "opencv_pins.npz"), **pins)
    print("OpenCV %s: wrote %s" % (cv2.__version__, os.path.join(OUT, "opencv_pins.npz")))
    for b in bad:
        print("MISMATCH:", b)
    print("oracle == OpenCV on every primitive" if not bad else "%d primitive(s) differ: see above" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(from_opencv() if "--from-opencv" in sys.argv[1:] else main())
