"""Pins each OpenCV primitive the oracle restates by an INDEPENDENT definition written
here in numpy/scipy — the reference ships no tests or golden vectors and OpenCV is absent, so this is the
strongest pin available ("parity unpinned").  CPU only."""
import numpy as np
import pytest
from scipy import ndimage

import oracle_lib as O

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ---------------------------------------------------------------- FAST ------------------------------------
def fast_bruteforce(img, t):
    """FAST-9/16 from the definition: corner <=> 9 contiguous ring pixels all > v+t or all < v-t;
    score = largest threshold at which the pixel is still a corner (cornerScore: that value, as max-min form)."""
    H, W = img.shape
    im = img.astype(np.int32)
    corner = np.zeros((H, W), bool)
    score = np.zeros((H, W), np.int32)
    ring = np.stack([im[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in RING])      # [16, h, w]
    v = im[3:H - 3, 3:W - 3]
    for tt in range(t, 256):                                   # largest tt with a 9-arc => score = tt
        br = ring > v + tt
        dk = ring < v - tt
        any_arc = np.zeros(v.shape, bool)
        for k in range(16):
            idx = [(k + j) % 16 for j in range(9)]
            any_arc |= br[idx].all(0) | dk[idx].all(0)
        if tt == t:
            corner[3:H - 3, 3:W - 3] = any_arc
        if not any_arc.any():
            break
        score[3:H - 3, 3:W - 3][any_arc] = tt
    return corner, score


def fast_nms_bruteforce(img, t):
    corner, score = fast_bruteforce(img, t)
    H, W = img.shape
    s = np.where(corner, score, 0)
    out = []
    for y in range(3, H - 3):
        for x in range(3, W - 3):
            if not corner[y, x]:
                continue
            nb = s[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s[y, x] > nb).all():
                out.append((x, y, s[y, x]))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 40), (3, 1)])
def test_fast_matches_definition(seed, t):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (37, 41), dtype=np.uint8)
    if seed == 2:   # smoother content: fewer, stronger corners
        img = ndimage.uniform_filter(img.astype(np.float32), 3).astype(np.uint8)
    want = fast_nms_bruteforce(img, t)
    got = O.fast(img, t, True)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == want
    assert all(k["size"] == 7 and k["angle"] == -1 and k["octave"] == 0 and k["class_id"] == -1 for k in got)
    # without NMS: every corner, raster order
    corner, score = fast_bruteforce(img, t)
    ys, xs = np.nonzero(corner)
    got2 = O.fast(img, t, False)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got2] == [(x, y, score[y, x]) for y, x in zip(ys, xs)]


def test_fast_edge_cases():
    assert len(O.fast(np.full((20, 20), 77, np.uint8), 7)) == 0                     # flat
    assert len(O.fast(np.zeros((6, 30), np.uint8), 7)) == 0                         # ROI too small to test a pixel
    img = np.zeros((15, 15), np.uint8); img[7, 7] = 255                            # isolated bright dot: dark ring
    k = O.fast(img, 20)
    assert len(k) == 1 and (k[0]["x"], k[0]["y"]) == (7, 7) and k[0]["response"] == 254
    img = np.full((15, 15), 255, np.uint8); img[7, 7] = 0
    k = O.fast(img, 20)
    assert len(k) == 1 and k[0]["response"] == 254
    # plateau: two equal adjacent corners suppress each other (strict >)
    img = np.zeros((15, 16), np.uint8); img[7, 7] = 200; img[7, 8] = 200
    assert len(O.fast(img, 20, True)) < len(O.fast(img, 20, False))


# ---------------------------------------------------------------- blur ------------------------------------
def test_gauss_taps():
    assert O.gauss_taps().tolist() == [18, 34, 49, 55, 49, 34, 18]


@pytest.mark.parametrize("shape", [(40, 53), (7, 9), (100, 64)])
def test_blur_matches_scipy_mirror_convolution(shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    h = ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")       # mirror == BORDER_REFLECT_101
    v = ndimage.correlate1d(h, k, axis=0, mode="mirror")
    want = np.minimum((v + 32768) >> 16, 255).astype(np.uint8)
    assert np.array_equal(O.gaussian_blur7(img), want)


def test_blur_flat_values():
    for v in (0, 1, 128, 254, 255):
        out = O.gaussian_blur7(np.full((12, 12), v, np.uint8))
        assert (out == min(255, (v * 257 * 257 + 32768) >> 16)).all()


# ---------------------------------------------------------------- resize ----------------------------------
def resize_numpy(src, dw, dh):
    """Independent vectorised statement of cv::resize INTER_LINEAR 8u."""
    sh, sw = src.shape

    def axis(sn, dn, zero_at_edges):
        scale = 1.0 / (np.float64(dn) / np.float64(sn))
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if zero_at_edges:
            lo, hi = s < 0, s >= sn - 1
            f = np.where(lo | hi, np.float32(0), f); s = np.where(lo, 0, s); s = np.where(hi, sn - 1, s)
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, w0, w1

    sx, a0, a1 = axis(sw, dw, True)
    sy, b0, b1 = axis(sh, dh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0 + S[:, sx1] * a1                                              # [sh, dw]
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    out = (((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("sshape,dshape", [((480, 640), (400, 533)), ((400, 533), (333, 444)), ((134, 179), (112, 149)),
                                           ((50, 70), (50, 70)), ((31, 47), (90, 100))])
def test_resize_matches_independent_statement(sshape, dshape):
    rng = np.random.default_rng(sshape[0] + dshape[1])
    src = rng.integers(0, 256, sshape, dtype=np.uint8)
    got = O.resize_linear(src, dshape[1], dshape[0])
    assert np.array_equal(got, resize_numpy(src, dshape[1], dshape[0]))


def test_resize_properties():
    # identity at equal size; constant image stays constant; result within the local min/max (convexity)
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    assert np.array_equal(O.resize_linear(src, 80, 60), src)
    assert (O.resize_linear(np.full((60, 80), 93, np.uint8), 67, 50) == 93).all()
    out = O.resize_linear(src, 67, 50)
    assert out.min() >= src.min() and out.max() <= src.max()
    # close to a float bilinear resample (|err| <= 1 grey level from the 11-bit weights)
    fy = (np.arange(50) + 0.5) * (60 / 50) - 0.5; fx = (np.arange(67) + 0.5) * (80 / 67) - 0.5
    ref = ndimage.map_coordinates(src.astype(np.float64), np.meshgrid(fy, fx, indexing="ij"), order=1, mode="nearest")
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0


def test_pyramid_border_is_reflect101():
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    o = O.Oracle(300)
    o.extract(img)
    for l in range(8):
        inner, full = o.level(l), o.level(l, bordered=True)
        assert np.array_equal(full, np.pad(inner, 19, mode="reflect"))             # numpy 'reflect' == REFLECT_101
        if l:
            w, h = o.level_size(l)
            assert np.array_equal(inner, O.resize_linear(o.level(l - 1), w, h))    # level l from level l-1 (:1183)
    assert np.array_equal(o.level(0), img)


# ---------------------------------------------------------------- atan2 / sincos --------------------------
def test_fast_atan2_accuracy_and_quadrants():
    rng = np.random.default_rng(11)
    y = rng.integers(-2900000, 2900000, 200000).astype(np.float32)
    x = rng.integers(-2900000, 2900000, 200000).astype(np.float32)
    got = O.fast_atan2(y, x)
    want = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    err = np.abs(((got - want) + 180) % 360 - 180)
    assert err.max() < 0.02            # the 7th-order polynomial is good to ~0.01 degrees
    assert (got >= 0).all() and (got <= 360).all()
    for yy, xx, a in [(0, 0, 0), (0, 1, 0), (1, 0, 90), (0, -1, 180), (-1, 0, 270)]:
        assert abs(O.fast_atan2(np.float32([yy]), np.float32([xx]))[0] - a) < 1e-4
    d = abs(O.fast_atan2(np.float32([5]), np.float32([5]))[0] - 45)
    assert d < 0.01


def test_restated_sincos_equals_host_libm_everywhere():
    # the routine the HIP kernel evaluates (glibc's published sinf/cosf algorithm) vs the libm the reference
    # would call (ORBextractor.cc:111), over EVERY float in [2^-13, 2*pi + margin]: 130 M values, ~1-2 s
    bad, n = O.sincos_check(2.0 ** -13, 6.2832)
    assert n > 130_000_000 and bad == 0
    bad, n = O.sincos_check(0.0, 2.0 ** -13)   # tiny angles: sin x = x, cos x = 1
    assert bad == 0


# ---------------------------------------------------------------- descriptor ------------------------------
def test_descriptor_definition():
    # independent numpy statement of computeOrbDescriptor (:106-145) with float32 arithmetic and no FMA
    import re, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "orbx_brief_pattern.inc")).read(), flags=re.S)
    pat = np.array([int(t) for t in re.findall(r"-?\d+", text)], np.float32).reshape(512, 2)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    for angle in [0.0, 17.52323, 90.0, 118.222176, 270.5, 359.99]:
        ang = np.float32(angle) * np.float32(np.pi / np.float32(180))
        s, c = O.sincos_restated(ang)
        a, b = np.float32(c), np.float32(s)
        rr = np.rint(pat[:, 0] * b + pat[:, 1] * a).astype(int)
        cc = np.rint(pat[:, 0] * a - pat[:, 1] * b).astype(int)
        vals = img[32 + rr, 30 + cc].astype(int)
        bits = (vals[0::2] < vals[1::2]).astype(np.uint8)
        want = np.packbits(bits, bitorder="little")
        assert np.array_equal(O.describe(img, 30, 32, angle), want)
