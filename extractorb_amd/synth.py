"""Deterministic synthetic frames for tests and bench.py.

``pix(f, y, x) = hash32(seed, f, y*W + x) >> 24`` with a counter-based hash (splitmix64 finaliser), so any
rank can generate exactly its own frames of the stream with no communication.  Three variants:
  noise     iid uniform u8 (dense corners at every threshold: the quad-tree always hits its quota)
  textured  the noise 5x5 box-filtered then contrast-stretched x4 about 128 (natural-image-like density)
  sparse    flat 128 with ~1 % of 9x9 bright squares (exercises the minThFAST retry and the
            "fewer candidates than quota" quad-tree exits)
"""
import numpy as np

SEED = 20261003
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash32(seed, f, idx):
    """32-bit hash of (seed, frame, linear pixel index); idx may be an array."""
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(f) * np.uint64(0xD1B54A32D192ED03)) & _M64
        return (_splitmix64((key + np.asarray(idx, np.uint64)) & _M64) >> np.uint64(32)).astype(np.uint32)


def noise_frame(f, rows, cols, seed=SEED):
    idx = np.arange(rows * cols, dtype=np.uint64)
    return (hash32(seed, f, idx) >> np.uint32(24)).astype(np.uint8).reshape(rows, cols)


def textured_frame(f, rows, cols, seed=SEED):
    n = noise_frame(f, rows, cols, seed).astype(np.int32)
    p = np.pad(n, 2, mode="reflect")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    box = c[5:, 5:] - c[:-5, 5:] - c[5:, :-5] + c[:-5, :-5]          # 5x5 box sums
    v = (box * 4 - 128 * 25 * 4) // 25 + 128                          # stretch x4 about 128
    return np.clip(v, 0, 255).astype(np.uint8)


def sparse_frame(f, rows, cols, seed=SEED):
    img = np.full((rows, cols), 128, np.uint8)
    n_sq = max(1, rows * cols // (100 * 81))
    h = hash32(seed ^ 0x5bd1e995, f, np.arange(3 * n_sq, dtype=np.uint64))
    ys = (h[0::3] % np.uint32(rows - 9)).astype(np.int64)
    xs = (h[1::3] % np.uint32(cols - 9)).astype(np.int64)
    vs = (160 + (h[2::3] % np.uint32(96))).astype(np.uint8)
    for y, x, v in zip(ys, xs, vs):
        img[y:y + 9, x:x + 9] = v
    return img


VARIANTS = {"noise": noise_frame, "textured": textured_frame, "sparse": sparse_frame}


def frames(variant, first, count, rows, cols, seed=SEED):
    """uint8 [count, rows, cols]: frames first .. first+count-1 of the stream."""
    gen = VARIANTS[variant]
    return np.stack([gen(first + i, rows, cols, seed) for i in range(count)])


def fixture_frame(f, rows, cols, seed=SEED):
    """A natural image for bench/tests: the reference's own 640x480 demo frame (tests/golden/robot_865_gray.png),
    cropped/tiled to the requested size and shifted by f pixels so consecutive frames differ."""
    import os
    from PIL import Image
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "robot_865_gray.png")
    img = np.array(Image.open(path))
    reps = (-(-rows // img.shape[0]), -(-(cols + 64) // img.shape[1]))
    big = np.tile(img, reps)
    s = int(f) % 64
    return np.ascontiguousarray(big[:rows, s:s + cols])


VARIANTS["natural"] = fixture_frame
