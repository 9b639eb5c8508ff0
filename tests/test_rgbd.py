"""RGB-D frames: Frame::ComputeStereoFromRGBD (reference src/Frame.cc:994-1015) with the depth conversion of
Tracking::GrabImageRGBD (src/Tracking.cc:1003-1004).  CPU: oracle against a direct numpy statement; GPU:
orbx_stereo_from_rgbd_device against the oracle, bit-exact, chained after extraction + undistortion."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth

TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, k1=0.262383, k2=-0.953104, p1=-0.005358, p2=0.002628, k3=1.163314)


def keys(rng, n, cols=640, rows=480):
    k = np.zeros(n, O.KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(0, cols - 0.01, n).astype(np.float32), rng.uniform(0, rows - 0.01, n).astype(np.float32)
    k["size"], k["class_id"] = 31, -1
    return k


@pytest.mark.parametrize("dtype,factor", [(np.float32, 1.0), (np.float32, 1.0 / 5000.0), (np.uint16, 1.0 / 5000.0), (np.uint16, 1.0)])
def test_oracle_rgbd_is_a_gather_with_one_rounded_scale(dtype, factor):
    rng = np.random.default_rng(3)
    k = keys(rng, 700)
    un = k.copy(); un["x"] += rng.normal(0, 0.5, len(k)).astype(np.float32)
    depth = rng.uniform(0, 4, (480, 640)).astype(np.float32) if dtype == np.float32 else rng.integers(0, 30000, (480, 640)).astype(np.uint16)
    depth[rng.random((480, 640)) < 0.2] = 0                          # holes: no depth
    mbf = np.float32(40.0)
    u, d = O.stereo_from_rgbd(k, un, depth, factor, mbf)
    raw = depth[k["y"].astype(np.int32), k["x"].astype(np.int32)].astype(np.float32)
    scaled = raw * np.float32(factor) if (dtype == np.uint16 or abs(np.float32(factor) - np.float32(1)) > 1e-5) else raw
    ok = scaled > 0
    assert np.array_equal(d[ok], scaled[ok]) and (d[~ok] == -1).all() and (u[~ok] == -1).all()
    assert np.array_equal(u[ok], (un["x"][ok] - mbf / scaled[ok]).astype(np.float32))
    assert ok.sum() > 400 and (~ok).sum() > 50


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,factor,pad", [(np.float32, 1.0, 0), (np.uint16, 1.0 / 5000.0, 0), (np.float32, 0.5, 12), (np.uint16, 1.0, 6)])
def test_gpu_rgbd_equals_oracle_after_extraction_and_undistortion(dtype, factor, pad):
    import torch
    B, rows, cols = 3, 480, 640
    frames = synth.frames("textured", 90, B, rows, cols)
    ex = X.ORBextractor(1000, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, B, rows, cols, d_k, d_d, d_n, d_m, cap)
    c = X.camera(**TUM1)
    bounds = X.compute_image_bounds(c, cols, rows)
    d_un = torch.zeros_like(d_k); d_off = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_in = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.frame_finish_device(B, d_k, d_n, cap, c, bounds, d_un, d_off, d_idx, d_in)
    rng = np.random.default_rng(11)
    elem = np.dtype(dtype).itemsize
    stride = cols * elem + pad * elem
    depth = np.zeros((B, rows, stride // elem), dtype)
    depth[:, :, :cols] = rng.uniform(0.3, 6, (B, rows, cols)).astype(np.float32) if dtype == np.float32 else rng.integers(1, 40000, (B, rows, cols))
    depth[:, :, :cols][rng.random((B, rows, cols)) < 0.25] = 0
    d_depth = torch.from_numpy(depth.view(np.int16) if dtype == np.uint16 else depth).cuda()
    d_u = torch.full((B, cap), 7.0, dtype=torch.float32, device="cuda"); d_z = torch.full((B, cap), 7.0, dtype=torch.float32, device="cuda")
    ex.stereo_from_rgbd_device(B, d_k, d_un, d_n, cap, d_depth, dtype == np.uint16, rows, cols, factor, 40.0, d_u, d_z, depth_stride_bytes=stride)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    view = lambda t, f: t[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
    for f in range(B):
        u_o, z_o = O.stereo_from_rgbd(view(d_k, f), view(d_un, f), depth[f, :, :cols], factor, 40.0)
        assert d_u[f, :n[f]].cpu().numpy().tobytes() == u_o.tobytes() and d_z[f, :n[f]].cpu().numpy().tobytes() == z_o.tobytes(), "frame %d" % f
        assert (d_u[f, n[f]:] == -1).all() and (d_z[f, n[f]:] == -1).all()          # unused slots are "no depth"
        assert (z_o > 0).sum() > 500 and (z_o < 0).sum() > 100


@pytest.mark.gpu
def test_gpu_rgbd_argument_errors():
    ex = X.ORBextractor(300)
    with pytest.raises(X.OrbxError):
        ex.stereo_from_rgbd_device(1, 1, 1, 1, ex.capacity, 1, True, 480, 640, 1.0, 40.0, 1, 1, depth_stride_bytes=640)       # shorter than a row
    with pytest.raises(X.OrbxError):
        ex.stereo_from_rgbd_device(1, 1, 1, 1, ex.capacity, 2, False, 480, 640, 1.0, 40.0, 1, 1)                             # misaligned float map
