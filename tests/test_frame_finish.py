"""Frame::UndistortKeyPoints / ComputeImageBounds / AssignFeaturesToGrid
(reference src/Frame.cc:748-811, 383-417, 726-736).  CPU: oracle against independent definitions and the host-only
C-ABI helper; GPU: orbx_frame_finish_device against the oracle, bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth

EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359, p2=1.76187114e-05)
TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, k1=0.262383, k2=-0.953104, p1=-0.005358, p2=0.002628, k3=1.163314)
PINHOLE = dict(fx=500.0, fy=500.0, cx=320.0, cy=240.0)


def distort(cam, x, y):
    """Forward radial-tangential model in float64 (the inverse of what undistortPoints solves)."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(v) for v in cam]
    xn, yn = (x - cx) / fx, (y - cy) / fy
    r2 = xn * xn + yn * yn
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return xd * fx + cx, yd * fy + cy


def keys(pts):
    k = np.zeros(len(pts), O.KEYPOINT_DTYPE)
    k["x"], k["y"] = pts[:, 0], pts[:, 1]
    k["size"], k["angle"], k["octave"], k["class_id"] = 31, 10, 0, -1
    return k


@pytest.mark.parametrize("cam", [EUROC, TUM1])
def test_oracle_undistortion_inverts_the_distortion_model(cam):
    c = O.camera(**cam)
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(20, 730, 500), rng.uniform(20, 460, 500)], 1).astype(np.float32)
    un, _, _ = O.frame_finish(c, keys(pts), O.image_bounds(c, 752, 480))
    xb, yb = distort(c, un["x"].astype(np.float64), un["y"].astype(np.float64))
    err = np.hypot(xb - pts[:, 0], yb - pts[:, 1])
    # five fixed-point iterations (cv::undistortPoints' default) converge to ~1e-5 px near the centre and leave up to
    # a few tenths of a pixel (EuRoC) / ~2 px (TUM1's strong k3) in the image corners
    assert np.median(err) < 1e-3 and np.percentile(err, 80) < 0.1 and err.max() < 3.0
    pp = keys(np.array([[c[2], c[3]]], np.float32))
    un, _, _ = O.frame_finish(c, pp, O.image_bounds(c, 752, 480))
    assert abs(un["x"][0] - c[2]) < 1e-4 and abs(un["y"][0] - c[3]) < 1e-4       # the principal point is a fixed point
    assert (un["octave"] == 0).all() and (un["angle"] == 10).all()                 # only pt changes (:771-776)


def test_bounds_and_host_helper():
    for cam, cols, rows in [(EUROC, 752, 480), (TUM1, 640, 480), (PINHOLE, 640, 480)]:
        c = O.camera(**cam)
        b = O.image_bounds(c, cols, rows)
        assert X.compute_image_bounds(X.camera(**cam), cols, rows).tobytes() == b.tobytes()
        if c[4] == 0:
            assert b.tolist() == [0, cols, 0, rows]
        else:
            assert b[0] != 0 and b[1] != cols      # the undistorted corners move


def test_oracle_grid_is_a_stable_bucket_sort():
    c = O.camera(**PINHOLE)
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(0, 640, 1500), rng.uniform(0, 480, 1500)], 1).astype(np.float32)
    pts[:5] = [[0, 0], [639.9, 479.9], [640, 10], [10, 480], [320, 240]]
    b = O.image_bounds(c, 640, 480)
    un, off, idx = O.frame_finish(c, keys(pts), b)
    assert un["x"].tobytes() == pts[:, 0].tobytes()             # k1 == 0: mvKeysUn = mvKeys
    px = np.floor((pts[:, 0] - b[0]) * np.float32(64.0 / 640.0) + 0.5).astype(int)
    py = np.floor((pts[:, 1] - b[2]) * np.float32(48.0 / 480.0) + 0.5).astype(int)
    ok = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    assert off[-1] == ok.sum() == len(idx) and not ok[1] and not ok[2] and not ok[3]   # round() pushes border points out
    cell = px * 48 + py
    for c_ in np.unique(cell[ok]):
        want = np.nonzero(ok & (cell == c_))[0]
        assert idx[off[c_]:off[c_ + 1]].tolist() == want.tolist()    # increasing keypoint index inside a cell
    assert (np.diff(off) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cam", [EUROC, TUM1, PINHOLE])
def test_gpu_frame_finish_equals_oracle(cam):
    import torch
    B = 3
    frames = synth.frames("textured", 60, B, 480, 640)
    ex = X.ORBextractor(1500, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, B, 480, 640, d_k, d_d, d_n, d_m, cap)
    c = X.camera(**cam)
    bounds = X.compute_image_bounds(c, 640, 480)
    d_un = torch.zeros_like(d_k); d_off = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_in = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.frame_finish_device(B, d_k, d_n, cap, c, bounds, d_un, d_off, d_idx, d_in)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    for f in range(B):
        k = d_k[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
        un_o, off_o, idx_o = O.frame_finish(O.camera(**cam), k, O.image_bounds(O.camera(**cam), 640, 480))
        un = d_un[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
        assert un.tobytes() == un_o.tobytes(), "mvKeysUn differs"
        assert d_off[f].cpu().numpy().tolist() == off_o.tolist()
        assert int(d_in[f]) == len(idx_o) and d_idx[f, :len(idx_o)].cpu().numpy().tolist() == idx_o.tolist()


def test_oracle_two_eyes_grid_against_an_independent_statement():
    """AssignFeaturesToGrid with Nleft != -1 (reference src/Frame.cc:404-414): left keys (raw, not undistorted) into mGrid with their own indices,
    right keys into mGridRight with indices i - Nleft.  Independent numpy statement: a stable bucket sort of each eye's raw positions."""
    rng = np.random.default_rng(5)
    c = O.camera(**EUROC)
    b = O.image_bounds(c, 752, 480)
    pl = np.stack([rng.uniform(-5, 760, 900), rng.uniform(-5, 490, 900)], 1).astype(np.float32)
    pr = np.stack([rng.uniform(-5, 760, 700), rng.uniform(-5, 490, 700)], 1).astype(np.float32)
    (offl, idxl), (offr, idxr) = O.assign_features_two_eyes(keys(pl), keys(pr), b)
    wInv, hInv = np.float32(64.0) / np.float32(b[1] - b[0]), np.float32(48.0) / np.float32(b[3] - b[2])
    for pts, off, idx in ((pl, offl, idxl), (pr, offr, idxr)):
        px = np.floor(((pts[:, 0] - b[0]) * wInv).astype(np.float64) + 0.5).astype(int)      # round() of a non-negative product; negatives fall out either way
        py = np.floor(((pts[:, 1] - b[2]) * hInv).astype(np.float64) + 0.5).astype(int)
        neg = ((pts[:, 0] - b[0]) * wInv < -0.5) | ((pts[:, 1] - b[2]) * hInv < -0.5)
        ok = ~neg & (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
        assert off[-1] == ok.sum() == len(idx)
        cell = px * 48 + py
        for c_ in np.unique(cell[ok]):
            assert idx[off[c_]:off[c_ + 1]].tolist() == np.nonzero(ok & (cell == c_))[0].tolist()
    # ... and it is NOT the Nleft == -1 grid of a distorted camera: that one buckets the undistorted keys
    _, off_un, _ = O.frame_finish(c, keys(pl), b)
    assert off_un.tolist() != offl.tolist()
    # with an undistorted camera the two branches agree per eye
    cp = O.camera(**PINHOLE)
    bp = O.image_bounds(cp, 640, 480)
    (o1, i1), (o2, i2) = O.assign_features_two_eyes(keys(pl), keys(pr), bp)
    for pts, off, idx in ((pl, o1, i1), (pr, o2, i2)):
        _, off_m, idx_m = O.frame_finish(cp, keys(pts), bp)
        assert off_m.tolist() == off.tolist() and idx_m.tolist() == idx.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("cam", [EUROC, TUM1, PINHOLE])
def test_gpu_two_eyes_grid_equals_oracle(cam):
    """orbx_frame_finish_two_eyes_device on extracted stereo pairs (frame 2p = left, 2p + 1 = right; 1200 features per eye, lapping {0, 0}): mGrid /
    mGridRight from the raw keys against the oracle's restatement of Frame.cc:404-414, mvKeysUn against UndistortKeyPoints - bit-exact."""
    import torch
    P = 2
    B = 2 * P
    frames = synth.frames("textured", 80, B, 480, 640)
    ex = X.ORBextractor(1200, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, B, 480, 640, d_k, d_d, d_n, d_m, cap, lapping=(0, 0))
    c = X.camera(**cam)
    bounds = X.compute_image_bounds(c, 640, 480)
    d_un = torch.zeros_like(d_k); d_off = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_in = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.frame_finish_two_eyes_device(P, d_k, d_n, cap, c, bounds, d_un, d_off, d_idx, d_in)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    view = lambda t, f: t[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
    oc = O.camera(**cam)
    ob = O.image_bounds(oc, 640, 480)
    for p in range(P):
        kl, kr = view(d_k, 2 * p), view(d_k, 2 * p + 1)
        (offl, idxl), (offr, idxr) = O.assign_features_two_eyes(kl, kr, ob)
        for f, off_o, idx_o, k in ((2 * p, offl, idxl, kl), (2 * p + 1, offr, idxr, kr)):
            assert d_off[f].cpu().numpy().tolist() == off_o.tolist(), "pair %d frame %d: grid offsets" % (p, f)
            assert int(d_in[f]) == len(idx_o) and d_idx[f, :len(idx_o)].cpu().numpy().tolist() == idx_o.tolist()
            un_o, _, _ = O.frame_finish(oc, k, ob)
            assert view(d_un, f).tobytes() == un_o.tobytes(), "mvKeysUn differs"
    with pytest.raises(X.OrbxError):
        ex.frame_finish_two_eyes_device(0, d_k, d_n, cap, c, bounds, d_un, d_off, d_idx, d_in)
