"""SURVEY.md §8f-1, the first "next" row: Frame::ComputeStereoMatches (reference src/Frame.cc:813-991).
CPU: the oracle restatement against known answers.  GPU: liborbx's orbx_stereo_match_* against the oracle, bit-exact
(floats included)."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth


def stereo_pair(disparity, seed=77, rows=480, cols=640, variant="textured", dy=0, noise=0):
    """Left / right views cut from one wider scene.  dy: the right view sits dy rows lower (imperfect rectification: the
    match of a left keypoint is then found through the +-2*scale row band of Frame.cc:826-840, not on its own row);
    noise: independent per-eye sensor noise of that amplitude, so the two eyes never hold identical pixels."""
    big = synth.VARIANTS[variant](seed, rows + 8, cols + 2 * 40)
    left = np.ascontiguousarray(big[4:4 + rows, 40:40 + cols])
    right = np.ascontiguousarray(big[4 + dy:4 + dy + rows, 40 + disparity:40 + disparity + cols])   # the right camera sees the scene shifted left
    if noise:
        rng = np.random.default_rng(seed * 7 + 1)
        left = np.clip(left.astype(np.int16) + rng.integers(-noise, noise + 1, left.shape), 0, 255).astype(np.uint8)
        right = np.clip(right.astype(np.int16) + rng.integers(-noise, noise + 1, right.shape), 0, 255).astype(np.uint8)
    return left, right


@pytest.mark.parametrize("dy", [-2, -1, 1, 2])
def test_oracle_matches_through_the_row_band(dy):
    # a vertical offset of 1-2 px between the eyes plus per-eye noise: matches must come from neighbouring rows of the band
    left, right = stereo_pair(15, dy=dy, noise=4)
    kL, u, d, kept = oracle_stereo(left, right)
    ok = u >= 0
    assert kept == ok.sum() and kept > 60
    est = kL["x"][ok] - u[ok]
    assert abs(np.median(est) - 15) < 0.3


def oracle_stereo(left, right, nf=1200, bf=40.0, b=0.1):
    oL, oR = O.Oracle(nf), O.Oracle(nf)
    _, kL, dL = oL.extract(left, (0, 0))
    _, kR, dR = oR.extract(right, (0, 0))
    u, d, kept = O.stereo_match(oL, oR, kL, dL, kR, dR, bf, b)
    return kL, u, d, kept


def test_descriptor_distance_is_hamming():
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
        assert O.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert O.descriptor_distance(a, a) == 0 and O.descriptor_distance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


@pytest.mark.parametrize("disp", [4, 12, 37])
def test_oracle_recovers_known_disparity(disp):
    left, right = stereo_pair(disp)
    kL, u, d, kept = oracle_stereo(left, right, bf=40.0, b=0.1)        # maxD = bf/b = 400 px
    ok = u >= 0
    assert kept == ok.sum() and kept > 150
    est = kL["x"][ok] - u[ok]
    assert abs(np.median(est) - disp) < 0.05 and np.abs(est - disp).max() < 3.6      # within one pixel of the coarsest level
    assert np.allclose(d[ok], np.float32(40.0) / est, rtol=1e-6)       # depth = bf / disparity
    assert (d[~ok] == -1).all() and (u[~ok] == -1).all()


def test_oracle_respects_disparity_range():
    left, right = stereo_pair(37)
    _, u, _, kept = oracle_stereo(left, right, bf=2.0, b=0.1)          # maxD = 20 px < true disparity
    assert kept < 40 or (u >= 0).sum() < 40                            # essentially nothing can match
    # unrelated images: few descriptor matches survive the 75-bit threshold and the SAD/median filters
    a = synth.frames("textured", 1, 1, 480, 640)[0]; b2 = synth.frames("textured", 2, 1, 480, 640)[0]
    _, u, _, kept = oracle_stereo(a, b2)
    assert kept < 60


@pytest.mark.gpu
@pytest.mark.parametrize("disp,variant,bf,b", [(12, "textured", 40.0, 0.1), (4, "noise", 40.0, 0.1), (37, "textured", 60.0, 0.5),
                                               (0, "textured", 40.0, 0.1), (25, "sparse", 40.0, 0.1)])
@pytest.mark.parametrize("dy,noise", [(0, 0), (1, 3), (-2, 5), (2, 0)])
def test_gpu_stereo_match_equals_oracle(disp, variant, bf, b, dy, noise):
    if (dy, noise) != (0, 0) and variant == "sparse":
        pytest.skip("offset cases run on the dense variants")
    left, right = stereo_pair(disp, variant=variant, dy=dy, noise=noise)
    kL, u_o, d_o, kept_o = oracle_stereo(left, right, bf=bf, b=b)
    ex = X.ORBextractor(1200, max_batch=2)
    res = ex.extract_batch(np.stack([left, right]), lapping=(0, 0))
    assert res[0][1].tobytes() == kL.tobytes()
    u, d, nm = ex.stereo_match_last(1, bf, b)
    n = len(kL)
    assert nm[0] == kept_o
    assert u[0, :n].tobytes() == u_o.tobytes(), "uRight differs: %d entries" % (u[0, :n] != u_o).sum()
    assert d[0, :n].tobytes() == d_o.tobytes()


@pytest.mark.gpu
def test_gpu_stereo_batch_of_pairs_device_path():
    import torch
    P = 3
    pairs = [stereo_pair(8 + 5 * p, seed=100 + p) for p in range(P)]
    frames = np.stack([img for pr in pairs for img in pr])
    ex = X.ORBextractor(1200, max_batch=2 * P)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros((2 * P, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((2 * P, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(2 * P, dtype=torch.int32, device="cuda"); d_m = torch.zeros(2 * P, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, 2 * P, 480, 640, d_k, d_d, d_n, d_m, cap, lapping=(0, 0))
    d_u = torch.zeros((P, cap), dtype=torch.float32, device="cuda"); d_z = torch.zeros((P, cap), dtype=torch.float32, device="cuda")
    d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    ex.stereo_match_device(P, d_k, d_d, d_n, cap, 40.0, 0.1, d_u, d_z, d_nm)
    torch.cuda.synchronize()
    for p in range(P):
        kL, u_o, d_o, kept_o = oracle_stereo(*pairs[p])
        n = len(kL)
        assert int(d_nm[p]) == kept_o
        assert d_u[p, :n].cpu().numpy().tobytes() == u_o.tobytes()
        assert d_z[p, :n].cpu().numpy().tobytes() == d_o.tobytes()


@pytest.mark.gpu
def test_gpu_stereo_argument_errors():
    left, right = stereo_pair(10)
    ex = X.ORBextractor(1200, max_batch=2)
    with pytest.raises(X.OrbxError):
        ex.stereo_match_last(1, 40.0, 0.1)                 # nothing extracted yet
    ex.extract_batch(np.stack([left, right]), lapping=(0, 0))
    with pytest.raises(X.OrbxError):
        ex.stereo_match_last(2, 40.0, 0.1)                 # only one pair in the last batch
    with pytest.raises(X.OrbxError):
        ex.stereo_match_last(1, 40.0, 0.0)                 # b must be positive (maxD = bf/b)
    u, d, nm = ex.stereo_match_last(1, 40.0, 0.1)
    assert nm[0] > 100
    # after the device-resident path the results live in the caller's buffers, not in the handle: _last must refuse
    # (it used to read a stale keypoint count out of the handle's staging memory) and _device is the call to use
    import torch
    cap = ex.capacity
    d_img = torch.from_numpy(np.stack([left, right])).cuda()
    d_k = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(2, dtype=torch.int32, device="cuda"); d_m = torch.zeros(2, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, 2, 480, 640, d_k, d_d, d_n, d_m, cap, lapping=(0, 0))
    ex.synchronize()
    with pytest.raises(X.OrbxError):
        ex.stereo_match_last(1, 40.0, 0.1)
    ex.extract_batch(np.stack([left, right]), lapping=(0, 0))          # the host path makes it valid again
    u2, d2, nm2 = ex.stereo_match_last(1, 40.0, 0.1)
    assert nm2[0] == nm[0] and u2.tobytes() == u.tobytes()
