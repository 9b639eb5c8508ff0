"""ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:706-821) with
Frame::GetFeaturesInArea (src/Frame.cc:655-724).  CPU: the oracle against independent definitions; GPU:
orbx_search_for_initialization_device against the oracle, bit-exact (vnMatches12, vbPrevMatched, nmatches)."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth

PINHOLE = dict(fx=500.0, fy=500.0, cx=320.0, cy=240.0)
EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359, p2=1.76187114e-05)


def shifted_pair(f, dx, dy, rows=480, cols=640, noise=0):
    """Two views of one textured scene: the second is the first translated by (dx, dy), plus optional noise."""
    big = synth.textured_frame(f, rows + 64, cols + 64)
    a = big[32:32 + rows, 32:32 + cols]
    b = big[32 - dy:32 - dy + rows, 32 - dx:32 - dx + cols].astype(np.int32)
    if noise:
        rng = np.random.default_rng(f)
        b = b + rng.integers(-noise, noise + 1, b.shape)
    return np.ascontiguousarray(a), np.clip(b, 0, 255).astype(np.uint8)


def random_frame(rng, n, level0_frac=0.3, cols=640, rows=480):
    k = np.zeros(n, O.KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(0, cols, n).astype(np.float32), rng.uniform(0, rows, n).astype(np.float32)
    k["octave"] = np.where(rng.random(n) < level0_frac, 0, rng.integers(1, 8, n))
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["size"], k["class_id"] = 31, -1
    return k


def test_oracle_features_in_area_is_a_box_query_in_grid_order():
    rng = np.random.default_rng(5)
    k = random_frame(rng, 1200)
    c = O.camera(**PINHOLE)
    b = O.image_bounds(c, 640, 480)
    un, off, idx = O.frame_finish(c, k, b)
    pos_in_grid = {int(i): p for p, i in enumerate(idx)}
    for x, y, r, lo, hi in [(320, 240, 100, 0, 0), (5, 5, 60, -1, -1), (639, 470, 30, 2, 4), (100, 400, 15, 0, 7), (-50, 240, 100, 0, 0),
                            (800, 240, 100, 0, 0), (320, 700, 100, -1, -1), (320, 240, 1000, 1, -1)]:
        got = O.features_in_area(un, off, idx, b, x, y, r, lo, hi)
        check = (lo > 0) or (hi >= 0)
        want = [i for i in range(len(un)) if i in pos_in_grid and abs(un["x"][i] - np.float32(x)) < r and abs(un["y"][i] - np.float32(y)) < r
                and (not check or (un["octave"][i] >= lo and (hi < 0 or un["octave"][i] <= hi)))]
        # a keypoint that passes the box test lies in a visited cell, so the query is exactly the box; the
        # traversal order is ascending grid position (cells x-major, push_back order inside)
        assert sorted(got.tolist()) == want
        assert [pos_in_grid[int(i)] for i in got] == sorted(pos_in_grid[int(i)] for i in got)


def brute_force_search(k1, d1, k2, d2, inside2, grid_pos2, prev, window, nnratio, check):
    """Independent restatement on plain Python containers: candidates by brute-force box test over the keypoints
    that are in the grid, visited in grid order."""
    n1 = len(k1)
    m12 = [-1] * n1; m21 = {}; mdist = {}
    hist = [[] for _ in range(30)]
    cand_all = sorted((i for i in range(len(k2)) if inside2[i] and k2["octave"][i] == 0), key=lambda i: grid_pos2[i])
    pc = np.unpackbits(d2, axis=1)
    nm = 0
    for i1 in range(n1):
        if k1["octave"][i1] > 0:
            continue
        x, y = np.float32(prev[i1][0]), np.float32(prev[i1][1])
        best, best2, bi = 1 << 30, 1 << 30, -1
        b1 = np.unpackbits(d1[i1])
        for i2 in cand_all:
            if not (abs(k2["x"][i2] - x) < window and abs(k2["y"][i2] - y) < window):
                continue
            dist = int((b1 != pc[i2]).sum())
            if mdist.get(i2, 1 << 30) <= dist:
                continue
            if dist < best:
                best2, best, bi = best, dist, i2
            elif dist < best2:
                best2 = dist
        if best <= 50 and best < np.float32(best2 if best2 < (1 << 30) else 2 ** 31) * np.float32(nnratio):
            if bi in m21:
                m12[m21[bi]] = -1; nm -= 1
            m12[i1] = bi; m21[bi] = i1; mdist[bi] = best; nm += 1
            if check:
                rot = np.float32(k1["angle"][i1]) - np.float32(k2["angle"][bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360))
                v = float(np.float32(rot * np.float32(1.0 / 30)))
                b = int(np.floor(v + 0.5))
                hist[0 if b == 30 else b].append(i1)
    if check:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-sizes[i], i))
        top = [order[0] if sizes[order[0]] > 0 else -1]
        mx = sizes[order[0]]
        for o in order[1:3]:
            top.append(o if sizes[o] > 0 and not sizes[o] < np.float32(0.1) * np.float32(mx) else -1)
        if top[1] == -1:
            top[2] = -1
        for b in range(30):
            if b in top:
                continue
            for i1 in hist[b]:
                if m12[i1] >= 0:
                    m12[i1] = -1; nm -= 1
    prev = np.array(prev, np.float32).copy()
    for i1 in range(n1):
        if m12[i1] >= 0:
            prev[i1] = (k2["x"][m12[i1]], k2["y"][m12[i1]])
    return nm, m12, prev


def make_descriptor_pair(rng, k1, n2, flip_bits=12):
    """Frame 2 = a permuted, jittered copy of frame 1 with a few descriptor bits flipped, plus distractors."""
    n1 = len(k1)
    k2 = random_frame(rng, n2)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    perm = rng.permutation(min(n1, n2))
    for j, i in enumerate(perm[: int(0.7 * len(perm))]):
        k2[j] = k1[i]
        k2["x"][j] += np.float32(rng.uniform(-20, 20)); k2["y"][j] += np.float32(rng.uniform(-20, 20))
        k2["angle"][j] = np.float32((k1["angle"][i] + rng.normal(0, 8)) % 360)
        bits = np.unpackbits(d1[i]); flip = rng.choice(256, rng.integers(0, flip_bits + 1), replace=False); bits[flip] ^= 1
        d2[j] = np.packbits(bits)
    return d1, k2, d2


@pytest.mark.parametrize("seed,check,nnratio,window", [(1, True, 0.9, 100), (2, False, 0.9, 100), (3, True, 0.6, 40), (4, True, 1.5, 300)])
def test_oracle_search_equals_the_brute_force_definition(seed, check, nnratio, window):
    rng = np.random.default_rng(seed)
    k1 = random_frame(rng, 400, level0_frac=0.5)
    d1, k2, d2 = make_descriptor_pair(rng, k1, 450)
    d2[5] = d2[6] = d2[7]                                     # exact descriptor ties between neighbours
    k2["x"][5:8] = k2["x"][7]; k2["y"][5:8] = k2["y"][7]; k2["octave"][5:8] = 0
    c = O.camera(**PINHOLE)
    b = O.image_bounds(c, 640, 480)
    un2, off2, idx2 = O.frame_finish(c, k2, b)
    inside2 = np.zeros(len(k2), bool); inside2[idx2] = True
    grid_pos2 = np.zeros(len(k2), int); grid_pos2[idx2] = np.arange(len(idx2))
    prev = np.stack([k1["x"], k1["y"]], 1)
    n, m12, prev_o = O.search_for_initialization(k1, d1, un2, d2, off2, idx2, b, prev, window, nnratio, check)
    n_b, m12_b, prev_b = brute_force_search(k1, d1, un2, d2, inside2, grid_pos2, prev, window, nnratio, check)
    assert m12.tolist() == m12_b and n == n_b == sum(m >= 0 for m in m12_b)
    assert prev_o.tobytes() == prev_b.tobytes()
    assert n > 40 or nnratio < 0.9
    assert (k1["octave"][m12 >= 0] == 0).all() and (un2["octave"][m12[m12 >= 0]] == 0).all()    # level 0 only (:722-726)
    assert len(set(m12[m12 >= 0].tolist())) == (m12 >= 0).sum()                                  # one-to-one (:763-769)


def test_oracle_search_second_call_tracks_the_previous_matches():
    """vbPrevMatched is in/out: matched entries move to the frame-2 position (:815-817), unmatched ones stay."""
    rng = np.random.default_rng(9)
    k1 = random_frame(rng, 300, level0_frac=0.6)
    d1, k2, d2 = make_descriptor_pair(rng, k1, 300)
    c = O.camera(**PINHOLE); b = O.image_bounds(c, 640, 480)
    un2, off2, idx2 = O.frame_finish(c, k2, b)
    prev0 = np.stack([k1["x"], k1["y"]], 1)
    n, m12, prev1 = O.search_for_initialization(k1, d1, un2, d2, off2, idx2, b, prev0)
    moved = m12 >= 0
    assert np.array_equal(prev1[~moved], prev0[~moved])
    assert np.array_equal(prev1[moved, 0], un2["x"][m12[moved]]) and np.array_equal(prev1[moved, 1], un2["y"][m12[moved]])
    n2, m12b, _ = O.search_for_initialization(k1, d1, un2, d2, off2, idx2, b, prev1, window=25)
    assert (m12b[moved] == m12[moved]).mean() > 0.9          # a narrow window around the tracked positions finds them again


# ---------------------------------------------------------------------------------------------------------------
def run_gpu_pairs(frames, pairs_spec, cam, nfeatures=1000, window=100, nnratio=0.9, check=True, rounds=1):
    import torch
    B = len(frames)
    rows, cols = frames.shape[1:]
    ex = X.ORBextractor(nfeatures, max_batch=B, max_width=cols, max_height=rows)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, B, rows, cols, d_k, d_d, d_n, d_m, cap)
    c = X.camera(**cam)
    bounds = X.compute_image_bounds(c, cols, rows)
    d_un = torch.zeros_like(d_k); d_off = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_in = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.frame_finish_device(B, d_k, d_n, cap, c, bounds, d_un, d_off, d_idx, d_in)
    (f1, s1), (f2, s2), P = pairs_spec
    pair_f1 = torch.tensor([f1 + p * s1 for p in range(P)], device="cuda")
    d_prev = d_un[pair_f1][:, :, :2].contiguous()            # Tracking.cc:2029-2031: vbPrevMatched = F1.mvKeysUn[i].pt
    prev0 = d_prev.cpu().numpy().copy()
    d_m12 = torch.full((P, cap), -7, dtype=torch.int32, device="cuda"); d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    outs = []
    for _ in range(rounds):
        ex.search_for_initialization_device(P, (f1, s1), (f2, s2), d_un, d_d, d_n, cap, d_off, d_idx, bounds, d_prev, d_m12, d_nm,
                                            window, nnratio, check)
        torch.cuda.synchronize()
        outs.append((d_m12.cpu().numpy().copy(), d_nm.cpu().numpy().copy(), d_prev.cpu().numpy().copy()))
    n = d_n.cpu().numpy()
    view = lambda t, f: t[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
    un = [view(d_un, f) for f in range(B)]
    desc = [d_d[f, :n[f]].cpu().numpy() for f in range(B)]
    off = d_off.cpu().numpy(); idx = d_idx.cpu().numpy(); nin = d_in.cpu().numpy()
    total = 0
    for p in range(P):
        a, b = f1 + p * s1, f2 + p * s2
        prev = prev0[p, :n[a]]
        for r in range(rounds):
            nm_o, m12_o, prev = O.search_for_initialization(un[a], desc[a], un[b], desc[b], off[b], idx[b, :nin[b]], bounds, prev,
                                                            window, nnratio, check)
            m12, nm, prv = outs[r]
            assert m12[p, :n[a]].tolist() == m12_o.tolist(), "vnMatches12 differs (pair %d, round %d)" % (p, r)
            assert int(nm[p]) == nm_o
            assert prv[p, :n[a]].tobytes() == prev.tobytes(), "vbPrevMatched differs"
            assert (m12[p, n[a]:] == -7).all()                 # nothing written past N1
            total += nm_o
    return total, P * rounds


@pytest.mark.gpu
@pytest.mark.parametrize("cam", [PINHOLE, EUROC])
def test_gpu_search_equals_oracle_on_shifted_pairs(cam):
    fr = []
    for f, (dx, dy, noise) in enumerate([(7, 3, 0), (-12, 5, 4), (25, -20, 8), (0, 0, 0)]):
        a, b = shifted_pair(300 + f, dx, dy, noise=noise)
        fr += [a, b]
    total, calls = run_gpu_pairs(np.stack(fr), ((0, 2), (1, 2), 4), cam)
    assert total / calls > 60                                # the shifted views really match


@pytest.mark.gpu
def test_gpu_search_one_initial_frame_against_a_stream_and_repeated_calls():
    base = synth.textured_frame(77, 480 + 64, 640 + 64)
    fr = [np.ascontiguousarray(base[32 + s:32 + s + 480, 32 + 2 * s:32 + 2 * s + 640]) for s in range(5)]
    total, calls = run_gpu_pairs(np.stack(fr), ((0, 0), (1, 1), 4), PINHOLE, rounds=2)     # F1 fixed, F2 = frames 1..4; prev carried over
    assert total / calls > 40


@pytest.mark.gpu
@pytest.mark.parametrize("variant,nfeatures,window,nnratio,check", [("textured", 1000, 100, 0.9, True), ("noise", 500, 40, 0.9, False),
                                                                     ("sparse", 1000, 100, 0.9, True), ("natural", 2000, 300, 1.2, True)])
def test_gpu_search_equals_oracle_on_stream_pairs(variant, nfeatures, window, nnratio, check):
    frames = synth.frames(variant, 10, 6, 480, 640)
    run_gpu_pairs(frames, ((0, 1), (1, 1), 5), PINHOLE, nfeatures, window, nnratio, check)


@pytest.mark.gpu
@pytest.mark.parametrize("nfeatures", [5000, 10000])
def test_gpu_search_on_initialisation_extractor_frames(nfeatures):
    """Tracking builds the two initialisation frames with mpIniORBextractor = ORBextractor(5 * nFeatures) (reference src/Tracking.cc:774,
    :2065): thousands of keypoints per frame, of which only the level-0 fifth are candidates."""
    fr = []
    for f, (dx, dy, noise) in enumerate([(9, 4, 0), (-15, 6, 3)]):
        a, b = shifted_pair(500 + f, dx, dy, noise=noise)
        fr += [a, b]
    total, calls = run_gpu_pairs(np.stack(fr), ((0, 2), (1, 2), 2), PINHOLE, nfeatures=nfeatures)
    assert total / calls > 150


@pytest.mark.gpu
def test_gpu_search_argument_errors():
    ex = X.ORBextractor(1000)
    with pytest.raises(X.OrbxError):
        ex.search_for_initialization_device(0, (0, 1), (1, 1), 1, 1, 1, ex.capacity, 1, 1, [0, 640, 0, 480], 1, 1, 1)
    with pytest.raises(X.OrbxError):
        ex.search_for_initialization_device(1, (0, 1), (1, 1), 1, 1, 1, 40000, 1, 1, [0, 640, 0, 480], 1, 1, 1)
