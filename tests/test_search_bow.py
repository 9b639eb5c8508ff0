"""The consumer of ComputeBoW's FeatureVector: ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches)
(reference src/ORBmatcher.cc:269-471; Tracking::TrackReferenceKeyFrame, Relocalization), Nleft == -1.
CPU: the oracle (the reference's loop over two std::maps) against an independent per-node numpy statement; GPU:
orbx_search_by_bow_device against the oracle on synthetic frames (planted correspondences, ties, closed keypoints, empty nodes) and on
frames extracted + ComputeBoW'ed by the HIP path."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth
from test_bow import make_vocab


def synthetic_pair(rng, n_kf=900, n_f=1000, n_nodes=60, tie_heavy=False):
    """Two frames with planted correspondences: descriptors of the frame are noisy copies of keyframe descriptors, both assigned to
    the same node most of the time; FeatureVectors in (node, index) order with some features left out (stopped words)."""
    dk = rng.integers(0, 256, (n_kf, 32), dtype=np.uint8)
    src = rng.integers(0, n_kf, n_f)
    df = dk[src].copy()
    nflip = rng.integers(0, 3 if tie_heavy else 40, n_f)
    for i in range(n_f):
        for b in rng.integers(0, 256, nflip[i]):
            df[i, b >> 3] ^= np.uint8(1 << (b & 7))
    if tie_heavy:      # many identical descriptors inside a node: best == second best, first position must win
        df[: n_f // 2] = df[rng.integers(0, 8, n_f // 2)]
    node_k = rng.integers(1, n_nodes + 1, n_kf).astype(np.uint32) * 7
    node_f = node_k[src].copy()
    wrong = rng.random(n_f) < 0.15
    node_f[wrong] = rng.integers(1, n_nodes + 40, wrong.sum()).astype(np.uint32) * 7      # other (sometimes keyframe-less) nodes
    keep_k = rng.random(n_kf) > 0.05; keep_f = rng.random(n_f) > 0.05                   # features of stopped words are in no node

    def fv(node, keep):
        idx = np.nonzero(keep)[0]
        order = np.lexsort((idx, node[idx]))
        return node[idx][order].astype(np.uint32), idx[order].astype(np.uint32)
    kps_k = np.zeros(n_kf, X.KEYPOINT_DTYPE); kps_f = np.zeros(n_f, X.KEYPOINT_DTYPE)
    kps_k["angle"] = rng.uniform(0, 360, n_kf).astype(np.float32)
    kps_f["angle"] = np.where(rng.random(n_f) < 0.8, np.mod(kps_k["angle"][src] + rng.normal(12, 4, n_f), 360), rng.uniform(0, 360, n_f)).astype(np.float32)
    flags = (rng.random(n_kf) < 0.8).astype(np.uint8) | (rng.integers(0, 2, n_kf).astype(np.uint8) << 1)      # bit 1 is noise
    return dict(dk=dk, df=df, fv_k=fv(node_k, keep_k), fv_f=fv(node_f, keep_f), kps_k=kps_k, kps_f=kps_f, flags=flags)


def brute(p, nnratio, th_low, check):
    """Independent statement: nodes are independent; inside a node the keyframe features go in list order."""
    (kn, ki), (fn, fi) = p["fv_k"], p["fv_f"]
    bits_k = np.unpackbits(p["dk"], axis=1).astype(np.int16); bits_f = np.unpackbits(p["df"], axis=1).astype(np.int16)
    matches = np.full(len(p["df"]), -1, np.int64)
    bins = {}
    for node in sorted(set(kn.tolist()) & set(fn.tolist())):
        fs = fi[fn == node]
        for k in ki[kn == node]:
            if not p["flags"][k] & 1:
                continue
            free = [int(f) for f in fs if matches[f] < 0]
            if not free:
                continue
            dist = np.abs(bits_f[free] - bits_k[k]).sum(1)
            order = np.argsort(dist, kind="stable")
            b1 = int(dist[order[0]]); b2 = int(dist[order[1]]) if len(free) > 1 else 256
            if b1 <= th_low and np.float32(b1) < np.float32(nnratio) * np.float32(b2):
                f = free[int(order[0])]
                matches[f] = k
                rot = np.float32(p["kps_k"]["angle"][k]) - np.float32(p["kps_f"]["angle"][f])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                x = float(np.float32(rot * np.float32(1.0 / 30)))
                bins[f] = int(np.floor(x + 0.5)) % 30
    if check:
        cnt = [sum(1 for b in bins.values() if b == i) for i in range(30)]
        m1 = m2 = m3 = 0; i1 = i2 = i3 = -1
        for i, s in enumerate(cnt):
            if s > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
            elif s > m2:
                m3, m2, i3, i2 = m2, s, i2, i
            elif s > m3:
                m3, i3 = s, i
        if m2 < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for f, b in bins.items():
            if b not in (i1, i2, i3):
                matches[f] = -1
    return int((matches >= 0).sum()), matches


CASES = [dict(seed=1), dict(seed=2, nnratio=0.9, check=False), dict(seed=3, tie_heavy=True), dict(seed=4, n_nodes=3, n_kf=300, n_f=500),
         dict(seed=5, n_nodes=2000), dict(seed=6, th_low=100, nnratio=0.6)]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_oracle_search_by_bow_equals_independent_statement(case):
    rng = np.random.default_rng(case["seed"])
    p = synthetic_pair(rng, case.get("n_kf", 900), case.get("n_f", 1000), case.get("n_nodes", 60), case.get("tie_heavy", False))
    nn, th, chk = case.get("nnratio", 0.7), case.get("th_low", 50), case.get("check", True)
    n, m = O.search_by_bow(p["fv_k"], p["fv_f"], p["flags"], p["kps_k"], p["dk"], p["kps_f"], p["df"], nn, th, chk)
    bn, bm = brute(p, nn, th, chk)
    assert n == bn and m.tolist() == bm.tolist()
    if case.get("n_nodes", 60) <= 60:
        assert n > 100
    assert n == int((m >= 0).sum())
    # a frame keypoint receives at most one MapPoint, a keyframe MapPoint goes to at most one keypoint of its node
    got = m[m >= 0]
    assert len(set(got.tolist())) == len(got)


def test_oracle_search_by_bow_degenerate_inputs():
    rng = np.random.default_rng(9)
    p = synthetic_pair(rng, 50, 60, 5)
    e = (np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    assert O.search_by_bow(e, p["fv_f"], p["flags"], p["kps_k"], p["dk"], p["kps_f"], p["df"])[0] == 0
    assert O.search_by_bow(p["fv_k"], e, p["flags"], p["kps_k"], p["dk"], p["kps_f"], p["df"])[0] == 0
    n, m = O.search_by_bow(p["fv_k"], p["fv_f"], np.zeros(50, np.uint8), p["kps_k"], p["dk"], p["kps_f"], p["df"])
    assert n == 0 and (m == -1).all()


def _run_gpu(pairs, nnratio, th_low, check, cap):
    """pairs: list of synthetic_pair dicts; frames 2p = keyframe, 2p + 1 = current frame of pair p."""
    import torch
    B = 2 * len(pairs)
    desc = np.zeros((B, cap, 32), np.uint8); kps = np.zeros((B, cap), X.KEYPOINT_DTYPE)
    fn = np.zeros((B, cap), np.uint32); fi = np.zeros((B, cap), np.uint32)
    nfeat = np.zeros(B, np.int32); nout = np.zeros(B, np.int32); flags = np.zeros((len(pairs), cap), np.uint8)
    for i, p in enumerate(pairs):
        for f, (d, k, fv) in ((2 * i, (p["dk"], p["kps_k"], p["fv_k"])), (2 * i + 1, (p["df"], p["kps_f"], p["fv_f"]))):
            desc[f, :len(d)] = d; kps[f, :len(k)] = k; nout[f] = len(d)
            fn[f, :len(fv[0])] = fv[0]; fi[f, :len(fv[1])] = fv[1]; nfeat[f] = len(fv[0])
        flags[i, :len(p["flags"])] = p["flags"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8) if a.dtype.fields else np.ascontiguousarray(a)).cuda()
    d_m = torch.full((len(pairs), cap), -7, dtype=torch.int32, device="cuda"); d_nm = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    ex = X.ORBextractor(1000)
    ex.search_by_bow_device(len(pairs), (0, 2), (1, 2), dev(fn.view(np.int32)), dev(fi.view(np.int32)), dev(nfeat), dev(flags), dev(kps), dev(desc),
                            dev(nout), cap, d_m, d_nm, nnratio=nnratio, th_low=th_low, check_orientation=check)
    ex.synchronize()
    return d_nm.cpu().numpy(), d_m.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_gpu_search_by_bow_equals_oracle(case):
    rng = np.random.default_rng(case["seed"] + 100)
    pairs = [synthetic_pair(rng, case.get("n_kf", 900) - 17 * i, case.get("n_f", 1000) - 31 * i, case.get("n_nodes", 60), case.get("tie_heavy", False))
             for i in range(3)]
    nn, th, chk = case.get("nnratio", 0.7), case.get("th_low", 50), case.get("check", True)
    cap = 1024
    nm, m = _run_gpu(pairs, nn, th, chk, cap)
    for i, p in enumerate(pairs):
        n, want = O.search_by_bow(p["fv_k"], p["fv_f"], p["flags"], p["kps_k"], p["dk"], p["kps_f"], p["df"], nn, th, chk)
        assert int(nm[i]) == n, "pair %d" % i
        assert m[i, :len(want)].tolist() == want.tolist(), "pair %d" % i
        assert (m[i, len(want):] == -1).all()


@pytest.mark.gpu
def test_gpu_search_by_bow_on_extracted_frames():
    """TrackReferenceKeyFrame's order on device buffers: ExtractORB (both frames), ComputeBoW, SearchByBoW."""
    import torch
    rng = np.random.default_rng(12)
    v = make_vocab(rng, k=10, L=4, ragged=False)
    voc = X.Vocabulary(arrays=v)
    base = synth.frames("textured", 70, 1, 520, 700)[0]
    fr = np.stack([base[20:500, 30:670], base[22:502, 33:673], base[20:500, 30:670], base[14:494, 38:678]])      # two (keyframe, frame) pairs
    B = 4
    ex = X.ORBextractor(1000, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_mono = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(torch.from_numpy(np.ascontiguousarray(fr)).cuda(), B, 480, 640, d_k, d_d, d_n, d_mono, cap)
    d_wid = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_ww = torch.zeros((B, cap), dtype=torch.float64, device="cuda")
    d_nw = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_fn = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); d_fi = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    d_nf = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.compute_bow_device(voc, B, d_d, d_n, cap, d_wid, d_ww, d_nw, d_fn, d_fi, d_nf, levels_up=2)
    flags = (rng.random((2, cap)) < 0.85).astype(np.uint8)
    d_m = torch.zeros((2, cap), dtype=torch.int32, device="cuda"); d_nm = torch.zeros(2, dtype=torch.int32, device="cuda")
    ex.search_by_bow_device(2, (0, 2), (1, 2), d_fn, d_fi, d_nf, torch.from_numpy(flags).cuda(), d_k, d_d, d_n, cap, d_m, d_nm)
    ex.synchronize()
    n = d_n.cpu().numpy(); nf = d_nf.cpu().numpy()
    kk = d_k.cpu().numpy(); dd = d_d.cpu().numpy(); fnn = d_fn.cpu().numpy().astype(np.uint32); fii = d_fi.cpu().numpy().astype(np.uint32)
    total = 0
    for p in range(2):
        a, b = 2 * p, 2 * p + 1
        kp = lambda f: kk[f, :n[f]].copy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
        want_n, want = O.search_by_bow((fnn[a, :nf[a]], fii[a, :nf[a]]), (fnn[b, :nf[b]], fii[b, :nf[b]]), flags[p, :n[a]], kp(a), dd[a, :n[a]],
                                       kp(b), dd[b, :n[b]])
        assert int(d_nm[p]) == want_n and d_m[p, :n[b]].cpu().numpy().tolist() == want.tolist(), "pair %d" % p
        total += want_n
    assert total > 150      # shifted views of one scene do match


# ---- the keyframe-to-keyframe overload: ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher.cc:823-963) ----

def brute_keyframes(p, flags2, nnratio, th_low, check):
    (kn, ki), (fn, fi) = p["fv_k"], p["fv_f"]
    bits_k = np.unpackbits(p["dk"], axis=1).astype(np.int16); bits_f = np.unpackbits(p["df"], axis=1).astype(np.int16)
    m12 = np.full(len(p["dk"]), -1, np.int64)
    matched2 = np.zeros(len(p["df"]), bool)
    bins = {}
    for node in sorted(set(kn.tolist()) & set(fn.tolist())):
        fs = fi[fn == node]
        for k in ki[kn == node]:
            if not p["flags"][k] & 1:
                continue
            free = [int(f) for f in fs if not matched2[f] and flags2[f] & 1]
            if not free:
                continue
            dist = np.abs(bits_f[free] - bits_k[k]).sum(1)
            order = np.argsort(dist, kind="stable")
            b1 = int(dist[order[0]]); b2 = int(dist[order[1]]) if len(free) > 1 else 256
            if b1 < th_low and np.float32(b1) < np.float32(nnratio) * np.float32(b2):
                f = free[int(order[0])]
                m12[k] = f; matched2[f] = True
                rot = np.float32(p["kps_k"]["angle"][k]) - np.float32(p["kps_f"]["angle"][f])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                bins[int(k)] = int(np.floor(float(np.float32(rot * np.float32(1.0 / 30))) + 0.5)) % 30
    if check:
        cnt = [sum(1 for b in bins.values() if b == i) for i in range(30)]
        m1 = m2 = m3 = 0; i1 = i2 = i3 = -1
        for i, s in enumerate(cnt):
            if s > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
            elif s > m2:
                m3, m2, i3, i2 = m2, s, i2, i
            elif s > m3:
                m3, i3 = s, i
        if m2 < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for k, b in bins.items():
            if b not in (i1, i2, i3):
                m12[k] = -1
    return int((m12 >= 0).sum()), m12


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_oracle_keyframe_search_by_bow_equals_independent_statement(case):
    rng = np.random.default_rng(case["seed"] + 40)
    p = synthetic_pair(rng, case.get("n_kf", 900), case.get("n_f", 1000), case.get("n_nodes", 60), case.get("tie_heavy", False))
    flags2 = (rng.random(len(p["df"])) < 0.7).astype(np.uint8)
    nn, th, chk = case.get("nnratio", 0.8), case.get("th_low", 50), case.get("check", True)
    n, m = O.search_by_bow_keyframes(p["fv_k"], p["fv_f"], p["flags"], flags2, p["kps_k"], p["dk"], p["kps_f"], p["df"], nn, th, chk)
    bn, bm = brute_keyframes(p, flags2, nn, th, chk)
    assert n == bn and m.tolist() == bm.tolist()
    got = m[m >= 0]
    assert len(set(got.tolist())) == len(got) and all(flags2[g] & 1 for g in got)
    if case.get("n_nodes", 60) <= 60:
        assert n > 80


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_gpu_keyframe_search_by_bow_equals_oracle(case):
    import torch
    rng = np.random.default_rng(case["seed"] + 140)
    pairs = [synthetic_pair(rng, case.get("n_kf", 900) - 13 * i, case.get("n_f", 1000) - 29 * i, case.get("n_nodes", 60), case.get("tie_heavy", False))
             for i in range(3)]
    nn, th, chk = case.get("nnratio", 0.8), case.get("th_low", 50), case.get("check", True)
    cap = 1024
    B = 2 * len(pairs)
    desc = np.zeros((B, cap, 32), np.uint8); kps = np.zeros((B, cap), X.KEYPOINT_DTYPE)
    fn = np.zeros((B, cap), np.uint32); fi = np.zeros((B, cap), np.uint32)
    nfeat = np.zeros(B, np.int32); nout = np.zeros(B, np.int32)
    flags1 = np.zeros((len(pairs), cap), np.uint8); flags2 = np.zeros((len(pairs), cap), np.uint8)
    for i, p in enumerate(pairs):
        for f, (d, k, fv) in ((2 * i, (p["dk"], p["kps_k"], p["fv_k"])), (2 * i + 1, (p["df"], p["kps_f"], p["fv_f"]))):
            desc[f, :len(d)] = d; kps[f, :len(k)] = k; nout[f] = len(d)
            fn[f, :len(fv[0])] = fv[0]; fi[f, :len(fv[1])] = fv[1]; nfeat[f] = len(fv[0])
        flags1[i, :len(p["flags"])] = p["flags"]
        flags2[i, :len(p["df"])] = (rng.random(len(p["df"])) < 0.7).astype(np.uint8) | 2      # bit 1 is noise
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8) if a.dtype.fields else np.ascontiguousarray(a)).cuda()
    d_m = torch.full((len(pairs), cap), -7, dtype=torch.int32, device="cuda"); d_nm = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    ex = X.ORBextractor(1000)
    ex.search_by_bow_keyframes_device(len(pairs), (0, 2), (1, 2), dev(fn.view(np.int32)), dev(fi.view(np.int32)), dev(nfeat), dev(flags1), dev(flags2),
                                      dev(kps), dev(desc), dev(nout), cap, d_m, d_nm, nnratio=nn, th_low=th, check_orientation=chk)
    ex.synchronize()
    m = d_m.cpu().numpy()
    for i, p in enumerate(pairs):
        n, want = O.search_by_bow_keyframes(p["fv_k"], p["fv_f"], p["flags"], flags2[i, :len(p["df"])], p["kps_k"], p["dk"], p["kps_f"], p["df"], nn, th, chk)
        assert int(d_nm[i]) == n, "pair %d" % i
        assert m[i, :len(want)].tolist() == want.tolist(), "pair %d" % i
        assert (m[i, len(want):] == -1).all()
