"""SURVEY.md §8f-2, second half: ORBmatcher::SearchByProjection — frame-to-frame (reference src/ORBmatcher.cc:1961-2177, called by
Tracking::TrackWithMotionModel) and map-to-frame (src/ORBmatcher.cc:44-135, called by Tracking::SearchLocalPoints), Nleft == -1.
CPU: the oracle against an independent brute-force statement; GPU: orbx_project_last_frame_device + orbx_search_by_projection_device
against the oracle, bit-exact (requests, matches, occupancy, match count)."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth

CAM = dict(fx=500.0, fy=500.0, cx=320.0, cy=240.0)
ROWS, COLS = 480, 640


def scales():
    return np.asarray(X.compute_tables(1000, 1.2, 8)["scale_factors"], np.float32)


def shifted_frames(n, seed, dx, dy, noise=2):
    """n views of one textured scene; view k is view 0 translated by k*(dx, dy) pixels (content moves right / down)."""
    big = synth.textured_frame(seed, ROWS + 128, COLS + 128)
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        v = big[64 - k * dy:64 - k * dy + ROWS, 64 - k * dx:64 - k * dx + COLS].astype(np.int32)
        if noise:
            v = v + rng.integers(-noise, noise + 1, v.shape)
        out.append(np.clip(v, 0, 255).astype(np.uint8))
    return np.stack(out)


def pose(tx=0.0, ty=0.0, tz=0.0, yaw=0.0):
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.zeros((3, 4), np.float32)
    T[:, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    T[:, 3] = (tx, ty, tz)
    return T


def frame_products(img, nf=1000):
    """mvKeys, mDescriptors, mvKeysUn, mGrid of one frame, from the oracle."""
    cam = O.camera(**CAM)
    b = O.image_bounds(cam, COLS, ROWS)
    _, k, d = O.Oracle(nf).extract(img, (0, 0))
    un, off, idx = O.frame_finish(cam, k, b)
    return dict(k=k, d=d, un=un, off=off, idx=idx, bounds=b, cam=cam)


def make_map(last, depth, rng, valid=0.75, with_obs=0.9, desc_flip=3):
    """A synthetic map for the last frame: every keypoint's MapPoint sits on the plane z = depth in the last camera's frame."""
    n = len(last["k"])
    flags = (rng.random(n) < valid).astype(np.uint8)
    flags |= ((rng.random(n) < with_obs).astype(np.uint8) << 1) & (flags << 1)
    u, v = last["un"]["x"].astype(np.float64), last["un"]["y"].astype(np.float64)
    z = np.full(n, depth) * (1 + 0.02 * rng.standard_normal(n))
    world = np.stack([(u - CAM["cx"]) / CAM["fx"] * z, (v - CAM["cy"]) / CAM["fy"] * z, z], 1).astype(np.float32)
    mpd = last["d"].copy()
    for i in range(n):                                   # MapPoint::GetDescriptor() is close to, not equal to, the frame's descriptor
        for bit in rng.integers(0, 256, desc_flip):
            mpd[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
    return flags, world, mpd


def brute_search(q, qd, un, d, inside_pos, bounds, ur, occ0, ratio, nnratio, check, max_dist=100):
    """Independent statement on Python containers: candidates = brute-force box + level test over the keypoints inside the grid, in
    grid order; sequential best / second; acceptance; occupancy; rotation histogram."""
    N = len(un)
    occ = [0] * N if occ0 is None else [int(x) for x in occ0]
    m = [-1] * N
    nm = 0
    hist = [[] for _ in range(30)]
    order = sorted(inside_pos, key=inside_pos.get)
    bits = np.unpackbits(d, axis=1) if N else np.zeros((0, 256), np.uint8)
    for i in range(len(q)):
        if not (q["flags"][i] & 1):
            continue
        x, y, r = np.float32(q["u"][i]), np.float32(q["v"][i]), np.float32(q["radius"][i])
        lo, hi = int(q["min_level"][i]), int(q["max_level"][i])
        check_lv = lo > 0 or hi >= 0
        qb = np.unpackbits(qd[i])
        best, best2, bl, bl2, bi = 256, 256, -1, -1, -1
        # the cell window only prunes; a keypoint passing the box test always lies in a visited cell, EXCEPT that an empty window
        # (request outside the grid) yields nothing: reproduce the early returns of GetFeaturesInArea
        wInv = np.float32(64) / np.float32(bounds[1] - bounds[0]); hInv = np.float32(48) / np.float32(bounds[3] - bounds[2])
        if max(0, int(np.floor((x - bounds[0] - r) * wInv))) >= 64 or min(63, int(np.ceil((x - bounds[0] + r) * wInv))) < 0:
            continue
        if max(0, int(np.floor((y - bounds[2] - r) * hInv))) >= 48 or min(47, int(np.ceil((y - bounds[2] + r) * hInv))) < 0:
            continue
        any_cand = False
        for i2 in order:
            if check_lv and (un["octave"][i2] < lo or (hi >= 0 and un["octave"][i2] > hi)):
                continue
            if not (abs(un["x"][i2] - x) < r and abs(un["y"][i2] - y) < r):
                continue
            any_cand = True
            if occ[i2]:
                continue
            if ur is not None and ur[i2] > 0 and abs(np.float32(q["ur"][i]) - ur[i2]) > r:
                continue
            dist = int((qb != bits[i2]).sum())
            if dist < best:
                best2, best, bl2, bl, bi = best, dist, bl, int(un["octave"][i2]), i2
            elif ratio and dist < best2:
                bl2, best2 = int(un["octave"][i2]), dist
        if not any_cand or best > max_dist:
            continue
        if ratio and bl == bl2 and np.float32(best) > np.float32(nnratio) * np.float32(best2):
            continue
        m[bi] = i
        occ[bi] = 1 if q["flags"][i] & 2 else 0
        nm += 1
        if not ratio and check:
            rot = np.float32(q["angle"][i]) - un["angle"][bi]
            if rot < 0:
                rot = np.float32(rot + np.float32(360))
            b = int(np.floor(float(rot * np.float32(1.0 / 30)) + 0.5))
            hist[0 if b == 30 else b].append(bi)
    if not ratio and check:
        sizes = [len(h) for h in hist]
        order3 = sorted(range(30), key=lambda b: (-sizes[b], b))
        keep = [order3[0]] if sizes[order3[0]] > 0 else []
        if len(keep) and sizes[order3[1]] >= 0.1 * sizes[order3[0]] and sizes[order3[1]] > 0:
            keep.append(order3[1])
            if sizes[order3[2]] >= 0.1 * sizes[order3[0]] and sizes[order3[2]] > 0:
                keep.append(order3[2])
        for b in range(30):
            if b not in keep:
                for i2 in hist[b]:
                    m[i2] = -1; occ[i2] = 0; nm -= 1
    return nm, m, occ


def random_scene(rng, n=900, nq=700, ratio=False):
    cam = O.camera(**CAM)
    b = O.image_bounds(cam, COLS, ROWS)
    k = np.zeros(n, O.KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(-5, COLS + 5, n).astype(np.float32), rng.uniform(-5, ROWS + 5, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n); k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["size"], k["class_id"] = 31, -1
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    un, off, idx = O.frame_finish(cam, k, b)
    sc = scales()
    q = np.zeros(nq, O.PROJ_QUERY_DTYPE); qd = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    for i in range(nq):
        t = int(rng.integers(0, n))                      # aim at keypoint t with a descriptor 0..60 bits away
        lvl = int(un["octave"][t])
        q["u"][i] = un["x"][t] + rng.uniform(-6, 6); q["v"][i] = un["y"][t] + rng.uniform(-6, 6)
        q["radius"][i] = np.float32(rng.choice([2.5, 4.0, 7.0, 15.0])) * sc[lvl]
        q["ur"][i] = q["u"][i] - 8.0
        mode = int(rng.integers(0, 4))
        q["min_level"][i], q["max_level"][i] = [(lvl - 1, lvl + 1), (lvl, -1), (0, lvl), (lvl - 1, lvl)][mode]
        q["flags"][i] = (1 if rng.random() < 0.9 else 0) | (2 if rng.random() < 0.85 else 0)
        q["angle"][i] = np.float32((un["angle"][t] + rng.choice([0, 0, 0, 12, 100, 200]) + rng.uniform(-3, 3)) % 360)
        qd[i] = d[t]
        for bit in rng.integers(0, 256, int(rng.integers(0, 60))):
            qd[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
    if nq > 10:                                           # exact descriptor ties and requests outside the image
        qd[5] = qd[4]; q[5] = q[4]
        q["u"][7] = -400.0; q["v"][8] = 5000.0
    ur = np.where(rng.random(n) < 0.6, un["x"] - 8.0 + rng.uniform(-20, 20, n), -1.0).astype(np.float32)
    occ = (rng.random(n) < 0.15).astype(np.uint8)
    return dict(un=un, d=d, off=off, idx=idx, bounds=b, q=q, qd=qd, ur=ur, occ=occ)


@pytest.mark.parametrize("seed,ratio,stereo,check,maxd", [(1, False, False, True, 100), (2, False, True, True, 100), (3, True, True, False, 100),
                                                          (4, True, False, False, 100), (5, False, True, False, 100), (7, False, False, True, 64)])
def test_oracle_search_equals_brute_force(seed, ratio, stereo, check, maxd):
    rng = np.random.default_rng(seed)
    s = random_scene(rng, ratio=ratio)
    if maxd != 100:      # the relocalisation form (ORBmatcher.cc:2179-2300): ORBdist, every MapPoint closes its keypoint
        s["q"]["flags"] |= 2
    inside_pos = {int(i): p for p, i in enumerate(s["idx"])}
    nm, m, occ = O.search_by_projection(s["q"], s["qd"], s["un"], s["d"], s["off"], s["idx"], s["bounds"], s["ur"] if stereo else None,
                                        s["occ"], ratio, 0.8, check, maxd)
    bn, bm, bocc = brute_search(s["q"], s["qd"], s["un"], s["d"], inside_pos, s["bounds"], s["ur"] if stereo else None, s["occ"], ratio, 0.8, check, maxd)
    assert m.tolist() == bm and nm == bn and occ.tolist() == bocc
    assert nm > 50                                        # the scene is built so that many requests find their keypoint


def test_oracle_projection_matches_double_precision_geometry():
    rng = np.random.default_rng(9)
    last = frame_products(shifted_frames(1, 3, 0, 0)[0])
    flags, world, _ = make_map(last, 5.0, rng)
    Tlw, Tcw = pose(), pose(tx=0.07, ty=-0.03, tz=0.0, yaw=0.004)
    q = O.project_last_frame(last["k"], last["un"], flags, world, Tcw, Tlw, last["cam"], last["bounds"], scales(), 40.0, 0.08, 15.0, True)
    X3 = world.astype(np.float64) @ Tcw[:, :3].astype(np.float64).T + Tcw[:, 3].astype(np.float64)
    u = CAM["fx"] * X3[:, 0] / X3[:, 2] + CAM["cx"]; v = CAM["fy"] * X3[:, 1] / X3[:, 2] + CAM["cy"]
    inb = (u >= 0) & (u <= COLS) & (v >= 0) & (v <= ROWS) & (flags & 1).astype(bool)
    on = (q["flags"] & 1).astype(bool)
    edge = (np.abs(u) < 1e-3) | (np.abs(u - COLS) < 1e-3) | (np.abs(v) < 1e-3) | (np.abs(v - ROWS) < 1e-3)
    assert ((on == inb) | edge).all() and on.sum() > 300
    assert np.abs(q["u"][on] - u[on]).max() < 2e-3 and np.abs(q["v"][on] - v[on]).max() < 2e-3
    assert np.abs(q["ur"][on] - (u[on] - 40.0 / X3[on, 2])).max() < 2e-3
    oct_ = last["k"]["octave"][on]
    assert np.array_equal(q["min_level"][on], oct_ - 1) and np.array_equal(q["max_level"][on], oct_ + 1)      # monocular: never forward / backward
    assert np.array_equal(q["radius"][on], np.float32(15.0) * scales()[oct_])
    assert np.array_equal(q["flags"][on], 1 | (flags[on] & 2))
    # stereo, camera moved forward by more than the baseline: only same-or-coarser levels are searched (:2018-2019)
    qf = O.project_last_frame(last["k"], last["un"], flags, world, pose(tz=-0.5), Tlw, last["cam"], last["bounds"], scales(), 40.0, 0.08, 7.0, False)
    onf = (qf["flags"] & 1).astype(bool)
    assert onf.sum() > 100 and (qf["max_level"][onf] == -1).all() and np.array_equal(qf["min_level"][onf], last["k"]["octave"][onf])
    qb = O.project_last_frame(last["k"], last["un"], flags, world, pose(tz=0.5), Tlw, last["cam"], last["bounds"], scales(), 40.0, 0.08, 7.0, False)
    onb = (qb["flags"] & 1).astype(bool)
    assert onb.sum() > 100 and (qb["min_level"][onb] == 0).all() and np.array_equal(qb["max_level"][onb], last["k"]["octave"][onb])


def test_oracle_tracks_a_translating_camera():
    """End to end on the CPU: consecutive views of a fronto-parallel plane, the true pose, th = 15: most MapPoints are found again at the
    keypoint that is their own image, shifted."""
    rng = np.random.default_rng(11)
    fr = shifted_frames(2, 21, 6, 3)
    last, cur = frame_products(fr[0]), frame_products(fr[1])
    Z = 5.0
    flags, world, mpd = make_map(last, Z, rng, valid=1.0)
    Tcw = pose(tx=6 * Z / CAM["fx"], ty=3 * Z / CAM["fy"])
    q = O.project_last_frame(last["k"], last["un"], flags, world, Tcw, pose(), last["cam"], last["bounds"], scales(), 40.0, 0.08, 15.0, True)
    nm, m, _ = O.search_by_projection(q, mpd, cur["un"], cur["d"], cur["off"], cur["idx"], cur["bounds"], None, None, False, 0.9, True)
    assert nm > 0.35 * len(last["k"])
    hit = np.nonzero(m >= 0)[0]
    dxy = np.stack([cur["un"]["x"][hit] - last["un"]["x"][m[hit]], cur["un"]["y"][hit] - last["un"]["y"][m[hit]]], 1)
    lvl = scales()[cur["un"]["octave"][hit]]
    assert (np.abs(dxy - np.array([6.0, 3.0])) <= 2.5 * lvl[:, None]).mean() > 0.97


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _frames_to_device(prods, cap):
    """Per-frame oracle products -> the frame-major device arrays the C ABI takes."""
    B = len(prods)
    un = np.zeros((B, cap), O.KEYPOINT_DTYPE); k = np.zeros((B, cap), O.KEYPOINT_DTYPE); d = np.zeros((B, cap, 32), np.uint8)
    n = np.zeros(B, np.int32); off = np.zeros((B, 64 * 48 + 1), np.int32); idx = np.zeros((B, cap), np.int32)
    for f, p in enumerate(prods):
        m = len(p["un"])
        un[f, :m], d[f, :m], n[f], off[f] = p["un"], p["d"], m, p["off"]
        if "k" in p:
            k[f, :m] = p["k"]
        idx[f, :len(p["idx"])] = p["idx"]
    return dict(un=_dev(un.view(np.uint8)), k=_dev(k.view(np.uint8)), d=_dev(d), n=_dev(n), off=_dev(off), idx=_dev(idx))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ratio,stereo,check,maxd", [(1, False, False, True, 100), (2, False, True, True, 100), (3, True, True, False, 100),
                                                          (4, True, False, False, 100), (6, False, True, False, 100), (8, False, False, True, 64)])
def test_gpu_search_equals_oracle_on_random_scenes(seed, ratio, stereo, check, maxd):
    import torch
    P, cap, qcap = 3, 1024, 768
    rng = np.random.default_rng(seed)
    scenes = [random_scene(rng, n=int(rng.integers(500, 1000)), nq=int(rng.integers(300, 768)), ratio=ratio) for _ in range(P)]
    if maxd != 100:      # the relocalisation form: ORBdist, every MapPoint closes its keypoint
        for s in scenes:
            s["q"]["flags"] |= 2
    dev = _frames_to_device([dict(un=s["un"], d=s["d"], off=s["off"], idx=s["idx"]) for s in scenes], cap)
    q = np.zeros((P, qcap), O.PROJ_QUERY_DTYPE); qd = np.zeros((P, qcap, 32), np.uint8); nq = np.zeros(P, np.int32)
    ur = np.full((P, cap), -1.0, np.float32); occ = np.zeros((P, cap), np.uint8)
    for p, s in enumerate(scenes):
        q[p, :len(s["q"])] = s["q"]; qd[p, :len(s["q"])] = s["qd"]; nq[p] = len(s["q"])
        ur[p, :len(s["ur"])] = s["ur"]; occ[p, :len(s["occ"])] = s["occ"]
    d_occ = _dev(occ)
    d_m = torch.full((P, cap), -7, dtype=torch.int32, device="cuda"); d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    ex = X.ORBextractor(1000)
    ex.search_by_projection_device(P, (0, 1), _dev(q.view(np.uint8)), _dev(qd), (0, 1), _dev(nq), qcap, dev["un"], dev["d"], dev["n"], cap,
                                   dev["off"], dev["idx"], scenes[0]["bounds"], _dev(ur) if stereo else None, d_occ, ratio, 0.8, check, d_m, d_nm, max_distance=maxd)
    ex.synchronize()
    for p, s in enumerate(scenes):
        nm, m, o = O.search_by_projection(s["q"], s["qd"], s["un"], s["d"], s["off"], s["idx"], s["bounds"], s["ur"] if stereo else None,
                                          s["occ"], ratio, 0.8, check, maxd)
        n = len(s["un"])
        assert int(d_nm[p]) == nm, "pair %d" % p
        assert d_m[p, :n].cpu().numpy().tolist() == m.tolist(), "pair %d" % p
        assert d_occ[p, :n].cpu().numpy().tolist() == o.tolist(), "pair %d" % p
        assert (d_m[p, n:].cpu().numpy() == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mono,tz,th", [(True, 0.0, 15.0), (False, 0.0, 7.0), (False, -0.5, 7.0), (False, 0.5, 15.0)])
def test_gpu_frame_to_frame_pipeline_equals_oracle(mono, tz, th):
    """The whole TrackWithMotionModel matching step on a device-resident batch extracted by the HIP path: extraction -> UndistortKeyPoints /
    AssignFeaturesToGrid -> projection of the last frame's MapPoints -> search, for the pairs (0,1), (1,2), (2,3) of one batch."""
    import torch
    B, nf = 4, 1000
    fr = shifted_frames(B, 33, 5, 2)
    ex = X.ORBextractor(nf, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_mo = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(_dev(fr), B, ROWS, COLS, d_k, d_d, d_n, d_mo, cap, lapping=(0, 0))
    cam = X.camera(**CAM); bounds = X.compute_image_bounds(cam, COLS, ROWS)
    d_un = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_goff = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda"); d_gidx = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    d_nin = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.frame_finish_device(B, d_k, d_n, cap, cam, bounds, d_un, d_goff, d_gidx, d_nin)
    # the map and the poses (tracker state: plain arrays)
    rng = np.random.default_rng(5)
    prods = [frame_products(fr[f], nf) for f in range(B)]
    Z = 5.0
    flags = np.zeros((B, cap), np.uint8); world = np.zeros((B, cap, 3), np.float32); mpd = np.zeros((B, cap, 32), np.uint8)
    poses = np.zeros((B, 3, 4), np.float32)
    ur = np.full((B, cap), -1.0, np.float32)
    for f in range(B):
        n = len(prods[f]["k"])
        fl, w, md = make_map(prods[f], Z, rng)
        # world points are expressed in the frame-0 camera; frame f sees the scene shifted by f*(5, 2) px
        w[:, 0] -= np.float32(f * 5 * Z / CAM["fx"]); w[:, 1] -= np.float32(f * 2 * Z / CAM["fy"])
        flags[f, :n], world[f, :n], mpd[f, :n] = fl, w, md
        poses[f] = pose(tx=f * 5 * Z / CAM["fx"], ty=f * 2 * Z / CAM["fy"], tz=f * tz)
        ur[f, :n] = np.where(rng.random(n) < 0.7, prods[f]["un"]["x"] - 40.0 / Z + rng.uniform(-12, 12, n), -1.0)
    P = B - 1
    d_q = torch.zeros((P, cap, 8), dtype=torch.float32, device="cuda")
    ex.project_last_frame_device(P, (0, 1), (1, 1), d_k, d_un, d_n, cap, _dev(flags), _dev(world), _dev(poses), cam, bounds, 40.0, 0.08, th, mono, d_q)
    d_m = torch.zeros((P, cap), dtype=torch.int32, device="cuda"); d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    ex.search_by_projection_device(P, (1, 1), d_q, _dev(mpd), (0, 1), None, cap, d_un, d_d, d_n, cap, d_goff, d_gidx, bounds,
                                   None if mono else _dev(ur), None, False, 0.9, True, d_m, d_nm)
    ex.synchronize()
    total = 0
    for p in range(P):
        last, cur = prods[p], prods[p + 1]
        nl, nc = len(last["k"]), len(cur["k"])
        q = O.project_last_frame(last["k"], last["un"], flags[p, :nl], world[p, :nl], poses[p + 1], poses[p], last["cam"], last["bounds"], scales(),
                                 40.0, 0.08, th, mono)
        got_q = d_q[p].cpu().numpy().view(np.uint8).reshape(cap, 32)
        assert got_q[:nl].tobytes() == q.tobytes(), "requests of pair %d" % p
        assert not got_q[nl:].any()
        nm, m, _ = O.search_by_projection(q, mpd[p, :nl], cur["un"], cur["d"], cur["off"], cur["idx"], cur["bounds"], None if mono else ur[p + 1, :nc],
                                          None, False, 0.9, True)
        assert int(d_nm[p]) == nm and d_m[p, :nc].cpu().numpy().tolist() == m.tolist(), "pair %d" % p
        total += nm
    # the scenario really matches; with the camera also moving along z (forward / backward level windows) the poses no longer
    # agree with the purely translated images, so fewer requests land on their keypoint
    assert total > (300 if tz == 0.0 else 50)


@pytest.mark.gpu
def test_gpu_search_argument_errors():
    import torch
    ex = X.ORBextractor(1000)
    z = torch.zeros(64, dtype=torch.int32, device="cuda")
    b = np.array([0, 640, 0, 480], np.float32)
    with pytest.raises(X.OrbxError):
        ex.search_by_projection_device(1, (0, 1), z, z, (0, 1), None, 16, z, z, z, 40000, z, z, b, None, None, False, 0.9, True, z, z)   # capacity
    with pytest.raises(X.OrbxError):
        ex.search_by_projection_device(0, (0, 1), z, z, (0, 1), None, 16, z, z, z, 16, z, z, b, None, None, False, 0.9, True, z, z)      # n_pairs
    with pytest.raises(X.OrbxError):
        ex.search_by_projection_device(1, (0, 1), None, z, (0, 1), None, 16, z, z, z, 16, z, z, b, None, None, False, 0.9, True, z, z)   # null


def crowd_scene(rng, n=400, nq=700, clusters=6, ratio=False):
    """Many requests compete for few keypoints: every cluster holds ~n/clusters keypoints within a few pixels, all on neighbouring levels,
    with near-identical descriptors, and ~nq/clusters requests that all aim at the cluster centre with a radius covering it.  In the
    reference each accepted request closes its keypoint and the next one has to settle for the next best: dependency chains as long as the
    cluster, best-key lists that run dry (the kernel's re-scan path), ties everywhere."""
    cam = O.camera(**CAM)
    b = O.image_bounds(cam, COLS, ROWS)
    centres = np.stack([rng.uniform(60, COLS - 60, clusters), rng.uniform(60, ROWS - 60, clusters)], 1)
    k = np.zeros(n, O.KEYPOINT_DTYPE)
    which = rng.integers(0, clusters, n)
    k["x"] = (centres[which, 0] + rng.uniform(-9, 9, n)).astype(np.float32); k["y"] = (centres[which, 1] + rng.uniform(-9, 9, n)).astype(np.float32)
    k["octave"] = rng.integers(1, 4, n); k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["size"], k["class_id"] = 31, -1
    proto = rng.integers(0, 256, (clusters, 32), dtype=np.uint8)
    d = proto[which].copy()
    for i in range(n):
        for bit in rng.integers(0, 256, int(rng.integers(0, 4))):
            d[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
    un, off, idx = O.frame_finish(cam, k, b)
    q = np.zeros(nq, O.PROJ_QUERY_DTYPE); qd = np.zeros((nq, 32), np.uint8)
    for i in range(nq):
        c = int(rng.integers(0, clusters))
        q["u"][i], q["v"][i] = centres[c, 0] + rng.uniform(-2, 2), centres[c, 1] + rng.uniform(-2, 2)
        q["radius"][i] = np.float32(rng.choice([9.0, 14.0, 30.0]))
        q["ur"][i] = q["u"][i] - 8.0
        q["min_level"][i], q["max_level"][i] = [(0, -1), (1, 3), (2, -1), (-1, -1)][int(rng.integers(0, 4))]
        q["flags"][i] = 1 | (2 if rng.random() < 0.9 else 0)
        q["angle"][i] = np.float32(rng.uniform(0, 360))
        qd[i] = proto[c]
        for bit in rng.integers(0, 256, int(rng.integers(0, 3))):
            qd[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
    ur = np.full(n, -1.0, np.float32)
    occ = (rng.random(n) < 0.05).astype(np.uint8)
    return dict(un=un, d=d, off=off, idx=idx, bounds=b, q=q, qd=qd, ur=ur, occ=occ)


@pytest.mark.parametrize("seed,ratio", [(21, False), (22, True)])
def test_oracle_crowded_requests_equal_brute_force(seed, ratio):
    rng = np.random.default_rng(seed)
    s = crowd_scene(rng, ratio=ratio)
    inside = {int(i): p for p, i in enumerate(s["idx"])}
    nm, m, o = O.search_by_projection(s["q"], s["qd"], s["un"], s["d"], s["off"], s["idx"], s["bounds"], None, s["occ"], ratio, 0.8, not ratio, 100)
    bn, bm, bo = brute_search(s["q"], s["qd"], s["un"], s["d"], inside, s["bounds"], None, s["occ"], ratio, 0.8, not ratio)
    assert nm == bn and m.tolist() == bm and o.tolist() == bo
    assert (m >= 0).sum() > 60           # the crowds do get matched, one keypoint after the other (random angles: the histogram drops many)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ratio,cap,qcap", [(21, False, 1024, 768), (22, True, 1024, 768), (23, False, 2200, 2200), (24, True, 2200, 1500)])
def test_gpu_search_settles_crowded_requests_like_the_walk(seed, ratio, cap, qcap):
    """The fixed point against the oracle's sequential walk where the chains are long; cap = 2200 takes the kernel variant without
    best-key lists (they do not fit in LDS next to the staged frame)."""
    import torch
    P = 3
    rng = np.random.default_rng(seed)
    scenes = [crowd_scene(rng, n=int(rng.integers(300, 500)), nq=int(rng.integers(500, 760)), clusters=int(rng.integers(3, 9)), ratio=ratio) for _ in range(P)]
    dev = _frames_to_device([dict(un=s["un"], d=s["d"], off=s["off"], idx=s["idx"]) for s in scenes], cap)
    q = np.zeros((P, qcap), O.PROJ_QUERY_DTYPE); qd = np.zeros((P, qcap, 32), np.uint8); nq = np.zeros(P, np.int32)
    occ = np.zeros((P, cap), np.uint8)
    for p, s in enumerate(scenes):
        q[p, :len(s["q"])] = s["q"]; qd[p, :len(s["q"])] = s["qd"]; nq[p] = len(s["q"]); occ[p, :len(s["occ"])] = s["occ"]
    d_occ = _dev(occ)
    d_m = torch.full((P, cap), -7, dtype=torch.int32, device="cuda"); d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    ex = X.ORBextractor(1000)
    ex.search_by_projection_device(P, (0, 1), _dev(q.view(np.uint8)), _dev(qd), (0, 1), _dev(nq), qcap, dev["un"], dev["d"], dev["n"], cap,
                                   dev["off"], dev["idx"], scenes[0]["bounds"], None, d_occ, ratio, 0.8, not ratio, d_m, d_nm)
    ex.synchronize()
    import ctypes as C
    rounds = (C.c_int * 4)()
    X.load_library().orbx_debug_search_rounds(rounds)
    assert rounds[0] > 8                 # pair 0 really needed many rounds
    for p, s in enumerate(scenes):
        nm, m, o = O.search_by_projection(s["q"], s["qd"], s["un"], s["d"], s["off"], s["idx"], s["bounds"], None, s["occ"], ratio, 0.8, not ratio, 100)
        n = len(s["un"])
        assert int(d_nm[p]) == nm, "pair %d" % p
        assert d_m[p, :n].cpu().numpy().tolist() == m.tolist(), "pair %d" % p
        assert d_occ[p, :n].cpu().numpy().tolist() == o.tolist(), "pair %d" % p
