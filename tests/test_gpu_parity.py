"""Parity of the HIP path (through the C ABI of liborbx.so) with the CPU oracle: bit-exact at every stage
boundary and in the final keypoints/descriptors.  Runs on the GPU box only."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth
from helpers import GOLDEN_CASES, assert_same_result, fail_with_dump, load_case, load_gray, sort_kps

pytestmark = pytest.mark.gpu


def oracle_run(img, nf=1000, lap=(0, 1000), nlevels=8, sf=1.2, ini=20, mn=7):
    o = O.Oracle(nf, sf, nlevels, ini, mn)
    return o, o.extract(img, lap)


def check_stages(ex, o, lvl_gpu, nlevels=8, frame=0):
    """Every stage boundary of one frame against the oracle.  The first mismatch raises with a replay file (helpers.dump_failure) that holds
    the inputs, the ORBX_* switches and, for EVERY level, both sides' candidates and kept keypoints — not only the arrays that differed."""
    def fail(msg, **failed):
        # `failed`: the arrays that FAILED the comparison, as they were read then (the loop below reads the device again for the
        # dump; if the two reads differ the fault was transient, and the dump shows it)
        both = {"failed_" + k: v for k, v in failed.items()}
        for l in range(nlevels):
            both["gpu_candidates_%d" % l], both["oracle_candidates_%d" % l] = ex.debug_candidates(l, frame), o.candidates(l)
            both["gpu_level_keys_%d" % l], both["oracle_level_keys_%d" % l] = lvl_gpu[l], o.level_keypoints(l)
            both["gpu_pyramid_%d" % l], both["oracle_pyramid_%d" % l] = ex.image_pyramid_level(l, frame, bordered=True), o.level(l, bordered=True)      # (bordered: the interior is inside)
        fail_with_dump(msg, frame=frame, quotas=o.features_per_level, **both)

    for l in range(nlevels):
        got = ex.image_pyramid_level(l, frame)
        if not np.array_equal(got, o.level(l)):
            fail("pyramid level %d" % l, gpu_interior=got, oracle_interior=o.level(l))
        got = ex.image_pyramid_level(l, frame, bordered=True)
        if not np.array_equal(got, o.level(l, bordered=True)):
            fail("border %d" % l, gpu_bordered=got, oracle_bordered=o.level(l, bordered=True))
        if len(o.level_keypoints(l)) and ex.blurred_level_exists(l):      # the reference only blurs levels that hold keypoints (:1122-1127); where the blur ran per keypoint no blurred level exists
            got = ex.debug_blurred(l, frame)
            if not np.array_equal(got, o.blurred(l)):
                fail("blur level %d" % l, gpu_blurred=got, oracle_blurred=o.blurred(l))
        # k_fast's per-cell segments, read segment by segment, ARE vToDistributeKeys in the reference's order (cell row,
        # cell column, then raster order inside the cell, ORBextractor.cc:797-864): compared without sorting
        cg, co = ex.debug_candidates(l, frame), o.candidates(l)
        if cg.tobytes() != co.tobytes():
            fail("FAST candidates level %d (%d vs %d)" % (l, len(cg), len(co)), gpu_candidates=cg, oracle_candidates=co)
        if lvl_gpu[l].tobytes() != o.level_keypoints(l).tobytes():
            fail("quad-tree/orientation level %d" % l, gpu_level_keys=lvl_gpu[l], oracle_level_keys=o.level_keypoints(l))


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_golden_fixtures(case):
    c = load_case(case)
    rows, cols = c["image"].shape
    ex = X.ORBextractor(c["nfeatures"], 1.2, 8, 20, 7, max_width=cols, max_height=rows)
    mono, k, d, lvl = ex(c["image"], None, c["lapping"])
    assert_same_result((mono, k, d), (c["mono_index"], c["keypoints"], c["descriptors"]), case)
    assert [len(x) for x in lvl] == c["level_counts"].tolist()
    assert [len(ex.debug_candidates(l)) for l in range(8)] == c["candidate_counts"].tolist()


@pytest.mark.parametrize("variant", ["noise", "textured", "sparse"])
@pytest.mark.parametrize("shape,nf", [((480, 640), 1000), ((512, 512), 1000), ((333, 517), 700)])
def test_synthetic_stagewise(variant, shape, nf):
    rows, cols = shape
    img = synth.frames(variant, 11, 1, rows, cols)[0]
    o, want = oracle_run(img, nf)
    ex = X.ORBextractor(nf, 1.2, 8, 20, 7, max_width=cols, max_height=rows)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl)
    assert_same_result((mono, k, d), want, "%s %s" % (variant, shape))


def test_full_hd_2000_features():
    # BASELINE.json configs[2]: two quad-tree roots per level, 6342 cell slots
    img = synth.frames("textured", 2, 1, 1080, 1920)[0]
    o, want = oracle_run(img, 2000)
    ex = X.ORBextractor(2000, 1.2, 8, 20, 7, max_width=1920, max_height=1080)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl)
    assert_same_result((mono, k, d), want, "1080p")
    assert mono > 0      # x > 1000 is outside the mono lapping window {0,1000}: both branches of :1147-1156 taken


def test_adversarial_images():
    rows, cols = 300, 400
    yy, xx = np.mgrid[0:rows, 0:cols]
    imgs = {
        "constant": np.full((rows, cols), 128, np.uint8),
        "black": np.zeros((rows, cols), np.uint8),
        "white": np.full((rows, cols), 255, np.uint8),
        "checker8": (((yy // 8 + xx // 8) % 2) * 255).astype(np.uint8),      # tie-heavy: equal scores everywhere
        "checker1": (((yy + xx) % 2) * 255).astype(np.uint8),
        "saturated_noise": np.where(synth.noise_frame(5, rows, cols) > 127, 255, 0).astype(np.uint8),
        "ramp": ((xx * 255) // (cols - 1)).astype(np.uint8),
        "dots": np.where((yy % 23 == 0) & (xx % 29 == 0), 255, 40).astype(np.uint8),
    }
    ex = X.ORBextractor(500, 1.2, 8, 20, 7, max_width=cols, max_height=rows)
    for name, img in imgs.items():
        o, want = oracle_run(img, 500)
        mono, k, d, lvl = ex(img)
        check_stages(ex, o, lvl)
        assert_same_result((mono, k, d), want, name)


@pytest.mark.parametrize("lap", [(0, 0), (0, 1000), (100, 300), (250, 250), (-5, 5000)])
def test_lapping_area_split(lap):
    img = load_gray("robot_865_gray.png")
    o, want = oracle_run(img, 1200, lap)
    ex = X.ORBextractor(1200, 1.2, 8, 20, 7)
    mono, k, d, _ = ex(img, None, lap)
    assert_same_result((mono, k, d), want, "lap %s" % (lap,))


@pytest.mark.parametrize("params", [dict(nf=300, nlevels=4, sf=1.5, ini=30, mn=10), dict(nf=2000, nlevels=8, sf=1.2, ini=12, mn=3),
                                    dict(nf=150, nlevels=1, sf=1.2, ini=20, mn=7), dict(nf=800, nlevels=12, sf=1.1, ini=20, mn=7)])
def test_other_constructor_parameters(params):
    img = load_gray("luna_gray.png")
    o, want = oracle_run(img, params["nf"], (0, 1000), params["nlevels"], params["sf"], params["ini"], params["mn"])
    ex = X.ORBextractor(params["nf"], params["sf"], params["nlevels"], params["ini"], params["mn"], max_width=512, max_height=512)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl, params["nlevels"])
    assert_same_result((mono, k, d), want, str(params))


def test_batch_equals_single_frames_and_stereo_pair():
    # BASELINE.json configs[3]: L+R in one launch sequence, 1200 features per eye, lapping {0,0} (Frame.cc:109-110)
    frames = synth.frames("textured", 40, 6, 480, 640)
    ex = X.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=6)
    out = ex.extract_batch(frames, lapping=(0, 0))
    for f in range(6):
        o, want = oracle_run(frames[f], 1200, (0, 0))
        assert_same_result(out[f][:3], want, "frame %d" % f)
        for l in range(8):
            assert out[f][3][l].tobytes() == o.level_keypoints(l).tobytes()
    pair = ex.extract_batch(frames[2:4], lapping=[(0, 0), (0, 1000)])      # per-frame lapping areas
    assert_same_result(pair[0][:3], oracle_run(frames[2], 1200, (0, 0))[1])
    assert_same_result(pair[1][:3], oracle_run(frames[3], 1200, (0, 1000))[1])
    # pyramids of both eyes stay readable afterwards (Frame::ComputeStereoMatches reads them: Frame.cc:910,929)
    o, _ = oracle_run(frames[3], 1200)
    assert np.array_equal(ex.image_pyramid_level(3, frame=1), o.level(3))


def test_row_stride_and_submatrix_input():
    big = synth.frames("noise", 1, 1, 500, 700)[0]
    view = big[10:490, 30:670]                      # 480 x 640 window with a 700-byte row step (cv::Mat::step)
    assert view.strides == (700, 1)
    o, want = oracle_run(np.ascontiguousarray(view))
    ex = X.ORBextractor(1000)
    assert_same_result(ex(view)[:3], want, "strided")


def test_size_changes_between_calls():
    ex = X.ORBextractor(600, 1.2, 8, 20, 7, max_width=640, max_height=480)
    for shape in [(480, 640), (300, 400), (480, 640), (241, 333)]:
        img = synth.frames("textured", shape[0], 1, *shape)[0]
        assert_same_result(ex(img)[:3], oracle_run(img, 600)[1], str(shape))


def test_error_behaviour_matches_reference():
    ex = X.ORBextractor(1000)
    mono, k, d, lvl = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0 and d.shape == (0, 32)          # ORBextractor.cc:1083-1084
    L = X.load_library()
    n, m = C.c_int(), C.c_int()
    kp = np.zeros(2000, X.KEYPOINT_DTYPE); de = np.zeros((2000, 32), np.uint8)
    img = synth.frames("noise", 0, 1, 480, 640)[0]
    args = lambda im, rows, cols, cap: (ex._h, im.ctypes.data_as(C.c_void_p), rows, cols, cols, 0, 1000,
                                         kp.ctypes.data_as(C.c_void_p), de.ctypes.data_as(C.c_void_p), cap,
                                         C.byref(n), C.byref(m), None, None)
    assert L.orbx_extract(ex._h, None, 480, 640, 640, 0, 1000, kp.ctypes.data_as(C.c_void_p),
                          de.ctypes.data_as(C.c_void_p), 2000, C.byref(n), C.byref(m), None, None) == -1
    assert L.orbx_extract(*args(img, 480, 640, 100)) == -3 and n.value > 100       # capacity too small, count reported
    small = np.zeros((120, 160), np.uint8)
    assert L.orbx_extract(*args(small, 120, 160, 2000)) == -5                       # reference would divide by zero
    with pytest.raises(X.OrbxError):
        ex(synth.frames("noise", 0, 1, 600, 800)[0])                                # larger than max_width/height
    with pytest.raises(ValueError):
        ex(np.zeros((480, 640), np.float32))                                        # assert(type==CV_8UC1) :1087
    assert_same_result(ex(img)[:3], oracle_run(img)[1], "after errors")              # handle still usable


def test_device_resident_path_with_torch_tensors():
    import torch
    B, rows, cols = 4, 480, 640
    frames = synth.frames("noise", 100, B, rows, cols)
    ex = X.ORBextractor(1000, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, B, rows, cols, d_k, d_d, d_n, d_m, cap)
    torch.cuda.synchronize()
    n, m = d_n.cpu().numpy(), d_m.cpu().numpy()
    kraw = d_k.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    for f in range(B):
        k = kraw[f, :n[f]].copy().view(X.KEYPOINT_DTYPE).reshape(-1)
        assert_same_result((int(m[f]), k, d_d[f, :n[f]].cpu().numpy()), oracle_run(frames[f])[1], "device frame %d" % f)


def test_quadtree_paths_wide_and_clustered():
    # (a) 10 quad-tree roots: too many for the LDS count pyramids -> the kernel sweeps the keys every pass
    wide = synth.frames("noise", 3, 1, 260, 2400)[0]
    o, want = oracle_run(wide, 1500, (0, 1000), nlevels=3)
    ex = X.ORBextractor(1500, 1.2, 3, 20, 7, max_width=2400, max_height=260)
    mono, k, d, lvl = ex(wide)
    check_stages(ex, o, lvl, 3)
    assert_same_result((mono, k, d), want, "wide")
    # (b) tight clusters: nodes deeper than the dense phase's 5 levels must be materialised and swept
    rng = np.random.default_rng(4)
    img = np.full((480, 640), 90, np.uint8)
    for cy, cx in [(100, 120), (300, 500), (240, 320), (400, 90)]:
        ys = np.clip(cy + rng.integers(-24, 24, 900), 20, 459); xs = np.clip(cx + rng.integers(-24, 24, 900), 20, 619)
        img[ys, xs] = rng.integers(150, 255, 900)
    o, want = oracle_run(img, 3000)
    ex = X.ORBextractor(3000)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl)
    assert_same_result((mono, k, d), want, "clustered")


def test_randomised_sizes_and_parameters():
    rng = np.random.default_rng(2026)
    for t in range(10):
        rows, cols = int(rng.integers(230, 700)), int(rng.integers(230, 900))
        if rows > 2 * cols - 40:
            continue
        nf = int(rng.integers(50, 2500))
        variant = ["noise", "textured", "sparse"][t % 3]
        lap = (int(rng.integers(-10, 400)), int(rng.integers(100, 1200)))
        img = synth.frames(variant, 1000 + t, 1, rows, cols)[0]
        try:
            o, want = oracle_run(img, nf, lap)
            ex = X.ORBextractor(nf, 1.2, 8, 20, 7, max_width=cols, max_height=rows)
        except X.OrbxError:
            continue          # geometry the library rejects (a level smaller than one cell)
        mono, k, d, lvl = ex(img, None, lap)
        check_stages(ex, o, lvl)
        assert_same_result((mono, k, d), want, "random case %d: %dx%d nf=%d %s lap=%s" % (t, cols, rows, nf, variant, lap))


def test_device_path_with_padded_rows_and_frames():
    import torch
    B, rows, cols, stride, fstride = 3, 300, 400, 448, 448 * 310
    frames = synth.frames("textured", 7, B, rows, cols)
    buf = np.zeros((B, fstride), np.uint8)
    for f in range(B):
        buf[f, :rows * stride].reshape(rows, stride)[:, :cols] = frames[f]
    ex = X.ORBextractor(600, max_width=cols, max_height=rows, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(buf).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_lk = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_lc = torch.zeros((B, 8), dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, B, rows, cols, d_k, d_d, d_n, d_m, cap, stride=stride, frame_stride=fstride,
                            lapping=(50, 250), d_level_kps=d_lk, d_level_counts=d_lc)
    ex.synchronize()
    n = d_n.cpu().numpy()
    for f in range(B):
        o, want = oracle_run(frames[f], 600, (50, 250))
        k = d_k[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
        assert_same_result((int(d_m[f]), k, d_d[f, :n[f]].cpu().numpy()), want, "padded frame %d" % f)
        assert d_lc[f].cpu().numpy().tolist() == [len(o.level_keypoints(l)) for l in range(8)]


@pytest.mark.parametrize("form", ["default", "ORBX_PYR_COLS=0", "ORBX_PYR_COLS=1,ORBX_PYR_COL_PX=112", "ORBX_PATCH_BLUR=1,ORBX_BLUR_SPLIT=3"])
@pytest.mark.parametrize("B,offset,stride", [(1, 1, 643), (2, 3, 641), (3, 2, 650), (1, 0, 640)])
def test_device_path_with_unaligned_pointer_and_stride(B, offset, stride, form, monkeypatch):
    """cv::Mat ROIs handed over on the device: a base pointer and a row step that are not multiples of 4 (every pyramid form stages
    through aligned dwords when it can and must fall back to byte reads here, never reading past a row of the caller's buffer)."""
    import torch
    if form != "default":
        for kv in form.split(","):
            monkeypatch.setenv(*kv.split("="))
    rows, cols = 480, 640
    fstride = stride * rows + 5
    frames = synth.frames("noise", 31, B, rows, cols)
    buf = np.full(offset + B * fstride, 255, np.uint8)       # 255 outside the images: a stray read would change FAST scores
    for f in range(B):
        buf[offset + f * fstride: offset + f * fstride + rows * stride].reshape(rows, stride)[:, :cols] = frames[f]
    # the buffer ends with the last image row's last pixel (+ padding of the frame stride): nothing readable behind it
    ex = X.ORBextractor(1000, max_batch=B)
    cap = ex.capacity
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    d_buf = torch.from_numpy(buf).cuda()
    d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_buf.data_ptr() + offset, B, rows, cols, d_k, d_d, d_n, d_m, cap, stride=stride, frame_stride=fstride)
    ex.synchronize()
    n = d_n.cpu().numpy()
    for f in range(B):
        o, want = oracle_run(frames[f], 1000)
        k = d_k[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28).copy().view(X.KEYPOINT_DTYPE).reshape(-1)
        assert_same_result((int(d_m[f]), k, d_d[f, :n[f]].cpu().numpy()), want, "unaligned frame %d" % f)
        for l in (0, 1, 4, 7):
            assert np.array_equal(ex.image_pyramid_level(l, frame=f, bordered=True), o.level(l, bordered=True)), "level %d" % l


def test_profile_api_and_algorithmic_bytes():
    ex = X.ORBextractor(1000, max_batch=4)
    fr = synth.frames("noise", 0, 4, 480, 640)
    ex.profile(True)
    ex.extract_batch(fr)
    ex.extract_batch(fr)
    p = ex.profile_read()
    # the slots are keyed by the kernels that ran, as rocprofv3 names them: the region-major pyramid and the resident 256-thread quad-tree here
    oct_keys = [k for k in p if k.startswith("k_octree_")]
    assert p["k_fast"][1] == 2 and p["k_pyr_cols"][1] == 2 and oct_keys == ["k_octree_256r"] and p[oct_keys[0]][1] == 2 and p["k_fast"][0] > 0
    assert "k_resize" not in p and "k_octree" not in p
    assert p["batch_total"][0] >= p["k_fast"][0]
    ex.profile(False)
    assert ex.algorithmic_bytes(480, 640, 1000) == 307200 + 2 * 950532 + 60000
    assert ex.algorithmic_bytes(1080, 1920, 2000) == 2073600 + 2 * 6419321 + 120000


def test_fast_on_dense_natural_and_sparse_content():
    # one FAST kernel serves every content (rounds 1-3 switched to a prefilter + compaction variant on streams with few candidates; it lost to
    # the packed pass on every content once that evaluated one polarity per pixel, and was removed): dense, natural and sparse frames through
    # ONE handle, twice each (a stream that changes content keeps giving the reference result)
    ex = X.ORBextractor(1000, max_batch=4)
    for variant in ("natural", "noise", "sparse", "natural"):
        fr = synth.frames(variant, 3, 4, 480, 640)
        for rep in range(2):
            out = ex.extract_batch(fr)
            ex.synchronize()
        for f in (0, 3):
            o, want = oracle_run(fr[f])
            assert_same_result(out[f][:3], want, "%s frame %d" % (variant, f))


@pytest.mark.parametrize("route", ["default", "shared-queue", "own-stream"])
def test_async_host_api_with_two_handles_and_pinned_input(route):
    """(round 6) route: input copies of 16 MiB and more go through ONE copy queue per device shared by its handles and bring their results back by DMA,
    smaller ones stay on the handle's stream and bring them back with a copy kernel (orbx_api.cpp: uploadFrames, k_copy.hip) - "shared-queue" /
    "own-stream" force either route for every batch of this test through the test aid `shared_upload_bytes`.
    orbx_extract_batch_begin / _end called directly (the test was lost with the prefilter variant): two handles in flight, a
    second begin on a busy handle and an end without a begin are errors, and every branch of the upload - pinned memory copied as it lies,
    pageable memory staged through the handle's pinned block, padded rows as 2-D copies - gives the reference result."""
    if route != "default":
        X.debug_set_option("shared_upload_bytes", 0 if route == "shared-queue" else 1 << 30)      # (conftest.py resets the aids after the test)
    B = 3
    fa, fb = synth.frames("textured", 200, B, 480, 640), synth.frames("noise", 300, B, 480, 640)
    pa, pb = X.pinned_empty(fa.shape), X.pinned_empty(fb.shape)
    pa[...] = fa; pb[...] = fb
    a, b = X.ORBextractor(1000, max_batch=B), X.ORBextractor(1000, max_batch=B)
    a.extract_batch_begin(pa)                    # pinned, tight rows, B > 1: one asynchronous copy straight from the caller's memory
    b.extract_batch_begin(pb)                    # both batches in flight
    with pytest.raises(X.OrbxError):
        a.extract_batch_begin(pa)                # one batch per handle
    ra, rb = a.extract_batch_end(), b.extract_batch_end()
    with pytest.raises(X.OrbxError):
        a.extract_batch_end()                    # nothing in flight any more
    for f in range(B):
        assert_same_result(ra[f], oracle_run(fa[f])[1], "handle a frame %d" % f)
        assert_same_result(rb[f], oracle_run(fb[f])[1], "handle b frame %d" % f)
    # pageable memory with padded rows (a cv::Mat ROI): gathered row by row in the handle's pinned staging, one copy
    wide = np.zeros((B, 480, 704), np.uint8); wide[:, :, 13:653] = fb
    a.extract_batch_begin(wide[:, :, 13:653])
    for f, r in enumerate(a.extract_batch_end()):
        assert_same_result(r, oracle_run(fb[f])[1], "pageable padded frame %d" % f)
    # pinned memory with padded rows: one 2-D copy per frame
    pw = X.pinned_empty(wide.shape); pw[...] = 0; pw[:, :, 7:647] = fa
    b.extract_batch_begin(pw[:, :, 7:647])
    for f, r in enumerate(b.extract_batch_end()):
        assert_same_result(r, oracle_run(fa[f])[1], "pinned padded frame %d" % f)
    # pageable memory past the staging block (8 MB): the runtime's own staged copy, tight rows, and padded rows as 2-D copies
    Bn = 30
    big = synth.frames("noise", 400, Bn, 480, 640)
    c = X.ORBextractor(1000, max_batch=Bn)
    c.extract_batch_begin(big)
    rc = c.extract_batch_end()
    bigw = np.zeros((Bn, 480, 672), np.uint8); bigw[:, :, 32:] = big
    c.extract_batch_begin(bigw[:, :, 32:])
    rw = c.extract_batch_end()
    for f in (0, 17, Bn - 1):
        want = oracle_run(big[f])[1]
        assert_same_result(rc[f], want, "large pageable frame %d" % f)
        assert_same_result(rw[f], want, "large pageable padded frame %d" % f)
    # the ping-pong a host-fed stream runs (bench.py: host_to_host_fps): two handles alternating for a dozen batches whose content changes every time,
    # every batch's every frame against the single-frame result of the same extractor parameters; the views are read before the handle is used again
    Bp = 6
    stream = [synth.frames("noise" if i % 2 else "natural", 500 + i, Bp, 480, 640) for i in range(6)]
    ref = X.ORBextractor(1000, max_batch=Bp)
    want = [ref.extract_batch(fr) for fr in stream]
    pins = [X.pinned_empty(stream[0].shape), X.pinned_empty(stream[0].shape)]
    hs = [X.ORBextractor(1000, max_batch=Bp), X.ORBextractor(1000, max_batch=Bp)]
    got = [None] * len(stream)

    def collect(view):      # (kps[B, cap], desc[B, cap, 32], n[B], mono[B]) views of the pinned slab -> copies, before the handle is used again
        kps, desc, n, mono = view
        return [(int(mono[f]), kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(Bp)]

    pins[0][...] = stream[0]
    hs[0].extract_batch_begin(pins[0])
    for i in range(1, len(stream)):
        pins[i & 1][...] = stream[i]
        hs[i & 1].extract_batch_begin(pins[i & 1])
        got[i - 1] = collect(hs[(i - 1) & 1].extract_batch_end_view())
    got[-1] = collect(hs[(len(stream) - 1) & 1].extract_batch_end_view())
    for i in range(len(stream)):
        for f in range(Bp):
            assert_same_result(got[i][f], want[i][f][:3], "%s: ping-pong batch %d frame %d" % (route, i, f))
    X.pinned_free(pa); X.pinned_free(pb); X.pinned_free(pw); X.pinned_free(pins[0]); X.pinned_free(pins[1])


@pytest.mark.parametrize("switch,value", [("ORBX_OCT_THREADS", "256"), ("ORBX_OCT_THREADS", "512"), ("ORBX_OCT_THREADS", "1024"),
                                          ("ORBX_RESIZE_BYTEWISE", "1"), ("ORBX_PYR_COLS", "0"), ("ORBX_PYR_COLS,ORBX_RESIZE_BYTEWISE", "0,1")] +
                                         [("aid:pyr_cols_shape", str(v)) for v in (1, 4, 6)] +     # every workgroup shape of k_pyr_cols
                                         [("ORBX_FAST_WIDE", "0"), ("ORBX_FAST_WIDE", "1")])      # FAST: a wave / a workgroup per cell
def test_tuning_switches_do_not_change_results(switch, value, monkeypatch):
    # the quad-tree kernel exists in three workgroup sizes, the resize kernel in a packed and a byte-gather form, and the pyramid is one
    # launch region by region (k_pyr_cols) or one launch per level; the host picks by batch size / image area / tap geometry, and every
    # choice must give the reference result
    for sw, v in zip(switch.split(","), value.split(",")):
        if sw.startswith("aid:"):
            X.debug_set_option(sw[4:], int(v))      # (test aids are not environment variables; conftest.py resets them after the test)
        else:
            monkeypatch.setenv(sw, v)
    for shape, nf, variant in (((480, 640), 1000, "noise"), ((333, 517), 700, "textured"), ((480, 640), 1200, "natural")):
        img = synth.frames(variant, 21, 1, *shape)[0]
        o, want = oracle_run(img, nf)
        ex = X.ORBextractor(nf, max_width=shape[1], max_height=shape[0])
        mono, k, d, lvl = ex(img)
        check_stages(ex, o, lvl)
        assert_same_result((mono, k, d), want, "%s=%s %s" % (switch, value, variant))
    # tight clusters: a node deeper than the dense phase's leaf grid has to split, so the keys are gathered late
    rng = np.random.default_rng(4)
    img = np.full((480, 640), 90, np.uint8)
    for cy, cx in [(100, 120), (300, 500), (240, 320), (400, 90)]:
        ys = np.clip(cy + rng.integers(-24, 24, 900), 20, 459); xs = np.clip(cx + rng.integers(-24, 24, 900), 20, 619)
        img[ys, xs] = rng.integers(150, 255, 900)
    o, want = oracle_run(img, 3000)
    ex = X.ORBextractor(3000)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl)
    assert_same_result((mono, k, d), want, "%s=%s clustered" % (switch, value))


@pytest.mark.parametrize("B,split", [(72, False), (136, False), (300, False), (300, True), (257, True), (-136, False), (-300, True), (300, "stagger"), (257, "stagger"),
                                     (-301, "stagger")])
def test_large_batches_pick_other_quadtree_sizes_and_agree_with_single_frames(B, split, monkeypatch):
    # 8 levels x B workgroups: all resident with 1024 threads up to B = 64, with 512 up to 128, 256 threads above;
    # a large batch overlaps inside the call: its blur on a side stream beside FAST and the quad-tree (the default, round 4), the largest
    # also with staggered tails (ORBX_SPLIT=3: for every large batch; odd B: unequal halves); ORBX_SPLIT_MIN_MPX=0 makes these small frames
    # count as large, ORBX_SPLIT=0 turns every overlap off
    if B < 0:      # (negative: the pyramid as one launch per level, the form of large batches of large frames, instead of the region-major one)
        monkeypatch.setenv("ORBX_PYR_COLS", "0")
        B = -B
    if split:
        monkeypatch.setenv("ORBX_SPLIT_MIN_MPX", "0")
        if split == "stagger":      # FAST in two halves back to back, the first half's quad-tree + description on the internal stream under the second half's FAST
            monkeypatch.setenv("ORBX_SPLIT", "3")
    else:
        monkeypatch.setenv("ORBX_SPLIT", "0")
    fr = synth.frames("textured", 40, B, 240, 320)
    ex = X.ORBextractor(500, max_width=320, max_height=240, max_batch=B)
    out = ex.extract_batch(fr)
    one = X.ORBextractor(500, max_width=320, max_height=240)
    for f in (0, 1, B // 2, B - 1):
        mono, k, d, _ = one(fr[f])
        assert_same_result(out[f][:3], (mono, k, d), "frame %d of %d" % (f, B))
    o, want = oracle_run(fr[B - 1], 500)
    assert_same_result(out[B - 1][:3], want, "last frame of %d vs oracle" % B)
    # calls back to back on the same handle: the next call's pyramid and blur must not overtake this call's description (side streams)
    fr2 = np.ascontiguousarray(fr[::-1])
    ex.extract_batch(fr)
    out2 = ex.extract_batch(fr2)
    assert_same_result(out2[0][:3], want, "second call, frame 0 = the first call's last")
    assert_same_result(out2[B - 1][:3], out[0][:3], "second call, last frame = the first call's frame 0")


@pytest.mark.parametrize("shape,nf", [((480, 640), 10000), ((376, 1241), 10000), ((480, 640), 25000), ((480, 752), 6000)])
def test_initialisation_extractor_quotas(shape, nf):
    # the reference builds its monocular initialisation extractor with 5 * nFeatures (Tracking.cc:774): 5 x 2000 = 10000 for a
    # KITTI-style settings file (1241x376: four quad-tree roots), 5 x 1200 = 6000 for EuRoC.  Per-level quotas in the thousands:
    # the quad-tree's node arrays fill a CU's LDS (10000) or move to the HBM arena (25000)
    rows, cols = shape
    img = synth.frames("noise", 77, 1, rows, cols)[0]
    o, want = oracle_run(img, nf)
    ex = X.ORBextractor(nf, 1.2, 8, 20, 7, max_width=cols, max_height=rows, max_batch=2)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl)
    assert_same_result((mono, k, d), want, "%s nf=%d" % (shape, nf))
    assert len(k) > 0.9 * nf or nf > 20000          # noise frames fill the quota
    tex = synth.frames("textured", 78, 2, rows, cols)                     # two frames per call: two arena slices
    out = ex.extract_batch(tex)
    for f in range(2):
        assert_same_result(out[f][:3], oracle_run(tex[f], nf)[1], "textured %s nf=%d frame %d" % (shape, nf, f))


def test_scale_factor_two_uses_the_byte_gather_resize():
    # scaleFactor 2.0: the taps of four adjacent pixels span more than 8 source bytes, so the packed resize does not apply
    img = synth.frames("textured", 5, 1, 480, 640)[0]
    o, want = oracle_run(img, 600, (0, 1000), 3, 2.0, 20, 7)
    ex = X.ORBextractor(600, 2.0, 3, 20, 7, max_width=640, max_height=480)
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl, 3)
    assert_same_result((mono, k, d), want, "scale 2.0")


@pytest.mark.parametrize("B", [1, 2, 7, 8, 9, 12])
def test_small_batches_around_the_leaf_table_limit(B, monkeypatch):
    """Up to ORBX_LEAF_FRAMES frames per call (default 128; 8 here) k_fast's emit builds the quad-tree's leaf tables and k_octree starts from them;
    one frame more and the kernel sweeps the segments itself.  Both sides of the limit, twice in a row on one handle (the tables must be clean
    again), every frame against the oracle."""
    monkeypatch.setenv("ORBX_LEAF_FRAMES", "8")
    frames = np.concatenate([synth.frames("noise", 3, (B + 1) // 2, 480, 640), synth.frames("sparse", 5, B // 2 + 1, 480, 640)])[:B]
    ex = X.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=12)
    for rep in range(2):
        out = ex.extract_batch(frames if rep == 0 else frames[::-1].copy())
        src = frames if rep == 0 else frames[::-1]
        for f in range(B):
            o, want = oracle_run(src[f], 1000)
            assert_same_result(out[f][:3], want, "B=%d rep %d frame %d" % (B, rep, f))


@pytest.mark.parametrize("px", [40, 56, 80, 112])
def test_region_major_pyramid_every_cut(px, monkeypatch):
    """k_pyr_cols (one workgroup = one region of the image through every level) with each of the region sizes the host chooses from, forced
    for every batch size: every bordered level of every frame against the oracle (ORBextractor.cc:1164-1219 incl. copyMakeBorder), then the
    final arrays.  Shapes: the benchmark's, an odd one (partial dwords at the right edge, regions of unequal size), 1080p; a batch; twelve
    levels at scale 1.1 (more levels than the kernel's unrolled ones); scale 2.0 (byte-gather steps); two levels."""
    monkeypatch.setenv("ORBX_PYR_COLS", "1")
    monkeypatch.setenv("ORBX_PYR_COL_PX", str(px))
    for shape, nf, variant, kw in (((480, 640), 1000, "noise", {}), ((333, 517), 700, "textured", {}), ((1080, 1920), 2000, "natural", {}),
                                   ((480, 640), 900, "natural", dict(nlevels=12, sf=1.1)), ((600, 800), 500, "noise", dict(nlevels=3, sf=2.0)),
                                   ((241, 322), 300, "textured", dict(nlevels=2, sf=1.2))):
        nlevels, sf = kw.get("nlevels", 8), kw.get("sf", 1.2)
        img = synth.frames(variant, 51, 1, *shape)[0]
        o, want = oracle_run(img, nf, (0, 0), nlevels, sf)
        ex = X.ORBextractor(nf, sf, nlevels, 20, 7, max_width=shape[1], max_height=shape[0])
        mono, k, d, lvl = ex(img, None, (0, 0))
        check_stages(ex, o, lvl, nlevels)
        assert_same_result((mono, k, d), want, "px %d %s %s" % (px, shape, kw))
    frames = synth.frames("noise", 52, 5, 480, 640)
    ex = X.ORBextractor(1000, max_batch=5)
    out = ex.extract_batch(frames)
    for f in range(5):
        o, want = oracle_run(frames[f], 1000)
        assert_same_result(out[f][:3], want, "px %d batch frame %d" % (px, f))
        for l in range(8):
            assert np.array_equal(ex.image_pyramid_level(l, frame=f, bordered=True), o.level(l, bordered=True)), "frame %d level %d" % (f, l)


def test_wide_frame_with_tiny_quotas_keeps_the_first_pass_nodes():
    """A 922 x 200 frame has five to six quad-tree roots per level (nIni = round(width / height), ORBextractor.cc:548); with 55 features the
    per-level quotas are 5 .. 14, so what a level keeps is set by the unconditional first pass (up to four nodes per root), not by quota + 3
    (found by the round-4 soak, where the ORACLE's Python wrapper had sized its output for quota + 3)."""
    img = synth.frames("textured", 341, 1, 200, 922)[0]
    o, want = oracle_run(img, 55, (122, 689), 6, 1.2, 24, 24)
    assert len(want[1]) > 55 + 3 * 6
    ex = X.ORBextractor(55, 1.2, 6, 24, 24, max_width=922, max_height=200)
    mono, k, d, lvl = ex(img, None, (122, 689))
    check_stages(ex, o, lvl, 6)
    assert_same_result((mono, k, d), want, "wide frame, tiny quotas")


@pytest.mark.parametrize("pyr_cols", ["1", "0"])
def test_patch_blur_inside_the_description(pyr_cols, monkeypatch):
    """k_describe<PB> (the default of large batches of large frames; forced here for every size): no blurred level is made, every keypoint's
    37 x 37 patch is blurred out of its raw 43 x 43 tile of the bordered pyramid.  Final arrays and per-level keypoints against the oracle:
    keypoints on the FAST rectangle's rim (their patches reach 2 px into the REFLECT_101 frame), odd shapes, many and few levels, batches."""
    monkeypatch.setenv("ORBX_PATCH_BLUR", "1")
    monkeypatch.setenv("ORBX_BLUR_SPLIT", "0")      # (every level per keypoint; the split by level has its own test below)
    monkeypatch.setenv("ORBX_PYR_COLS", pyr_cols)
    rim = 0
    for shape, nf, variant, kw in (((480, 640), 1000, "noise", {}), ((333, 517), 700, "textured", {}), ((1080, 1920), 2000, "natural", {}),
                                   ((480, 640), 900, "natural", dict(nlevels=12, sf=1.1)), ((241, 322), 300, "sparse", dict(nlevels=2, sf=1.2)),
                                   ((482, 643), 1500, "noise", {})):
        nlevels, sf = kw.get("nlevels", 8), kw.get("sf", 1.2)
        img = synth.frames(variant, 57, 1, *shape)[0]
        o, want = oracle_run(img, nf, (0, 0), nlevels, sf)
        ex = X.ORBextractor(nf, sf, nlevels, 20, 7, max_width=shape[1], max_height=shape[0])
        mono, k, d, lvl = ex(img, None, (0, 0))
        assert ex.last_forms()[2] == 3
        with pytest.raises(X.OrbxError):
            ex.debug_blurred(0)            # no blurred level exists in this form, and the library says so
        check_stages(ex, o, lvl, nlevels)
        assert_same_result((mono, k, d), want, "patch blur %s %s" % (shape, kw))
        for l in range(nlevels):      # keypoints within 20 px of a level's edge: their raw tiles reach into the bordered frame (REFLECT_101)
            w_l, h_l = o.level_size(l)
            if len(lvl[l]):
                rim += int(((lvl[l]["x"] <= 20) | (lvl[l]["y"] <= 20) | (lvl[l]["x"] >= w_l - 21) | (lvl[l]["y"] >= h_l - 21)).sum())
    assert rim >= 20, rim
    frames = synth.frames("noise", 58, 9, 480, 640)
    ex = X.ORBextractor(1200, max_batch=9)
    out = ex.extract_batch(frames, lapping=(0, 0))
    for f in (0, 4, 8):
        o, want = oracle_run(frames[f], 1200, (0, 0))
        assert_same_result(out[f][:3], want, "patch blur batch frame %d" % f)


@pytest.mark.parametrize("split", [1, 2, 3, 4, 5, 7])
def test_patch_blur_split_by_level(split, monkeypatch):
    """Round 6: the blur per keypoint on the levels below the split, k_blur + the description from the blurred level on the levels from the split
    on (ORBextractor.cc:1122-1132 blurs a level, then describes its keypoints: both forms compute that level's blurred pixels with the same
    arithmetic).  Forced at every split level: final arrays and per-level keypoints against the oracle, the blurred levels that exist against
    the oracle's, the ones that do not refused; levels without keypoints, two levels (no level to split off), twelve levels, odd shapes, a
    batch with per-frame lapping areas, a second call on the same handle, and a back-only pass (orbx_compute_keypoints_octree) after a split call."""
    monkeypatch.setenv("ORBX_PATCH_BLUR", "1")
    monkeypatch.setenv("ORBX_BLUR_SPLIT", str(split))
    for shape, nf, variant, kw in (((480, 640), 1000, "noise", {}), ((333, 517), 700, "textured", {}), ((1080, 1920), 2000, "natural", {}),
                                   ((480, 640), 900, "natural", dict(nlevels=12, sf=1.1)), ((241, 322), 300, "sparse", dict(nlevels=2, sf=1.2)),
                                   ((482, 643), 1500, "noise", {})):
        nlevels, sf = kw.get("nlevels", 8), kw.get("sf", 1.2)
        img = synth.frames(variant, 61, 1, *shape)[0]
        o, want = oracle_run(img, nf, (0, 0), nlevels, sf)
        ex = X.ORBextractor(nf, sf, nlevels, 20, 7, max_width=shape[1], max_height=shape[0])
        assert "BLUR_SPLIT=%d(env)" % split in ex.policy()
        mono, k, d, lvl = ex(img, None, (0, 0))
        active = split < nlevels
        assert ex.last_forms()[2] == (5 if active else 3), (ex.last_forms(), split, nlevels)
        for l in range(nlevels):
            if active and l >= split:
                if len(o.level_keypoints(l)):      # (the oracle, like the reference, only blurs levels that hold keypoints, :1122-1127)
                    assert np.array_equal(ex.debug_blurred(l), o.blurred(l)), "split %d %s %s: blurred level %d" % (split, shape, kw, l)
            else:
                with pytest.raises(X.OrbxError):
                    ex.debug_blurred(l)            # blurred per keypoint: no such level exists, and the library says so
        check_stages(ex, o, lvl, nlevels)
        assert_same_result((mono, k, d), want, "split %d %s %s" % (split, shape, kw))
        mono2, k2, d2, lvl2 = ex(img, None, (0, 0))      # the same handle again
        assert_same_result((mono2, k2, d2), want, "split %d %s %s, second call" % (split, shape, kw))
        # a back-only pass describes from what the call in front of it left: the same split (levels below it have no blurred image)
        again = ex.ComputeKeyPointsOctTree()
        assert all(a.tobytes() == b.tobytes() for a, b in zip(again, lvl)), "split %d %s %s: ComputeKeyPointsOctTree after a split call" % (split, shape, kw)
        assert ex.last_forms()[2] == (5 if active else 3)
    B = 9
    frames = np.concatenate([synth.frames("noise", 62, 4, 480, 640), synth.frames("natural", 63, 3, 480, 640), synth.frames("sparse", 64, 2, 480, 640)])
    lap = [(0, 1000) if f % 2 == 0 else (100 + 7 * f, 400) for f in range(B)]
    ex = X.ORBextractor(1200, max_batch=B)
    out = ex.extract_batch(frames, lap)
    assert ex.last_forms()[2] == 5
    for f in range(B):
        o, want = oracle_run(frames[f], 1200, lap[f])
        assert_same_result(out[f][:3], want, "split %d batch frame %d" % (split, f))
        assert [len(a) for a in out[f][3]] == [len(o.level_keypoints(l)) for l in range(8)]


def test_patch_blur_split_in_a_large_batch_with_every_overlap(monkeypatch):
    """The split inside the launch DAG of large batches: k_blur of the coarse levels on the side stream beside FAST + quad-tree, the fine levels'
    description in front of the join, staggered tails (two halves, each with its own two description launches); 300 frames of 320x240 count
    as a large batch with ORBX_SPLIT_MIN_MPX=0.  Every frame against the serial single-frame result, frames spread over the batch against the oracle."""
    monkeypatch.setenv("ORBX_PATCH_BLUR", "1")
    monkeypatch.setenv("ORBX_BLUR_SPLIT", "3")
    B = 301
    frames = np.concatenate([synth.frames("noise", 71, 150, 240, 320), synth.frames("natural", 72, B - 150, 240, 320)])
    results = {}
    for mode in ("0", "1", "3"):
        monkeypatch.setenv("ORBX_SPLIT", mode)
        monkeypatch.setenv("ORBX_SPLIT_MIN_MPX", "0")
        ex = X.ORBextractor(500, max_width=320, max_height=240, max_batch=B)
        results[mode] = ex.extract_batch(frames)
        assert ex.last_forms()[2] == 5, ex.last_forms()
    for f in range(B):
        for mode in ("1", "3"):
            a, b = results[mode][f], results["0"][f]
            assert a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]), "ORBX_SPLIT=%s frame %d differs from the serial launches" % (mode, f)
    for f in (0, 1, 149, 150, 151, 299, 300):
        o, want = oracle_run(frames[f], 500)
        assert_same_result(results["3"][f][:3], want, "frame %d" % f)


def test_fast_with_a_workgroup_per_cell_in_batches(monkeypatch):
    """k_fast_wide (four waves per FAST cell; default only while a call holds few cells) forced for a batch: candidates IN ORDER, every stage and the
    final arrays of every frame against the oracle — dense, sparse (cells that fall back to minThFAST, empty cells) and natural content, with
    and without the leaf tables (nine frames are past ORBX_LEAF_FRAMES = 8)."""
    monkeypatch.setenv("ORBX_LEAF_FRAMES", "8")
    monkeypatch.setenv("ORBX_FAST_WIDE", "1")
    for B in (3, 9):
        frames = np.concatenate([synth.frames("noise", 7, B // 3, 480, 640), synth.frames("sparse", 8, B // 3, 480, 640), synth.frames("natural", 9, B - 2 * (B // 3), 480, 640)])
        ex = X.ORBextractor(1000, max_batch=B)
        out = ex.extract_batch(frames)
        for f in range(B):
            o, want = oracle_run(frames[f], 1000)
            assert_same_result(out[f][:3], want, "B=%d frame %d" % (B, f))
            for l in range(8):
                assert np.array_equal(ex.debug_candidates(l, f), o.candidates(l)), "candidates of level %d, frame %d" % (l, f)


@pytest.mark.parametrize("B", [128, 130])
def test_the_default_leaf_table_limit(B):
    """The default limit (128 frames per call): a call at the limit builds the leaf tables for every frame, a larger one for none; twice on one handle;
    frames spread over the batch against the oracle, all frames of both calls against each other."""
    frames = np.concatenate([synth.frames("noise", 11, B // 2, 240, 320), synth.frames("natural", 12, B - B // 2, 240, 320)])
    ex = X.ORBextractor(500, max_width=320, max_height=240, max_batch=B)
    a = ex.extract_batch(frames)
    b = ex.extract_batch(frames)
    for f in range(B):
        assert a[f][0] == b[f][0] and np.array_equal(a[f][1], b[f][1]) and np.array_equal(a[f][2], b[f][2]), "frame %d differs between two calls" % f
    for f in (0, 1, B // 2 - 1, B // 2, B - 2, B - 1):
        o, want = oracle_run(frames[f], 500)
        assert_same_result(a[f][:3], want, "B=%d frame %d" % (B, f))


@pytest.mark.parametrize("shape,nf", [((1944, 2592), 3000), ((3000, 4000), 5000), ((480, 4000), 1500), ((4000, 6000), 5000), ((300, 9000), 2500), ((4200, 2200), 3000), ((600, 4500), 30000)])
def test_large_and_very_wide_images(shape, nf):
    """5- and 12-megapixel frames and a 4000-px-wide strip (several quad-tree roots, hundreds of pyramid regions, thousands of FAST cells), and -
    round 6 - frames BEYOND 4096 px (the reference has no size limit, ORBextractor.cc:1171): a 24-megapixel 6000 x 4000 frame, a 9000-px strip
    with 33 quad-tree roots, a 4200-px-high frame; their candidates travel in the two-dword format, the quad-tree runs in its ...b builds
    (orbx_device.hpp: CandFmt); 4500 x 600 with 30 000 features takes the build whose node arrays live in an HBM arena (`k_octree_1024gb`).  Every
    stage and the final arrays of one frame against the oracle."""
    img = synth.frames("natural", 77, 1, *shape)[0]
    o, want = oracle_run(img, nf)
    ex = X.ORBextractor(nf, max_width=shape[1], max_height=shape[0])
    mono, k, d, lvl = ex(img)
    check_stages(ex, o, lvl)
    assert_same_result((mono, k, d), want, str(shape))
    if max(shape) > 4096:
        # the same handle on a frame WITHIN 4096 px: one-dword candidates in the arenas sized for two; and two big frames in one call
        small = synth.frames("textured", 78, 1, min(shape[0], 480), min(shape[1], 640))[0]
        o2, want2 = oracle_run(small, nf)
        mono2, k2, d2, lvl2 = ex(small)
        check_stages(ex, o2, lvl2)
        assert_same_result((mono2, k2, d2), want2, "%s on the handle of %s" % (small.shape, shape))
        if shape[0] * shape[1] <= 4_000_000:
            two = np.stack([img, img[::-1].copy()])
            ex2 = X.ORBextractor(nf, max_width=shape[1], max_height=shape[0], max_batch=2)
            out = ex2.extract_batch(two)
            assert_same_result(out[0][:3], want, "%s, frame 0 of two" % (shape,))
            assert_same_result(out[1][:3], oracle_run(two[1], nf)[1], "%s, frame 1 of two" % (shape,))


def test_frames_beyond_16384_px_are_refused():
    with pytest.raises(X.OrbxError):
        X.ORBextractor(1000, max_width=16500, max_height=480)
