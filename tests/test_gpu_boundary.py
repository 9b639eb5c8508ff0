"""The boundary of round 4 on the GPU, through the C ABI: the reference class's two public stage methods (ComputePyramid,
ComputeKeyPointsOctTree: inc/ORBextractor.h:87-90), the one-copy pyramid fetch behind mvImagePyramid (:85), the host-in / host-out call of
one frame (results written by the kernels into the pinned result slab: no copy command), the result slab of a batch (one D2H copy), the
leaf-table guard, and the launch forms a call reports."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth
from helpers import assert_same_result, load_gray

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_run(img, nf=1000, lap=(0, 1000)):
    o = O.Oracle(nf, 1.2, 8, 20, 7)
    return o, o.extract(img, lap)


@pytest.mark.parametrize("shape,nf,variant", [((480, 640), 1000, "textured"), ((512, 512), 1500, "tum"), ((1080, 1920), 2000, "noise"), ((333, 517), 700, "sparse")])
def test_stage_methods_equal_the_oracle(shape, nf, variant):
    """main_orb_extractor.cpp:43-46: ComputePyramid(image), then ComputeKeyPointsOctTree(allKeypoints): level coordinates, angles set."""
    img = load_gray("tum_room4_gray.png") if variant == "tum" else synth.frames(variant, 5, 1, *shape)[0]
    o, _ = oracle_run(img, nf)
    ex = X.ORBextractor(nf, 1.2, 8, 20, 7, max_width=img.shape[1], max_height=img.shape[0])
    ex.ComputePyramid(img)
    for l, (got, gotb) in enumerate(zip(ex.fetch_pyramid(), ex.fetch_pyramid(bordered=True))):      # mvImagePyramid right after ComputePyramid
        assert np.array_equal(got, o.level(l)) and np.array_equal(gotb, o.level(l, bordered=True)), "level %d" % l
    lvl = ex.ComputeKeyPointsOctTree()
    assert [len(x) for x in lvl] == [len(o.level_keypoints(l)) for l in range(8)]
    for l in range(8):
        assert lvl[l].tobytes() == o.level_keypoints(l).tobytes(), "level %d" % l
    if variant == "tum":
        assert sum(len(x) for x in lvl) == 1420          # img_folder/Screenshot.png (tests/test_reference_pin.py)
    # the stages again on the same pyramid give the same answer; a full call afterwards is unaffected
    assert all(a.tobytes() == b.tobytes() for a, b in zip(ex.ComputeKeyPointsOctTree(), lvl))
    mono, k, d, lvl2 = ex(img)
    assert_same_result((mono, k, d), o.extract(img, (0, 1000)), "operator() after the stage methods")
    assert all(a.tobytes() == b.tobytes() for a, b in zip(lvl2, lvl))


def test_stage_method_on_the_pyramid_of_a_full_call_and_without_one():
    img = synth.frames("textured", 8, 1, 480, 640)[0]
    o, want = oracle_run(img)
    ex = X.ORBextractor(1000)
    with pytest.raises(X.OrbxError):
        ex.ComputeKeyPointsOctTree()                      # no pyramid yet
    ex(img)
    lvl = ex.ComputeKeyPointsOctTree()                    # the pyramid operator() left
    assert all(lvl[l].tobytes() == o.level_keypoints(l).tobytes() for l in range(8))
    # ... and frame 0 of a batch's
    frames = synth.frames("noise", 20, 3, 480, 640)
    exb = X.ORBextractor(1000, max_batch=3)
    exb.extract_batch(frames)
    o0, _ = oracle_run(frames[0])
    lvl = exb.ComputeKeyPointsOctTree()
    assert all(lvl[l].tobytes() == o0.level_keypoints(l).tobytes() for l in range(8))


def test_pyramid_fetch_of_every_frame_of_a_batch():
    frames = synth.frames("textured", 30, 5, 300, 400)
    ex = X.ORBextractor(500, max_width=400, max_height=300, max_batch=8)
    ex.extract_batch(frames)
    for f in (0, 3, 4):
        o, _ = oracle_run(frames[f], 500)
        for l, (a, b) in enumerate(zip(ex.fetch_pyramid(f), ex.fetch_pyramid(f, bordered=True))):
            assert np.array_equal(a, o.level(l)) and np.array_equal(b, o.level(l, bordered=True)), (f, l)
    with pytest.raises(X.OrbxError):
        ex.fetch_pyramid(5)                               # not a frame of the last batch


def _extract_view(ex, img, lap=(0, 1000), want_levels=1):
    L, h = ex._L, ex._h
    k, d, lk, lc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    n, mono = C.c_int(), C.c_int()
    rc = L.orbx_extract_view(h, img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], img.strides[0], lap[0], lap[1], want_levels, C.byref(k), C.byref(d),
                             C.byref(n), C.byref(mono), C.byref(lk), C.byref(lc))
    assert rc == 0, L.orbx_last_error(h)
    kk = np.frombuffer((C.c_uint8 * (28 * n.value)).from_address(k.value), X.KEYPOINT_DTYPE).copy()
    dd = np.frombuffer((C.c_uint8 * (32 * n.value)).from_address(d.value), np.uint8).reshape(-1, 32).copy()
    lv = None
    if want_levels:
        counts = np.frombuffer((C.c_int32 * ex.nlevels).from_address(lc.value), np.int32).copy()
        lv = np.frombuffer((C.c_uint8 * (28 * n.value)).from_address(lk.value), X.KEYPOINT_DTYPE).copy(), counts
    return mono.value, kk, dd, lv


@pytest.mark.parametrize("zero_copy", ["1", "0"])
def test_one_frame_host_call_forms(zero_copy, monkeypatch):
    """orbx_extract_view / orbx_extract on pageable, strided and pinned images: results written by the kernels into pinned host memory
    (default) or copied back in one piece (ORBX_ZERO_COPY=0) are the oracle's."""
    monkeypatch.setenv("ORBX_ZERO_COPY", zero_copy)
    big = synth.frames("textured", 60, 1, 500, 700)[0]
    sub = big[7:487, 13:653]                              # a cv::Mat ROI: 640x480, step 700, unaligned start
    o, want = oracle_run(np.ascontiguousarray(sub), 1000, (100, 300))
    ex = X.ORBextractor(1000)
    for img in (sub, np.ascontiguousarray(sub)):
        mono, k, d, (lk, counts) = _extract_view(ex, img, (100, 300))
        assert_same_result((mono, k, d), want, "view, zero copy %s" % zero_copy)
        assert counts.tolist() == [len(o.level_keypoints(l)) for l in range(8)]
        assert lk.tobytes() == b"".join(o.level_keypoints(l).tobytes() for l in range(8))
        mono, k, d, _ = _extract_view(ex, img, (100, 300), want_levels=0)
        assert_same_result((mono, k, d), want, "view without levels")
        assert_same_result(ex(img, None, (100, 300))[:3], want, "orbx_extract")
    pin = X.pinned_empty((480, 640))
    pin[...] = sub
    assert_same_result(_extract_view(ex, pin, (100, 300))[:3], want, "pinned image")
    X.pinned_free(pin)


def test_batch_results_come_back_in_one_slab():
    """A batch through the host-buffer path (one D2H copy of the result slab), with and without the per-level arrays, odd sizes."""
    frames = synth.frames("noise", 70, 5, 480, 640)
    ex = X.ORBextractor(1000, max_batch=8)
    out = ex.extract_batch(frames[:5], lapping=(0, 0))
    for f in range(5):
        o, want = oracle_run(frames[f], 1000, (0, 0))
        assert_same_result(out[f][:3], want, "frame %d" % f)
        assert all(out[f][3][l].tobytes() == o.level_keypoints(l).tobytes() for l in range(8))
    one = ex.extract_batch(frames[4:5])                   # one frame through the batch entry point (zero-copy slab again)
    assert_same_result(one[0][:3], oracle_run(frames[4])[1], "batch of one")


def test_a_call_that_dies_between_fast_and_quadtree_leaves_no_stale_leaf_tables():
    """the leaf tables k_fast fills are only zero again once k_octree has consumed them.  A call that returns in between
    (injected: the test aid "fail_after_fast" of orbx_debug_set_option, one shot) must not inflate the next call's counts."""
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
os.environ["ORBX_TEST_FAIL_AFTER_FAST"] = "1"      # (the environment cannot inject it any more: must have no effect)
import numpy as np
import extractorb_amd as X, oracle_lib as O
from extractorb_amd import synth
img = synth.frames("noise", 3, 2, 480, 640)
ex0 = X.ORBextractor(1000)
ex0(img[0])                                          # no failure: the aid is not reachable from the environment
assert "test_aids" not in ex0.policy(), ex0.policy()
X.debug_set_option("fail_after_fast", 1)
ex = X.ORBextractor(1000)
assert "fail_after_fast:1" in ex.policy(), ex.policy()
try:
    ex(img[0]); print("NOFAIL")
except X.OrbxError as e:
    assert "fail_after_fast" in str(e), e
mono, k, d, lvl = ex(img[1])
wm, wk, wd = O.Oracle(1000).extract(img[1], (0, 1000))
assert mono == wm and k.tobytes() == wk.tobytes() and np.array_equal(d, wd), "stale leaf tables"
print("OK")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "NOFAIL" not in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("shape,nf,B,form", [((480, 640), 1000, 1, 0), ((480, 640), 1000, 64, 0), ((720, 1280), 1500, 1, 0), ((1080, 1920), 2000, 1, 0),
                                             ((1080, 1920), 2000, 24, 1)])
def test_reported_launch_forms(shape, nf, B, form):
    """the published single-frame and traffic figures assume the region-major pyramid; a sizing regression that quietly
    falls back to the tile forms must fail a test, not just change a timing."""
    frames = synth.frames("noise", 0, min(B, 2), *shape)
    frames = np.concatenate([frames] * ((B + len(frames) - 1) // len(frames)))[:B]
    ex = X.ORBextractor(nf, max_width=shape[1], max_height=shape[0], max_batch=B)
    out = ex.extract_batch(frames)
    pyr, cut, blur = ex.last_forms()
    assert pyr == form, (pyr, cut, blur)
    if form == 0:
        assert cut in (40, 56, 80, 112)
    o, want = oracle_run(frames[B - 1], nf)
    assert_same_result(out[B - 1][:3], want, "%s x %d" % (shape, B))


@pytest.mark.parametrize("shape,nf,B", [((480, 640), 1000, 512), ((1080, 1920), 2000, 64), ((1080, 1920), 2000, 128)])
def test_the_benchmarks_own_batch_shapes(shape, nf, B):
    """BASELINE.md §3 / bench.py's lines under the DEFAULT launch policy: 512 x 640x480 x 1000 (region-major pyramid,
    256-thread queued quad-tree, no leaf tables, the blur on its side stream, staggered tails) and 64 / 128 x 1920x1080 x 2000 (per-level pyramid
    launches, 1024-thread queued quad-tree): first / middle / last frames of the batch against the oracle, as bench.py's `verified` does."""
    base = synth.frames("noise", 0, 16, *shape)
    frames = np.concatenate([base] * (B // 16))
    frames[B // 2 - 1] = synth.frames("textured", 1, 1, *shape)[0]      # (not only noise)
    ex = X.ORBextractor(nf, max_width=shape[1], max_height=shape[0], max_batch=B)
    out = ex.extract_batch(frames)
    picks = sorted({0, B // 2 - 1, B // 2, B - 1})
    for f in picks:
        o, want = oracle_run(frames[f], nf)
        assert_same_result(out[f][:3], want, "%s x %d frame %d (forms %s)" % (shape, B, f, ex.last_forms()))
    # the blur per keypoint inside k_describe: 1080p x 2000 on every level (form 3: the coarsest level's patches hold 0.99 of its pixels); 512 x 640x480 x
    # 1000 split by level (form 5, round 6: levels 0-2 per keypoint, k_blur for levels 3-7, whose patches hold 1.6-3.4 x the level's pixels)
    assert ex.last_forms()[2] == (5 if shape == (480, 640) else 3), ex.last_forms()
    assert X.load_library().orbx_debug_last_split_level(ex._h) == (3 if shape == (480, 640) else 0)
    out2 = ex.extract_batch(frames[::-1].copy())          # the same handle again: what the timed steps of bench.py do
    assert_same_result(out2[B - 1][:3], out[0][:3], "second call")


def test_first_call_on_fresh_handles_has_every_border_byte():
    """The first call of a NEW handle follows the geometry tables' upload and the arenas' zero fill, which go through the null stream while the
    kernels run on a non-blocking stream (orbx_create / installGeometry end with a device synchronisation: DESIGN.md section 4j).  Thirty fresh handles,
    three sizes, first call each: every bordered level (interior + its BORDER_REFLECT_101 frame) against the oracle."""
    import test_gpu_parity as P
    for rows, cols, nl, sf in ((432, 1014, 2, 1.5), (480, 640, 8, 1.2), (333, 517, 4, 1.3)):
        img = synth.frames("natural", 3, 1, rows, cols)[0]
        o, _ = P.oracle_run(img, 600, (0, 1000), nl, sf, 20, 7)
        want = [o.level(l, bordered=True) for l in range(nl)]
        for rep in range(10):
            ex = X.ORBextractor(600, sf, nl, 20, 7, max_width=cols, max_height=rows)
            ex(img)
            for l in range(nl):
                got = ex.image_pyramid_level(l, 0, bordered=True)
                assert np.array_equal(got, want[l]), "handle %d, %dx%d level %d: %d bytes of the bordered level differ" % (rep, cols, rows, l, int((got != want[l]).sum()))


def test_clock_probe_and_policy_string():
    """Round 5 introspection bench.py relies on: the shader-clock probe (one sleeping wave per CU stamps s_memtime against the 100-MHz s_memrealtime,
    asynchronous, on a stream of its own) reports a plausible clock beside a running batch, and the policy string lists the switches orbx_create read."""
    ex = X.ORBextractor(1000, max_batch=8)
    frames = synth.frames("noise", 5, 8, 480, 640)
    for slot in range(3):
        ex.clock_probe(slot)
        ex.extract_batch(frames)
    ghz = ex.clock_read(4)
    assert all(1.0 < v < 3.5 for v in ghz[:3]), ghz          # MI355X: up to 2.4 GHz
    assert ghz[3] == 0.0                                     # a slot never probed
    with pytest.raises(X.OrbxError):
        ex.clock_probe(64)
    pol = ex.policy()
    for key in ("SPLIT=1", "PATCH_BLUR=-1", "BLUR_SPLIT=-1", "LEAF_FRAMES=128", "ZERO_COPY=1"):
        assert key in pol, pol
    assert "(env)" not in pol and "test_aids" not in pol, pol
