"""The C++ shim (include/orbx_extractor.hpp): compiles on CPU against stand-in cv types; on the GPU box the
compiled program drives the reference call shape of Frame::ExtractORB and is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import extractorb_amd as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="shim_test"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(X.library_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe, "-L" + libdir, "-lorbx",
                           "-Wl,-rpath," + libdir])
    return exe


def _build_cv(tmp_path, name="cvtraits_typecheck"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(X.library_path())
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "tests", "cpp", "opencv_standin"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + libdir, "-lorbx", "-Wl,-rpath," + libdir])
    return exe


def test_drop_in_name_compiles_against_opencv_shaped_headers(tmp_path):
    """ORB_SLAM3::ORBextractor = BasicORBextractor<CvTraits>: the branch a maintainer with OpenCV compiles, type-checked here against a stand-in
    header with OpenCV 3's names (C++11, as the reference builds: CMakeLists.txt:10-12)."""
    assert os.path.exists(_build_cv(tmp_path))


def test_reference_demo_and_stereo_access_compile(tmp_path):
    """The reference's own call sequences against the drop-in header (C++11, -Wall -Werror): the demo's ComputePyramid + ComputeKeyPointsOctTree
    (main_orb_extractor.cpp:43-53) and Frame::ComputeStereoMatches' indexing of mvImagePyramid right after operator() (Frame.cc:820,905-932)."""
    assert os.path.exists(_build_cv(tmp_path, "reference_demo_sequence"))
    assert os.path.exists(_build_cv(tmp_path, "stereo_pyramid_access"))


@pytest.mark.gpu
def test_reference_demo_sequence_prints_1420(tmp_path):
    """main_orb_extractor.cpp:43-53 verbatim against the shim, on the frame and with the parameters of the reference's screenshot
    (tests/test_reference_pin.py): the program prints the reference's own line, and the per-level vectors and the pyramid equal the oracle's."""
    import oracle_lib as O
    from helpers import load_gray
    exe = _build_cv(tmp_path, "reference_demo_sequence")
    img = load_gray("tum_room4_gray.png")
    (tmp_path / "in.gray").write_bytes(img.tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(tmp_path / "in.gray"), str(img.shape[0]), str(img.shape[1]), "1500", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "ORB_SLAM3 has total 1420 keypoints"
    raw = out.read_bytes()
    total = int(np.frombuffer(raw[:4], np.int32)[0]); counts = np.frombuffer(raw[4:36], np.int32)
    assert total == 1420 == counts.sum()
    o = O.Oracle(1500, 1.2, 8, 20, 7)
    o.extract(img, (0, 1000))
    p = 36
    for l in range(8):
        k = np.frombuffer(raw[p:p + 28 * counts[l]], X.KEYPOINT_DTYPE); p += 28 * counts[l]
        assert k.tobytes() == o.level_keypoints(l).tobytes(), "level %d" % l      # level coordinates, octave, size, angle (ORBextractor.cc:872-887)
    for l in range(8):
        w, h = np.frombuffer(raw[p:p + 8], np.int32); p += 8
        assert np.array_equal(np.frombuffer(raw[p:p + w * h], np.uint8).reshape(h, w), o.level(l)); p += w * h
    assert p == len(raw)


@pytest.mark.gpu
def test_stereo_constructor_reads_mvImagePyramid_without_an_added_call(tmp_path):
    """Frame.cc:820,905-932 against the shim: two extractors, then rowRange / colRange windows of both eyes' mvImagePyramid with no call
    in between; windows and the REFLECT_101 frame around the views equal the oracle's levels."""
    import oracle_lib as O
    from extractorb_amd import synth
    exe = _build_cv(tmp_path, "stereo_pyramid_access")
    left = synth.frames("textured", 41, 1, 480, 640)[0]
    right = synth.frames("textured", 42, 1, 480, 640)[0]
    (tmp_path / "l.gray").write_bytes(left.tobytes()); (tmp_path / "r.gray").write_bytes(right.tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([exe, str(tmp_path / "l.gray"), str(tmp_path / "r.gray"), "480", "640", "1200", str(out)])
    raw = out.read_bytes()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    assert n > 800 and len(raw) == 4 + n * (12 + 121 + 121 + 1)
    oL, oR = O.Oracle(1200), O.Oracle(1200)
    oL.extract(left, (0, 0)); oR.extract(right, (0, 0))
    lv = [(oL.level(l), oR.level(l), oL.level(l, bordered=True)) for l in range(8)]
    p, seen = 4, set()
    for _ in range(n):
        octave, v, u = np.frombuffer(raw[p:p + 12], np.int32); p += 12
        il = np.frombuffer(raw[p:p + 121], np.uint8).reshape(11, 11); p += 121
        ir = np.frombuffer(raw[p:p + 121], np.uint8).reshape(11, 11); p += 121
        frame = raw[p]; p += 1
        assert np.array_equal(il, lv[octave][0][v - 5:v + 6, u - 5:u + 6]) and np.array_equal(ir, lv[octave][1][v - 5:v + 6, u - 5:u + 6])
        assert frame == lv[octave][2][19 + v, 0]
        seen.add(int(octave))
    assert seen == set(range(8))


@pytest.mark.gpu
def test_drop_in_name_runs_the_frame_call_shape(tmp_path):
    import oracle_lib as O
    from extractorb_amd import synth
    exe = _build_cv(tmp_path)
    img = synth.frames("textured", 12, 1, 480, 640)[0]
    (tmp_path / "in.gray").write_bytes(img.tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([exe, str(tmp_path / "in.gray"), "480", "640", "1000", str(out)])
    raw = out.read_bytes()
    mono, n = np.frombuffer(raw[:8], np.int32)
    k = np.frombuffer(raw[8:8 + 28 * n], X.KEYPOINT_DTYPE)
    d = np.frombuffer(raw[8 + 28 * n:8 + 60 * n], np.uint8).reshape(n, 32)
    wm, wk, wd = O.Oracle(1000).extract(img, (0, 1000))
    assert mono == wm and k.tobytes() == wk.tobytes() and np.array_equal(d, wd)


def test_thread_program_compiles(tmp_path):
    assert os.path.exists(_build(tmp_path, "shim_threads_test"))


def test_shim_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    # no GPU here: the program must fail loudly at orbx_create, not fall back
    import torch
    if not torch.cuda.is_available():
        img = tmp_path / "in.gray"
        np.zeros((240, 320), np.uint8).tofile(img)
        r = subprocess.run([exe, str(img), "240", "320", "300", "0", "1000", str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_shim_matches_oracle(tmp_path):
    import oracle_lib as O
    from extractorb_amd import synth
    exe = _build(tmp_path)
    img = synth.frames("textured", 9, 1, 480, 640)[0]
    (tmp_path / "in.gray").write_bytes(img.tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([exe, str(tmp_path / "in.gray"), "480", "640", "1000", "100", "300", str(out)])
    raw = out.read_bytes()
    mono, n = np.frombuffer(raw[:8], np.int32)
    k = np.frombuffer(raw[8:8 + 28 * n], X.KEYPOINT_DTYPE)
    d = np.frombuffer(raw[8 + 28 * n:8 + 60 * n], np.uint8).reshape(n, 32)
    p = 8 + 60 * n
    counts = np.frombuffer(raw[p:p + 32], np.int32); p += 32
    w, h = np.frombuffer(raw[p:p + 8], np.int32); p += 8
    l3 = np.frombuffer(raw[p:p + w * h], np.uint8).reshape(h, w); p += w * h
    l3b = np.frombuffer(raw[p:p + (w + 38) * (h + 38)], np.uint8).reshape(h + 38, w + 38); p += (w + 38) * (h + 38)
    assert p == len(raw)
    o = O.Oracle(1000)
    wm, wk, wd = o.extract(img, (100, 300))
    assert mono == wm and k.tobytes() == wk.tobytes() and np.array_equal(d, wd)
    assert counts.tolist() == [len(o.level_keypoints(l)) for l in range(8)]
    assert np.array_equal(l3, o.level(3)) and np.array_equal(l3b, o.level(3, bordered=True))      # the view and the (w+38) x (h+38) buffer it sits in (ORBextractor.cc:1173-1177)


@pytest.mark.gpu
def test_two_extractors_on_two_threads_match_oracle(tmp_path):
    """Frame.cc:109-112: mpORBextractorLeft and mpORBextractorRight are called from two std::threads at once.  Two shim instances
    (two handles, two streams), different images, 20 rounds: every round equals the first, and both eyes equal the oracle.
    Both instances start with 320x240 arenas and grow to the 640x480 images inside their first call."""
    import oracle_lib as O
    from extractorb_amd import synth
    exe = _build(tmp_path, "shim_threads_test")
    left = synth.frames("textured", 31, 1, 480, 640)[0]
    right = synth.frames("noise", 32, 1, 480, 640)[0]
    (tmp_path / "l.gray").write_bytes(left.tobytes()); (tmp_path / "r.gray").write_bytes(right.tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([exe, str(tmp_path / "l.gray"), str(tmp_path / "r.gray"), "480", "640", "1200", "20", str(out)])
    raw = out.read_bytes()
    p = 0
    for img in (left, right):
        mono, n = np.frombuffer(raw[p:p + 8], np.int32); p += 8
        k = np.frombuffer(raw[p:p + 28 * n], X.KEYPOINT_DTYPE); p += 28 * n
        d = np.frombuffer(raw[p:p + 32 * n], np.uint8).reshape(n, 32); p += 32 * n
        wm, wk, wd = O.Oracle(1200).extract(img, (0, 0))
        assert mono == wm and k.tobytes() == wk.tobytes() and np.array_equal(d, wd)
    assert p == len(raw)


@pytest.mark.gpu
def test_shim_grows_its_arenas(tmp_path):
    """The reference's operator() takes any image size; the shim pre-sizes for 1920x1080 and re-creates its arenas for more."""
    import oracle_lib as O
    from extractorb_amd import synth
    exe = _build(tmp_path)
    img = synth.frames("textured", 3, 1, 1100, 2000)[0]              # larger than the default arenas in both dimensions
    (tmp_path / "in.gray").write_bytes(img.tobytes())
    out = tmp_path / "out.bin"
    env = dict(os.environ, ORBX_SHIM_TEST_SMALL_ARENAS="1")           # shim_test then constructs with the default 1920x1080
    subprocess.check_call([exe, str(tmp_path / "in.gray"), "1100", "2000", "1500", "0", "1000", str(out)], env=env)
    raw = out.read_bytes()
    mono, n = np.frombuffer(raw[:8], np.int32)
    k = np.frombuffer(raw[8:8 + 28 * n], X.KEYPOINT_DTYPE)
    d = np.frombuffer(raw[8 + 28 * n:8 + 60 * n], np.uint8).reshape(n, 32)
    wm, wk, wd = O.Oracle(1500).extract(img, (0, 1000))
    assert mono == wm and k.tobytes() == wk.tobytes() and np.array_equal(d, wd)
