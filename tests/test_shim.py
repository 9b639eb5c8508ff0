"""The C++ shim (include/orbx_extractor.hpp): compiles on CPU against stand-in cv types; on the GPU box the
compiled program drives the reference call shape of Frame::ExtractORB and is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import extractorb_amd as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "shim_test")
    libdir = os.path.dirname(X.library_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"), "-o", exe, "-L" + libdir, "-lorbx",
                           "-Wl,-rpath," + libdir])
    return exe


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
    l3 = np.frombuffer(raw[p:p + w * h], np.uint8).reshape(h, w)
    o = O.Oracle(1000)
    wm, wk, wd = o.extract(img, (100, 300))
    assert mono == wm and k.tobytes() == wk.tobytes() and np.array_equal(d, wd)
    assert counts.tolist() == [len(o.level_keypoints(l)) for l in range(8)]
    assert np.array_equal(l3, o.level(3))
