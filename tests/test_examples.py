"""examples/orbx_stream.cpp: a pure C++ host program on the C ABI.  It must build with hipcc here; on the GPU box its
first frame must equal the oracle's result for the same synthetic frame."""
import os
import re
import subprocess

import numpy as np
import pytest

import extractorb_amd as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "orbx_stream")
    libdir = os.path.dirname(X.library_path())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "orbx_stream.cpp"), "-o", exe, "-L" + libdir, "-lorbx",
                           "-Wl,-rpath," + libdir])
    return exe


def test_cpp_example_builds(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_example_matches_oracle(tmp_path):
    import oracle_lib as O
    from extractorb_amd import synth
    exe = _build(tmp_path)
    out = subprocess.check_output([exe, "4", "3", "480", "640", "1000"], text=True)
    n = int(re.search(r"frame0_keypoints=(\d+)", out).group(1))
    fnv = int(re.search(r"frame0_descriptor_fnv=(\d+)", out).group(1))
    assert float(re.search(r"frames_per_sec=([\d.]+)", out).group(1)) > 100
    frame = synth.noise_frame(0, 480, 640)          # the C++ generator restates this one
    mono, k, d = O.Oracle(1000).extract(frame)
    assert n == len(k)
    s = 0
    for b in d.reshape(-1).tolist():
        s = (s * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
    assert s == fnv
