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


def _build_latency(tmp_path):
    exe = str(tmp_path / "orbx_frame_latency")
    libdir = os.path.dirname(X.library_path())
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "orbx_frame_latency.cpp"),
                           "-o", exe, "-L" + libdir, "-lorbx", "-Wl,-rpath," + libdir])
    return exe


def test_frame_latency_example_builds(tmp_path):
    assert os.path.exists(_build_latency(tmp_path))


@pytest.mark.gpu
def test_frame_latency_example_runs(tmp_path):
    """examples/orbx_frame_latency.cpp: the drop-in class called frame by frame with pageable images from C++ (the figure DESIGN.md §5 quotes)."""
    out = subprocess.check_output([_build_latency(tmp_path), "480", "640", "1000", "100"], text=True)
    m = re.search(r"frame_call_us_median=([\d.]+).*keypoints=(\d+)", out)
    assert m and 10 < float(m.group(1)) < 2000 and int(m.group(2)) > 900, out
    (open(os.path.join(ROOT, "gpurun_out", "frame_latency.txt"), "w") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else open(os.devnull, "w")).write(out)
