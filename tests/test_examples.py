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


def _build_multi(tmp_path):
    exe = str(tmp_path / "orbx_multi_gpu")
    libdir = os.path.dirname(X.library_path())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-pthread", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "orbx_multi_gpu.cpp"), "-o", exe, "-L" + libdir, "-lorbx", "-Wl,-rpath," + libdir])
    return exe


def test_multi_gpu_example_builds(tmp_path):
    assert os.path.exists(_build_multi(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [0, 3])
def test_multi_gpu_example_gathers_what_the_oracle_computes(tmp_path, ranks):
    """examples/orbx_multi_gpu.cpp: one C++ process, one std::thread + one handle per device, frames sharded by sharding.shard_range's
    rule, every rank's result slab copied into device 0's memory.  ranks = 0: one rank per device the box has (1 on the lease); ranks = 3: three
    ranks sharing the box's devices round-robin - the control flow of N > 1 (unequal shards of 10 frames: 4 + 3 + 3, double-buffered slabs, the
    waits before a slab is overwritten) on one GPU.  The first frame of EVERY rank's gathered slab against the oracle on that frame of the stream."""
    import oracle_lib as O
    from extractorb_amd import sharding, synth
    total = 10
    out = subprocess.check_output([_build_multi(tmp_path), str(total), "4", "480", "640", "1000", str(ranks)], text=True, timeout=600)
    head = re.search(r"devices=(\d+) ranks=(\d+) total_frames=(\d+)", out)
    ndev, nranks = int(head.group(1)), int(head.group(2))
    assert nranks == (ranks or ndev) and int(head.group(3)) == total
    assert float(re.search(r"frames_per_sec=([\d.]+)", out).group(1)) > 50
    per_rank = [float(v) for v in re.search(r"per_rank_ms_per_step=([\d.,]+)", out).group(1).split(",")]
    assert len(per_rank) == nranks and all(v > 0 for v in per_rank)
    rows = re.findall(r"rank=(\d+) device=(\d+) frames=(\d+) first_frame=(\d+) keypoints=(\d+) mono=(-?\d+) descriptor_fnv=(\d+)", out)
    assert len(rows) == nranks, out
    orc = O.Oracle(1000)
    for r, dev, nfr, first, n, mono, fnv in rows:
        lo, hi = sharding.shard_range(total, int(r), nranks)
        assert (int(first), int(nfr)) == (lo, hi - lo) and int(dev) == int(r) % ndev
        wm, wk, wd = orc.extract(synth.noise_frame(lo, 480, 640))
        s = 0
        for b in wd.reshape(-1).tolist():
            s = (s * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
        assert (int(n), int(mono), int(fnv)) == (len(wk), wm, s), "rank %s: frame %d of the stream as gathered on device 0" % (r, lo)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        open(os.path.join(ROOT, "gpurun_out", "multi_gpu_example_%d.txt" % nranks), "w").write(out)
