"""Case files for the host emulation of the quad-tree kernel (tools/octree_emu/octree_emu.cpp): inputs = the oracle's FAST candidates
per level (the reference's vToDistributeKeys), expected = the oracle's DistributeOctTree result per level, in list order."""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "build")


def build(kinds=("plain", "asan", "tsan")):
    subprocess.check_call(["make", "-s", "-C", HERE] + ["build/emu_%s" % k for k in kinds])
    return {k: os.path.join(BUILD, "emu_%s" % k) for k in kinds}


def write_case(path, oracle, rows, cols, nfeatures, scale, nlevels, threads, roomy, lapping=(0, 1000), poison=0xA5, max_rows=None, max_cols=None,
               leaf_tables=False):
    """`oracle`: an oracle_lib.Oracle that has just extracted the image."""
    with open(path, "wb") as f:
        f.write(struct.pack("<12if3i", 0x4f435445, nfeatures, nlevels, rows, cols, max_rows or rows, max_cols or cols, threads, int(roomy),
                            int(lapping[0]), int(lapping[1]), poison, scale, int(leaf_tables), 0, 0))
        for l in range(nlevels):
            c = oracle.candidates(l)
            w = (c["x"].astype(np.uint32) | (c["y"].astype(np.uint32) << 12) | (c["response"].astype(np.uint32) << 24)).astype("<u4")
            f.write(struct.pack("<i", len(w)))
            f.write(w.tobytes())


def read_result(path, nlevels):
    out = []
    with open(path, "rb") as f:
        for _ in range(nlevels):
            n, nlap = struct.unpack("<ii", f.read(8))
            sel = np.frombuffer(f.read(8 * n), dtype="<u4").reshape(n, 2)
            out.append((sel, nlap))
    return out


def expected_level(oracle, level, lapping):
    """(x | y << 16, response) per kept keypoint in the reference's list order - the selection entry's .x and the top byte of its .y
    (extractorb_amd/csrc/orbx_device.hpp) - and the lapping flags (ORBextractor.cc:1143-1147)."""
    k = oracle.level_keypoints(level)
    w = (k["x"].astype(np.uint32) | (k["y"].astype(np.uint32) << 16), k["response"].astype(np.uint32))
    xs = k["x"] if level == 0 else (k["x"] * oracle.scale_factors[level]).astype(np.float32)
    lap = (xs >= np.float32(lapping[0])) & (xs <= np.float32(lapping[1]))
    return w, lap


def run_case(binary, case_path, out_path, timeout=600, env=None):
    """Returns (returncode, stderr).  A hang (a wave collective some lane never reaches) ends in the timeout."""
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([binary, case_path, out_path], capture_output=True, text=True, timeout=timeout, env=e)
    return r.returncode, r.stderr


def check(oracle, nlevels, lapping, result):
    """Bit-for-bit comparison of the emulated kernel's selection with the oracle; returns a list of mismatch descriptions."""
    bad = []
    for l in range(nlevels):
        (want, want_resp), lap = expected_level(oracle, l, lapping)
        sel, nlap = result[l]
        if len(sel) != len(want):
            bad.append("level %d: %d kept, oracle %d" % (l, len(sel), len(want)))
            continue
        if not np.array_equal(sel[:, 0], want):
            i = int(np.nonzero(sel[:, 0] != want)[0][0])
            bad.append("level %d: list position %d differs (0x%08x vs 0x%08x)" % (l, i, sel[i, 0], want[i]))
        if not np.array_equal(sel[:, 1] >> 24, want_resp):
            bad.append("level %d: responses differ" % l)
        if not np.array_equal(((sel[:, 1] >> 23) & 1).astype(bool), lap) or nlap != int(lap.sum()):
            bad.append("level %d: lapping flags / count differ" % l)
        rank = np.cumsum(lap) - lap
        if not np.array_equal(sel[:, 1] & 0x7fffff, rank.astype(np.uint32)):
            bad.append("level %d: lapping ranks differ" % l)
    return bad
