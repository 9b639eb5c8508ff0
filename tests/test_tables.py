"""Constructor tables, pyramid sizes and cell grids: host logic of liborbx.so vs the oracle vs SURVEY.md
Appendix B (values derived there from reference ORBextractor.cc:419-474, :781-795, :1171).  CPU only."""
import hashlib
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_brief_pattern_sha256():
    text = open(os.path.join(ROOT, "include", "orbx_brief_pattern.inc")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    nums = [int(t) for t in re.findall(r"-?\d+", text)]
    assert len(nums) == 1024 and max(abs(n) for n in nums) <= 13
    assert hashlib.sha256(",".join(map(str, nums)).encode()).hexdigest() == \
        "88df8ca875cc8db56799edd57bb914edad8acb2d48c202b7a464a575b55dbdb8"
    pts = np.array(nums).reshape(512, 2)
    assert abs(np.sqrt((pts ** 2).sum(1)).max() - 18.385) < 1e-3   # footprint the blur/border must cover


@pytest.mark.parametrize("nf,expect", [
    (1000, [217, 181, 151, 126, 105, 87, 73, 60]),
    (1200, [261, 217, 181, 151, 126, 105, 87, 72]),
    (2000, [434, 362, 302, 251, 209, 175, 145, 122]),
    (7500, [1629, 1357, 1131, 943, 785, 655, 545, 455]),
])
def test_feature_quotas(nf, expect):
    o = O.Oracle(nf)
    assert o.features_per_level.tolist() == expect
    assert X.compute_tables(nf)["features_per_level"].tolist() == expect
    assert sum(expect) == nf


def test_scale_tables_match_oracle_bitwise():
    for nf, sf, nl in [(1000, 1.2, 8), (2000, 1.2, 8), (500, 1.5, 5), (1000, 1.1, 12), (300, 2.0, 3)]:
        o = O.Oracle(nf, sf, nl)
        t = X.compute_tables(nf, sf, nl)
        for a, b in [(o.scale_factors, t["scale_factors"]), (o.inv_scale_factors, t["inv_scale_factors"]),
                     (o.level_sigma2, t["level_sigma2"]), (o.inv_level_sigma2, t["inv_level_sigma2"])]:
            assert a.tobytes() == b.tobytes()
        assert o.features_per_level.tolist() == t["features_per_level"].tolist()
        assert o.umax.tolist() == t["umax"].tolist()


def test_scale_and_umax_values():
    o = O.Oracle(1000)
    want = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.allclose(o.scale_factors, want, rtol=0, atol=1e-7)
    assert o.umax.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # 749 pixels in the orientation disc
    assert 31 + 2 * sum(2 * u + 1 for u in o.umax[1:]) == 749
    # keypoint size per level = (int)(31*scale)
    assert [int(np.float32(31) * s) for s in o.scale_factors] == [31, 37, 44, 53, 64, 77, 92, 111]


LEVELS = {
    (480, 640): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
    (1080, 1920): [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)],
    (512, 512): [(512, 512), (427, 427), (356, 356), (296, 296), (247, 247), (206, 206), (171, 171), (143, 143)],
}
GRIDS = {   # (nCols, nRows, wCell, hCell) per level, SURVEY.md Appendix B
    (480, 640): [(20, 14, 31, 32), (16, 12, 32, 31), (13, 10, 32, 31), (11, 8, 31, 31), (9, 6, 31, 34), (7, 5, 33, 33),
                 (6, 4, 31, 33), (4, 3, 37, 34)],
    (1080, 1920): [(62, 34, 31, 31), (52, 28, 31, 31), (43, 23, 31, 32), (35, 19, 31, 32), (29, 16, 31, 31),
                   (24, 13, 31, 31), (20, 11, 31, 30), (16, 8, 32, 34)],
    (512, 512): [(16, 16, 30, 30), (13, 13, 31, 31), (10, 10, 33, 33), (8, 8, 33, 33), (7, 7, 31, 31), (5, 5, 35, 35),
                 (4, 4, 35, 35), (3, 3, 37, 37)],
}
CELLS = {(480, 640): 815, (1080, 1920): 6342, (512, 512): 688}
NINI = {(480, 640): 1, (1080, 1920): 2, (512, 512): 1}


@pytest.mark.parametrize("shape", list(LEVELS))
def test_level_sizes_and_cell_grid(shape):
    rows, cols = shape
    assert X.compute_level_sizes(rows, cols) == LEVELS[shape]
    total = 0
    for l in range(8):
        g = X.compute_cell_grid(rows, cols, l)
        assert (g["n_cols"], g["n_rows"], g["w_cell"], g["h_cell"]) == GRIDS[shape][l]
        assert g["n_ini"] == NINI[shape]
        total += g["n_cols"] * g["n_rows"]
        # cells the reference's loop skips (`continue` at ORBextractor.cc:802-803, :811-812) make no cv::FAST call
        w, h = LEVELS[shape][l]
        max_bx, max_by = w - 16, h - 16
        real = sum(1 for i in range(g["n_rows"]) for j in range(g["n_cols"])
                   if 16 + i * g["h_cell"] < max_by - 3 and 16 + j * g["w_cell"] < max_bx - 6)
        assert g["n_cells"] == real
    assert total == CELLS[shape]
    assert sum(w * h for w, h in LEVELS[shape]) == {(480, 640): 950532, (1080, 1920): 6419321, (512, 512): 811960}[shape]


def test_level_sizes_match_oracle_on_odd_shapes():
    rng = np.random.default_rng(1)
    for rows, cols in [(333, 517), (241, 223), (600, 800), (719, 1279)]:
        o = O.Oracle(500)
        o.extract(rng.integers(0, 256, (rows, cols), dtype=np.uint8))
        assert [o.level_size(l) for l in range(8)] == X.compute_level_sizes(rows, cols)


def test_candidate_capacity_bounds_oracle_counts():
    # the arena bound (no two 8-adjacent NMS survivors inside a cell) must dominate what FAST can emit
    from extractorb_amd import synth
    for variant in ("noise", "textured"):
        f = synth.frames(variant, 3, 1, 480, 640)[0]
        o = O.Oracle(1000)
        o.extract(f)
        for l in range(8):
            assert len(o.candidates(l)) <= X.compute_cell_grid(480, 640, l)["cand_cap"]


def test_unsupported_geometries_are_rejected():
    with pytest.raises(X.OrbxError):
        X.compute_cell_grid(120, 160, 7)          # coarsest level narrower than one cell
    with pytest.raises(X.OrbxError):
        X.compute_cell_grid(2000, 400, 0)         # nIni == 0 in the reference (undefined behaviour there)


def test_patch_blur_runs_cover_every_pixel_the_rotated_pattern_can_read():
    """k_describe<PB> (round 5) blurs only the part of a keypoint's 37 x 37 patch the rotated pattern can reach: the run table c_pbRun of
    k_describe_body.hpp - (column group, first row, rows) per lane - must hold, per group, exactly the rows |r| <= rmax(group) of the bound
    derived from the pattern's largest radius, and no sample of any rotation may fall outside it (swept every 0.0005 degrees with the kernel's
    float arithmetic: products and sums rounded separately, half-even rounding)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "extractorb_amd", "csrc", "k_describe_body.hpp")).read()
    body = text[text.index("c_pbRun[32] = {"):]
    entries = [int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]{6}", body[:body.index("};")])]
    assert len(entries) == 32
    pat = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "orbx_brief_pattern.inc")).read(), flags=re.S)
    pts = np.array([int(t) for t in re.findall(r"-?\d+", pat)]).reshape(-1, 2)
    r2 = int((pts ** 2).sum(1).max())
    assert r2 == 338
    covered = {g: set() for g in range(10)}
    for e in entries:
        g, o0, n = e & 0xff, (e >> 8) & 0xff, e >> 16
        assert 0 <= g < 10 and 1 <= n <= 12 and 0 <= o0 and o0 + n <= 37          # at most six trips of the two-row loop; inside the 37-row tile
        rows = set(range(o0, o0 + n))
        assert not rows & covered[g]
        covered[g] |= rows
    for g in range(10):
        qmin = min(abs(4 * g - 18 + k) for k in range(4) if 4 * g - 18 + k <= 18)
        rmax = min(18, int(np.floor(0.5 + np.sqrt(r2 + 0.5 - max(qmin - 0.5, 0) ** 2))))
        assert covered[g] == set(range(18 - rmax, 18 + rmax + 1)), g
    ang = np.arange(0, 360, 0.0005, dtype=np.float32)
    rad = (ang * np.float32(np.pi / 180.0)).astype(np.float32)
    a, b = np.cos(rad).astype(np.float32), np.sin(rad).astype(np.float32)
    need = np.zeros((37, 37), bool)
    for x, y in pts.astype(np.float32):
        r = np.rint((x * b).astype(np.float32) + (y * a).astype(np.float32)).astype(int)
        q = np.rint((x * a).astype(np.float32) - (y * b).astype(np.float32)).astype(int)
        assert np.abs(r).max() <= 18 and np.abs(q).max() <= 18
        need[r + 18, q + 18] = True
    for row, col in zip(*np.nonzero(need)):
        assert int(row) in covered[int(col) // 4], (row - 18, col - 18)
