"""DistributeOctTree (reference ORBextractor.cc:544-771) as restated by the oracle: hand-checked small cases and
invariants that hold for any input.  CPU only."""
import numpy as np
import pytest

import oracle_lib as O


def keys(pts):
    k = np.zeros(len(pts), O.KEYPOINT_DTYPE)
    for i, (x, y, r) in enumerate(pts):
        k[i] = (x, y, 7, -1, r, 0, -1)
    return k


def test_empty_and_single():
    assert len(O.distribute(keys([]), 16, 624, 16, 464, 100)) == 0
    out = O.distribute(keys([(10, 20, 50)]), 16, 624, 16, 464, 100)
    assert len(out) == 1 and (out[0]["x"], out[0]["y"], out[0]["response"]) == (10, 20, 50)


def test_one_split_order_and_best_response():
    # 608 x 448 root -> halfX = 304, halfY = 224.  One key per quadrant + an extra weaker key in n1.
    pts = [(10, 10, 30), (20, 20, 90), (400, 10, 40), (10, 300, 50), (400, 300, 60)]
    out = O.distribute(keys(pts), 16, 624, 16, 464, 4)
    # children are pushed to the front in order n1..n4, so the list reads n4, n3, n2, n1 (:626-665)
    assert [(int(k["x"]), int(k["y"])) for k in out] == [(400, 300), (10, 300), (400, 10), (20, 20)]
    assert out[3]["response"] == 90          # best response inside n1


def test_first_key_wins_response_ties():
    pts = [(10, 10, 70), (20, 20, 70), (400, 300, 10)]
    out = O.distribute(keys(pts), 16, 624, 16, 464, 2)
    assert [(int(k["x"]), int(k["y"])) for k in out] == [(400, 300), (10, 10)]


def test_two_roots_for_wide_images():
    # 1888 x 1048 rectangle -> nIni = round(1.80) = 2 roots of width 944
    pts = [(100, 100, 10), (1000, 100, 20)]
    out = O.distribute(keys(pts), 16, 1904, 16, 1064, 10)
    assert [(int(k["x"]), int(k["y"])) for k in out] == [(100, 100), (1000, 100)]


@pytest.mark.parametrize("seed,n,N", [(0, 3000, 217), (1, 500, 217), (2, 150, 217), (3, 20000, 60), (4, 64, 5), (5, 5000, 1)])
def test_invariants(seed, n, N):
    rng = np.random.default_rng(seed)
    W, H = 608, 448
    flat = rng.choice(W * H, n, replace=False)
    pts = [(int(p % W), int(p // W), int(r)) for p, r in zip(flat, rng.integers(7, 255, n))]
    k = keys(pts)
    out = O.distribute(k, 16, 16 + W, 16, 16 + H, N)
    # every output is an input key; no duplicates
    inp = {(x, y): r for x, y, r in pts}
    got = [(int(a["x"]), int(a["y"])) for a in out]
    assert len(set(got)) == len(got) and all(g in inp for g in got)
    assert all(int(a["response"]) == inp[(int(a["x"]), int(a["y"]))] for a in out)
    # size: never more than N+3 once past the first pass (4 roots' children); all keys kept if fewer than N
    assert len(out) <= max(N + 3, 4)
    if n <= N:
        assert len(out) == n
    else:
        assert len(out) >= min(N, n)
    # deterministic
    assert out.tobytes() == O.distribute(k, 16, 16 + W, 16, 16 + H, N).tobytes()
    # order of the input keys only matters through response ties: a permutation that keeps equal-response
    # keys in relative order yields the same result when responses are distinct
    if len(set(r for _, _, r in pts)) == n:
        perm = rng.permutation(n)
        assert out.tobytes() == O.distribute(k[perm], 16, 16 + W, 16, 16 + H, N).tobytes()


def test_spatial_spread():
    # the point of the quad-tree: a dense cluster cannot take all N slots
    rng = np.random.default_rng(8)
    cluster = [(int(x), int(y), 200) for x, y in zip(rng.integers(0, 40, 400), rng.integers(0, 40, 400))]
    spread = [(int(x), int(y), 10) for x, y in zip(rng.integers(50, 600, 100), rng.integers(50, 440, 100))]
    pts = list({(x, y): (x, y, r) for x, y, r in cluster + spread}.values())
    out = O.distribute(keys(pts), 16, 624, 16, 464, 100)
    outside = sum(1 for k in out if k["x"] >= 50 or k["y"] >= 50)
    assert outside >= 60
