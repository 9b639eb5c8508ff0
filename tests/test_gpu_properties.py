"""Size-independent properties at BASELINE.json's full batch sizes, where running the oracle on every frame would
take minutes: idempotence, independence of a frame from its batch neighbours, bounds, and spot parity."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X
from extractorb_amd import synth
from helpers import assert_same_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stream64():
    return synth.frames("noise", 0, 64, 480, 640)


def test_batch64_idempotent_and_order_independent(stream64):
    ex = X.ORBextractor(1000, max_batch=64)
    a = ex.extract_batch(stream64)
    b = ex.extract_batch(stream64)
    perm = np.random.default_rng(0).permutation(64)
    c = ex.extract_batch(stream64[perm])
    for f in range(64):
        assert_same_result(a[f][:3], b[f][:3], "rerun %d" % f)
        assert_same_result(c[f][:3], a[perm[f]][:3], "permuted %d" % f)      # frames are independent units


def test_batch64_bounds_and_spot_parity(stream64):
    ex = X.ORBextractor(1000, max_batch=64)
    out = ex.extract_batch(stream64)
    quota = ex.mnFeaturesPerLevel
    for f, (mono, k, d, lvl) in enumerate(out):
        assert mono == 0 and 1000 <= len(k) <= 1000 + 3 * 8                 # dense noise always fills the quota
        assert all(len(lvl[l]) <= quota[l] + 3 for l in range(8))
        assert (k["x"] >= 19).all() and (k["x"] < 640 - 19).all() and (k["y"] >= 19).all() and (k["y"] < 480 - 19).all()
        assert ((k["response"] >= 7) & (k["response"] <= 254)).all()
        assert len({(x, y, o) for x, y, o in zip(k["x"], k["y"], k["octave"])}) == len(k)
        bits = np.unpackbits(d, axis=1).mean()
        assert 0.35 < bits < 0.65
    for f in (0, 31, 63):
        o = O.Oracle(1000)
        assert_same_result(out[f][:3], o.extract(stream64[f]), "spot %d" % f)


def test_max_batch_smaller_than_request_is_rejected(stream64):
    ex = X.ORBextractor(1000, max_batch=8)
    with pytest.raises(X.OrbxError):
        ex.extract_batch(stream64[:9])
    assert len(ex.extract_batch(stream64[:8])) == 8


def test_two_handles_interleaved(stream64):
    # two cameras == two handles (Frame.cc:109-112 runs them from two threads)
    a, b = X.ORBextractor(1000), X.ORBextractor(1200)
    ra, rb = a(stream64[0]), b(stream64[1])
    assert_same_result(ra[:3], O.Oracle(1000).extract(stream64[0]))
    assert_same_result(rb[:3], O.Oracle(1200).extract(stream64[1]))
    assert np.array_equal(a.image_pyramid_level(2), O_level(stream64[0], 2))


def O_level(img, l):
    o = O.Oracle(1000)
    o.extract(img)
    return o.level(l)


def test_parity_over_a_longer_stream():
    # 96 frames (3 variants x 32) against the oracle: catches rare-path differences (rounding ties in the rotated
    # pattern, response ties in the quad-tree, empty-cell retries) that a handful of frames may never reach
    ex = X.ORBextractor(1000, max_batch=32)
    total_kp = 0
    for variant in ("noise", "textured", "sparse"):
        fr = synth.frames(variant, 500, 32, 480, 640)
        out = ex.extract_batch(fr)
        o = O.Oracle(1000)
        for f in range(32):
            assert_same_result(out[f][:3], o.extract(fr[f]), "%s frame %d" % (variant, f))
            total_kp += len(out[f][1])
    assert total_kp > 70000
