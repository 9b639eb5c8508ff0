"""Host-side check of the argument k_fast's score pass rests on since round 4 (extractorb_amd/csrc/k_fast.hip: pairScore): a FAST-9/16 pixel needs
only ONE polarity evaluated, chosen by the opposite ring pairs - "max over the eight pairs (k, k+8) of the pair's minimum < v" selects dark - with a
dark pixel's ring and centre complemented (x ^ 255) so that dark becomes bright.  Claim: the score so computed equals the two-polarity score
S = max(max_arcs min_arc(r) - v, v - min_arcs max_arc(r), 0) for EVERY input, not only for corners.  No GPU: plain numpy, exhaustive over small value
ranges (which make corners frequent) and random over the full range."""
import numpy as np


def both_polarities(r, v):
    """S of the reference's cornerScore<16>: r [n, 16] ring values, v [n] centres"""
    idx = (np.arange(16)[:, None] + np.arange(9)[None, :]) % 16          # the 16 arcs of 9 contiguous ring pixels
    arcs = r[:, idx]                                                     # [n, 16, 9]
    bright = arcs.min(axis=2).max(axis=1) - v                            # max over arcs of the arc minimum, minus the centre
    dark = v - arcs.max(axis=2).min(axis=1)
    return np.maximum(np.maximum(bright, dark), 0)


def one_polarity(r, v):
    """k_fast's form: the pair test picks the polarity, dark pixels are complemented, the bright score is evaluated (centre included: >= 0)"""
    pair_min = np.minimum(r[:, :8], r[:, 8:])
    dark = pair_min.max(axis=1) < v
    flip = np.where(dark, 255, 0)
    x = r ^ flip[:, None]
    c = v ^ flip
    idx = (np.arange(16)[:, None] + np.arange(9)[None, :]) % 16
    max_min = x[:, idx].min(axis=2).max(axis=1)
    return np.maximum(max_min, c) - c


def check(r, v):
    want, got = both_polarities(r, v), one_polarity(r, v)
    bad = np.nonzero(want != got)[0]
    assert bad.size == 0, "ring %s centre %d: two polarities %d, one polarity %d" % (r[bad[0]].tolist(), v[bad[0]], want[bad[0]], got[bad[0]])
    return int((want > 7).sum())


def test_one_polarity_equals_two_on_random_rings():
    rng = np.random.default_rng(4)
    corners = 0
    for span in (2, 3, 5, 16, 64, 256):                                  # few distinct values: many arcs of equal / extreme pixels, many corners
        lo = rng.integers(0, 256 - span + 1, (60000, 1))
        r = (lo + rng.integers(0, span, (60000, 16))).astype(np.int64)
        v = (lo[:, 0] + rng.integers(0, span, 60000)).astype(np.int64)
        corners += check(r, v)
    assert corners > 300


def test_one_polarity_equals_two_on_structured_rings():
    # every ring that is an arc of n bright (or dark) pixels on a flat background, at every rotation and every contrast sign, centre on both sides
    rows, cents = [], []
    for n in range(0, 17):
        for rot in range(16):
            for fg, bg in ((200, 100), (100, 200), (255, 0), (0, 255), (128, 127), (127, 128)):
                ring = np.full(16, bg)
                ring[(rot + np.arange(n)) % 16] = fg
                for v in (bg, fg, (fg + bg) // 2, 0, 255):
                    rows.append(ring); cents.append(v)
    check(np.array(rows, np.int64), np.array(cents, np.int64))


def test_one_polarity_equals_two_over_brighter_equal_darker_patterns():
    # rings over {90, 100, 110} with the centre at 100 - every mixture of brighter / equal / darker ring pixels: all 2^16 rings without an equal pixel
    # exhaustively, and every 43rd of the 3^16 rings with them (the full set passes too; it takes minutes)
    two = (np.arange(1 << 16, dtype=np.int64)[:, None] >> np.arange(16, dtype=np.int64)[None, :]) & 1
    check(90 + 20 * two, np.full(1 << 16, 100, np.int64))
    codes = np.arange(0, 3 ** 16, 43, dtype=np.int64)
    three = (codes[:, None] // (3 ** np.arange(16, dtype=np.int64))[None, :]) % 3
    check(90 + 10 * three, np.full(three.shape[0], 100, np.int64))
