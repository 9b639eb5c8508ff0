"""Bounded, seeded randomised parity sweep under -m gpu: tools/fuzz_parity.py's generator (random image sizes, feature counts,
level counts, scale factors 1.1-2.0, thresholds, lapping areas, content), every stage and the final arrays against the oracle,
under each kernel-variant switch.  The totals are written to gpurun_out/r06_fuzz_parity.json on the GPU box (copied to
profiles/r02_fuzz_parity.md)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# (switches, cases, seed): ORBX_* = launch-policy switches in the environment, read once by orbx_create; "aid:*" = test aids, set through
# orbx_debug_set_option (they cannot come from the environment): helpers.apply_switches
CONFIGS = [({}, 14, 101), ({"ORBX_OCT_THREADS": "256"}, 8, 102), ({"ORBX_OCT_THREADS": "512"}, 8, 103),
           ({"ORBX_OCT_THREADS": "1024"}, 8, 104), ({}, 8, 105), ({}, 8, 106),
           # FAST with a workgroup per cell (the form of calls with few cells, round 3) whatever the size, and never
           ({"ORBX_FAST_WIDE": "1"}, 8, 121), ({"ORBX_FAST_WIDE": "0"}, 8, 122), ({"ORBX_FAST_WIDE": "1", "ORBX_LEAF_FRAMES": "0", "aid:lds_pollute": "77"}, 8, 123),
           ({"ORBX_RESIZE_BYTEWISE": "1"}, 8, 107),
           # the 128-VGPR quad-tree variants (short phase-2 passes) at every workgroup size; seed 103 draws sparse levels
           ({"ORBX_OCT_THREADS": "256", "ORBX_OCT_ROOMY": "1"}, 8, 103), ({"ORBX_OCT_THREADS": "512", "ORBX_OCT_ROOMY": "1"}, 8, 103),
           ({"ORBX_OCT_THREADS": "1024", "ORBX_OCT_ROOMY": "1"}, 8, 108),
           # the pyramid as one launch per level (the form of large batches of large frames), and region by region with the coarser cuts
           ({"ORBX_PYR_COLS": "0"}, 8, 117), ({"ORBX_PYR_COLS": "1", "ORBX_PYR_COL_PX": "56"}, 8, 118), ({"ORBX_PYR_COLS": "1", "ORBX_PYR_COL_PX": "112"}, 8, 119),
           ({"ORBX_PYR_COLS": "1", "ORBX_PYR_COL_PX": "80", "ORBX_RESIZE_BYTEWISE": "1", "aid:lds_pollute": "201"}, 8, 120),
           # every device allocation of the handle filled with a byte pattern: nothing may depend on what hipMalloc returns
           ({"aid:poison": "165"}, 8, 111), ({"aid:poison": "255"}, 8, 112),
           # ... nor on what the previous workgroup left in LDS (every CU's LDS filled with a byte in front of every kernel)
           ({"aid:lds_pollute": "165"}, 8, 113), ({"aid:lds_pollute": "255", "aid:poison": "90"}, 8, 114),
           # small batches start the quad-tree from k_fast's leaf tables by default (round 3); these keep the kernel's own first sweep under test
           ({"ORBX_LEAF_FRAMES": "0"}, 8, 115), ({"ORBX_LEAF_FRAMES": "0", "ORBX_OCT_THREADS": "512", "ORBX_OCT_ROOMY": "1"}, 8, 103),
           # ... and the leaf tables with poisoned allocations (they must be zero between calls whatever hipMalloc returned)
           ({"aid:poison": "77", "ORBX_OCT_THREADS": "256", "ORBX_OCT_ROOMY": "1"}, 8, 116),
           # round 6: the blur per keypoint split by level (k_describe<PB> below the split, k_blur + the plain description from it on), every workgroup
           # shape of the region-major pyramid pinned in turn; the copy-back form of the one-frame host call
           ({"ORBX_PATCH_BLUR": "1", "ORBX_BLUR_SPLIT": "3", "aid:pyr_cols_shape": "4"}, 8, 124),
           ({"ORBX_PATCH_BLUR": "1", "ORBX_BLUR_SPLIT": "1", "ORBX_PYR_COL_PX": "80", "aid:pyr_cols_shape": "6", "aid:lds_pollute": "99"}, 8, 125),
           ({"ORBX_PATCH_BLUR": "1", "ORBX_BLUR_SPLIT": "5", "aid:pyr_cols_shape": "1", "aid:poison": "119"}, 8, 127),
           ({"ORBX_ZERO_COPY": "0", "aid:poison": "33"}, 8, 126),
           # round 6: frames beyond 4096 px (FUZZ_BIG: the generator draws 4100-7000-px strips and 4100-5200-px-high frames), plain and with polluted LDS
           ({"FUZZ_BIG": "1"}, 4, 131), ({"FUZZ_BIG": "1", "aid:lds_pollute": "201", "aid:poison": "77"}, 3, 132)]
_totals = []


@pytest.mark.parametrize("env,n,seed", CONFIGS, ids=[",".join("%s=%s" % kv for kv in c[0].items()) or "default" for c in CONFIGS])
def test_seeded_fuzz_sweep(env, n, seed, monkeypatch):
    import fuzz_parity
    import helpers
    helpers.apply_switches(env, monkeypatch)
    done, skipped, nkp, nbytes = fuzz_parity.run(n, seed)
    assert done + skipped == n and done >= n // 2
    _totals.append(dict(switches=env, seed=seed, drawn=n, bit_exact=done, rejected_geometries=skipped, keypoints=nkp,
                        descriptor_bytes=nbytes))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(_totals, open(os.path.join(out, "r06_fuzz_parity.json"), "w"), indent=1)
    except OSError:
        pass


def test_seeded_batch_shape_sweep():
    """tools/fuzz_batches.py: random frames per call (1 .. 160) x image size x feature count x content, i.e. the host's launch policies crossed at
    their thresholds; three frames of every batch against the oracle, two calls on one handle against each other."""
    import fuzz_batches
    done, skipped, checked = fuzz_batches.run(24, 31)
    assert done + skipped == 24 and done >= 20 and checked >= 2 * done


@pytest.mark.parametrize("env", [{"ORBX_SPLIT_MIN_MPX": "0"}, {"ORBX_SPLIT_MIN_MPX": "0", "ORBX_SPLIT": "3"},
                                 # round 6: the blur split by level under both overlaps (k_blur of the coarse levels aside, two description launches per half)
                                 {"ORBX_PATCH_BLUR": "1", "ORBX_BLUR_SPLIT": "3", "ORBX_SPLIT_MIN_MPX": "0", "aid:lds_pollute": "119"},
                                 {"ORBX_PATCH_BLUR": "1", "ORBX_BLUR_SPLIT": "2", "ORBX_SPLIT_MIN_MPX": "0", "ORBX_SPLIT": "3", "aid:poison": "201"}],
                         ids=["blur-aside", "staggered-tails", "split-blur-aside", "split-staggered"])
def test_seeded_batch_shape_sweep_under_the_overlap_policies(env, monkeypatch):
    """The same sweep with every batch counted as large (ORBX_SPLIT_MIN_MPX=0), so that the overlap forms of large batches - the blur on its side
    stream (the default), staggered tails - and the blur split by level meet every batch shape, not only the benchmark's."""
    import fuzz_batches
    import helpers
    helpers.apply_switches(env, monkeypatch)
    done, skipped, checked = fuzz_batches.run(10, 37)
    assert done + skipped == 10 and done >= 8 and checked >= 2 * done
