"""A guard instead of a memory for docs/history/DESIGN_rounds_1-5.md §4i's quad-tree anomaly: the register / scratch / memory-
instruction table of every kernel in the SHIPPED liborbx.so — read out of the code objects embedded in the library itself — against
extractorb_amd/csrc/kernel_table.json, which is checked in next to the kernels.  CPU only (llvm-readelf / llvm-objdump from the ROCm image)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "isa"))
import kernel_table as KT  # noqa: E402
import extractorb_amd as X  # noqa: E402


@pytest.fixture(scope="module")
def current():
    if not os.path.exists(os.path.join(KT.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    return KT.table(X.library_path())


def test_the_library_holds_the_kernels_of_the_hot_path(current):
    names = set(current)
    for k in ("k_octree_256", "k_octree_512", "k_octree_1024", "k_octree_256r", "k_octree_512r", "k_octree_1024r", "k_octree_1024g", "k_describe<0>", "k_describe<1>"):
        assert k in names, k
    for prefix in ("k_fast<", "k_fast_wide<", "k_pyr_cols<", "k_blur<", "k_resize<", "k_pyr_first<"):
        assert any(n.startswith(prefix) for n in names), prefix
    assert len(names) >= 38


def test_no_flat_memory_instructions_on_the_extraction_path(current):
    """The fault of §4i needed FLAT accesses to LDS (a pointer table indexed at run time = generic pointers) beside scratch traffic.  No kernel of
    the extraction path contains a FLAT instruction; the three next-row kernels that use a generic pointer by design have no scratch."""
    flat = {k: v["flat"] for k, v in current.items() if v.get("flat", 0)}
    assert set(flat) <= KT.FLAT_ALLOWED, flat
    for k in KT.FLAT_ALLOWED & set(current):
        assert current[k]["scratch_bytes"] == 0 and current[k]["sgpr_spill"] == 0, (k, current[k])
    for k, v in current.items():
        if k.startswith(("k_octree", "k_fast", "k_pyr_", "k_blur", "k_resize", "k_describe")):
            assert v.get("flat", 0) == 0, (k, v)


def test_scratch_and_spills_stay_within_the_checked_in_table(current):
    allowed = json.load(open(KT.TABLE))
    bad = KT.check(current, allowed)
    assert not bad, "\n".join(bad)


def test_only_the_queued_quadtree_variants_use_scratch(current):
    """Every kernel but the register-starved quad-tree variants is scratch-free; those hold their spills in scratch with DS (not FLAT) node accesses."""
    with_scratch = sorted(k for k, v in current.items() if v["scratch_bytes"])
    assert all(k.startswith("k_octree_") for k in with_scratch), with_scratch
    for k in with_scratch:
        assert current[k].get("flat", 0) == 0
