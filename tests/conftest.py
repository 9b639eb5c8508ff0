import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The oracle is compiled on demand; the HIP library must already exist (build() makes it)."""
    import oracle_lib
    oracle_lib.build_oracle()
    from extractorb_amd import library_path, build_library
    if not os.path.exists(library_path()):
        build_library()
    import helpers
    helpers.record_inputs()        # a failing comparison writes its inputs to gpurun_out/fail_*.npz (helpers.dump_failure)


@pytest.fixture(autouse=True)
def _no_test_aid_outlives_its_test():
    yield
    import helpers
    helpers.reset_aids()
