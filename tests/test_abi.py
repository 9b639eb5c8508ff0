"""The C-ABI shared library: loads, exports every symbol include/orbx.h declares, fails loudly without a GPU.
CPU only: no compute entry point is called."""
import ctypes
import os
import subprocess

import pytest

import extractorb_amd as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_reports_abi():
    L = X.load_library()
    assert L.orbx_abi_version() == 1


def test_every_declared_symbol_is_exported():
    L = X.load_library()
    names = X.header_symbols()
    assert len(names) >= 25 and "orbx_extract" in names and "orbx_extract_batch_device" in names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    out = subprocess.check_output(["nm", "-D", "--defined-only", X.library_path()], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(names) <= exported


def test_product_library_does_not_link_the_oracle():
    out = subprocess.check_output(["ldd", X.library_path()], text=True)
    assert "orb_oracle" not in out
    syms = subprocess.check_output(["nm", "-D", X.library_path()], text=True)
    assert "oracle_" not in syms
    for dirpath, _, files in os.walk(os.path.join(ROOT, "extractorb_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "liborb_oracle" not in text, f


def test_keypoint_layout():
    assert X.KEYPOINT_DTYPE.itemsize == 28
    assert [X.KEYPOINT_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def _gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_gpu(), reason="checks the no-device error path")
def test_create_fails_loudly_without_device():
    with pytest.raises(X.OrbxError) as e:
        X.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == -7 and "no HIP device" in str(e.value)


def test_bad_arguments_rejected_before_touching_the_gpu():
    L = X.load_library()
    h = ctypes.c_void_p()
    assert L.orbx_create(ctypes.byref(h), 0, 1.2, 8, 20, 7, 640, 480, 1, -1) == -2
    assert L.orbx_create(ctypes.byref(h), 1000, 1.0, 8, 20, 7, 640, 480, 1, -1) == -2
    assert L.orbx_create(ctypes.byref(h), 1000, 1.2, 17, 20, 7, 640, 480, 1, -1) == -2
    assert L.orbx_create(ctypes.byref(h), 1000, 1.2, 8, 20, 0, 640, 480, 1, -1) == -2
    assert L.orbx_create(None, 1000, 1.2, 8, 20, 7, 640, 480, 1, -1) == -2
    assert h.value is None
