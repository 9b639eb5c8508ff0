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


def test_a_parity_failure_describes_itself(tmp_path, monkeypatch):
    """helpers.dump_failure: what a failing GPU comparison leaves behind (inputs, ORBX_* switches, both arrays)."""
    import json
    import numpy as np
    import helpers
    monkeypatch.setattr(helpers, "ROOT", str(tmp_path))
    monkeypatch.setenv("ORBX_OCT_THREADS", "512")
    helpers.LAST_INPUT.clear(); helpers.LAST_INPUT.update(images=np.zeros((4, 5), np.uint8), nfeatures=77)
    k = np.zeros(2, helpers_keypoint_dtype()); k2 = k.copy(); k2["x"][1] = 3
    with pytest.raises(AssertionError) as e:
        helpers.assert_same_result((0, k, np.zeros((2, 32), np.uint8)), (0, k2, np.zeros((2, 32), np.uint8)), "case A")
    assert "replay file" in str(e.value)
    files = list((tmp_path / "gpurun_out").glob("fail_*.npz"))
    assert len(files) == 1
    z = np.load(files[0])
    assert json.loads(str(z["orbx_env"]))["ORBX_OCT_THREADS"] == "512" and int(z["in_nfeatures"]) == 77
    assert z["in_images"].shape == (4, 5) and z["want_keypoints"]["x"][1] == 3 and "case A" in str(z["what"])


def helpers_keypoint_dtype():
    import oracle_lib
    return oracle_lib.KEYPOINT_DTYPE
