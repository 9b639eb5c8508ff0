"""Shared test helpers (fixtures loading, keypoint comparison)."""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_gray(name):
    return np.array(Image.open(os.path.join(GOLDEN, name)))


def load_case(case):
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    return dict(image=load_gray(str(z["image"])), nfeatures=int(z["nfeatures"]), lapping=tuple(z["lapping"].tolist()),
                mono_index=int(z["mono_index"]), keypoints=z["keypoints"], descriptors=z["descriptors"],
                level_counts=z["level_counts"], candidate_counts=z["candidate_counts"])


GOLDEN_CASES = ["luna_1000", "luna_1000_lap00", "luna_7500", "robot_865_1000", "robot_865_1200_lap", "tum_corridor_1000", "tum_room4_1500"]


def sort_kps(k):
    return k[np.lexsort((k["x"], k["y"]))]


# ---- self-describing failures -------------------------------------------------------------------------------------------------
# A parity mismatch writes everything needed to replay it to gpurun_out/fail_<n>_<what>.npz (merged back from the GPU box): the
# input image(s) and constructor parameters of the LAST extraction through the Python mirror (recorded by `record_inputs`, installed
# by conftest.py), every ORBX_* environment switch, the stage / level that differed and both arrays.  One failure then explains
# itself; nothing has to be re-run to find out what was compared.
LAST_INPUT = {}
_fail_count = [0]

# ---- switches of a test row ---------------------------------------------------------------------------------------------------
# Launch-policy switches (ORBX_*) go to the environment, which orbx_create reads once; TEST AIDS ("aid:poison", "aid:lds_pollute",
# "aid:fail_after_fast") cannot be set from the environment and go through orbx_debug_set_option (include/orbx.h).  conftest.py resets the
# aids after every test.
ACTIVE_AIDS = {}


def apply_switches(switches, monkeypatch):
    import extractorb_amd as X
    for k, v in switches.items():
        if k.startswith("aid:"):
            X.debug_set_option(k[4:], int(v))
            ACTIVE_AIDS[k[4:]] = int(v)
        else:
            monkeypatch.setenv(k, v)


def reset_aids():
    """after every test (conftest.py): also the aids a test set through X.debug_set_option directly"""
    import extractorb_amd as X
    X.debug_reset_options()
    ACTIVE_AIDS.clear()


def record_inputs():
    """Wraps extractorb_amd.ORBextractor.__call__ / extract_batch so that the last inputs are on file when a comparison fails."""
    import extractorb_amd as X
    if getattr(X.ORBextractor, "_records_inputs", False):
        return
    call, batch = X.ORBextractor.__call__, X.ORBextractor.extract_batch

    def params(ex):
        return dict(nfeatures=ex.nfeatures, scaleFactor=ex.scaleFactor, nlevels=ex.nlevels, iniThFAST=ex.iniThFAST, minThFAST=ex.minThFAST,
                    max_width=ex.max_width, max_height=ex.max_height, max_batch=ex.max_batch)

    def __call__(self, image, mask=None, lapping=(0, 1000)):
        LAST_INPUT.clear(); LAST_INPUT.update(images=np.array(image, copy=True), lapping=np.asarray(lapping), form="operator()", **params(self))
        return call(self, image, mask, lapping)

    def extract_batch(self, images, lapping=None):
        LAST_INPUT.clear(); LAST_INPUT.update(images=np.array(images, copy=True), lapping=np.asarray(-1 if lapping is None else lapping),
                                              form="extract_batch", **params(self))
        return batch(self, images, lapping)

    X.ORBextractor.__call__, X.ORBextractor.extract_batch = __call__, extract_batch
    X.ORBextractor._records_inputs = True


def dump_failure(what, **arrays):
    """Writes gpurun_out/fail_<n>_<what>.npz and returns its path (or a note why not)."""
    import json
    import re
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        _fail_count[0] += 1
        tag = re.sub(r"[^A-Za-z0-9_.-]+", "_", what)[:60]
        path = os.path.join(out, "fail_%d_%d_%s.npz" % (os.getpid(), _fail_count[0], tag))
        env = {k: v for k, v in os.environ.items() if k.startswith("ORBX_")}
        env.update({"aid:" + k: v for k, v in ACTIVE_AIDS.items()})
        test = os.environ.get("PYTEST_CURRENT_TEST", "")
        keep = {("in_" + k): np.asarray(v) for k, v in LAST_INPUT.items()}
        keep.update({k: np.asarray(v) for k, v in arrays.items() if v is not None})
        np.savez_compressed(path, what=np.array(what), test=np.array(test), orbx_env=np.array(json.dumps(env)), **keep)
        return path
    except Exception as e:       # never mask the parity failure itself
        return "(no dump: %r)" % (e,)


def fail_with_dump(msg, **arrays):
    raise AssertionError("%s  [replay file: %s]" % (msg, dump_failure(msg, **arrays)))


def assert_same_result(got, want, what=""):
    """got/want: (mono_index, keypoints, descriptors).  Bit-exact: integer and float fields alike."""
    arrays = dict(got_mono=got[0], want_mono=want[0], got_keypoints=got[1], want_keypoints=want[1], got_descriptors=got[2], want_descriptors=want[2])
    if got[0] != want[0]:
        fail_with_dump("%s mono index %d != %d" % (what, got[0], want[0]), **arrays)
    if len(got[1]) != len(want[1]):
        fail_with_dump("%s keypoint count %d != %d" % (what, len(got[1]), len(want[1])), **arrays)
    if got[1].tobytes() != want[1].tobytes():
        fail_with_dump("%s keypoints differ" % what, **arrays)
    if not (got[2].shape == want[2].shape and np.array_equal(got[2], want[2])):
        fail_with_dump("%s descriptors differ" % what, **arrays)
