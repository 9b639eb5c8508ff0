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


def assert_same_result(got, want, what=""):
    """got/want: (mono_index, keypoints, descriptors).  Bit-exact: integer and float fields alike."""
    assert got[0] == want[0], "%s mono index %d != %d" % (what, got[0], want[0])
    assert len(got[1]) == len(want[1]), "%s keypoint count %d != %d" % (what, len(got[1]), len(want[1]))
    assert got[1].tobytes() == want[1].tobytes(), "%s keypoints differ" % what
    assert got[2].shape == want[2].shape and np.array_equal(got[2], want[2]), "%s descriptors differ" % what
