"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package `extractorb_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liborb_oracle.so")

# numpy mirror of cv::KeyPoint (28 bytes)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "orb_oracle.cpp")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liborb_oracle.so"])
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        u8p, vp, ip, fp = C.POINTER(C.c_uint8), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_set_mutation.restype = C.c_int
        L.oracle_set_mutation.argtypes = [vp, C.c_int]
        L.oracle_extract.restype = C.c_int
        L.oracle_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, ip]
        L.oracle_get_tables.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.oracle_level_size.argtypes = [vp, C.c_int, ip, ip]
        L.oracle_get_level.argtypes = [vp, C.c_int, C.c_int, vp]
        L.oracle_get_blurred.argtypes = [vp, C.c_int, vp]
        L.oracle_num_candidates.restype = C.c_int
        L.oracle_num_candidates.argtypes = [vp, C.c_int]
        L.oracle_get_candidates.argtypes = [vp, C.c_int, vp]
        L.oracle_num_level_keys.restype = C.c_int
        L.oracle_num_level_keys.argtypes = [vp, C.c_int]
        L.oracle_get_level_keys.argtypes = [vp, C.c_int, vp]
        L.oracle_resize_linear.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.oracle_gaussian_blur7.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.oracle_gauss_taps.argtypes = [vp]
        L.oracle_fast.restype = C.c_int
        L.oracle_fast.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.oracle_fast_atan2.restype = C.c_float
        L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_fast_atan2_array.argtypes = [vp, vp, vp, C.c_int]
        L.oracle_distribute.restype = C.c_int
        L.oracle_distribute.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.oracle_sincos_restated.argtypes = [C.c_float, fp, fp]
        L.oracle_sincos_check.restype = C.c_long
        L.oracle_sincos_check.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_long)]
        L.oracle_describe.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, vp]
        L.oracle_stereo_match.restype = C.c_int
        L.oracle_stereo_match.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp]
        L.oracle_descriptor_distance.restype = C.c_int
        L.oracle_descriptor_distance.argtypes = [vp, vp]
        L.oracle_image_bounds.argtypes = [vp, C.c_int, C.c_int, vp]
        L.oracle_frame_finish.restype = C.c_int
        L.oracle_frame_finish.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
        L.oracle_assign_features_two_eyes.restype = C.c_int
        L.oracle_assign_features_two_eyes.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
        L.oracle_search_for_initialization.restype = C.c_int
        L.oracle_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]
        L.oracle_features_in_area.restype = C.c_int
        L.oracle_features_in_area.argtypes = [vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, vp, C.c_int]
        L.oracle_project_last_frame.restype = None
        L.oracle_project_last_frame.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_int, vp]
        L.oracle_search_by_projection.restype = C.c_int
        L.oracle_search_by_projection.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp]
        L.oracle_search_by_bow.restype = C.c_int
        L.oracle_search_by_bow.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp]
        L.oracle_search_by_bow_keyframes.restype = C.c_int
        L.oracle_search_by_bow_keyframes.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp]
        L.oracle_compute_bow.restype = C.c_int
        L.oracle_compute_bow.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, ip]
        L.oracle_stereo_from_rgbd.restype = None
        L.oracle_stereo_from_rgbd.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_float, C.c_float, vp, vp]
        L.oracle_gray_from_color.restype = None
        L.oracle_gray_from_color.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, vp, C.c_long]
        L.oracle_time_frames.restype = C.c_double
        L.oracle_time_frames.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CPU restatement of ORB_SLAM3::ORBextractor (reference inc/ORBextractor.h:44-111)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.oracle_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        sf = np.zeros(nlevels, np.float32); isf = sf.copy(); s2 = sf.copy(); is2 = sf.copy()
        nf = np.zeros(nlevels, np.int32); um = np.zeros(16, np.int32)
        self.L.oracle_get_tables(self.h, _ptr(sf), _ptr(isf), _ptr(s2), _ptr(is2), _ptr(nf), _ptr(um))
        self.scale_factors, self.inv_scale_factors = sf, isf
        self.level_sigma2, self.inv_level_sigma2 = s2, is2
        self.features_per_level, self.umax = nf, um

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    def set_mutation(self, m):
        """Sensitivity knob for tests/test_reference_pin.py (oracle/orb_oracle.cpp `enum Mutation`); 0 = the faithful restatement."""
        return self.L.oracle_set_mutation(self.h, int(m))

    def extract(self, image, lapping=(0, 1000)):
        """Returns (mono_index, keypoints[structured], descriptors[n,32]); mono_index -1 for empty."""
        image = np.asarray(image)
        if image.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2 and image.strides[1] == 1
        # a level keeps at most quota + 3 keypoints - or, when the quota is tiny, what the unconditional first pass leaves: up to four nodes per
        # root (nIni = round(width / height) roots, ORBextractor.cc:548-599; a 922 x 200 frame with 55 features has five to six roots and quotas
        # of 5 .. 14 per level: found by the round-4 soak)
        n_ini = max(1, int(round(image.shape[1] / max(1, image.shape[0])))) + 1
        cap = self.nfeatures + (3 + 4 * n_ini) * self.nlevels + 64
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        mono = C.c_int(0)
        n = self.L.oracle_extract(self.h, _ptr(image), image.shape[0], image.shape[1], image.strides[0],
                                  int(lapping[0]), int(lapping[1]), _ptr(kps), _ptr(desc), cap, C.byref(mono))
        if n < 0:
            raise RuntimeError("oracle_extract failed: %d" % n)
        return mono.value, kps[:n].copy(), desc[:n].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        self.L.oracle_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level(self, level, bordered=False):
        w, h = self.level_size(level)
        if bordered:
            out = np.zeros((h + 38, w + 38), np.uint8)
        else:
            out = np.zeros((h, w), np.uint8)
        self.L.oracle_get_level(self.h, level, int(bordered), _ptr(out))
        return out

    def blurred(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        self.L.oracle_get_blurred(self.h, level, _ptr(out))
        return out

    def candidates(self, level):
        n = self.L.oracle_num_candidates(self.h, level)
        out = np.zeros(n, KEYPOINT_DTYPE)
        if n:
            self.L.oracle_get_candidates(self.h, level, _ptr(out))
        return out

    def level_keypoints(self, level):
        n = self.L.oracle_num_level_keys(self.h, level)
        out = np.zeros(n, KEYPOINT_DTYPE)
        if n:
            self.L.oracle_get_level_keys(self.h, level, _ptr(out))
        return out


def stereo_match(o_left, o_right, kps_l, desc_l, kps_r, desc_r, bf, b):
    """Frame::ComputeStereoMatches (reference src/Frame.cc:813-991) on two Oracle instances that just extracted
    the left / right image.  Returns (uRight[N], depth[N], matches_kept)."""
    kps_l = np.ascontiguousarray(kps_l, KEYPOINT_DTYPE); kps_r = np.ascontiguousarray(kps_r, KEYPOINT_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8); desc_r = np.ascontiguousarray(desc_r, np.uint8)
    u = np.zeros(len(kps_l), np.float32); d = np.zeros(len(kps_l), np.float32)
    kept = lib().oracle_stereo_match(o_left.h, o_right.h, _ptr(kps_l), _ptr(desc_l), len(kps_l), _ptr(kps_r), _ptr(desc_r),
                                     len(kps_r), float(bf), float(b), _ptr(u), _ptr(d))
    return u, d, kept


def camera(fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
    """Frame::mK / mDistCoef as the 9-float struct both the oracle and liborbx take."""
    return np.array([fx, fy, cx, cy, k1, k2, p1, p2, k3], np.float32)


def image_bounds(cam, cols, rows):
    b = np.zeros(4, np.float32)
    lib().oracle_image_bounds(_ptr(np.ascontiguousarray(cam, np.float32)), cols, rows, _ptr(b))
    return b


def frame_finish(cam, kps, bounds):
    """UndistortKeyPoints + AssignFeaturesToGrid (reference src/Frame.cc:748-782, 383-417).
    Returns (mvKeysUn, grid_offsets[3073], grid_indices[n_inside])."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    un = np.zeros(len(kps), KEYPOINT_DTYPE); off = np.zeros(64 * 48 + 1, np.int32); idx = np.zeros(max(len(kps), 1), np.int32)
    n = lib().oracle_frame_finish(_ptr(np.ascontiguousarray(cam, np.float32)), _ptr(kps), len(kps),
                                  _ptr(np.ascontiguousarray(bounds, np.float32)), _ptr(un), _ptr(off), _ptr(idx))
    return un, off, idx[:n].copy()


def assign_features_two_eyes(kps_left, kps_right, bounds):
    """Frame::AssignFeaturesToGrid, the Nleft != -1 branch (reference src/Frame.cc:404-414): raw keys of both eyes.
    Returns ((mGrid offsets[3073], indices), (mGridRight offsets[3073], indices relative to the right eye))."""
    kl = np.ascontiguousarray(kps_left, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kps_right, KEYPOINT_DTYPE)
    off = np.zeros(64 * 48 + 1, np.int32); idx = np.zeros(max(len(kl), 1), np.int32)
    offr = np.zeros(64 * 48 + 1, np.int32); idxr = np.zeros(max(len(kr), 1), np.int32)
    nr = C.c_int()
    nl = lib().oracle_assign_features_two_eyes(_ptr(kl), len(kl), _ptr(kr), len(kr), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                               _ptr(off), _ptr(idx), _ptr(offr), _ptr(idxr), C.byref(nr))
    return (off, idx[:nl].copy()), (offr, idxr[:nr.value].copy())


def features_in_area(kps_un, grid_off, grid_idx, bounds, x, y, r, min_level=-1, max_level=-1):
    """Frame::GetFeaturesInArea (reference src/Frame.cc:655-724): keypoint indices in traversal order."""
    kps_un = np.ascontiguousarray(kps_un, KEYPOINT_DTYPE)
    out = np.zeros(max(len(kps_un), 1), np.int32)
    n = lib().oracle_features_in_area(_ptr(kps_un), _ptr(np.ascontiguousarray(grid_off, np.int32)),
                                      _ptr(np.ascontiguousarray(grid_idx, np.int32)), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                      x, y, r, min_level, max_level, _ptr(out), len(out))
    return out[:n].copy()


def search_for_initialization(kps_un1, desc1, kps_un2, desc2, grid_off2, grid_idx2, bounds, prev_matched, window=100, nnratio=0.9,
                              check_orientation=True):
    """ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:706-821).
    Returns (nmatches, vnMatches12[N1], updated vbPrevMatched[N1,2])."""
    k1 = np.ascontiguousarray(kps_un1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(kps_un2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    prev = np.array(prev_matched, np.float32).reshape(len(k1), 2).copy()
    m12 = np.zeros(max(len(k1), 1), np.int32)
    gi = np.ascontiguousarray(grid_idx2, np.int32) if len(grid_idx2) else np.zeros(1, np.int32)
    n = lib().oracle_search_for_initialization(_ptr(k1), _ptr(d1), len(k1), _ptr(k2), _ptr(d2), len(k2),
                                               _ptr(np.ascontiguousarray(grid_off2, np.int32)), _ptr(gi),
                                               _ptr(np.ascontiguousarray(bounds, np.float32)), _ptr(prev), window, nnratio,
                                               int(check_orientation), _ptr(m12))
    return n, m12[:len(k1)].copy(), prev


PROJ_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                             ("flags", "<i4"), ("angle", "<f4")])
assert PROJ_QUERY_DTYPE.itemsize == 32


def project_last_frame(kps_last, kps_un_last, mp_flags, world, Tcw, Tlw, cam, bounds, scale_factors, mbf, mb, th, mono):
    """Front half of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference src/ORBmatcher.cc:1961-2023):
    one search request per keypoint of the last frame.  Tcw / Tlw: 3x4 poses of the current / last frame."""
    kl = np.ascontiguousarray(kps_last, KEYPOINT_DTYPE); ku = np.ascontiguousarray(kps_un_last, KEYPOINT_DTYPE)
    q = np.zeros(max(len(kl), 1), PROJ_QUERY_DTYPE)
    cam4 = np.ascontiguousarray(np.asarray(cam, np.float32)[:4])
    lib().oracle_project_last_frame(_ptr(kl), _ptr(ku), len(kl), _ptr(np.ascontiguousarray(mp_flags, np.uint8)),
                                    _ptr(np.ascontiguousarray(world, np.float32)), _ptr(np.ascontiguousarray(Tcw, np.float32)),
                                    _ptr(np.ascontiguousarray(Tlw, np.float32)), _ptr(cam4), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                    _ptr(np.ascontiguousarray(scale_factors, np.float32)), mbf, mb, th, int(mono), _ptr(q))
    return q[:len(kl)].copy()


def search_by_projection(queries, qdesc, kps_un, desc, grid_off, grid_idx, bounds, u_right=None, occupied=None, ratio_mode=False,
                         nnratio=0.9, check_orientation=True, max_distance=100):
    """ORBmatcher::SearchByProjection, the search half (reference src/ORBmatcher.cc:2025-2175 with ratio_mode False, :60-135 with True).
    Returns (nmatches, matches[N] = request index per keypoint or -1, occupied[N] afterwards)."""
    q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE); ku = np.ascontiguousarray(kps_un, KEYPOINT_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8); d = np.ascontiguousarray(desc, np.uint8)
    N = len(ku)
    m = np.zeros(max(N, 1), np.int32)
    occ = np.zeros(max(N, 1), np.uint8) if occupied is None else np.ascontiguousarray(occupied, np.uint8).copy()
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    gi = np.ascontiguousarray(grid_idx, np.int32) if len(grid_idx) else np.zeros(1, np.int32)
    n = lib().oracle_search_by_projection(_ptr(q), _ptr(qd), len(q), _ptr(ku), _ptr(d), N, _ptr(np.ascontiguousarray(grid_off, np.int32)), _ptr(gi),
                                          _ptr(np.ascontiguousarray(bounds, np.float32)), None if ur is None else _ptr(ur), _ptr(occ),
                                          int(ratio_mode), nnratio, max_distance, int(check_orientation), _ptr(m))
    return n, m[:N].copy(), occ[:N].copy()


def compute_bow(vocab, desc, levelsup=4):
    """Frame::ComputeBoW (reference src/Frame.cc:739-746; DBoW2 TemplatedVocabulary::transform).  vocab: dict with k, L, scoring, weighting,
    parent[n], is_leaf[n], desc[n,32], weight[n].  Returns (word_ids, word_weights, fv_nodes, fv_idx)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    N = len(d)
    wid = np.zeros(max(N, 1), np.uint32); ww = np.zeros(max(N, 1), np.float64)
    fn = np.zeros(max(N, 1), np.uint32); fi = np.zeros(max(N, 1), np.uint32)
    nf = C.c_int()
    nw = lib().oracle_compute_bow(vocab["k"], vocab["L"], vocab["scoring"], vocab["weighting"], len(vocab["parent"]),
                                  _ptr(np.ascontiguousarray(vocab["parent"], np.int32)), _ptr(np.ascontiguousarray(vocab["is_leaf"], np.uint8)),
                                  _ptr(np.ascontiguousarray(vocab["desc"], np.uint8)), _ptr(np.ascontiguousarray(vocab["weight"], np.float64)),
                                  _ptr(d), N, levelsup, _ptr(wid), _ptr(ww), _ptr(fn), _ptr(fi), C.byref(nf))
    return wid[:nw].copy(), ww[:nw].copy(), fn[:nf.value].copy(), fi[:nf.value].copy()


def search_by_bow(kf_fv, f_fv, kf_flags, kps_kf, desc_kf, kps_f, desc_f, nnratio=0.7, th_low=50, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (reference src/ORBmatcher.cc:269-471).  kf_fv / f_fv = (nodes, idx) flattened
    FeatureVectors; kf_flags[i] bit 0 = the keyframe's keypoint i holds a good MapPoint.  Returns (nmatches, matches[NF])."""
    kn, ki = (np.ascontiguousarray(a, np.uint32) for a in kf_fv)
    fn, fi = (np.ascontiguousarray(a, np.uint32) for a in f_fv)
    kk = np.ascontiguousarray(kps_kf, KEYPOINT_DTYPE); kf = np.ascontiguousarray(kps_f, KEYPOINT_DTYPE)
    dk = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32); df = np.ascontiguousarray(desc_f, np.uint8).reshape(-1, 32)
    fl = np.ascontiguousarray(kf_flags, np.uint8)
    NF = len(kf)
    m = np.full(max(NF, 1), -1, np.int32)
    pad = lambda a: a if len(a) else np.zeros(1, a.dtype)
    n = lib().oracle_search_by_bow(_ptr(pad(kn)), _ptr(pad(ki)), len(kn), _ptr(pad(fn)), _ptr(pad(fi)), len(fn), _ptr(pad(fl)), _ptr(pad(kk)), _ptr(pad(dk)),
                                   _ptr(pad(kf)), _ptr(pad(df)), NF, nnratio, th_low, int(check_orientation), _ptr(m))
    return n, m[:NF].copy()


def search_by_bow_keyframes(fv1, fv2, flags1, flags2, kps1, desc1, kps2, desc2, nnratio=0.8, th_low=50, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (reference src/ORBmatcher.cc:823-963).  Returns (nmatches, matches12[N1])."""
    n1, i1 = (np.ascontiguousarray(a, np.uint32) for a in fv1)
    n2, i2 = (np.ascontiguousarray(a, np.uint32) for a in fv2)
    k1 = np.ascontiguousarray(kps1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(kps2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    f1 = np.ascontiguousarray(flags1, np.uint8); f2 = np.ascontiguousarray(flags2, np.uint8)
    m = np.full(max(len(k1), 1), -1, np.int32)
    pad = lambda a: a if len(a) else np.zeros(1, a.dtype)
    n = lib().oracle_search_by_bow_keyframes(_ptr(pad(n1)), _ptr(pad(i1)), len(n1), _ptr(pad(n2)), _ptr(pad(i2)), len(n2), _ptr(pad(f1)), _ptr(pad(f2)),
                                             _ptr(pad(k1)), _ptr(pad(d1)), len(k1), _ptr(pad(k2)), _ptr(pad(d2)), len(k2), nnratio, th_low,
                                             int(check_orientation), _ptr(m))
    return n, m[:len(k1)].copy()


def stereo_from_rgbd(kps, kps_un, depth, factor, mbf):
    """Frame::ComputeStereoFromRGBD after GrabImageRGBD's convertTo (reference src/Frame.cc:994-1015, src/Tracking.cc:1003-1004).
    depth: H x W float32 or uint16.  Returns (mvuRight, mvDepth)."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); kps_un = np.ascontiguousarray(kps_un, KEYPOINT_DTYPE)
    depth = np.ascontiguousarray(depth)
    assert depth.dtype in (np.float32, np.uint16)
    u = np.zeros(max(len(kps), 1), np.float32); d = np.zeros(max(len(kps), 1), np.float32)
    lib().oracle_stereo_from_rgbd(_ptr(kps), _ptr(kps_un), len(kps), _ptr(depth), int(depth.dtype == np.uint16), depth.shape[0], depth.shape[1],
                                  depth.strides[0], factor, mbf, _ptr(u), _ptr(d))
    return u[:len(kps)].copy(), d[:len(kps)].copy()


def gray_from_color(img, red_first):
    """cv::cvtColor(... 2GRAY) of an 8-bit H x W x {3,4} image (reference src/Tracking.cc:915-941)."""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(img.shape[:2], np.uint8)
    lib().oracle_gray_from_color(_ptr(img), img.shape[0], img.shape[1], img.shape[2], int(red_first), img.strides[0], _ptr(out), out.strides[0])
    return out


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_descriptor_distance(_ptr(a), _ptr(b))


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().oracle_gaussian_blur7(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), src.shape[1])
    return dst


def gauss_taps():
    t = np.zeros(7, np.int32)
    lib().oracle_gauss_taps(_ptr(t))
    return t


def fast(img, threshold, nms=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().oracle_fast(_ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nms), _ptr(out), cap)
    return out[:n].copy()


def fast_atan2(y, x):
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(y)
    lib().oracle_fast_atan2_array(_ptr(y), _ptr(x), _ptr(out), y.size)
    return out


def distribute(keys, min_x, max_x, min_y, max_y, n_target):
    keys = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
    cap = max(len(keys), 1)
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().oracle_distribute(_ptr(keys), len(keys), min_x, max_x, min_y, max_y, n_target, _ptr(out), cap)
    return out[:n].copy()


def sincos_restated(x):
    s, c = C.c_float(), C.c_float()
    lib().oracle_sincos_restated(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def sincos_check(lo, hi):
    n = C.c_long()
    bad = lib().oracle_sincos_check(float(lo), float(hi), C.byref(n))
    return bad, n.value


def describe(blurred, x, y, angle):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().oracle_describe(_ptr(blurred), blurred.strides[0], float(x), float(y), float(angle), _ptr(d))
    return d


def time_frames(frames, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, lapping=(0, 1000),
                nthreads=1):
    frames = np.ascontiguousarray(frames, np.uint8)
    n, rows, cols = frames.shape
    tot = C.c_long()
    sec = lib().oracle_time_frames(nfeatures, scale_factor, nlevels, ini_th, min_th, _ptr(frames), n, rows, cols,
                                   int(lapping[0]), int(lapping[1]), nthreads, C.byref(tot))
    return sec, tot.value
