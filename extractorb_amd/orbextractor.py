"""ctypes binding of liborbx.so and the Python mirror of ORB_SLAM3::ORBextractor.

Reference interface mirrored here: inc/ORBextractor.h:44-111 (constructor, operator(), scale getters,
mvImagePyramid), called from Frame::ExtractORB (src/Frame.cc:419-427).
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_LIB = os.environ.get("ORBX_LIBRARY") or os.path.join(_PKG, "liborbx.so")   # ORBX_LIBRARY: another build of the same library (tuning experiments)
_HEADER = os.path.join(_ROOT, "include", "orbx.h")

# numpy mirror of cv::KeyPoint / orbx_keypoint (28 bytes)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

ORBX_OK = 0
ORBX_ERR_EMPTY_IMAGE = -1
ORBX_NUM_KERNELS = 10


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbx error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return _LIB


def build_library(force=False):
    """Compiles liborbx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    args = ["make", "-C", csrc]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return _LIB


def source_hash():
    """Short hash of every source file liborbx.so is built from.  Counter files under profiles/ carry the hash of the
    sources they were measured on; bench.py refuses (null) counters whose hash differs from the tree it runs in."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_PKG, "csrc", "*.hip")) + glob.glob(os.path.join(_PKG, "csrc", "*.hpp")) +
                   glob.glob(os.path.join(_PKG, "csrc", "*.inc")) + glob.glob(os.path.join(_PKG, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_PKG, "csrc", "Makefile")) + glob.glob(os.path.join(_ROOT, "include", "*")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class Vocabulary:
    """DBoW2 ORB vocabulary on the device (reference inc/ORBVocabulary.h; System.cc:81-82 loads ORBvoc.txt once per process)."""

    def __init__(self, path=None, arrays=None, device=-1):
        self._L = load_library()
        self._v = C.c_void_p()
        if path is not None:
            rc = self._L.orbx_vocabulary_load_text(C.byref(self._v), str(path).encode(), device)
        else:
            a = arrays
            par = np.ascontiguousarray(a["parent"], np.int32); leaf = np.ascontiguousarray(a["is_leaf"], np.uint8)
            d = np.ascontiguousarray(a["desc"], np.uint8); w = np.ascontiguousarray(a["weight"], np.float64)
            rc = self._L.orbx_vocabulary_create(C.byref(self._v), a["k"], a["L"], a["scoring"], a["weighting"], len(par), _ptr(par), _ptr(leaf),
                                                _ptr(d), _ptr(w), device)
        if rc != ORBX_OK:
            raise OrbxError(rc, (self._L.orbx_last_error(None) or b"").decode())

    def info(self):
        k, L, n, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._L.orbx_vocabulary_info(self._v, C.byref(k), C.byref(L), C.byref(n), C.byref(w))
        return dict(k=k.value, L=L.value, n_nodes=n.value, n_words=w.value)

    def __del__(self):
        try:
            if getattr(self, "_v", None):
                self._L.orbx_vocabulary_destroy(self._v)
                self._v = None
        except Exception:
            pass


def header_symbols():
    """Names of every function include/orbx.h declares."""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library():
    """Loads liborbx.so; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise OrbxError(-100, "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C extractorb_amd/csrc` (the HIP library is the only compute path)" % _LIB)
    L = C.CDLL(_LIB)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.orbx_abi_version.restype = C.c_int
    L.orbx_create.restype = C.c_int
    L.orbx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_last_error.restype = C.c_char_p
    L.orbx_last_error.argtypes = [vp]
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.restype = C.c_float
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_tables.argtypes = [vp] + [vp] * 6
    L.orbx_max_keypoints.argtypes = [vp]
    L.orbx_compute_tables.argtypes = [C.c_int, C.c_float, C.c_int] + [vp] * 6
    L.orbx_compute_level_sizes.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, vp, vp]
    L.orbx_compute_cell_grid.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_int] + [ip] * 7
    L.orbx_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, vp, vp, C.c_int, ip, ip, vp, vp]
    L.orbx_extract_view.argtypes = [vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), ip, ip,
                                    C.POINTER(vp), C.POINTER(vp)]
    L.orbx_compute_pyramid.argtypes = [vp, vp, C.c_int, C.c_int, C.c_ssize_t]
    L.orbx_compute_keypoints_octree.argtypes = [vp, vp, C.c_int, vp]
    L.orbx_fetch_pyramid.argtypes = [vp, C.c_int, C.POINTER(vp), vp, vp, vp, vp]
    L.orbx_debug_last_forms.argtypes = [vp, ip, ip, ip]
    L.orbx_debug_last_split_level.argtypes = [vp]
    L.orbx_debug_set_option.argtypes = [C.c_char_p, C.c_int]
    L.orbx_debug_clock_probe.argtypes = [vp, C.c_int]
    L.orbx_debug_clock_read.argtypes = [vp, C.c_int, vp]
    L.orbx_debug_policy.restype = C.c_char_p
    L.orbx_debug_policy.argtypes = [vp]
    L.orbx_extract_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, vp, vp, vp, C.c_int,
                                     vp, vp, vp, vp]
    L.orbx_extract_batch_begin.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, vp, C.c_int]
    L.orbx_extract_batch_end.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp]
    L.orbx_extract_batch_end_view.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), ip, C.POINTER(vp), C.POINTER(vp)]
    L.orbx_host_alloc.restype = vp
    L.orbx_host_alloc.argtypes = [C.c_size_t]
    L.orbx_host_free.argtypes = [vp]
    L.orbx_extract_batch_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, vp, vp, vp,
                                            C.c_int, vp, vp, vp, vp]
    L.orbx_get_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_ssize_t, ip, ip]
    L.orbx_stereo_match_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
    L.orbx_project_last_frame_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp,
                                                 C.c_float, C.c_float, C.c_float, C.c_int, vp]
    L.orbx_search_by_projection_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int,
                                                   vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp]
    L.orbx_vocabulary_load_text.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int]
    L.orbx_vocabulary_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int]
    L.orbx_vocabulary_destroy.argtypes = [vp]
    L.orbx_vocabulary_destroy.restype = None
    L.orbx_vocabulary_info.argtypes = [vp, ip, ip, ip, ip]
    L.orbx_compute_bow_device.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orbx_search_by_bow_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float,
                                            C.c_int, C.c_int, vp, vp]
    L.orbx_search_by_bow_keyframes_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int,
                                                      C.c_float, C.c_int, C.c_int, vp, vp]
    L.orbx_stereo_match_last.argtypes = [vp, C.c_int, C.c_float, C.c_float, vp, vp, C.c_int, vp]
    L.orbx_compute_image_bounds.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orbx_frame_finish_device.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orbx_frame_finish_two_eyes_device.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orbx_search_for_initialization_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp,
                                                        vp, vp, C.c_int, C.c_float, C.c_int, vp, vp]
    L.orbx_stereo_from_rgbd_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t,
                                               C.c_float, C.c_float, vp, vp]
    L.orbx_gray_from_color_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, vp,
                                              C.c_ssize_t, C.c_ssize_t]
    L.orbx_set_stream.argtypes = [vp, vp]
    L.orbx_get_stream.restype = vp
    L.orbx_get_stream.argtypes = [vp]
    L.orbx_synchronize.argtypes = [vp]
    L.orbx_debug_num_candidates.argtypes = [vp, C.c_int, C.c_int, ip]
    L.orbx_debug_get_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.orbx_debug_get_blurred.argtypes = [vp, C.c_int, C.c_int, vp, C.c_ssize_t]
    L.orbx_profile_enable.argtypes = [vp, C.c_int]
    L.orbx_profile_reset.argtypes = [vp]
    L.orbx_profile_read.argtypes = [vp, vp, vp]
    L.orbx_profile_kernel_name.restype = C.c_char_p
    L.orbx_profile_kernel_name.argtypes = [C.c_int]
    L.orbx_profile_kernel_name_of.restype = C.c_char_p
    L.orbx_profile_kernel_name_of.argtypes = [vp, C.c_int]
    L.orbx_algorithmic_bytes.restype = C.c_long
    L.orbx_algorithmic_bytes.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- test aids (include/orbx.h: orbx_debug_set_option).  Not reachable from the environment; handles created AFTERWARDS carry the setting ----
TEST_AIDS = ("poison", "lds_pollute", "fail_after_fast")


def debug_set_option(name, value):
    rc = load_library().orbx_debug_set_option(name.encode(), int(value))
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_debug_set_option(%r)" % name)


def debug_reset_options():
    debug_set_option("poison", -1)
    debug_set_option("lds_pollute", -1)
    debug_set_option("fail_after_fast", 0)
    debug_set_option("pyr_cols_shape", -1)
    debug_set_option("shared_upload_bytes", -1)


# ---- handle-free host helpers (no GPU needed) -------------------------------------------------------------
def compute_tables(nfeatures=1000, scale_factor=1.2, nlevels=8):
    L = load_library()
    sf = np.zeros(nlevels, np.float32); isf = sf.copy(); s2 = sf.copy(); is2 = sf.copy()
    q = np.zeros(nlevels, np.int32); um = np.zeros(16, np.int32)
    rc = L.orbx_compute_tables(nfeatures, scale_factor, nlevels, _ptr(sf), _ptr(isf), _ptr(s2), _ptr(is2), _ptr(q), _ptr(um))
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_compute_tables")
    return dict(scale_factors=sf, inv_scale_factors=isf, level_sigma2=s2, inv_level_sigma2=is2,
                features_per_level=q, umax=um)


def compute_level_sizes(rows, cols, scale_factor=1.2, nlevels=8):
    L = load_library()
    w = np.zeros(nlevels, np.int32); h = np.zeros(nlevels, np.int32)
    rc = L.orbx_compute_level_sizes(scale_factor, nlevels, rows, cols, _ptr(w), _ptr(h))
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_compute_level_sizes")
    return list(zip(w.tolist(), h.tolist()))


def compute_cell_grid(rows, cols, level, scale_factor=1.2, nlevels=8):
    L = load_library()
    v = [C.c_int() for _ in range(7)]
    rc = L.orbx_compute_cell_grid(scale_factor, nlevels, rows, cols, level, *[C.byref(x) for x in v])
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_compute_cell_grid")
    keys = ["n_cols", "n_rows", "w_cell", "h_cell", "n_cells", "n_ini", "cand_cap"]
    return dict(zip(keys, [x.value for x in v]))


def pinned_empty(shape, dtype=np.uint8):
    """numpy array backed by pinned host memory (orbx_host_alloc): H2D copies from it overlap with kernels."""
    L = load_library()
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    p = L.orbx_host_alloc(nbytes)
    if not p:
        raise OrbxError(-6, "orbx_host_alloc failed")
    buf = (C.c_uint8 * nbytes).from_address(p)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)   # free with pinned_free(arr); it must not be used afterwards


def pinned_free(arr):
    load_library().orbx_host_free(C.c_void_p(arr.ctypes.data))


def camera(fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
    """orbx_camera: Frame::mK and Frame::mDistCoef as nine floats."""
    return np.array([fx, fy, cx, cy, k1, k2, p1, p2, k3], np.float32)


def compute_image_bounds(cam, cols, rows):
    """Frame::ComputeImageBounds (reference src/Frame.cc:784-811): (mnMinX, mnMaxX, mnMinY, mnMaxY).  Host only."""
    L = load_library()
    b = np.zeros(4, np.float32)
    rc = L.orbx_compute_image_bounds(_ptr(np.ascontiguousarray(cam, np.float32)), cols, rows, _ptr(b))
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_compute_image_bounds")
    return b


class ORBextractor:
    """Mirror of ``ORB_SLAM3::ORBextractor`` (reference inc/ORBextractor.h:44-111) on one MI355X.

    ``ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)`` as in the reference; the
    extra keyword arguments size the device arenas once (the reference reallocates per call).
    ``extractor(image, mask=None, lapping=(0, 1000))`` mirrors ``operator()``: it returns
    ``(mono_index, keypoints, descriptors, all_levels_keypoints)`` where the reference fills its output
    arguments; ``mono_index`` is the reference's return value (-1 for an empty image).
    """

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, *,
                 max_width=640, max_height=480, max_batch=1, device=-1):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.orbx_create(C.byref(self._h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                                 max_width, max_height, max_batch, device)
        if rc != ORBX_OK:
            self._h = None
            raise OrbxError(rc, (self._L.orbx_last_error(None) or b"").decode())
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.scaleFactor, self.iniThFAST, self.minThFAST = scaleFactor, iniThFAST, minThFAST
        self.max_width, self.max_height, self.max_batch = max_width, max_height, max_batch
        self.capacity = self._L.orbx_max_keypoints(self._h)
        t = compute_tables(nfeatures, scaleFactor, nlevels)
        self.mvScaleFactor, self.mvInvScaleFactor = t["scale_factors"], t["inv_scale_factors"]
        self.mvLevelSigma2, self.mvInvLevelSigma2 = t["level_sigma2"], t["inv_level_sigma2"]
        self.mnFeaturesPerLevel, self.umax = t["features_per_level"], t["umax"]

    # ---- reference getters (inc/ORBextractor.h:63-83) ----
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self._L.orbx_get_scale_factor(self._h)
    def GetScaleFactors(self): return self.mvScaleFactor.copy()
    def GetInverseScaleFactors(self): return self.mvInvScaleFactor.copy()
    def GetScaleSigmaSquares(self): return self.mvLevelSigma2.copy()
    def GetInverseScaleSigmaSquares(self): return self.mvInvLevelSigma2.copy()

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != ORBX_OK:
            raise OrbxError(rc, (self._L.orbx_last_error(self._h) or b"").decode())

    # ---- operator() ----
    def __call__(self, image, mask=None, lapping=(0, 1000)):
        image = np.asarray(image)
        if image.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8), [np.zeros(0, KEYPOINT_DTYPE)] * self.nlevels
        if image.dtype != np.uint8 or image.ndim != 2 or image.strides[1] != 1:
            raise ValueError("image must be a 2-D uint8 array with contiguous rows (CV_8UC1)")   # assert at ORBextractor.cc:1087
        cap = self.capacity
        kps = np.zeros(cap, KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        lvl = np.zeros(cap, KEYPOINT_DTYPE); counts = np.zeros(self.nlevels, np.int32)
        n, mono = C.c_int(), C.c_int()
        self._check(self._L.orbx_extract(self._h, _ptr(image), image.shape[0], image.shape[1], image.strides[0],
                                         int(lapping[0]), int(lapping[1]), _ptr(kps), _ptr(desc), cap,
                                         C.byref(n), C.byref(mono), _ptr(lvl), _ptr(counts)))
        per_level, o = [], 0
        for c in counts.tolist():
            per_level.append(lvl[o:o + c].copy()); o += c
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy(), per_level

    def extract_batch(self, images, lapping=None):
        """images: uint8 [B, rows, cols].  Returns a list of (mono_index, keypoints, descriptors, per_level)."""
        images = np.ascontiguousarray(images, np.uint8)
        B, rows, cols = images.shape
        cap = self.capacity
        kps = np.zeros((B, cap), KEYPOINT_DTYPE); desc = np.zeros((B, cap, 32), np.uint8)
        lvl = np.zeros((B, cap), KEYPOINT_DTYPE); counts = np.zeros((B, self.nlevels), np.int32)
        n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
        lap = None
        if lapping is not None:
            lap = np.ascontiguousarray(np.broadcast_to(np.asarray(lapping, np.int32).reshape(-1, 2), (B, 2)))
        self._check(self._L.orbx_extract_batch(self._h, B, _ptr(images), rows, cols, cols, rows * cols, _ptr(lap),
                                               _ptr(kps), _ptr(desc), cap, _ptr(n), _ptr(mono), _ptr(lvl), _ptr(counts)))
        out = []
        for f in range(B):
            per_level, o = [], 0
            for c in counts[f].tolist():
                per_level.append(lvl[f, o:o + c].copy()); o += c
            out.append((int(mono[f]), kps[f, :n[f]].copy(), desc[f, :n[f]].copy(), per_level))
        return out

    def extract_batch_begin(self, images, lapping=None, want_levels=False):
        """Asynchronous host-buffer form: enqueue H2D + path + D2H and return (one batch in flight per handle)."""
        # rows may be padded (a view into a wider buffer, as a cv::Mat ROI): stride / frame_stride are passed as they are
        assert images.dtype == np.uint8 and images.ndim == 3 and images.strides[2] == 1 and images.strides[1] >= images.shape[2] and images.strides[0] > 0
        B, rows, cols = images.shape
        lap = None
        if lapping is not None:
            lap = np.ascontiguousarray(np.broadcast_to(np.asarray(lapping, np.int32).reshape(-1, 2), (B, 2)))
        self._check(self._L.orbx_extract_batch_begin(self._h, B, _ptr(images), rows, cols, images.strides[1], images.strides[0], _ptr(lap), int(want_levels)))
        self._pending = (B, images, lap)     # keep the buffers alive until the batch ends

    def extract_batch_end(self):
        if getattr(self, "_pending", None) is None:
            raise OrbxError(-2, "no batch in flight: call extract_batch_begin first")
        B = self._pending[0]
        cap = self.capacity
        kps = np.zeros((B, cap), KEYPOINT_DTYPE); desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
        self._check(self._L.orbx_extract_batch_end(self._h, _ptr(kps), _ptr(desc), cap, _ptr(n), _ptr(mono), None, None))
        self._pending = None
        return [(int(mono[f]), kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(B)]

    def extract_batch_end_view(self):
        """Zero-copy form of extract_batch_end: waits, then returns numpy views (keypoints[B, cap], descriptors[B, cap, 32], n[B],
        mono[B]) into the handle's pinned staging, valid until the next extract_batch_begin on this extractor."""
        if getattr(self, "_pending", None) is None:
            raise OrbxError(-2, "no batch in flight: call extract_batch_begin first")
        B = self._pending[0]
        vk, vd, vn, vm, vc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        self._check(self._L.orbx_extract_batch_end_view(self._h, C.byref(vk), C.byref(vd), C.byref(vc), C.byref(vn), C.byref(vm)))
        self._pending = None
        cap = vc.value
        as_np = lambda ptr, nbytes: np.frombuffer((C.c_uint8 * nbytes).from_address(ptr.value), np.uint8)
        kps = as_np(vk, B * cap * 28).view(KEYPOINT_DTYPE).reshape(B, cap)
        desc = as_np(vd, B * cap * 32).reshape(B, cap, 32)
        return kps, desc, as_np(vn, 4 * B).view(np.int32), as_np(vm, 4 * B).view(np.int32)

    def extract_batch_device(self, d_images, n_frames, rows, cols, d_kps, d_desc, d_n, d_mono, capacity,
                             stride=None, frame_stride=None, lapping=None, d_level_kps=0, d_level_counts=0):
        """Device-pointer form (ints or objects with .data_ptr()); asynchronous on the handle's stream."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        stride = cols if stride is None else stride
        frame_stride = rows * stride if frame_stride is None else frame_stride
        lap = None
        if lapping is not None:
            lap = np.ascontiguousarray(np.broadcast_to(np.asarray(lapping, np.int32).reshape(-1, 2), (n_frames, 2)))
        self._check(self._L.orbx_extract_batch_device(self._h, n_frames, dp(d_images), rows, cols, stride, frame_stride,
                                                      _ptr(lap), dp(d_kps), dp(d_desc), capacity, dp(d_n), dp(d_mono),
                                                      dp(d_level_kps), dp(d_level_counts)))

    # ---- Frame::ComputeStereoMatches (reference src/Frame.cc:813-991) on the last batch ----
    def stereo_match_last(self, n_pairs, bf, b):
        """Frames 2p / 2p+1 of the last extract_batch call are the left / right eye of pair p.
        Returns a list of (uRight[nL], depth[nL], n_matched) per pair."""
        cap = self.capacity
        u = np.zeros((n_pairs, cap), np.float32); d = np.zeros((n_pairs, cap), np.float32)
        nm = np.zeros(n_pairs, np.int32)
        self._check(self._L.orbx_stereo_match_last(self._h, n_pairs, bf, b, _ptr(u), _ptr(d), cap, _ptr(nm)))
        return u, d, nm

    def stereo_match_device(self, n_pairs, d_kps, d_desc, d_n, capacity, bf, b, d_u_right, d_depth, d_n_matched):
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self._L.orbx_stereo_match_device(self._h, n_pairs, dp(d_kps), dp(d_desc), dp(d_n), capacity, bf, b,
                                                     dp(d_u_right), dp(d_depth), dp(d_n_matched)))

    def frame_finish_device(self, n_frames, d_kps, d_n, capacity, cam, bounds, d_kps_un, d_grid_off, d_grid_idx, d_n_inside):
        """UndistortKeyPoints + AssignFeaturesToGrid (reference src/Frame.cc:748-782, 383-417) on device buffers."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        cam = np.ascontiguousarray(cam, np.float32); bounds = np.ascontiguousarray(bounds, np.float32)
        self._check(self._L.orbx_frame_finish_device(self._h, n_frames, dp(d_kps), dp(d_n), capacity, _ptr(cam), _ptr(bounds),
                                                     dp(d_kps_un), dp(d_grid_off), dp(d_grid_idx), dp(d_n_inside)))

    def frame_finish_two_eyes_device(self, n_pairs, d_kps, d_n, capacity, cam, bounds, d_kps_un, d_grid_off, d_grid_idx, d_n_inside):
        """AssignFeaturesToGrid's Nleft != -1 branch (reference src/Frame.cc:404-414: mGrid / mGridRight from the RAW keys of frames 2p / 2p + 1)
        + UndistortKeyPoints, on device buffers of 2 * n_pairs frames."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        cam = np.ascontiguousarray(cam, np.float32); bounds = np.ascontiguousarray(bounds, np.float32)
        self._check(self._L.orbx_frame_finish_two_eyes_device(self._h, n_pairs, dp(d_kps), dp(d_n), capacity, _ptr(cam), _ptr(bounds),
                                                              dp(d_kps_un), dp(d_grid_off), dp(d_grid_idx), dp(d_n_inside)))

    def search_for_initialization_device(self, n_pairs, frames1, frames2, d_kps_un, d_desc, d_n, capacity, d_grid_off, d_grid_idx,
                                         bounds, d_prev_matched, d_matches12, d_n_matches, window=100, nnratio=0.9,
                                         check_orientation=True):
        """ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:706-821) for n_pairs pairs of device-resident
        frames; frames1 / frames2 = (first, step) of the F1 / F2 frame index of pair p."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        bounds = np.ascontiguousarray(bounds, np.float32)
        self._check(self._L.orbx_search_for_initialization_device(
            self._h, n_pairs, frames1[0], frames1[1], frames2[0], frames2[1], dp(d_kps_un), dp(d_desc), dp(d_n), capacity,
            dp(d_grid_off), dp(d_grid_idx), _ptr(bounds), dp(d_prev_matched), window, nnratio, int(check_orientation),
            dp(d_matches12), dp(d_n_matches)))

    def project_last_frame_device(self, n_pairs, last, cur, d_kps, d_kps_un, d_n, capacity, d_mp_flags, d_world, d_poses, cam, bounds,
                                  mbf, mb, th, mono, d_queries):
        """Front half of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference src/ORBmatcher.cc:1961-2023);
        last / cur = (first, step) of the last / current frame index of pair p."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        cam = np.ascontiguousarray(cam, np.float32); bounds = np.ascontiguousarray(bounds, np.float32)
        self._check(self._L.orbx_project_last_frame_device(self._h, n_pairs, last[0], last[1], cur[0], cur[1], dp(d_kps), dp(d_kps_un), dp(d_n),
                                                           capacity, dp(d_mp_flags), dp(d_world), dp(d_poses), _ptr(cam), _ptr(bounds),
                                                           mbf, mb, th, int(mono), dp(d_queries)))

    def search_by_projection_device(self, n_pairs, cur, d_queries, d_query_desc, desc_blocks, d_n_queries, query_capacity, d_kps_un, d_desc,
                                    d_n, capacity, d_grid_off, d_grid_idx, bounds, d_u_right, d_occupied, ratio_mode, nnratio,
                                    check_orientation, d_matches, d_n_matches, max_distance=100):
        """ORBmatcher::SearchByProjection, the search (reference src/ORBmatcher.cc:2025-2175 / :44-135); cur and desc_blocks = (first, step)."""
        def dp(x):
            return C.c_void_p(0 if x is None else (x.data_ptr() if hasattr(x, "data_ptr") else int(x)))
        bounds = np.ascontiguousarray(bounds, np.float32)
        self._check(self._L.orbx_search_by_projection_device(
            self._h, n_pairs, cur[0], cur[1], dp(d_queries), dp(d_query_desc), desc_blocks[0], desc_blocks[1], dp(d_n_queries), query_capacity,
            dp(d_kps_un), dp(d_desc), dp(d_n), capacity, dp(d_grid_off), dp(d_grid_idx), _ptr(bounds), dp(d_u_right), dp(d_occupied),
            int(ratio_mode), nnratio, max_distance, int(check_orientation), dp(d_matches), dp(d_n_matches)))

    def compute_bow_device(self, vocab, n_frames, d_desc, d_n, capacity, d_word_ids, d_word_weights, d_n_words, d_feat_nodes, d_feat_idx,
                           d_n_feat, levels_up=4):
        """Frame::ComputeBoW (reference src/Frame.cc:739-746) for n_frames device-resident frames."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self._L.orbx_compute_bow_device(self._h, vocab._v, n_frames, dp(d_desc), dp(d_n), capacity, levels_up, dp(d_word_ids),
                                                    dp(d_word_weights), dp(d_n_words), dp(d_feat_nodes), dp(d_feat_idx), dp(d_n_feat)))

    def search_by_bow_device(self, n_pairs, kf, cur, d_feat_nodes, d_feat_idx, d_n_feat, d_kf_mp_flags, d_kps, d_desc, d_n, capacity,
                             d_matches, d_n_matches, nnratio=0.7, th_low=50, check_orientation=True):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (reference src/ORBmatcher.cc:269-471); kf and cur = (first, step)."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self._L.orbx_search_by_bow_device(self._h, n_pairs, kf[0], kf[1], cur[0], cur[1], dp(d_feat_nodes), dp(d_feat_idx),
                                                      dp(d_n_feat), dp(d_kf_mp_flags), dp(d_kps), dp(d_desc), dp(d_n), capacity,
                                                      C.c_float(nnratio), th_low, int(check_orientation), dp(d_matches), dp(d_n_matches)))

    def search_by_bow_keyframes_device(self, n_pairs, kf1, kf2, d_feat_nodes, d_feat_idx, d_n_feat, d_kf1_mp_flags, d_kf2_mp_flags, d_kps, d_desc,
                                       d_n, capacity, d_matches12, d_n_matches, nnratio=0.8, th_low=50, check_orientation=True):
        """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (reference src/ORBmatcher.cc:823-963); kf1 and kf2 = (first, step)."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self._L.orbx_search_by_bow_keyframes_device(self._h, n_pairs, kf1[0], kf1[1], kf2[0], kf2[1], dp(d_feat_nodes), dp(d_feat_idx),
                                                                dp(d_n_feat), dp(d_kf1_mp_flags), dp(d_kf2_mp_flags), dp(d_kps), dp(d_desc), dp(d_n),
                                                                capacity, C.c_float(nnratio), th_low, int(check_orientation), dp(d_matches12),
                                                                dp(d_n_matches)))

    def stereo_from_rgbd_device(self, n_frames, d_kps, d_kps_un, d_n, capacity, d_depth, depth_is_u16, rows, cols, depth_map_factor, mbf,
                                d_u_right, d_depth_out, depth_stride_bytes=None, depth_frame_stride_bytes=None):
        """Frame::ComputeStereoFromRGBD with GrabImageRGBD's depth conversion (reference src/Frame.cc:994-1015, src/Tracking.cc:1003-1004)."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        elem = 2 if depth_is_u16 else 4
        depth_stride_bytes = cols * elem if depth_stride_bytes is None else depth_stride_bytes
        depth_frame_stride_bytes = rows * depth_stride_bytes if depth_frame_stride_bytes is None else depth_frame_stride_bytes
        self._check(self._L.orbx_stereo_from_rgbd_device(self._h, n_frames, dp(d_kps), dp(d_kps_un), dp(d_n), capacity, dp(d_depth),
                                                         int(depth_is_u16), rows, cols, depth_stride_bytes, depth_frame_stride_bytes,
                                                         depth_map_factor, mbf, dp(d_u_right), dp(d_depth_out)))

    def gray_from_color_device(self, n_frames, d_src, rows, cols, channels, red_first, d_gray, src_stride=None, src_frame_stride=None,
                               gray_stride=None, gray_frame_stride=None):
        """cv::cvtColor(RGB/BGR/RGBA/BGRA -> GRAY) as Tracking::GrabImage* applies it (reference src/Tracking.cc:915-941)."""
        def dp(x):
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        src_stride = cols * channels if src_stride is None else src_stride
        src_frame_stride = rows * src_stride if src_frame_stride is None else src_frame_stride
        gray_stride = cols if gray_stride is None else gray_stride
        gray_frame_stride = rows * gray_stride if gray_frame_stride is None else gray_frame_stride
        self._check(self._L.orbx_gray_from_color_device(self._h, n_frames, dp(d_src), rows, cols, channels, int(red_first), src_stride,
                                                        src_frame_stride, dp(d_gray), gray_stride, gray_frame_stride))

    def set_stream(self, stream_ptr):
        self._check(self._L.orbx_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def synchronize(self):
        self._check(self._L.orbx_synchronize(self._h))

    # ---- mvImagePyramid (inc/ORBextractor.h:85) ----
    def image_pyramid_level(self, level, frame=0, bordered=False):
        w, h = C.c_int(), C.c_int()
        sizes = compute_level_sizes(self.max_height, self.max_width, self.scaleFactor, self.nlevels)
        buf = np.zeros((sizes[0][1] + 38, sizes[0][0] + 38), np.uint8)
        self._check(self._L.orbx_get_level(self._h, frame, level, int(bordered), _ptr(buf), buf.strides[0], C.byref(w), C.byref(h)))
        if bordered:
            return buf[:h.value + 38, :w.value + 38].copy()
        return buf[:h.value, :w.value].copy()

    def fetch_pyramid(self, frame=0, bordered=False):
        """All levels of one frame in ONE device-to-host copy (orbx_fetch_pyramid): a list of arrays, each a copy of the w x h level (or of
        the (w + 38) x (h + 38) buffer with the BORDER_REFLECT_101 frame the reference's views sit in, ORBextractor.cc:1173-1177)."""
        base = C.c_void_p()
        off = np.zeros(self.nlevels, np.uint64); st = np.zeros(self.nlevels, np.int32)
        w = np.zeros(self.nlevels, np.int32); h = np.zeros(self.nlevels, np.int32)
        self._check(self._L.orbx_fetch_pyramid(self._h, frame, C.byref(base), _ptr(off), _ptr(st), _ptr(w), _ptr(h)))
        out = []
        for l in range(self.nlevels):
            e = 19 if bordered else 0
            rows, cols, stride = int(h[l]) + 2 * e, int(w[l]) + 2 * e, int(st[l])
            start = base.value + int(off[l]) - e * stride - e
            buf = (C.c_uint8 * (stride * rows)).from_address(start)
            out.append(np.frombuffer(buf, np.uint8).reshape(rows, stride)[:, :cols].copy())
        return out

    @property
    def mvImagePyramid(self):
        return self.fetch_pyramid()

    # ---- the two public stage methods (inc/ORBextractor.h:87-90; src/orb_extractor/main_orb_extractor.cpp:43-46 calls them) ----
    def ComputePyramid(self, image):
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 2 or image.strides[1] != 1 or image.size == 0:
            raise ValueError("image must be a non-empty 2-D uint8 array with contiguous rows (CV_8UC1)")
        self._check(self._L.orbx_compute_pyramid(self._h, _ptr(image), image.shape[0], image.shape[1], image.strides[0]))

    def ComputeKeyPointsOctTree(self):
        """allKeypoints: one array per level, level coordinates, angles set (ORBextractor.cc:773-888)."""
        cap = self.capacity
        lvl = np.zeros(cap, KEYPOINT_DTYPE); counts = np.zeros(self.nlevels, np.int32)
        self._check(self._L.orbx_compute_keypoints_octree(self._h, _ptr(lvl), cap, _ptr(counts)))
        per_level, o = [], 0
        for c in counts.tolist():
            per_level.append(lvl[o:o + c].copy()); o += c
        return per_level

    def policy(self):
        """The launch-policy switches as orbx_create read them (include/orbx.h: orbx_debug_policy)."""
        return (self._L.orbx_debug_policy(self._h) or b"").decode()

    def clock_probe(self, slot):
        """Asynchronous sample of the shader clock beside the handle's running work (include/orbx.h: orbx_debug_clock_probe)."""
        self._check(self._L.orbx_debug_clock_probe(self._h, int(slot)))

    def clock_read(self, n_slots):
        ghz = np.zeros(n_slots, np.float64)
        self._check(self._L.orbx_debug_clock_read(self._h, int(n_slots), _ptr(ghz)))
        return ghz

    def last_forms(self):
        """(pyramid form, region side of k_pyr_cols, blur form) of the last call: include/orbx.h, orbx_debug_last_forms."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.orbx_debug_last_forms(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def blurred_level_exists(self, level):
        """whether the last call left a blurred image of `level` (it did not where the blur ran per keypoint inside k_describe)"""
        form = self.last_forms()[2]
        return form not in (3, 5) or (form == 5 and level >= self._L.orbx_debug_last_split_level(self._h))

    # ---- introspection for tests/bench ----
    def debug_candidates(self, level, frame=0):
        n = C.c_int()
        self._check(self._L.orbx_debug_num_candidates(self._h, frame, level, C.byref(n)))
        out = np.zeros(max(n.value, 1), KEYPOINT_DTYPE)
        self._check(self._L.orbx_debug_get_candidates(self._h, frame, level, _ptr(out), len(out)))
        return out[:n.value].copy()

    def debug_blurred(self, level, frame=0):
        w, h = self.image_pyramid_level(level, frame).shape[::-1]
        out = np.zeros((h, w), np.uint8)
        self._check(self._L.orbx_debug_get_blurred(self._h, frame, level, _ptr(out), out.strides[0]))
        return out

    def profile(self, enable=True):
        self._check(self._L.orbx_profile_enable(self._h, int(enable)))
        self._check(self._L.orbx_profile_reset(self._h))

    def profile_read(self):
        ms = np.zeros(ORBX_NUM_KERNELS, np.float64); n = np.zeros(ORBX_NUM_KERNELS, np.int64)
        self._check(self._L.orbx_profile_read(self._h, _ptr(ms), _ptr(n)))
        # keyed by the kernel that ran in each slot on this handle, as rocprofv3 names it (k_pyr_cols, k_octree_256, ...)
        return {self._L.orbx_profile_kernel_name_of(self._h, i).decode(): (float(ms[i]), int(n[i])) for i in range(ORBX_NUM_KERNELS)}

    def algorithmic_bytes(self, rows, cols, n_out):
        return int(self._L.orbx_algorithmic_bytes(self._h, rows, cols, n_out))
