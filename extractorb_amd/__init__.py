"""extractorb_amd — MI355X-native ORB feature extraction (host-side mirror of the reference interface).

The compute lives in ``liborbx.so`` (hand-written HIP kernels behind the C ABI of ``include/orbx.h``);
this package only binds it with ctypes and mirrors ``ORB_SLAM3::ORBextractor``
(reference inc/ORBextractor.h:44-111) so Python callers and the tests read like the reference's callers.

There is no CPU fallback: importing works anywhere (so the C ABI can be inspected), but constructing an
extractor without the built library or without a HIP device raises.
"""
from .orbextractor import (KEYPOINT_DTYPE, ORBextractor, OrbxError, build_library, library_path, load_library,
                           compute_tables, compute_level_sizes, compute_cell_grid, header_symbols, camera,
                           compute_image_bounds, pinned_empty, pinned_free, source_hash, Vocabulary, debug_set_option,
                           debug_reset_options)

__all__ = ["KEYPOINT_DTYPE", "ORBextractor", "OrbxError", "build_library", "library_path", "load_library",
           "compute_tables", "compute_level_sizes", "compute_cell_grid", "header_symbols", "camera", "compute_image_bounds", "pinned_empty", "pinned_free", "source_hash", "Vocabulary",
           "debug_set_option", "debug_reset_options"]
__version__ = "0.1.0"
