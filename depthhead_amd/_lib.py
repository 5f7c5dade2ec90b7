"""ctypes binding of libdepthhead_hip.so (include/depthhead_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing the product
path raises.  `load()` never builds anything by itself; `depthhead_amd.build.build()` (called by
`__graft_entry__.build()`) does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdepthhead_hip.so")

DH_OK = 0
ERRORS = {-1: "DH_EINVAL", -2: "DH_EFOREST", -3: "DH_EHIP", -4: "DH_ENOMEM", -5: "DH_ESIZE", -6: "DH_ESTATE"}

POSE_DTYPE = np.dtype([("mid_point", "<f4", (3,)), ("reserved", "<u4"), ("rotation", "<f8", (3,))], align=True)
assert POSE_DTYPE.itemsize == 40


class ForestDesc(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("roots", C.c_void_p), ("n_nodes", C.c_uint32), ("nodes", C.c_void_p),
                ("n_leaves", C.c_uint32), ("leaf_prob", C.c_void_p), ("off_begin", C.c_void_p),
                ("rot_begin", C.c_void_p), ("offsets", C.c_void_p), ("rotations", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("stepwidth", C.c_uint32), ("subimage_width", C.c_uint32), ("subimage_height", C.c_uint32),
                ("gaussian_sigma", C.c_float), ("meanshift_iterations", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("traverse_ms", C.c_float), ("vote_ms", C.c_float), ("cluster_ms", C.c_float),
                ("total_ms", C.c_float), ("n_frames", C.c_uint32), ("boxsum_ms", C.c_float),
                ("emit_ms", C.c_float), ("reserved", C.c_uint32)]


# every symbol include/depthhead_hip.h declares
EXPORTS = [
    "dh_last_error", "dh_version", "dh_forest_create", "dh_forest_destroy", "dh_forest_info",
    "dh_predictor_create", "dh_predictor_destroy", "dh_predictor_update_sigma", "dh_predictor_sigma",
    "dh_predict_batch", "dh_predict_batch_device", "dh_predict_batch_rle", "dh_biwi_decode_depth_device", "dh_host_alloc", "dh_host_free", "dh_predictor_reserve", "dh_predictor_set_forking", "dh_patch_grid",
    "dh_predict_mask", "dh_predict_mask_device", "dh_hough_image", "dh_hough_image_device", "dh_build_hough_image", "dh_build_hough_image_device",
    "dh_predict_from2dhough", "dh_predict_from2dhough_device",
    "dh_biwi_decode_depth", "dh_biwi_parse_cal", "dh_biwi_parse_pose",
    "dh_graph_capture", "dh_graph_launch", "dh_graph_destroy",
    "dh_set_profiling", "dh_get_timing", "dh_debug_enable", "dh_debug_leaf_indices", "dh_debug_patch_flags",
    "dh_debug_grids", "dh_debug_guesses", "dh_debug_votes", "dh_debug_meanshift", "dh_debug_hit_counts", "dh_debug_geometry",
]


class DepthheadError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load():
    """Load the HIP library; raises if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("DH_NO_TORCH_PRELOAD") != "1":
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one process do
        # not both see the GPU, so when torch is installed it is imported FIRST: the loader then
        # resolves this library's libamdhip64.so.7 to the copy torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = os.environ.get("DH_LIB_PATH") or LIB_PATH   # tools/ only: the -DDH_PROFILING_KNOBS twin
    if not os.path.exists(path):
        raise ImportError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the depthhead_amd product path has no CPU fallback)")
    lib = C.CDLL(path)
    for name in EXPORTS:
        fn = getattr(lib, name)   # AttributeError if an export is missing
        fn.restype = C.c_int
    lib.dh_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int):
    if rc != DH_OK:
        raise DepthheadError(rc, load().dh_last_error().decode("utf-8", "replace"))


def vp(x):
    """void* from a numpy array, an int address (device pointer) or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    return C.c_void_p(int(x))


def pinned_empty(shape, dtype) -> np.ndarray:
    """A numpy array in page-locked host memory (dh_host_alloc): frame batches filled in place are uploaded by
    asynchronous DMA at PCIe speed.  The memory is released when the array (and every view of it) is gone."""
    import weakref
    lib = load()
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    ptr = C.c_void_p()
    check(lib.dh_host_alloc(C.c_size_t(max(1, count * dt.itemsize)), C.byref(ptr)))
    raw = (C.c_uint8 * max(1, count * dt.itemsize)).from_address(ptr.value)
    weakref.finalize(raw, lib.dh_host_free, C.c_void_p(ptr.value))
    return np.frombuffer(raw, dtype=dt, count=count).reshape(shape)
