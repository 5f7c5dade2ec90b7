"""ctypes front-end of the CPU ORACLE (oracle/dh_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under depthhead_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libdh_oracle.so")

RECT_FAITHFUL = 0
RECT_SAT = 1


class _Forest(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("roots", C.c_void_p), ("n_nodes", C.c_uint32), ("nodes", C.c_void_p),
                ("n_leaves", C.c_uint32), ("leaf_prob", C.c_void_p), ("off_begin", C.c_void_p),
                ("rot_begin", C.c_void_p), ("offsets", C.c_void_p), ("rotations", C.c_void_p)]


class _Model(C.Structure):
    _fields_ = [("stepwidth", C.c_uint32), ("subimage_width", C.c_uint32), ("subimage_height", C.c_uint32),
                ("gaussian_sigma", C.c_float), ("meanshift_iterations", C.c_uint32)]


class _Taps(C.Structure):
    _fields_ = [("leaf_idx", C.c_void_p), ("patch_flags", C.c_void_p), ("pos_grid", C.c_void_p),
                ("rot_grid", C.c_void_p), ("guess_mid", C.c_void_p), ("guess_rot_deg", C.c_void_p),
                ("guess_rot", C.c_void_p), ("mid_cells", C.c_void_p), ("mid_cap", C.c_uint32),
                ("mid_count", C.c_void_p), ("rot_cells", C.c_void_p), ("rot_cap", C.c_uint32),
                ("rot_count", C.c_void_p), ("ms_trace_mid", C.c_void_p), ("ms_steps_mid", C.c_void_p),
                ("ms_trace_rot", C.c_void_p), ("ms_steps_rot", C.c_void_p)]


class _Pose(C.Structure):
    _fields_ = [("mid_point", C.c_float * 3), ("rotation", C.c_double * 3)]


POSE_DTYPE = np.dtype([("mid_point", "<f4", (3,)), ("rotation", "<f8", (3,))], align=True)
assert POSE_DTYPE.itemsize == C.sizeof(_Pose) == 40


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, n) for n in ("dh_oracle.c", "dh_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_predict.restype = C.c_int
        _lib.orc_predict_batch.restype = C.c_int
        _lib.orc_mat3_det_f64.restype = C.c_double
        _lib.orc_trace_f64.restype = C.c_double
        _lib.orc_average_value_in_rect.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _forest_struct(f) -> _Forest:
    return _Forest(f.n_trees, f.roots.ctypes.data, f.n_nodes, f.nodes.ctypes.data, f.n_leaves,
                   f.leaf_prob.ctypes.data, f.off_begin.ctypes.data, f.rot_begin.ctypes.data,
                   f.offsets.ctypes.data, f.rotations.ctypes.data)


def _model_struct(m) -> _Model:
    return _Model(m.stepwidth, m.subimage_width, m.subimage_height, m.gaussian_sigma, m.meanshift_iterations)


@dataclass
class OracleResult:
    mid_point: np.ndarray
    rotation: np.ndarray
    leaf_idx: np.ndarray = None
    patch_flags: np.ndarray = None
    pos_grid: np.ndarray = None
    rot_grid: np.ndarray = None
    guess_mid: np.ndarray = None
    guess_rot: np.ndarray = None
    guess_rot_deg: np.ndarray = None
    mid_cells: np.ndarray = None   # [k,4] sorted (x,y,z,value)
    rot_cells: np.ndarray = None
    ms_trace_mid: np.ndarray = None
    ms_trace_rot: np.ndarray = None
    extra: dict = field(default_factory=dict)


def predict(forest, model, img: np.ndarray, K: np.ndarray, midp_guess=None, rot_guess=None,
            rect_mode: int = RECT_SAT, taps: bool = True, cell_cap: int = 1 << 22) -> OracleResult:
    """One frame through the oracle, optionally with every intermediate."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    mg = None if midp_guess is None else np.ascontiguousarray(midp_guess, dtype=np.float32)
    rg = None if rot_guess is None else np.ascontiguousarray(rot_guess, dtype=np.float64)
    fs, ms = _forest_struct(forest), _model_struct(model)
    pose = _Pose()
    tp = None
    keep = {}
    if taps:
        nx, ny = model.patch_grid(w, h)
        npatch, T, it = nx * ny, forest.n_trees, model.meanshift_iterations
        keep = dict(
            leaf_idx=np.full((max(npatch, 1), T), -2, dtype=np.int32), patch_flags=np.zeros(max(npatch, 1), dtype=np.uint8),
            pos_grid=np.zeros(400, dtype=np.uint32), rot_grid=np.zeros(8000, dtype=np.uint32),
            guess_mid=np.zeros(3, dtype=np.int32), guess_rot_deg=np.zeros(3, dtype=np.float64),
            guess_rot=np.zeros(3, dtype=np.int32), mid_cells=np.zeros((cell_cap, 4), dtype=np.int32),
            mid_count=np.zeros(1, dtype=np.uint32), rot_cells=np.zeros((cell_cap, 4), dtype=np.int32),
            rot_count=np.zeros(1, dtype=np.uint32), ms_trace_mid=np.zeros((it + 1, 3), dtype=np.int32),
            ms_steps_mid=np.zeros(1, dtype=np.uint32), ms_trace_rot=np.zeros((it + 1, 3), dtype=np.int32),
            ms_steps_rot=np.zeros(1, dtype=np.uint32))
        k = keep
        tp = _Taps(k["leaf_idx"].ctypes.data, k["patch_flags"].ctypes.data, k["pos_grid"].ctypes.data,
                   k["rot_grid"].ctypes.data, k["guess_mid"].ctypes.data, k["guess_rot_deg"].ctypes.data,
                   k["guess_rot"].ctypes.data, k["mid_cells"].ctypes.data, cell_cap, k["mid_count"].ctypes.data,
                   k["rot_cells"].ctypes.data, cell_cap, k["rot_count"].ctypes.data, k["ms_trace_mid"].ctypes.data,
                   k["ms_steps_mid"].ctypes.data, k["ms_trace_rot"].ctypes.data, k["ms_steps_rot"].ctypes.data)
    rc = lib().orc_predict(C.byref(fs), C.byref(ms), _p(img), C.c_uint32(w), C.c_uint32(h), _p(K), _p(mg), _p(rg),
                           C.c_int(rect_mode), C.byref(pose), C.byref(tp) if tp is not None else None)
    if rc != 0:
        raise ValueError(f"orc_predict failed: {rc}")
    res = OracleResult(np.array(pose.mid_point[:], dtype=np.float32), np.array(pose.rotation[:], dtype=np.float64))
    if taps:
        nx, ny = model.patch_grid(w, h)
        npatch = nx * ny
        res.leaf_idx = keep["leaf_idx"][:npatch]
        res.patch_flags = keep["patch_flags"][:npatch]
        for name in ("pos_grid", "rot_grid", "guess_mid", "guess_rot", "guess_rot_deg"):
            setattr(res, name, keep[name])
        nm, nr = int(keep["mid_count"][0]), int(keep["rot_count"][0])
        if nm > cell_cap or nr > cell_cap:
            raise ValueError("cell_cap too small")
        res.mid_cells = keep["mid_cells"][:nm].copy()
        res.rot_cells = keep["rot_cells"][:nr].copy()
        res.ms_trace_mid = keep["ms_trace_mid"][: int(keep["ms_steps_mid"][0]) + 1].copy()
        res.ms_trace_rot = keep["ms_trace_rot"][: int(keep["ms_steps_rot"][0]) + 1].copy()
    return res


def predict_batch(forest, model, imgs: np.ndarray, K: np.ndarray, midp_guess=None, rot_guess=None,
                  rect_mode: int = RECT_SAT, threads: int = 0) -> np.ndarray:
    """Frame-parallel oracle over a batch; returns a POSE_DTYPE array."""
    imgs = np.ascontiguousarray(imgs, dtype=np.uint16)
    n, h, w = imgs.shape
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    mg = None if midp_guess is None else np.ascontiguousarray(midp_guess, dtype=np.float32).reshape(n, 3)
    rg = None if rot_guess is None else np.ascontiguousarray(rot_guess, dtype=np.float64).reshape(n, 3)
    fs, ms = _forest_struct(forest), _model_struct(model)
    out = np.zeros(n, dtype=POSE_DTYPE)
    rc = lib().orc_predict_batch(C.byref(fs), C.byref(ms), _p(imgs), C.c_uint32(n), C.c_uint32(w), C.c_uint32(h),
                                 _p(K), _p(mg), _p(rg), C.c_int(rect_mode), C.c_int(threads), _p(out))
    if rc != 0:
        raise ValueError(f"orc_predict_batch failed: {rc}")
    return out


def predict_mask(forest, model, img: np.ndarray, rect_mode: int = RECT_SAT) -> np.ndarray:
    """HoughPrediction::predict_mask (prediction.rs:850-905) -> uint8 [h, w]."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    fs, ms = _forest_struct(forest), _model_struct(model)
    out = np.zeros((h, w), dtype=np.uint8)
    rc = lib().orc_predict_mask(C.byref(fs), C.byref(ms), _p(img), C.c_uint32(w), C.c_uint32(h), C.c_int(rect_mode), _p(out))
    if rc != 0:
        raise ValueError(f"orc_predict_mask failed: {rc}")
    return out


def hough_image(forest, model, img: np.ndarray, K: np.ndarray, rect_mode: int = RECT_SAT) -> np.ndarray:
    """Voting stage of HoughPrediction::build_hough_image (prediction.rs:760-840), before the blur."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    fs, ms = _forest_struct(forest), _model_struct(model)
    out = np.zeros((h, w), dtype=np.uint16)
    rc = lib().orc_hough_image(C.byref(fs), C.byref(ms), _p(img), C.c_uint32(w), C.c_uint32(h), _p(K), C.c_int(rect_mode), _p(out))
    if rc != 0:
        raise ValueError(f"orc_hough_image failed: {rc}")
    return out


def gaussian_kernel(sigma: float) -> np.ndarray:
    """imageproc 0.12.0 filter::gaussian_kernel_f32 (restated; parity unpinned)."""
    n = C.c_uint32()
    if lib().orc_gaussian_kernel_f32(C.c_float(sigma), None, C.c_uint32(0), C.byref(n)) != 0:
        raise ValueError("sigma must be > 0")
    out = np.zeros(n.value, dtype=np.float32)
    lib().orc_gaussian_kernel_f32(C.c_float(sigma), _p(out), C.c_uint32(out.size), C.byref(n))
    return out


def gaussian_blur_u16(img: np.ndarray, sigma: float) -> np.ndarray:
    """imageproc 0.12.0 filter::gaussian_blur_f32 on a u16 image (restated; parity unpinned)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    out = np.zeros((h, w), dtype=np.uint16)
    if lib().orc_gaussian_blur_u16(_p(img), C.c_uint32(w), C.c_uint32(h), C.c_float(sigma), _p(out)) != 0:
        raise ValueError("orc_gaussian_blur_u16 failed")
    return out


def build_hough_image(forest, model, img: np.ndarray, K: np.ndarray, rect_mode: int = RECT_SAT) -> np.ndarray:
    """HoughPrediction::build_hough_image in full (prediction.rs:760-845)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    fs, ms = _forest_struct(forest), _model_struct(model)
    out = np.zeros((h, w), dtype=np.uint16)
    rc = lib().orc_build_hough_image(C.byref(fs), C.byref(ms), _p(img), C.c_uint32(w), C.c_uint32(h), _p(K), C.c_int(rect_mode), _p(out))
    if rc != 0:
        raise ValueError(f"orc_build_hough_image failed: {rc}")
    return out


def predict_from2dhough(forest, model, img: np.ndarray, K: np.ndarray, rect_mode: int = RECT_SAT):
    """HoughPrediction::predict_parameter_from2dhough (prediction.rs:343-367) -> (mid_point f32[3], rotation f64[3])."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    fs, ms = _forest_struct(forest), _model_struct(model)
    out = np.zeros(1, dtype=POSE_DTYPE)
    rc = lib().orc_predict_from2dhough(C.byref(fs), C.byref(ms), _p(img), C.c_uint32(w), C.c_uint32(h), _p(K), C.c_int(rect_mode), _p(out))
    if rc != 0:
        raise ValueError(f"orc_predict_from2dhough failed: {rc}")
    return out["mid_point"][0].copy(), out["rotation"][0].copy()
