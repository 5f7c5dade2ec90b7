"""Host-side mirror of the reference's prediction API on top of the C ABI.

Names, argument meaning and error behaviour follow `src/hough/prediction.rs` of the reference:

* `IntrinsicMatrix`  (src/types.rs:405-446)
* `PredictionResult` (prediction.rs:259-267)
* `HoughPrediction.predict_parameter_parallel(img, intrinsic, midp_guess, rot_guess)` (:397-409),
  its serial twin `predict_parameter` (:376-388; identical results by construction), `update_sigma`
  / `sigma` (:320-331)

plus the batch entry points this framework adds (`predict_batch`, `predict_batch_device`).  All
arithmetic happens in libdepthhead_hip.so on the GPU; this module only marshals pointers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import POSE_DTYPE, check, vp
from .forest import Forest
from .synth import ModelParams


class IntrinsicMatrix:
    """Row-major 3x3 f32 camera matrix (src/types.rs:405)."""

    def __init__(self, mat):
        self.mat = np.ascontiguousarray(mat, dtype=np.float32).reshape(3, 3)

    @staticmethod
    def default_kinect_intrinsic() -> "IntrinsicMatrix":
        """src/types.rs:418-420"""
        return IntrinsicMatrix([[560.0, 0.0, 320.0], [0.0, 560.0, 240.0], [0.0, 0.0, 1.0]])


@dataclass
class PredictionResult:
    """prediction.rs:259-267; `bounding_box` is always Rect(0,0,0,0) in the reference (:491)."""
    mid_point: np.ndarray   # float32[3], mm
    rotation: np.ndarray    # float64[3], radians
    bounding_box: tuple = (0, 0, 0, 0)


class HoughPrediction:
    """GPU-resident predictor.  Like the reference type it is not thread-safe (`!Sync`,
    prediction.rs:253): use one instance per host thread / stream."""

    def __init__(self, forest: Forest, params: ModelParams | None = None, device: int = 0):
        self._lib = _lib.load()
        src = params or ModelParams()   # private copy: update_sigma must not alias the caller's object
        self.params = ModelParams(src.stepwidth, src.subimage_width, src.subimage_height, src.gaussian_sigma,
                                  src.meanshift_iterations)
        self.forest = forest
        self.device = device
        self._fh = C.c_void_p()
        self._ph = C.c_void_p()
        desc = _lib.ForestDesc(forest.n_trees, forest.roots.ctypes.data, forest.n_nodes, forest.nodes.ctypes.data,
                               forest.n_leaves, forest.leaf_prob.ctypes.data, forest.off_begin.ctypes.data,
                               forest.rot_begin.ctypes.data, forest.offsets.ctypes.data, forest.rotations.ctypes.data)
        check(self._lib.dh_forest_create(C.byref(desc), C.byref(self._fh)))
        prm = self._cparams()
        try:
            check(self._lib.dh_predictor_create(self._fh, C.byref(prm), C.c_int(device), C.byref(self._ph)))
        except Exception:
            self._lib.dh_forest_destroy(self._fh)
            self._fh = C.c_void_p()
            raise

    def _cparams(self) -> _lib.Params:
        p = self.params
        return _lib.Params(p.stepwidth, p.subimage_width, p.subimage_height, p.gaussian_sigma, p.meanshift_iterations)

    def close(self):
        if getattr(self, "_ph", None) and self._ph.value:
            self._lib.dh_predictor_destroy(self._ph)
            self._ph = C.c_void_p()
        if getattr(self, "_fh", None) and self._fh.value:
            self._lib.dh_forest_destroy(self._fh)
            self._fh = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- reference API -------------------------------------------------------------------
    @property
    def stepwidth(self) -> int:
        return self.params.stepwidth

    @property
    def meanshift_iterations(self) -> int:
        return self.params.meanshift_iterations

    def update_sigma(self, val: float) -> None:
        """prediction.rs:320-326"""
        check(self._lib.dh_predictor_update_sigma(self._ph, C.c_float(val)))
        out = C.c_float()
        check(self._lib.dh_predictor_sigma(self._ph, C.byref(out)))
        self.params.gaussian_sigma = float(out.value)

    def sigma(self) -> float:
        """prediction.rs:329-331"""
        out = C.c_float()
        check(self._lib.dh_predictor_sigma(self._ph, C.byref(out)))
        return float(out.value)

    def predict_parameter_parallel(self, img, intrinsic: IntrinsicMatrix, midp_guess=None, rot_guess=None) -> PredictionResult:
        """prediction.rs:397-409: one depth frame (H x W uint16) -> head position and rotation."""
        img = np.ascontiguousarray(img, dtype=np.uint16)
        if img.ndim != 2:
            raise ValueError("img must be a 2-D uint16 depth image")
        mg = None if midp_guess is None else np.asarray(midp_guess, dtype=np.float32).reshape(1, 3)
        rg = None if rot_guess is None else np.asarray(rot_guess, dtype=np.float64).reshape(1, 3)
        poses = self.predict_batch(img[None], intrinsic, mg, rg)
        return PredictionResult(poses["mid_point"][0].copy(), poses["rotation"][0].copy())

    # the serial twin returns identical results (prediction.rs:386 vs :407)
    predict_parameter = predict_parameter_parallel

    # ---- batch API -----------------------------------------------------------------------
    def predict_batch(self, frames, intrinsic: IntrinsicMatrix, midp_guess=None, rot_guess=None, guess_mask=None) -> np.ndarray:
        """Host frames [n, H, W] uint16 -> POSE_DTYPE[n].  Guess arrays are [n, 3] or None;
        guess_mask [n] uint8 (bit0 mid, bit1 rot) marks which frames carry a guess."""
        frames = np.ascontiguousarray(frames, dtype=np.uint16)
        if frames.ndim != 3:
            raise ValueError("frames must be [n, H, W]")
        n, h, w = frames.shape
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        mg = None if midp_guess is None else np.ascontiguousarray(midp_guess, dtype=np.float32).reshape(n, 3)
        rg = None if rot_guess is None else np.ascontiguousarray(rot_guess, dtype=np.float64).reshape(n, 3)
        gm = None if guess_mask is None else np.ascontiguousarray(guess_mask, dtype=np.uint8).reshape(n)
        out = np.zeros(n, dtype=POSE_DTYPE)
        check(self._lib.dh_predict_batch(self._ph, vp(frames), C.c_int(n), C.c_int(w), C.c_int(h), vp(K), vp(mg),
                                         vp(rg), vp(gm), vp(out)))
        return out

    @staticmethod
    def _payload_arrays(payloads):
        bufs = [np.frombuffer(b, dtype=np.uint8) for b in payloads]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * len(bufs))(*[b.size for b in bufs])
        return bufs, ptrs, lens

    def predict_batch_rle(self, payloads, intrinsic: IntrinsicMatrix, midp_guess=None, rot_guess=None, guess_mask=None) -> np.ndarray:
        """Frames handed over as BIWI run-length coded depth payloads (the bytes of the `.bin` files,
        biwi.rs:81-103): decoded on the device, then predicted.  -> POSE_DTYPE[n]."""
        n = len(payloads)
        bufs, ptrs, lens = self._payload_arrays(payloads)
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        mg = None if midp_guess is None else np.ascontiguousarray(midp_guess, dtype=np.float32).reshape(n, 3)
        rg = None if rot_guess is None else np.ascontiguousarray(rot_guess, dtype=np.float64).reshape(n, 3)
        gm = None if guess_mask is None else np.ascontiguousarray(guess_mask, dtype=np.uint8).reshape(n)
        out = np.zeros(n, dtype=POSE_DTYPE)
        check(self._lib.dh_predict_batch_rle(self._ph, ptrs, lens, C.c_int(n), vp(K), vp(mg), vp(rg), vp(gm), vp(out)))
        return out

    def decode_depth_device(self, payloads, frames_ptr: int | None = None, cap_px: int = 0) -> tuple[int, int]:
        """BIWI payloads -> device frames at `frames_ptr` ([n][h][w] uint16); returns (w, h).  With
        frames_ptr None only validates and reports the size."""
        bufs, ptrs, lens = self._payload_arrays(payloads)
        w, h = C.c_uint32(), C.c_uint32()
        check(self._lib.dh_biwi_decode_depth_device(self._ph, ptrs, lens, C.c_int(len(payloads)), vp(frames_ptr), C.c_size_t(cap_px),
                                                    C.byref(w), C.byref(h)))
        return int(w.value), int(h.value)

    def predict_batch_device(self, frames_ptr: int, n: int, w: int, h: int, intrinsic: IntrinsicMatrix, out_ptr: int,
                             midp_guess_ptr: int | None = None, rot_guess_ptr: int | None = None,
                             guess_mask_ptr: int | None = None, stream: int = 0) -> None:
        """Device-resident batch: raw device addresses (e.g. `tensor.data_ptr()`) and a hipStream_t
        handle (`torch.cuda.current_stream().cuda_stream`).  Asynchronous."""
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        check(self._lib.dh_predict_batch_device(self._ph, vp(frames_ptr), C.c_int(n), C.c_int(w), C.c_int(h), vp(K),
                                                vp(midp_guess_ptr), vp(rot_guess_ptr), vp(guess_mask_ptr), vp(out_ptr),
                                                C.c_void_p(stream) if stream else None))

    # ---- sibling consumers of the walk (prediction.rs:760-905) -----------------------------
    def predict_mask(self, img) -> np.ndarray:
        """prediction.rs:850-905: uint8 mask [H, W] (or [n, H, W] for a batch) of per-window head
        probability * 255."""
        frames = np.ascontiguousarray(img, dtype=np.uint16)
        single = frames.ndim == 2
        if single:
            frames = frames[None]
        n, h, w = frames.shape
        out = np.zeros((n, h, w), dtype=np.uint8)
        check(self._lib.dh_predict_mask(self._ph, vp(frames), C.c_int(n), C.c_int(w), C.c_int(h), vp(out)))
        return out[0] if single else out

    def build_hough_votes(self, img, intrinsic: IntrinsicMatrix) -> np.ndarray:
        """Voting stage of `build_hough_image` (prediction.rs:760-840): the uint16 image BEFORE
        imageproc's gaussian_blur_f32 (:844), which is an external crate and is not applied here."""
        frames = np.ascontiguousarray(img, dtype=np.uint16)
        single = frames.ndim == 2
        if single:
            frames = frames[None]
        n, h, w = frames.shape
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        out = np.zeros((n, h, w), dtype=np.uint16)
        check(self._lib.dh_hough_image(self._ph, vp(frames), C.c_int(n), C.c_int(w), C.c_int(h), vp(K), vp(out)))
        return out[0] if single else out

    def build_hough_image(self, img, intrinsic: IntrinsicMatrix) -> np.ndarray:
        """prediction.rs:760-845 in full: the votes blurred by imageproc's gaussian_blur_f32(sigma = gaussian_sigma)
        (external crate, restated from its published algorithm: parity unpinned)."""
        frames = np.ascontiguousarray(img, dtype=np.uint16)
        single = frames.ndim == 2
        if single:
            frames = frames[None]
        n, h, w = frames.shape
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        out = np.zeros((n, h, w), dtype=np.uint16)
        check(self._lib.dh_build_hough_image(self._ph, vp(frames), C.c_int(n), C.c_int(w), C.c_int(h), vp(K), vp(out)))
        return out[0] if single else out

    def predict_parameter_from2dhough(self, img, intrinsic: IntrinsicMatrix):
        """prediction.rs:343-367: head position from the argmax of the blurred 2-D Hough image; rotation is always
        zero there.  One frame -> PredictionResult, a batch [n, H, W] -> POSE_DTYPE[n]."""
        frames = np.ascontiguousarray(img, dtype=np.uint16)
        single = frames.ndim == 2
        if single:
            frames = frames[None]
        n, h, w = frames.shape
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        out = np.zeros(n, dtype=POSE_DTYPE)
        check(self._lib.dh_predict_from2dhough(self._ph, vp(frames), C.c_int(n), C.c_int(w), C.c_int(h), vp(K), vp(out)))
        return PredictionResult(out["mid_point"][0].copy(), out["rotation"][0].copy()) if single else out

    def graph_capture(self, frames_ptr: int, n: int, w: int, h: int, intrinsic: IntrinsicMatrix, out_ptr: int,
                      midp_guess_ptr: int | None = None, rot_guess_ptr: int | None = None,
                      guess_mask_ptr: int | None = None) -> None:
        """Capture one device-resident batch into a hipGraph (pointers are baked in)."""
        K = np.ascontiguousarray(intrinsic.mat, dtype=np.float32).reshape(9)
        check(self._lib.dh_graph_capture(self._ph, vp(frames_ptr), C.c_int(n), C.c_int(w), C.c_int(h), vp(K),
                                         vp(midp_guess_ptr), vp(rot_guess_ptr), vp(guess_mask_ptr), vp(out_ptr)))

    def graph_launch(self, stream: int = 0) -> None:
        check(self._lib.dh_graph_launch(self._ph, C.c_void_p(stream) if stream else None))

    def reserve(self, n: int, w: int, h: int) -> None:
        check(self._lib.dh_predictor_reserve(self._ph, C.c_int(n), C.c_int(w), C.c_int(h)))

    def set_forking(self, chunks: int) -> None:
        """Forked sub-batches inside one device call (`dh_predictor_set_forking`): 0 automatic, 1 never, 2 .. 8 forced."""
        check(self._lib.dh_predictor_set_forking(self._ph, C.c_int(chunks)))

    def patch_grid(self, w: int, h: int) -> tuple[int, int]:
        nx, ny = C.c_int(), C.c_int()
        prm = self._cparams()
        check(self._lib.dh_patch_grid(C.byref(prm), C.c_int(w), C.c_int(h), C.byref(nx), C.byref(ny)))
        return nx.value, ny.value

    # ---- profiling -----------------------------------------------------------------------
    def set_profiling(self, on: bool) -> None:
        check(self._lib.dh_set_profiling(self._ph, C.c_int(1 if on else 0)))

    def timing(self) -> dict:
        t = _lib.Timing()
        check(self._lib.dh_get_timing(self._ph, C.byref(t)))
        return {"boxsum_ms": t.boxsum_ms, "traverse_ms": t.traverse_ms, "emit_ms": t.emit_ms, "vote_ms": t.vote_ms,
                "cluster_ms": t.cluster_ms, "total_ms": t.total_ms, "n_frames": t.n_frames}

    # ---- parity taps (tests) -------------------------------------------------------------
    def debug_enable(self, on: bool = True) -> None:
        check(self._lib.dh_debug_enable(self._ph, C.c_int(1 if on else 0)))

    def debug_leaf_indices(self, n: int, w: int, h: int) -> np.ndarray:
        nx, ny = self.patch_grid(w, h)
        out = np.zeros((n, nx * ny, self.forest.n_trees), dtype=np.int32)
        check(self._lib.dh_debug_leaf_indices(self._ph, vp(out), C.c_size_t(out.size)))
        return out

    def debug_patch_flags(self, n: int, w: int, h: int) -> np.ndarray:
        nx, ny = self.patch_grid(w, h)
        out = np.zeros((n, nx * ny), dtype=np.uint8)
        check(self._lib.dh_debug_patch_flags(self._ph, vp(out), C.c_size_t(out.size)))
        return out

    def debug_grids(self, n: int):
        pos = np.zeros((n, 400), dtype=np.uint32)
        rot = np.zeros((n, 8000), dtype=np.uint32)
        check(self._lib.dh_debug_grids(self._ph, vp(pos), vp(rot)))
        return pos, rot

    def debug_guesses(self, n: int) -> np.ndarray:
        out = np.zeros((n, 6), dtype=np.int32)
        check(self._lib.dh_debug_guesses(self._ph, vp(out)))
        return out

    def debug_geometry(self) -> dict:
        out = np.zeros(10, dtype=np.int32)
        check(self._lib.dh_debug_geometry(self._ph, vp(out)))
        keys = ("uniform", "px", "py", "tiles_x", "tiles_y", "swz_log2", "swz_q", "ss_row", "rw", "rh")
        geo = dict(zip(keys, (int(x) for x in out)))
        path = geo["uniform"]
        geo.update(uniform=path & 1, walk_table=(path >> 8) & 1, top_levels=path >> 16)
        return geo

    def debug_hit_counts(self, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint32)
        check(self._lib.dh_debug_hit_counts(self._ph, vp(out)))
        return out

    def debug_votes(self, frame: int, which: int, cap: int = 1 << 22) -> np.ndarray:
        """All votes of one frame as [k, 4] (x, y, z, value), unaggregated."""
        out = np.zeros((cap, 4), dtype=np.int32)
        cnt = C.c_size_t()
        check(self._lib.dh_debug_votes(self._ph, C.c_int(frame), C.c_int(which), vp(out), C.c_size_t(cap), C.byref(cnt)))
        if cnt.value > cap:
            return self.debug_votes(frame, which, int(cnt.value))
        return out[: cnt.value].copy()

    def debug_meanshift(self, n: int, which: int):
        it = self.params.meanshift_iterations
        trace = np.zeros((n, it + 1, 3), dtype=np.int32)
        steps = np.zeros(n, dtype=np.uint32)
        check(self._lib.dh_debug_meanshift(self._ph, C.c_int(which), vp(trace), vp(steps)))
        return trace, steps


def aggregate_votes(votes: np.ndarray) -> np.ndarray:
    """Sum (x,y,z,value) records per cell with u32 wrap-around -> sorted [k,4], the form in which
    the oracle exports the reference's SparseArray3D (src/meanshift.rs:14-68)."""
    if votes.shape[0] == 0:
        return np.zeros((0, 4), dtype=np.int32)
    order = np.lexsort((votes[:, 2], votes[:, 1], votes[:, 0]))
    v = votes[order]
    new = np.ones(v.shape[0], dtype=bool)
    new[1:] = np.any(v[1:, :3] != v[:-1, :3], axis=1)
    idx = np.flatnonzero(new)
    sums = np.add.reduceat(v[:, 3].astype(np.uint32).astype(np.uint64), idx) & np.uint64(0xFFFFFFFF)
    out = np.empty((idx.size, 4), dtype=np.int32)
    out[:, :3] = v[idx, :3]
    out[:, 3] = sums.astype(np.uint32).view(np.int32)
    return out
