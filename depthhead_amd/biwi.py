"""BIWI Kinect Head Pose Database ingest.

Mirrors `/root/reference/src/db_reader/biwi.rs`: `read_depth` (:81-103), `read_cal` (:27-60),
`read_gt` (:63-77) and the directory walk of `BiwiReader::person` (:263-314).  The three format
decoders run in libdepthhead_hip.so (csrc/dh_biwi.cpp); this module marshals bytes and walks
directories.  `encode_depth` is the inverse of `read_depth`, used to write fixtures and to
round-trip test the decoder (the database itself cannot be downloaded here).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, vp

_BIWI_NAME = re.compile(r"(frame_\d+)_.+")   # biwi.rs:267


def read_depth(data: bytes) -> np.ndarray:
    """Run-length coded depth `.bin` -> uint16 [h, w] (biwi.rs:81-103)."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    check(lib.dh_biwi_decode_depth(vp(buf), C.c_size_t(buf.size), None, C.c_size_t(0), C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value), dtype=np.uint16)
    check(lib.dh_biwi_decode_depth(vp(buf), C.c_size_t(buf.size), vp(out), C.c_size_t(out.size), C.byref(w), C.byref(h)))
    return out


def read_cal(text: bytes | str) -> np.ndarray:
    """`depth.cal` -> float32 3x3 intrinsic (biwi.rs:27-60)."""
    if isinstance(text, str):
        text = text.encode()
    K = np.zeros(9, dtype=np.float32)
    check(_lib.load().dh_biwi_parse_cal(C.c_char_p(text), C.c_size_t(len(text)), vp(K)))
    return K.reshape(3, 3)


@dataclass
class GT:
    """biwi.rs `GT`: ground-truth head pose of one frame."""
    pos3d: np.ndarray   # float32[3], mm
    pos2d: np.ndarray   # float32[2], pixels
    rot: np.ndarray     # float32[3], degrees


def read_gt(data: bytes, intrinsic: np.ndarray) -> GT:
    """24-byte `*_pose.bin` (biwi.rs:63-77)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    K = np.ascontiguousarray(intrinsic, dtype=np.float32).reshape(9)
    p3, p2, rot = np.zeros(3, np.float32), np.zeros(2, np.float32), np.zeros(3, np.float32)
    check(_lib.load().dh_biwi_parse_pose(vp(buf), C.c_size_t(buf.size), vp(K), vp(p3), vp(p2), vp(rot)))
    return GT(p3, p2, rot)


def encode_depth(img: np.ndarray) -> bytes:
    """Inverse of `read_depth`: alternate runs of zero and non-zero pixels in row-major order."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    flat = img.ravel()
    out = [np.array([w, h], dtype="<u4").tobytes()]
    nz = flat != 0
    edges = np.flatnonzero(np.diff(np.concatenate([[False], nz, [False]]).astype(np.int8)))   # run starts / ends of the non-zero runs
    pos = 0
    for a, b in zip(edges[0::2], edges[1::2]):
        out.append(np.array([a - pos, b - a], dtype="<u4").tobytes())
        out.append(flat[a:b].astype("<u2").tobytes())
        pos = b
    if pos < flat.size or flat.size == 0:
        out.append(np.array([flat.size - pos, 0], dtype="<u4").tobytes())
    return b"".join(out)


@dataclass
class DepthTrue:
    """reader.rs:69-78 `DepthTrue` (the mask is only needed for training and is left as a path)."""
    trans: GT
    depth: np.ndarray
    mask_path: str
    intrinsic: np.ndarray
    name: str


class BiwiReader:
    """biwi.rs:188-339.  `person(nr)` yields the frames of one subject in file-name order, skipping
    frames without a mask or a pose file exactly like the reference (:281-284)."""

    def __init__(self, mask_dir: str, depth_dir: str, truth_dir: str):
        self.mask_dir, self.depth_dir, self.truth_dir = mask_dir, depth_dir, truth_dir

    @staticmethod
    def _check_common(path: str) -> int:
        if not os.path.isdir(path):
            raise NotADirectoryError(path)
        dirs = [d for d in os.listdir(path) if os.path.isdir(os.path.join(path, d))]
        for i in range(1, len(dirs) + 1):
            if f"{i:02d}" not in dirs:
                raise ValueError(f"invalid directory structure: {path}")
        return len(dirs)

    def is_valid(self) -> bool:
        n = [self._check_common(p) for p in (self.mask_dir, self.depth_dir, self.truth_dir)]
        return n[0] == n[1] == n[2]

    def person_count(self) -> int:
        return self._check_common(self.depth_dir) if self.is_valid() else 0

    def person(self, nr: int):
        sub = f"{nr:02d}"
        ddir = os.path.join(self.depth_dir, sub)
        names = sorted(f for f in os.listdir(ddir) if f.endswith(".bin"))
        with open(os.path.join(ddir, "depth.cal"), "rb") as fh:
            K = read_cal(fh.read())
        for fname in names:
            m = _BIWI_NAME.match(fname)
            if not m:
                raise ValueError("Invalid filename found")
            prefix = m.group(1)
            mask = os.path.join(self.mask_dir, sub, f"{prefix}_depth_mask.png")
            truth = os.path.join(self.truth_dir, sub, f"{prefix}_pose.bin")
            if not os.path.exists(mask) or not os.path.exists(truth):
                continue
            with open(truth, "rb") as fh:
                gt = read_gt(fh.read(), K)
            with open(os.path.join(ddir, fname), "rb") as fh:
                depth = read_depth(fh.read())
            yield DepthTrue(gt, depth, mask, K, prefix)
