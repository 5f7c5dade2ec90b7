"""CPU ORACLE (test infrastructure) for the BIWI file formats: a pure-Python restatement of
/root/reference/src/db_reader/biwi.rs read_depth (:81-103), read_cal (:27-60), read_gt (:63-77).
Only tests/ import this.  Parity unpinned: the reference holds no fixture for these parsers."""
import re
import struct

import numpy as np

_FLOAT = re.compile(r"(\d+[\.\d+]*)")     # biwi.rs:31


def read_depth(data: bytes) -> np.ndarray:
    w, h = struct.unpack_from("<II", data, 0)                     # :83-84
    pos, p = 8, 0
    out = np.zeros(w * h, dtype=np.uint16)                        # :86
    while p < w * h:                                              # :89
        (n_empty,) = struct.unpack_from("<I", data, pos); pos += 4     # :90 (struct.error == io::Error on a short file)
        if p + n_empty > w * h:
            raise IndexError("iterator exhausted")                # it.next().unwrap() panics (:92)
        p += n_empty
        (n_full,) = struct.unpack_from("<I", data, pos); pos += 4      # :94
        for _ in range(n_full):                                   # :95-98
            (v,) = struct.unpack_from("<H", data, pos); pos += 2
            if p >= w * h:
                raise IndexError("iterator exhausted")
            out[p] = v
            p += 1
    return out.reshape(h, w)


def read_cal(text: str) -> np.ndarray:
    lines = text.split("\n")
    res = np.zeros((3, 3), dtype=np.float32)
    for j in range(3):                                            # :35
        line = lines[j] if j < len(lines) else ""
        found = 0
        for m in _FLOAT.finditer(line):                           # :38
            if found == 3:
                raise ValueError("Unsupported Calibration-File")  # res[j][3] is out of bounds (:46) / :48
            tok = m.group(1)
            if not re.fullmatch(r"\d+(\.\d*)?", tok):
                raise ValueError("invalid float literal")         # f32::from_str (:46)
            res[j, found] = np.float32(tok)
            found += 1
        if found != 3:
            raise ValueError("Unsupported Calibration-File")      # :54
    return res


def read_gt(data: bytes, K: np.ndarray):
    v = np.array(struct.unpack_from("<6f", data, 0), dtype=np.float32)      # :66-68
    K = np.asarray(K, dtype=np.float32)
    r = np.zeros(3, dtype=np.float32)
    for j in range(3):                                            # Mat3 * Vec3, meancov_estimation.rs:201-216
        t = np.float32(v[0] * K[j, 0])
        t = np.float32(t + np.float32(v[1] * K[j, 1]))
        t = np.float32(t + np.float32(v[2] * K[j, 2]))
        r[j] = t
    with np.errstate(all="ignore"):
        p2 = np.array([r[0] / r[2], r[1] / r[2]], dtype=np.float32)         # types.rs:426-427
    return v[:3].copy(), p2, v[3:].copy()
