"""Row N3 (frame ingest): the BIWI format decoders of the C-ABI library against the pure-Python
oracle restatement of src/db_reader/biwi.rs, plus encode -> decode round trips and the error
behaviour on malformed files.  Host-only: no GPU needed."""
import os
import struct

import numpy as np
import pytest

from depthhead_amd import biwi, synth
from depthhead_amd._lib import DepthheadError
from oracle import biwi_oracle as bo

CAL = """575.816 0 320
0 575.816 240
0 0 1

0 0 0 0

1 0 0
0 1 0
0 0 1

0 0 0

640 480
"""


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded(hip_lib):
    return hip_lib


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (320, 240, 2), (17, 5, 3), (1, 1, 4)])
def test_depth_roundtrip_and_oracle(w, h, seed):
    img = synth.biwi_like(max(w, 96), max(h, 96), synth.FRAME_SEED_BASE + seed)[:h, :w].copy()
    blob = biwi.encode_depth(img)
    got = biwi.read_depth(blob)
    assert got.dtype == np.uint16 and np.array_equal(got, img)
    assert np.array_equal(bo.read_depth(blob), img)
    if w * h >= 1000:
        assert len(blob) < img.nbytes or (img > 0).mean() > 0.5   # the format compresses sparse frames


def test_depth_edge_cases():
    zeros = np.zeros((4, 6), dtype=np.uint16)
    full = np.full((3, 5), 65535, dtype=np.uint16)
    for img in (zeros, full):
        blob = biwi.encode_depth(img)
        assert np.array_equal(biwi.read_depth(blob), img) and np.array_equal(bo.read_depth(blob), img)
    # hand-written stream: 2x3 image, runs (1 empty, 2 full), (3 empty, 0 full)
    blob = struct.pack("<II", 3, 2) + struct.pack("<II2H", 1, 2, 7, 9) + struct.pack("<II", 3, 0)
    exp = np.array([[0, 7, 9], [0, 0, 0]], dtype=np.uint16)
    assert np.array_equal(biwi.read_depth(blob), exp) and np.array_equal(bo.read_depth(blob), exp)
    # zero-size image: the loop does not run (biwi.rs:89)
    assert biwi.read_depth(struct.pack("<II", 0, 7)).shape == (7, 0)


def test_depth_malformed_inputs_fail_cleanly():
    good = biwi.encode_depth(synth.biwi_like(96, 96, 5)[:20, :30].copy())
    for cut in (0, 4, 7, 8, 11, len(good) - 1):                    # truncated: io::Error in the reference
        with pytest.raises(DepthheadError):
            biwi.read_depth(good[:cut])
        with pytest.raises((struct.error, IndexError)):
            bo.read_depth(good[:cut])
    over = struct.pack("<II", 3, 2) + struct.pack("<II", 7, 0)     # 7 empty pixels in a 6-pixel image: unwrap panics
    with pytest.raises(DepthheadError):
        biwi.read_depth(over)
    over2 = struct.pack("<II", 3, 2) + struct.pack("<II", 5, 2) + struct.pack("<2H", 1, 2)
    with pytest.raises(DepthheadError):
        biwi.read_depth(over2)


def test_calibration_parser():
    K = biwi.read_cal(CAL)
    assert K.dtype == np.float32 and np.array_equal(K, bo.read_cal(CAL))
    assert np.array_equal(K, np.array([[575.816, 0, 320], [0, 575.816, 240], [0, 0, 1]], dtype=np.float32))
    # the regex has no sign: "-5" parses as 5 (biwi.rs:31); exponents split a number in two
    assert biwi.read_cal("1 -5 2\n3 4 5\n6 7 8\n")[0, 1] == 5.0 == bo.read_cal("1 -5 2\n3 4 5\n6 7 8\n")[0, 1]
    assert biwi.read_cal("1. 2 3\n4 5 6\n7 8 9")[0, 0] == 1.0
    for bad in ("1 2\n3 4 5\n6 7 8\n", "1 2 3 4\n5 6 7\n8 9 1\n", "1 2 3\n4 5 6\n", "1e5 2 3\n4 5 6\n7 8 9\n",
                "1.2.3 4 5\n6 7 8\n9 1 2\n", "1+2 3 4\n5 6 7\n8 9 1\n", ""):
        with pytest.raises(DepthheadError):
            biwi.read_cal(bad)
        with pytest.raises(ValueError):
            bo.read_cal(bad)


def test_pose_parser():
    K = biwi.read_cal(CAL)
    vals = np.array([12.5, -40.25, 880.0, 5.5, -10.25, 3.0], dtype="<f4")
    gt = biwi.read_gt(vals.tobytes(), K)
    p3, p2, rot = bo.read_gt(vals.tobytes(), K)
    assert np.array_equal(gt.pos3d, p3) and np.array_equal(gt.pos2d, p2) and np.array_equal(gt.rot, rot)
    assert np.array_equal(gt.pos3d, vals[:3]) and np.array_equal(gt.rot, vals[3:])
    # src/types.rs:476-488 (test_intrinsic): dense K, point (11, 12, 32.2) -> (1.15896578, 0.21143073)
    Kd = np.array([[22.0, 11.4, 12.11], [2.1, 4.1, 2.11], [1.3, 3.1, 19.0]], dtype=np.float32)
    gt = biwi.read_gt(np.array([11.0, 12.0, 32.2, 0, 0, 0], dtype="<f4").tobytes(), Kd)
    assert abs(gt.pos2d[0] - 1.15896578) < 1e-4 and abs(gt.pos2d[1] - 0.21143073) < 1e-4
    with pytest.raises(DepthheadError):
        biwi.read_gt(vals.tobytes()[:23], K)


def test_directory_walk(tmp_path):
    """BiwiReader::person (biwi.rs:263-314) on a two-subject database written with the encoder."""
    roots = {k: tmp_path / k for k in ("head_pose_masks", "hpdb", "db_annotations")}
    frames = {}
    for person in (1, 2):
        for r in roots.values():
            (r / f"{person:02d}").mkdir(parents=True)
        (roots["hpdb"] / f"{person:02d}" / "depth.cal").write_text(CAL)
        for i in (3, 4, 10):
            img = synth.biwi_like(160, 120, 1000 * person + i)
            name = f"frame_{i:05d}"
            (roots["hpdb"] / f"{person:02d}" / f"{name}_depth.bin").write_bytes(biwi.encode_depth(img))
            if i != 10:                                            # frame 10 has no mask: skipped (:281-284)
                (roots["head_pose_masks"] / f"{person:02d}" / f"{name}_depth_mask.png").write_bytes(b"png")
            (roots["db_annotations"] / f"{person:02d}" / f"{name}_pose.bin").write_bytes(
                np.array([i, person, 900.0, 1, 2, 3], dtype="<f4").tobytes())
            frames[(person, name)] = img
    rd = biwi.BiwiReader(str(roots["head_pose_masks"]), str(roots["hpdb"]), str(roots["db_annotations"]))
    assert rd.is_valid() and rd.person_count() == 2
    for person in (1, 2):
        got = list(rd.person(person))
        assert [g.name for g in got] == ["frame_00003", "frame_00004"]
        for g in got:
            assert np.array_equal(g.depth, frames[(person, g.name)])
            assert g.trans.pos3d[1] == person and g.intrinsic[0, 0] == np.float32(575.816)
    os.rmdir(roots["hpdb"] / "02" / "x") if (roots["hpdb"] / "02" / "x").exists() else None
    (roots["hpdb"] / "04").mkdir()                                 # 01, 02, 04: not a dense numbering (:245-252)
    with pytest.raises(ValueError):
        rd.is_valid()
