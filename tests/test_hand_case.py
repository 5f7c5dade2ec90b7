"""One end-to-end case small enough to verify on paper (SURVEY.md section 8c asks for it because
the reference holds no golden vector past its linear-algebra helpers).

Setup: 81x81 frame, 80x80 patch, stride 1 -> exactly one window position, centre (40, 40).
Left half of the frame (x < 40) is 1000 mm, right half 500 mm.  K = [[560,0,40],[0,560,40],[0,0,1]].
One tree, one split node: r1 = left half of the patch, r2 = right half, threshold 0:
    avg(r1) - avg(r2) = 1000 - 500 = 500 > 0  -> Binar::One -> leaf A (prob 1.0), else leaf B (prob 0).
Leaf A: offsets (10.5,-20.5,100.5), (11.5,-21.5,101.5); rotations (10,20,30), (12,22,32) degrees.

By hand (src/hough/prediction.rs line numbers):
  prob = 1.0 > 0.7 (:584); valtoadd = (1000*1.0) as usize / 2 = 500 (:594).
  rotations: mean (11,21,31), covariance diagonal (1+1)/1 = 2 each, trace 6 <= 400 (:600).
    bins (:605-613): 10*120/360 = 3.33 -> 3+60 = 63; 20 -> 6+60 = 66; 30 -> 10+60 = 70;
                     12 -> 4+60 = 64; 22 -> 7+60 = 67; 32 -> 10+60 = 70.
    rot = {(63,66,70): 500, (64,67,70): 500}; rough (:630) = (10,11,11) for both -> 1000.
    guess (:745-747): (10*360+180)/20 = 189, 207, 207 deg -> *120/360 = (63, 69, 69) (:458).
    mean shift, sigma 8 used as variance (weights exp(-d2/16) * 500):
      it 1 from (63,69,69): A d=(0,-3,1) d2=10 w=267.63; B d=(1,-2,1) d2=6 w=343.64
            x = 63.56 -> 63, y = 66.56 -> 66, z = 70
      it 2 from (63,66,70): A d2=0 w=500; B d2=2 w=441.25 -> (63.47, 66.47, 70) -> (63,66,70): fixed.
    rotation = ((63,66,70) - 60)/60 * 3.14159 = (0.1570795, 0.314159, 0.5235983...) (:477-482).
  offsets: centre z = img[40][40] = 500, p3 = (~0, ~0, 500) (:554); covariance diagonal 0.5, trace 1.5.
    votes (:647, :667): (-10.5,20.5,399.5) -> (-10,20,399); (-11.5,21.5,398.5) -> (-11,21,398).
    2-D (:661): K*np / z = (10100/399.5, 27460/399.5) = (25.28, 68.74) -> grid (25*20/81, 68*20/81) = (6,16)
                         (9500/398.5, 27980/398.5) = (23.84, 70.21) -> (5,17).
    first maximum (:694-702) is cell (6,16); cell size 81/20 = 4 -> pixels x 24..27, y 64..67, all 1000
    -> meanz 1000; centre (26, 66) -> 3-D ((26-40)/560*1000, (66-40)/560*1000, 1000) = (-25.0, 46.4, 1000)
    -> guess (-25, 46, 1000) (:750).  No vote within +-10 of it -> "zero sum" break on iteration 0
    (meanshift.rs:385-388) -> mid_point = the guess.
"""
import numpy as np

from depthhead_amd import synth
from depthhead_amd.forest import Forest, NODE_DTYPE


def hand_case():
    nodes = np.zeros(1, dtype=NODE_DTYPE)
    nodes[0] = ((0, 0, 40, 80), (40, 0, 80, 80), 0.0, ~1, ~0)     # child_zero -> leaf B (1), child_one -> leaf A (0)
    forest = Forest(np.array([0], dtype=np.int32), nodes, np.array([1.0, 0.0]),
                    np.array([0, 2, 2], dtype=np.uint32), np.array([0, 2, 2], dtype=np.uint32),
                    np.array([[10.5, -20.5, 100.5], [11.5, -21.5, 101.5]], dtype=np.float32),
                    np.array([[10.0, 20.0, 30.0], [12.0, 22.0, 32.0]]))
    img = np.full((81, 81), 500, dtype=np.uint16)
    img[:, :40] = 1000
    K = np.array([[560, 0, 40], [0, 560, 40], [0, 0, 1]], dtype=np.float32)
    model = synth.ModelParams(stepwidth=1)
    return forest, model, img, K


EXPECT_MID = np.array([-25.0, 46.0, 1000.0], dtype=np.float32)
EXPECT_ROT = np.array([(63 - 60.0) / 60.0 * 3.14159, (66 - 60.0) / 60.0 * 3.14159, (70 - 60.0) / 60.0 * 3.14159])


def check(res_mid, res_rot):
    assert np.array_equal(res_mid, EXPECT_MID), res_mid
    assert np.array_equal(res_rot, EXPECT_ROT), res_rot


def test_oracle_matches_paper(oracle):
    forest, model, img, K = hand_case()
    for mode in (oracle.RECT_FAITHFUL, oracle.RECT_SAT):
        r = oracle.predict(forest, model, img, K, rect_mode=mode)
        assert r.leaf_idx.tolist() == [[0]] and r.patch_flags.tolist() == [3]
        assert r.rot_cells.tolist() == [[63, 66, 70, 500], [64, 67, 70, 500]]
        assert r.mid_cells.tolist() == [[-11, 21, 398, 500], [-10, 20, 399, 500]]
        assert r.rot_grid[11 * 400 + 11 * 20 + 10] == 1000 and r.rot_grid.sum() == 1000
        assert r.pos_grid[16 * 20 + 6] == 500 and r.pos_grid[17 * 20 + 5] == 500 and r.pos_grid.sum() == 1000
        assert r.guess_rot.tolist() == [63, 69, 69] and r.guess_mid.tolist() == [-25, 46, 1000]
        assert r.ms_trace_rot[:3].tolist() == [[63, 69, 69], [63, 66, 70], [63, 66, 70]]
        assert r.ms_trace_mid.tolist() == [[-25, 46, 1000]]          # zero-sum break before any update
        check(r.mid_point, r.rotation)
