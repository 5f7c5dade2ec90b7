"""Extended randomized differential run (GPU box): like tests/test_gpu_parity.py::test_randomized_differential but
with many more seeded configurations, biased towards multi-tile frames and uniform forests.  Usage:
    python tools/fuzz_parity.py [first_case] [n_cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import biwi, synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
from oracle import pyoracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # larger frames, up to 70 trees
tiny = len(sys.argv) > 3 and sys.argv[3] == "tiny"    # frames barely larger than the patch, small and odd patches
api = len(sys.argv) > 3 and sys.argv[3] == "api"      # also: guesses, dense intrinsics, predict_mask, 2-D Hough votes
bad = 0
refused = 0
frames_done = 0
hits_total = 0
t0 = time.time()
for case in range(first, first + count):
    if (case - first) % 1000 == 0:
        print(f"... case {case}, {bad} mismatching frames so far, {time.time() - t0:.0f} s", flush=True)
    rs = np.random.RandomState(777000 + case)
    sw, sh = int(rs.randint(16, 97)), int(rs.randint(16, 97))
    w, h = sw + int(rs.randint(1, 420)), sh + int(rs.randint(1, 300))
    if big:
        w, h = sw + int(rs.randint(300, 900)), sh + int(rs.randint(200, 640))
    if tiny:
        sw, sh = int(rs.randint(4, 60)), int(rs.randint(4, 60))
        w, h = sw + int(rs.randint(1, 40)), sh + int(rs.randint(1, 40))
    step = int(rs.choice([1, 2, 3, 4, 4, 4, 5, 6, 8, 10, 12]))
    trees, depth = int(rs.randint(1, 18)), int(rs.randint(1, 12))
    if big and rs.rand() < 0.3:
        trees = int(rs.randint(18, 71))
    mixed = rs.rand() < 0.2
    forest = synth.synth_forest(trees, depth, 9000 + case, patch=(sw, sh), rect_scale=float(rs.uniform(0.1, 0.6)),
                                rect_scale_max=float(rs.uniform(0.6, 0.9)) if mixed else None,
                                full_depth=int(rs.randint(0, depth + 1)), p_split=float(rs.uniform(0.4, 0.95)))
    # a third of the cases: some thresholds become small integers, whose product with any rectangle area is an integer -- such
    # nodes have an ambiguity band that only the f64 arithmetic decides (k_nodes_compact), and D = threshold * area is reachable
    # (own generator: the other draws of a case stay what they were)
    rs2 = np.random.RandomState((case * 2654435761) & 0x7fffffff)
    if len(forest.nodes) and rs2.rand() < 0.33:
        from depthhead_amd.forest import Forest
        nodes = forest.nodes.copy()
        pick = rs2.choice(len(nodes), int(rs2.randint(1, len(nodes) + 1)), replace=False)
        nodes["threshold"][pick] = rs2.randint(-2, 3, len(pick)).astype(np.float64)
        forest = Forest(forest.roots, nodes, forest.leaf_prob, forest.off_begin, forest.rot_begin, forest.offsets, forest.rotations)
    model = synth.ModelParams(stepwidth=step, subimage_width=sw, subimage_height=sh,
                              gaussian_sigma=float(rs.uniform(0.5, 30.0)), meanshift_iterations=int(rs.randint(0, 25)))
    n = int(rs.randint(1, 4))
    frames = np.stack([synth.biwi_like(max(w, 96), max(h, 96), 19000 + case * 7 + i)[:h, :w] for i in range(n)]).copy()
    if big:
        frames = np.tile(frames, (1, 1, 1))
    if rs.rand() < 0.25:
        frames[0] = (rs.rand(h, w) < 0.01) * rs.randint(1, 65536, (h, w))
    if rs.rand() < 0.25:
        frames[-1, :, : w // 2] = 0
    f = float(rs.uniform(200, 900))
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], dtype=np.float32)
    midp = rot = None
    if api:
        if rs.rand() < 0.5:
            K = np.array([[f, rs.uniform(-2, 2), w / 2 + rs.uniform(-20, 20)], [rs.uniform(-1, 1), f * rs.uniform(0.9, 1.1), h / 2], [0, 0, 1]], dtype=np.float32)
        if rs.rand() < 0.4:
            midp = rs.uniform(-300, 1500, (n, 3)).astype(np.float32)
        if rs.rand() < 0.4:
            rot = rs.uniform(-1.5, 1.5, (n, 3))
    if os.environ.get("FUZZ_TRACE"):
        with open(os.environ["FUZZ_TRACE"], "a") as tf:
            tf.write(f"case {case}: {w}x{h} patch {sw}x{sh} step {step} trees {trees} depth {depth} mixed {mixed} n {n} rect {forest.nodes['r1'][0] if forest.n_nodes else None}\n")
    try:
        with HoughPrediction(forest, model, device=0) as hp:
            hp.debug_enable(True)
            poses = hp.predict_batch(frames.astype(np.uint16), IntrinsicMatrix(K), midp, rot)
            leaf = hp.debug_leaf_indices(n, w, h)
            pg, rg = hp.debug_grids(n)
            hits_total += int(hp.debug_hit_counts(n).sum())
            # product mode (taps off: flagged-tile lists, no workgroups for empty tiles) must give the same poses
            hp.debug_enable(False)
            if hp.predict_batch(frames.astype(np.uint16), IntrinsicMatrix(K), midp, rot).tobytes() != poses.tobytes():
                bad += 1
                print(f"MISMATCH case {case}: product mode differs from the run with the taps on")
            hp.debug_enable(True)
            if api:                                   # (after the taps: these runs invalidate them)
                masks = hp.predict_mask(frames.astype(np.uint16))
                hough = hp.build_hough_votes(frames.astype(np.uint16), IntrinsicMatrix(K))
                blurred = hp.build_hough_image(frames.astype(np.uint16), IntrinsicMatrix(K)) if model.gaussian_sigma > 0 else None
                p2d = hp.predict_parameter_from2dhough(frames.astype(np.uint16), IntrinsicMatrix(K)) if model.gaussian_sigma > 0 else None
                hp.debug_enable(False)
                rle = hp.predict_batch_rle([biwi.encode_depth(f) for f in frames.astype(np.uint16)], IntrinsicMatrix(K), midp, rot)
                if rle.tobytes() != poses.tobytes():
                    bad += 1
                    print(f"MISMATCH case {case}: dh_predict_batch_rle differs from dh_predict_batch")
    except Exception as e:   # geometry refused (e.g. patch too large): fine as long as it is a clean error
        refused += 1
        print("case", case, "refused:", str(e)[:80])
        continue
    for i in range(n):
        frames_done += 1
        mg_i = None if midp is None else midp[i]
        rg_i = None if rot is None else rot[i]
        try:
            ref = pyoracle.predict(forest, model, frames[i], K, mg_i, rg_i)
        except ValueError:                       # more distinct accumulator cells than the default tap capacity
            ref = pyoracle.predict(forest, model, frames[i], K, mg_i, rg_i, cell_cap=1 << 25)
        ok = (np.array_equal(leaf[i], ref.leaf_idx) and np.array_equal(pg[i], ref.pos_grid) and np.array_equal(rg[i], ref.rot_grid)
              and np.array_equal(poses["mid_point"][i], ref.mid_point) and np.array_equal(poses["rotation"][i], ref.rotation))
        if api and ok:
            ok = np.array_equal(masks[i], pyoracle.predict_mask(forest, model, frames[i])) and np.array_equal(hough[i], pyoracle.hough_image(forest, model, frames[i], K))
            if not ok:
                print("   (mask / hough image differ)")
            if ok and blurred is not None:
                m2, _ = pyoracle.predict_from2dhough(forest, model, frames[i], K)
                ok = np.array_equal(blurred[i], pyoracle.build_hough_image(forest, model, frames[i], K)) and np.array_equal(p2d["mid_point"][i], m2, equal_nan=True)
                if not ok:
                    print("   (blurred hough image / 2-D prediction differ)")
        if not ok:
            bad += 1
            what = [nm for nm, eq in (("leaf", np.array_equal(leaf[i], ref.leaf_idx)), ("pos_grid", np.array_equal(pg[i], ref.pos_grid)),
                                      ("rot_grid", np.array_equal(rg[i], ref.rot_grid)), ("mid", np.array_equal(poses["mid_point"][i], ref.mid_point)),
                                      ("rot", np.array_equal(poses["rotation"][i], ref.rotation))) if not eq]
            print("   differs in:", what, "| leaf diffs:", int((leaf[i] != ref.leaf_idx).sum()), "| gpu pose", poses["mid_point"][i], poses["rotation"][i],
                  "| ref", ref.mid_point, ref.rotation)
            print(f"MISMATCH case {case} frame {i}: {w}x{h} patch {sw}x{sh} step {step} trees {trees} depth {depth} mixed {mixed}")
print(f"{count} cases from {first}: {frames_done} frames compared, {refused} cases refused, {hits_total} hit records in total, {bad} mismatching frames, {time.time() - t0:.0f} s")
