"""Does the order of a forest's nodes in memory matter to k_traverse?  The bench forest (nodes breadth-first per tree) against the
same forest renumbered depth-first (pre-order, as a recursive serialiser writes it) and in random order.  GPU box, repo root."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthhead_amd import synth
from depthhead_amd.forest import Forest
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix


def renumber(f, order):
    """order[i] = old index of the node that becomes node i."""
    new_of_old = np.empty(len(order), dtype=np.int64)
    new_of_old[order] = np.arange(len(order))
    nodes = f.nodes[order].copy()
    for key in ("child_zero", "child_one"):
        c = nodes[key].astype(np.int64)
        nodes[key] = np.where(c >= 0, new_of_old[np.maximum(c, 0)], c).astype(np.int32)
    roots = np.where(f.roots >= 0, new_of_old[np.maximum(f.roots, 0)], f.roots).astype(np.int32)
    return Forest(roots, nodes, f.leaf_prob, f.off_begin, f.rot_begin, f.offsets, f.rotations)


def dfs_order(f):
    out = []
    for r in f.roots:
        st = [int(r)]
        while st:
            n = st.pop()
            if n < 0:
                continue
            out.append(n)
            st.append(int(f.nodes["child_one"][n])); st.append(int(f.nodes["child_zero"][n]))
    return np.array(out)


def blocked_order(f, levels=3):
    """Subtrees of `levels` levels (7 nodes = 112 bytes of walk table, one 128-byte line) stored together, blocks breadth-first."""
    cz, co = f.nodes["child_zero"], f.nodes["child_one"]
    out = []
    for r in f.roots:
        if r < 0:
            continue
        blocks = [int(r)]
        while blocks:
            nxt = []
            for top in blocks:
                level = [top]
                for d in range(levels):
                    out.extend(level)
                    kids = [int(c) for n in level for c in (cz[n], co[n]) if c >= 0]
                    if d == levels - 1:
                        nxt.extend(kids)
                    level = kids
            blocks = nxt
    return np.array(out)


W, H, NF = 640, 480, 256
kind = sys.argv[1] if len(sys.argv) > 1 else "fitted"
base = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2) if kind == "fitted" else synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
dev = torch.device("cuda:0")
frames = torch.from_numpy(np.concatenate([synth.biwi_batch(64, W, H)] * 4).view(np.int16)).to(dev)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
out = torch.zeros(NF * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
rs = np.random.RandomState(1)
variants = {"breadth-first (as built)": base, "depth-first": renumber(base, dfs_order(base)), "random": renumber(base, rs.permutation(len(base.nodes))),
            "blocks of 3 levels": renumber(base, blocked_order(base, 3)), "blocks of 2 levels": renumber(base, blocked_order(base, 2))}
ref = None
for rep in range(2):
    for name, f in variants.items():
        with HoughPrediction(f, model, device=0) as hp:
            hp.reserve(NF, W, H)
            for _ in range(3):
                hp.predict_batch_device(frames.data_ptr(), NF, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
            hp.set_profiling(True)
            acc = 0.0
            for _ in range(10):
                hp.predict_batch_device(frames.data_ptr(), NF, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
                acc += hp.timing()["traverse_ms"] / 10
            st.synchronize()
            poses = out.cpu().numpy().tobytes()
            ref = ref or poses
            print(f"{name:28s} k_traverse {acc:.4f} ms   poses identical: {poses == ref}")
