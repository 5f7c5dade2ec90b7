"""What the per-step all-gather costs the pipelined bench loop, and which part of it (round 3).  One GPU, one rank:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/experiments/dist_overhead.py
Variants of ShardedPredictor.submit over the headline workload (4 predictors, 256 frames each, two alternating batches):
    none      no collective (what `bench.py --gpus 1` times)
    c10d      all_gather_into_tensor(async_op=True) under the predictor's stream, buffer reuse waits for the work (shipped in round 2)
    nowait    the same without the reuse wait (NOT a valid schedule: measurement only)
    side      the collective issued under ONE side stream that waits for an event of the predictor's stream
    events    no collective, but the same event plumbing as `side` (record, wait, record, wait)
    syncop    all_gather_into_tensor(async_op=False) under the predictor's stream (recent c10d runs it ON that stream: no events)
    hostq     c10d, but the buffer-reuse wait is a HOST query of the work (is_completed) -- shipped at the end of round 3
and what the barrier of the timed region's fence costs on an idle GPU.
"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from depthhead_amd import synth
from depthhead_amd.dist import POSE_BYTES
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group(backend="nccl", device_id=dev)
W, H, NF, DEPTH = 640, 480, 256, 4
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
fa = torch.from_numpy(synth.biwi_batch(64, W, H, first=0).repeat(4, axis=0).view(np.int16)).to(dev)
fb = torch.from_numpy(np.roll(synth.biwi_batch(64, W, H, first=10000).repeat(4, axis=0), 7, axis=0).view(np.int16)).to(dev)
batches = [fa, fb]
hps = [HoughPrediction(forest, model, device=0) for _ in range(DEPTH)]
for q in hps:
    q.reserve(NF, W, H)
streams = [torch.cuda.Stream(dev) for _ in range(DEPTH)]
side = torch.cuda.Stream(dev)
nb = NF * POSE_BYTES
pose = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(DEPTH)]
gath = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(DEPTH)]


def run(variant, steps=60):
    pending = [None] * DEPTH
    done = [None] * DEPTH
    count = [0]
    qd = [0, 0]

    def step():
        i = count[0]; count[0] += 1
        k = i % DEPTH
        st = streams[k]
        if variant in ("c10d",) and pending[k] is not None:
            with torch.cuda.stream(st):
                pending[k].wait()
        if variant == "hostq" and pending[k] is not None:
            qd[1] += 1
            if not pending[k].is_completed():
                qd[0] += 1
                with torch.cuda.stream(st):
                    pending[k].wait()
        if variant in ("side", "events") and done[k] is not None:
            st.wait_event(done[k])
        hps[k].predict_batch_device(batches[(i // DEPTH) % 2].data_ptr(), NF, W, H, intr, pose[k].data_ptr(), stream=st.cuda_stream)
        if variant in ("c10d", "nowait", "hostq"):
            with torch.cuda.stream(st):
                pending[k] = dist.all_gather_into_tensor(gath[k], pose[k], async_op=True)
        elif variant == "syncop":
            with torch.cuda.stream(st):
                dist.all_gather_into_tensor(gath[k], pose[k])
        elif variant in ("side", "events"):
            ev = torch.cuda.Event(); ev.record(st)
            side.wait_event(ev)
            if variant == "side":
                with torch.cuda.stream(side):
                    pending[k] = dist.all_gather_into_tensor(gath[k], pose[k], async_op=True)
                    pending[k].wait()           # (stream-side: `side` waits for RCCL's stream)
            d = torch.cuda.Event(); d.record(side); done[k] = d

    for _ in range(3 * DEPTH):
        step()
    torch.cuda.synchronize(dev)
    res = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ts = time.perf_counter()
        torch.cuda.synchronize(dev)
        e = time.perf_counter() - t0
        res.append((NF * steps / e, (ts - t0) / steps * 1e3))
    print("%-8s %s%s" % (variant, "  ".join("%7.0f f/s (host %.3f ms/step)" % r for r in res), "  [not completed at the query: %d of %d]" % tuple(qd) if qd[1] else ""), flush=True)


for v in (sys.argv[1:] or ["none", "c10d", "syncop", "hostq", "nowait", "none"]):
    run(v)
torch.cuda.synchronize(dev)
for _ in range(3):
    t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(dev); t1 = time.perf_counter()
    print("dist.barrier() + synchronize on an idle GPU: %.3f ms" % ((t1 - t0) * 1e3), flush=True)
dist.destroy_process_group()
