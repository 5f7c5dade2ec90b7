#!/usr/bin/env python3
"""bench.py -- depth frames/s of the Hough-forest head-pose path on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole path (frame batch already resident in HBM -> 40-byte pose per
frame, device-resident) over this rank's batch of synthetic 640x480 BIWI-shaped frames.  The
workload is BASELINE.json configs[1] per GPU: 256 frames, 10-tree depth-15 forest, stride-4
patches.  Frames shard across ranks (every rank holds its own 256 frames and a forest replica:
weak scaling); the only collective is the per-step RCCL all-gather of the pose records.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
`roofline` (dominant kernel vs the HBM roof, duration measured live with HIP events on the launch
stream) and `cpu_baseline` (the CPU oracle, frame-parallel over the host cores, on a bounded
sample; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

KERNEL_OF = {"boxsum_ms": "k_boxsum", "traverse_ms": "k_traverse", "emit_ms": "k_emit", "vote_ms": "k_vote", "cluster_ms": "k_cluster"}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SCLK_HZ = 2.4e9         # max shader clock (MI355X_MICROARCH.md); the chip may run lower under load, so *_frac are lower bounds
N_CUS, N_SIMDS = 256, 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of 100 steps (36 ms at the headline workload) -- the region starts on an idle GPU and ends with the
    # last batch's tail kernels running alone, which costs 20 steps 3 % and 100 steps 0.6 % (708 k / 721 k / 728 k frames/s at
    # 20 / 60 / 200 steps on one box, tools/experiments/dist_steps.sh)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None,
                    help="frames per GPU per step; default 256 (BASELINE configs[1]), and 512 with --gpus 8 (configs[3]: 4 096 frames over 8 GPUs)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--trees", type=int, default=10)
    ap.add_argument("--depth", type=int, default=15)
    ap.add_argument("--stride", type=int, default=4)
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames generated per rank (tiled to --frames)")
    ap.add_argument("--forest", choices=["fitted", "random"], default="fitted",
                    help="fitted: trainer-sized forest fitted to synthetic subjects (coherent votes, the stand-in for the "
                         "unavailable pretrained forest); random: 13x larger random-split stress forest")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (the real multi-GPU path); gloo = rehearsal of the N>1 control flow "
                         "with every rank on one device (poses staged through host memory)")
    ap.add_argument("--graph", action="store_true", help="replay the batch from a captured hipGraph (launch-bound configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra legs (other configs, PCIe-inclusive rates): profiling runs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline wall time")
    ap.add_argument("--pipeline", type=int, default=4, help="batches in flight per GPU (predictors / streams that consecutive steps alternate between)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the per-step all-gather even with ONE rank (launch through "
                         "torch.distributed.run --nproc-per-node 1): executes the RCCL path on a one-GPU box")
    ap.add_argument("--dump-poses", default=None, help="rank 0 writes the gathered pose records of the last timed step (all ranks, rank order) as .npy")
    return ap.parse_args()


def main():
    args = parse()
    if args.frames is None:
        args.frames = 512 if args.gpus == 8 else 256
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(1)
    if args.backend == "gloo":
        local_rank = 0                                            # rehearsal: every rank shares device 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    from depthhead_amd import synth
    from depthhead_amd._lib import POSE_DTYPE
    from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

    W, H, NF = args.width, args.height, args.frames
    if args.forest == "fitted":
        forest = synth.fit_forest(args.trees, args.depth, synth.FOREST_SEED_BASE + 2)
    else:
        forest = synth.synth_forest(args.trees, args.depth, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=args.stride)
    K = synth.default_intrinsic(W, H)
    intr = IntrinsicMatrix(K)
    nd = min(args.distinct, NF)
    distinct = synth.biwi_batch(nd, W, H, first=rank * nd)               # different frames on every rank
    frames_np = np.concatenate([distinct] * ((NF + nd - 1) // nd))[:NF]
    frames = torch.from_numpy(frames_np.view(np.int16)).to(dev)          # resident in HBM before timing
    # a second, different batch (other subjects, other head positions): consecutive batches of a predictor alternate between
    # the two, so that nothing that depends on the previous batch's content (k_boxsum does not rewrite zero over zero) sees the
    # same frames twice in a row
    other_np = np.roll(np.concatenate([synth.biwi_batch(nd, W, H, first=10000 + rank * nd)] * ((NF + nd - 1) // nd))[:NF], 7, axis=0)
    frames_b = torch.from_numpy(other_np.view(np.int16)).to(dev)
    batches = [frames, frames_b]
    step_no = [0]
    hp = HoughPrediction(forest, model, device=local_rank)
    hp.reserve(NF, W, H)
    # pipeline depth: consecutive steps alternate between this many predictors (own workspace, own stream), so that the
    # latency-bound tail kernels of one batch run beside the head kernels of the next; every step is still one whole batch
    depth = 1 if args.graph else max(1, args.pipeline)
    hps = [hp] + [HoughPrediction(forest, model, device=local_rank) for _ in range(depth - 1)]
    for q in hps[1:]:
        q.reserve(NF, W, H)
    stream = torch.cuda.current_stream(dev)
    # the shard -> predict -> gather loop is the package's (depthhead_amd.dist.ShardedPredictor): two pose buffers, the
    # RCCL all-gather of step i overlaps the kernels of step i + 1
    from depthhead_amd.dist import ShardedPredictor
    sp = ShardedPredictor(hps, NF, W, H, intr, device=dev, force_collective=args.force_dist)
    poses = sp.pose_bufs[0]
    if args.graph:
        sp.capture(frames.data_ptr())

    def step():
        i = step_no[0]
        step_no[0] += 1
        # (the frames are resident and nothing is pending on `stream`: deeper pipelines need not wait for it)
        sp.submit((frames if args.graph else batches[(i // depth) % 2]).data_ptr(), stream if depth == 1 else None)

    fence = sp.fence
    gdev = dev if args.backend == "nccl" else torch.device("cpu")

    # Priming (initialisation, before the W warm-up steps): every predictor of the pipeline runs each of the two batches once,
    # so that its first-launch work (lazy module loads, workspace first touch) and the first fill of its box-sum image are
    # not charged to whichever of the W + K steps happens to be its first.  Reported as config.priming_steps.
    priming = 0 if args.graph else 2 * depth
    for _ in range(priming):
        step()
    fence()
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_submitted = time.perf_counter()     # (host side of the K steps: all launches enqueued; the GPU is still working)
    fence()
    elapsed = time.perf_counter() - t0
    host_submit_ms = (t_submitted - t0) / args.steps * 1e3
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gpu_poses = sp.last_poses() if rank == 0 else None     # gathered records of the last timed step (all ranks, rank order)
    last_was_b = (not args.graph) and (((step_no[0] - 1) // depth) % 2 == 1)   # which of the two batches that step processed
    if args.dump_poses and rank == 0:
        np.save(args.dump_poses, gpu_poses)
    # run-to-run spread: the same K-step region twice more (untimed for `value`, which stays the first region)
    repeats = [elapsed / args.steps * 1e3]
    if not args.no_extras:
        for _ in range(2):
            fence()
            tr = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            e = time.perf_counter() - tr
            if dist is not None:
                t = torch.tensor([e], dtype=torch.float64, device=gdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e = float(t.item())
            repeats.append(e / args.steps * 1e3)

    # ---- per-kernel durations with HIP events on the launch stream (outside the timed region)
    hp.set_profiling(True)
    acc = {"boxsum_ms": 0.0, "traverse_ms": 0.0, "emit_ms": 0.0, "vote_ms": 0.0, "cluster_ms": 0.0, "total_ms": 0.0}
    reps = max(3, min(10, args.steps))
    for r in range(reps):
        hp.predict_batch_device(batches[r % 2].data_ptr(), NF, W, H, intr, poses.data_ptr(), stream=stream.cuda_stream)   # alternating, as in the timed region
        tm = hp.timing()
        for k in acc:
            acc[k] += tm[k] / reps
    hp.set_profiling(False)
    torch.cuda.synchronize(dev)
    # the same events with `depth` batches in flight (one launch per predictor, started together, a few rounds): kernels of
    # consecutive batches share the chip, so every launch takes longer while several run at once
    acc_fl = None
    if depth > 1:
        acc_fl = {k: 0.0 for k in acc}
        rounds = 4
        for q in hps:
            q.set_profiling(True)
        for r in range(rounds):
            for k, q in enumerate(hps):
                q.predict_batch_device(batches[r % 2].data_ptr(), NF, W, H, intr, sp.pose_bufs[k % sp.slots].data_ptr(), stream=sp.streams[k].cuda_stream)
            torch.cuda.synchronize(dev)
            for q in hps:
                tm = q.timing()
                for k in acc_fl:
                    acc_fl[k] += tm[k] / (rounds * depth)
        for q in hps:
            q.set_profiling(False)

    # ---- PCIe-inclusive rate: the same batch through the host-buffer entry point (H2D copy of the
    # frames, compute, D2H of the poses).  Reported next to `value`, never as `value`.
    pcie_fps = pcie_pageable_fps = h2d_ceiling_fps = rle_fps = rle_ratio = None
    if world == 1 and not args.no_extras:
        # host frames in page-locked memory (dh_host_alloc: where a capture / reader thread would put them): chunked
        # asynchronous upload overlapping the kernels
        from depthhead_amd._lib import pinned_empty
        pinned = pinned_empty(frames_np.shape, np.uint16)
        pinned[...] = frames_np

        def host_rate(fn, reps=5):
            fn()
            best = 1e9
            for _ in range(reps):
                t1 = time.perf_counter()
                fn()
                best = min(best, time.perf_counter() - t1)
            return NF / best

        pcie_fps = host_rate(lambda: hp.predict_batch(pinned, intr))
        pcie_pageable_fps = host_rate(lambda: hp.predict_batch(frames_np, intr))
        # the ceiling: the bare host-to-device copy of the same bytes from page-locked memory, nothing else running
        tp = torch.from_numpy(pinned.view(np.int16))
        td = torch.empty_like(tp, device=dev)
        h2d_ceiling_fps = host_rate(lambda: (td.copy_(tp, non_blocking=True), torch.cuda.synchronize(dev)))
        # the same frames as BIWI run-length coded payloads (the database's own `.bin` format, biwi.rs:81-103): payloads
        # and run table cross PCIe, the frames are rebuilt on the device (dh_predict_batch_rle)
        from depthhead_amd import biwi
        enc = [biwi.encode_depth(f) for f in distinct]
        payloads = [enc[i % nd] for i in range(NF)]
        rle_ratio = sum(len(b) for b in payloads) / frames_np.nbytes
        rp = hp.predict_batch_rle(payloads, intr)
        assert rp.tobytes() == hp.predict_batch(frames_np, intr).tobytes()
        rle_fps = host_rate(lambda: hp.predict_batch_rle(payloads, intr))

    also = also_l2 = single_frame_us = None
    if world == 1 and not args.no_extras and not custom_workload():
        for q in hps:
            q.close()
        hp = None
        also, also_l2, single_frame_us = other_configs(args, dev, stream)

    if rank == 0:
        total_frames = world * NF * args.steps
        fps = total_frames / elapsed
        kernels = {k: round(v, 4) for k, v in acc.items()}
        dom = max(("boxsum_ms", "traverse_ms", "emit_ms", "vote_ms", "cluster_ms"), key=lambda k: acc[k])
        # algorithmic bytes per launch: every depth pixel read once, one pose
        # record written per frame, the forest read once per launch
        b_alg = NF * (W * H * 2 + 36) + forest.nbytes()
        prof = Profiles()
        traffic = prof.traffic(KERNEL_OF[dom])
        achieved = b_alg / (acc[dom] * 1e-3) / 1e9 if acc[dom] > 0 else 0.0
        roof = {"bound": "hbm", "kernel": KERNEL_OF[dom],
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "traffic_source": prof.source if traffic is not None else None, "profile_head": prof.head,
                "profile_matches_build": prof.matches, "kernel_source_sha256_16": prof.build_hash,
                "algorithmic_bytes_per_launch": b_alg, "launch_ms": round(acc[dom], 4),
                "launch_ms_is": "the kernel's duration with ONE batch in flight (HIP events around every kernel of a dedicated pass, "
                                "as in profiles/*_kernel_stats.csv = rocprofv3 --stats of `bench.py --pipeline 1`); in the timed region "
                                f"{depth} batches are in flight and kernels of consecutive batches share the chip "
                                "(profiles/*_kernel_stats_inflight.csv)",
                # the same algorithmic bytes over the whole step (all five kernels): what the job as a whole reaches
                "whole_step_frac": round(b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                "note": "achieved = algorithmic bytes of the launch / duration of the dominant kernel (live HIP events). "
                        "That kernel is not HBM-bound (DESIGN.md section 5): its own measured traffic over its own time is in "
                        "kernels_hbm, its vector-issue load in valu_issue_frac; k_boxsum is the kernel at the HBM roof"}
        sq = prof.sq(KERNEL_OF[dom])
        if sq and acc[dom] > 0:
            cycles = acc[dom] * 1e-3 * SCLK_HZ
            # second roof of the dominant kernel: a wave64 VALU instruction holds its SIMD's vector issue for 4 cycles
            roof["valu_issue_frac"] = round(sq.get("SQ_INSTS_VALU", 0.0) * 4 / (N_SIMDS * cycles), 4)
            if sq.get("SQ_LDS_IDX_ACTIVE"):
                roof["lds_busy_frac"] = round(sq["SQ_LDS_IDX_ACTIVE"] / (N_CUS * cycles), 4)
                roof["lds_bank_conflict_share"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"], 4)
            if sq.get("SQ_INSTS_VMEM_RD"):
                # 16-byte-per-lane gathers occupy the CU's vector-memory path ~16 cycles each (profiles/r02_ubench_gather.txt)
                roof["vmem_issue_frac"] = round(sq["SQ_INSTS_VMEM_RD"] * 16 / (N_CUS * cycles), 4)
        out = {
            "metric": "depth frames/sec (640x480, 10-tree forest)",
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/f64",
            "data": f"synthetic (two different {NF}-frame batches per GPU, alternating)",
            "config": {"workload": f"{workload_name(args, world, NF)}: {NF} synthetic {W}x{H} u16 depth frames per GPU, "
                                   f"{args.trees}-tree depth-{args.depth} {args.forest} synthetic forest "
                                   f"({forest.n_nodes} nodes, {forest.n_leaves} leaves), stride-{args.stride} "
                                   f"80x80 patches, 20 mean-shift iterations",
                       "frames_per_gpu": NF, "width": W, "height": H, "trees": args.trees, "max_depth": args.depth,
                       "stride": args.stride, "parallelism": f"frame-sharded x{world}, RCCL all-gather of poses",
                       "batches_in_flight": depth, "priming_steps": priming,
                       "collective": ("RCCL all_gather_into_tensor of the pose records every step" if sp.collective and args.backend == "nccl"
                                      else "gloo all-gather through host memory (rehearsal)" if sp.collective else "none (one rank, no process group)")},
            "roofline": roof,
            "host_submit_ms_per_step": round(host_submit_ms, 4),
            "repeat_ms_per_step": [round(r, 4) for r in repeats],
            "last_step_batch": "b" if last_was_b else "a",      # which of the two alternating batches the last timed step processed (--dump-poses)
            "kernels_ms": kernels,
            "kernels_ms_in_flight": None if acc_fl is None else {k: round(v, 4) for k, v in acc_fl.items()},   # per launch, `batches_in_flight` at once
            # measured HBM traffic (PMC, gfx950-corrected) over the live duration of every kernel: shows which
            # kernel actually runs at memory speed (k_boxsum) and which are latency / issue bound
            "kernels_hbm": {KERNEL_OF[k]: {"traffic": t, "GB/s": round(t / (acc[k] * 1e-3) / 1e9, 1), "frac": round(t / (acc[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                            for k in KERNEL_OF for t in [prof.traffic(KERNEL_OF[k])] if t and acc[k] > 0},
            # whole job through the host entry points (frames in host memory -> poses in host memory), best of 5 calls:
            "pcie_inclusive_frames_per_s": None if pcie_fps is None else round(pcie_fps, 1),                 # page-locked host frames
            "pcie_inclusive_pageable_frames_per_s": None if pcie_pageable_fps is None else round(pcie_pageable_fps, 1),
            "h2d_copy_ceiling_frames_per_s": None if h2d_ceiling_fps is None else round(h2d_ceiling_fps, 1),  # the bare copy of the frames
            "rle_inclusive_frames_per_s": None if rle_fps is None else round(rle_fps, 1),
            "rle_payload_ratio": None if rle_ratio is None else round(rle_ratio, 4),
        }
        if also is not None:
            out["also"] = also
            out["also_pose_l2_max"] = also_l2          # every leg's last poses vs the oracle on a sample of its frames: all 0.0
            out["single_frame_us"] = single_frame_us
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_poses = cpu_baseline(forest, model, frames_np, K, args.cpu_seconds)
            if last_was_b:
                from oracle import pyoracle as po
                ref_poses = po.predict_batch(forest, model, other_np, K, rect_mode=po.RECT_SAT, threads=host_cores())
            # the second half of BASELINE.json's metric ("pose L2 vs ref"): the GPU poses of the last timed step against
            # the oracle's for the very same frames (both are integer-grid values: the difference must be exactly 0)
            dm = np.linalg.norm(gpu_poses["mid_point"].astype(np.float64) - ref_poses["mid_point"].astype(np.float64), axis=1)
            dr = np.linalg.norm(gpu_poses["rotation"] - ref_poses["rotation"], axis=1)
            out["pose_l2_max_vs_oracle"] = {"mid_point_mm": float(dm.max()), "rotation_rad": float(dr.max()), "frames": int(dm.size),
                                            "tolerance": 1e-4}
        print(json.dumps(out), flush=True)

    if hp is not None:
        for q in hps:
            q.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def workload_name(args, world: int, nf: int) -> str:
    """Which BASELINE.json config the job as a whole is."""
    if (args.width, args.height, args.trees, args.depth, args.stride) == (640, 480, 10, 15, 4):
        if world == 8 and nf == 512:
            return "BASELINE configs[3] (4 096-frame stream frame-sharded over 8 GPUs, 512 per rank; per GPU the configs[1] geometry)"
        if nf == 256:
            return "BASELINE configs[1]"
    return "custom workload"


WORKLOAD_FLAGS = ("--frames", "--width", "--height", "--trees", "--depth", "--stride", "--forest", "--graph")


def custom_workload() -> bool:
    return any(a in sys.argv for a in WORKLOAD_FLAGS)


def kernel_source_hash() -> str:
    """Identifies the kernels a profile was taken on: sha256 over the kernel / runtime sources."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "depthhead_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            h.update(open(os.path.join(csrc, fn), "rb").read())
    return h.hexdigest()[:16]


class Profiles:
    """The committed rocprofv3 counter summaries under profiles/ (written by tools/profile_round.sh from separate
    `--pmc` passes of this same command, default workload).  A summary is only used when the sources it records
    (`_meta.kernel_source_sha256_16`) are the sources of this build: otherwise `traffic` is null, never stale.
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are
    doubled; both counters are in KB."""

    def __init__(self):
        import glob
        self.build_hash = kernel_source_hash()
        self.mem, self.sqd, self.source, self.head, self.matches = None, None, None, None, False
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), key=os.path.getmtime)
        for f in reversed(files):
            try:
                d = json.load(open(f))
            except (OSError, ValueError):
                continue
            meta = d.get("_meta", {})
            if self.source is None:
                self.source, self.head = os.path.basename(f), meta.get("git_head")
            if meta.get("kernel_source_sha256_16") == self.build_hash:
                self.source, self.head, self.matches, self.mem = os.path.basename(f), meta.get("git_head"), True, d
                try:
                    self.sqd = json.load(open(f.replace("_pmc_summary.json", "_sq_summary.json")))
                except (OSError, ValueError):
                    self.sqd = None
                break
        if custom_workload():
            self.mem = self.sqd = None

    def traffic(self, kernel: str):
        if not self.mem:
            return None
        try:
            e = next(v for k, v in self.mem.items() if kernel in k)
            return int((2 * e["FETCH_SIZE_KB_avg_per_launch"] + e["WRITE_SIZE_KB_avg_per_launch"]) * 1024)
        except (StopIteration, KeyError, ValueError):
            return None

    def sq(self, kernel: str):
        if not self.sqd:
            return None
        return next((v for k, v in self.sqd.items() if kernel in k), None)


def rate(hp, frames_t, n, w, h, intr, stream, steps=5, warmup=2, graph=False, latency=False):
    """frames/s of `steps` device-resident batches (host wall clock around synchronised launches) and the pose records of the
    last one.  latency=True: the host waits for every batch before it submits the next (one frame at a time, as the reference's
    only caller does: examples/live_prediction.rs:76-86); returns microseconds per batch instead."""
    out = torch.zeros(n * 40, dtype=torch.uint8, device=frames_t.device)
    if graph:
        hp.graph_capture(frames_t.data_ptr(), n, w, h, intr, out.data_ptr())
    run = (lambda: hp.graph_launch(stream.cuda_stream)) if graph else \
          (lambda: hp.predict_batch_device(frames_t.data_ptr(), n, w, h, intr, out.data_ptr(), stream=stream.cuda_stream))
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
        if latency:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from depthhead_amd._lib import POSE_DTYPE
    poses = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE).copy()
    return (round(dt / steps * 1e6, 1) if latency else round(steps * n / dt, 1)), poses


def pose_l2(poses, ref) -> float:
    """Largest L2 distance of mid_point (mm) / rotation (rad) between two pose arrays (0.0 = identical)."""
    dm = np.linalg.norm(poses["mid_point"].astype(np.float64) - ref["mid_point"].astype(np.float64), axis=1)
    dr = np.linalg.norm(poses["rotation"] - ref["rotation"], axis=1)
    return float(max(dm.max(), dr.max()))


def other_configs(args, dev, stream):
    """The other BASELINE configs and variants on this one GPU, a few untimed-for-`value` steps each, so that the
    driver's line carries them (frames/s).  Same generators and seeds as the parity tests.  Every leg's last poses are compared
    with the oracle's on a small sample of its frames (`also_pose_l2_max`: must be 0.0)."""
    from depthhead_amd import synth
    from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
    from oracle import pyoracle as po
    res, l2 = {}, {}
    W, H = 640, 480
    K = synth.default_intrinsic(W, H)
    intr = IntrinsicMatrix(K)
    base = synth.biwi_batch(64, W, H)
    fr256 = torch.from_numpy(np.concatenate([base] * 4).view(np.int16)).to(dev)
    fitted = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
    rnd = synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 2)
    cores = host_cores()

    def env_set(env):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        return old

    def env_restore(old):
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def one(name, forest, stride, frames_t, frames_np, n, w, h, K_, env=None, graph=False, steps=5, check=4, latency=False):
        old = env_set(env)
        try:
            with HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=dev.index or 0) as hp:
                hp.reserve(n, w, h)
                v, poses = rate(hp, frames_t, n, w, h, IntrinsicMatrix(K_), stream, steps=steps, graph=graph, latency=latency)
        finally:
            env_restore(old)
        res[name] = v
        if check:
            k = min(check, n)
            l2[name] = pose_l2(poses[:k], po.predict_batch(forest, synth.ModelParams(stepwidth=stride), frames_np[:k], K_, rect_mode=po.RECT_SAT, threads=cores))

    def in_flight(name, forest, stride, frames_t, n, w, h, depth=4, rounds=4):
        """the way the headline is run: `depth` predictors on their own streams, batches in flight"""
        hps, streams = [], [torch.cuda.Stream(dev) for _ in range(depth)]
        try:
            for _ in range(depth):
                hps.append(HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=dev.index or 0))
                hps[-1].reserve(n, w, h)
            outs = [torch.zeros(n * 40, dtype=torch.uint8, device=dev) for _ in range(depth)]

            def sweep(k):
                for i in range(k):
                    hps[i % depth].predict_batch_device(frames_t.data_ptr(), n, w, h, intr, outs[i % depth].data_ptr(), stream=streams[i % depth].cuda_stream)
                torch.cuda.synchronize()
            sweep(2 * depth)
            t0 = time.perf_counter()
            sweep(rounds * depth)
            res[name] = round(rounds * depth * n / (time.perf_counter() - t0), 1)
        finally:
            for q in hps:
                q.close()

    base256 = np.concatenate([base] * 4)
    one("synth_forest_10_15", rnd, 4, fr256, base256, 256, W, H, K)                 # BASELINE.md section 4's random stress forest
    in_flight("synth_forest_10_15_4_in_flight", rnd, 4, fr256, 256, W, H)
    one("c1_stride10", fitted, 10, fr256, base256, 256, W, H, K)                    # configs[0] geometry (the trainer's step width)
    one("general_path", fitted, 4, fr256, base256, 256, W, H, K, env={"DH_FORCE_GENERAL": "1"})   # mixed-rectangle traversal, forced
    small_np = synth.biwi_batch(64, 320, 240)
    small = torch.from_numpy(small_np.view(np.int16)).to(dev)
    Ks = synth.default_intrinsic(320, 240)
    one("c5_320x240_s1_graph", fitted, 1, small, small_np, 64, 320, 240, Ks, graph=True)   # configs[4] on one GPU, hipGraph replay
    one("c5_320x240_s1_single_frame_graph", fitted, 1, small, small_np, 1, 320, 240, Ks, graph=True, steps=50, check=1)
    one("c5_320x240_s1_graph_synth_forest", rnd, 1, small, small_np, 64, 320, 240, Ks, graph=True, check=2)   # ... with BASELINE.md section 4's random stress forest
    c3 = synth.synth_forest(50, 20, synth.FOREST_SEED_BASE + 3)
    one("c3_50x20_s2_32f", c3, 2, fr256, base256, 32, W, H, K, steps=3, check=2)    # configs[2]: 50 trees, depth 20, stride 2, batch 32
    in_flight("c3_50x20_s2_32f_4_in_flight", c3, 2, fr256, 32, W, H)
    res["unit"] = "frames/s"
    # latency of ONE frame, the host waiting for each pose before the next frame (live_prediction.rs:76-86): microseconds per frame
    lat = {}
    res_keep = dict(res)
    one("lat_320", fitted, 1, small, small_np, 1, 320, 240, Ks, graph=True, steps=200, check=0, latency=True)
    one("lat_640", fitted, 4, fr256, base256, 1, W, H, K, graph=True, steps=200, check=0, latency=True)
    lat = {"320x240_stride1": res.pop("lat_320"), "640x480_stride4": res.pop("lat_640"), "unit": "us per frame, hipGraph replay, host sync after every frame"}
    assert set(res) == set(res_keep)
    return res, l2, lat


def host_cores() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(forest, model, frames_np, K, target_s):
    """The CPU oracle (a C restatement of the reference, NOT the Rust binary: no Rust toolchain
    exists here) frame-parallel over the host cores, on a bounded sample of the same frames: the
    benchmark's batch, repeated until about `target_s` seconds of CPU work have been spent."""
    from oracle import pyoracle as po
    cores = host_cores()
    probe = frames_np[: min(cores, frames_np.shape[0])]
    t0 = time.perf_counter()
    po.predict_batch(forest, model, probe, K, rect_mode=po.RECT_FAITHFUL, threads=cores)
    tp = max(time.perf_counter() - t0, 1e-3)
    per_batch = tp / probe.shape[0] * frames_np.shape[0]
    reps = int(max(1, min(16, round(target_s * 0.8 / per_batch))))
    n = frames_np.shape[0] * reps
    t0 = time.perf_counter()
    for _ in range(reps):
        po.predict_batch(forest, model, frames_np, K, rect_mode=po.RECT_FAITHFUL, threads=cores)
    tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = po.predict_batch(forest, model, frames_np, K, rect_mode=po.RECT_SAT, threads=cores)
    ts = time.perf_counter() - t0
    return {"value": round(n / tf, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"the benchmark's {frames_np.shape[0]} frames x {reps} passes, frame-parallel OpenMP over {cores} threads; "
                      f"value = faithful O(area) rectangle loops as src/types.rs:317-339; sat_value = same results with a "
                      f"summed-area table (the fair CPU ceiling)",
            "sat_value": round(n / ts, 2), "faithful_s": round(tf, 2), "sat_s": round(ts, 2)}, ref


if __name__ == "__main__":
    main()
