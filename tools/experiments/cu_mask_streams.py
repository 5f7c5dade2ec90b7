"""Experiment (GPU box): four predictors with four batches in flight -- on ordinary streams (kernels of all batches share every
CU) against streams restricted to a quarter of the CUs each (hipExtStreamCreateWithCUMask): does partitioning the chip beat sharing
it for this pipeline of latency- and gather-bound kernels?  Two bit layouts of the quarter masks are tried (contiguous / interleaved),
since the mapping of mask bits to XCDs is not documented."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

hip = C.CDLL("libamdhip64.so")
NF, W, H = 256, 640, 480
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
b0 = synth.biwi_batch(64, W, H)
b1 = synth.biwi_batch(64, W, H, first=5000)
fr = [torch.from_numpy(np.concatenate([b] * 4).view(np.int16)).to(dev) for b in (b0, b1)]
N = 4


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << (i - 32 * w) for i in bits if 32 * w <= i < 32 * w + 32) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), words)
    assert rc == 0, rc
    return s


def plain_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)) == 0       # hipStreamNonBlocking
    return s


layouts = {
    "shared (no masks)": [None] * N,
    "quarters, contiguous bits": [list(range(64 * k, 64 * k + 64)) for k in range(N)],
    "quarters, bits k mod 4": [[i for i in range(256) if i % 4 == k] for k in range(N)],
    "quarters, XCD pairs (bits i mod 8 in {2k, 2k+1})": [[i for i in range(256) if i % 8 in (2 * k, 2 * k + 1)] for k in range(N)],
}
hps = [HoughPrediction(forest, model) for _ in range(N)]
outs = [torch.zeros(NF * 40, dtype=torch.uint8, device=dev) for _ in range(N)]
for hp in hps:
    hp.reserve(NF, W, H)
ref = None
for name, masks in layouts.items():
    streams = [plain_stream() if m is None else masked_stream(m) for m in masks]

    def run(steps):
        for i in range(steps):
            k = i % N
            hps[k].predict_batch_device(fr[(i // N) % 2].data_ptr(), NF, W, H, intr, outs[k].data_ptr(), stream=streams[k].value)
    run(2 * N); torch.cuda.synchronize()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter(); run(48); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, 48 * NF / dt)
    got = b"".join(o.cpu().numpy().tobytes() for o in outs)
    ref = ref or got
    print(f"{name:52s} {best:9.0f} frames/s   poses identical: {got == ref}")
    for s in streams:
        hip.hipStreamDestroy(s)
