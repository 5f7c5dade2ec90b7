"""Builds libdepthhead_hip.so (the C-ABI library of include/depthhead_hip.h) in-tree with hipcc
for gfx950.  hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the built
.so travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdepthhead_hip.so")
# profiling-only twin with the kernel-truncating DH_*_STOP / DH_TRAV_STAMPS switches compiled in (tools/pmc_phases.sh loads
# it through DH_LIB_PATH); the product library above never contains them
LIB_KNOBS = os.path.join(HERE, "libdepthhead_hip_knobs.so")
# host runtime, host-only logic (plain C++: also built with g++ under sanitizers by tests/test_host_sanitize.py), one
# translation unit per kernel family
SOURCES = ["dh_api.hip", "dh_host.cpp", "dh_biwi.cpp", "k_forest.hip", "k_prepare.hip", "k_traverse.hip", "k_emit.hip",
           "k_vote.hip", "k_cluster.hip", "k_aux.hip"]
HEADERS = ["dh_internal.h", "dh_host.h", "dh_device.h", os.path.join("..", "..", "include", "depthhead_hip.h")]

# -ffp-contract=off: no FMA contraction on host or device -- every float expression keeps the
# reference's separate multiply / add rounding (the kernels additionally use explicit *_rn
# intrinsics).  No fast-math: IEEE-correct f32/f64 division.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
OBJ_DIR = os.path.join(HERE, "_obj")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, knobs: bool = False) -> str:
    lib = LIB_KNOBS if knobs else LIB
    if not force and not needs_build(lib):
        return lib
    # one object per source, compiled in parallel (objects whose source and headers are older than them are kept), then one link
    from concurrent.futures import ThreadPoolExecutor
    odir = os.path.join(OBJ_DIR, "knobs" if knobs else "product")
    os.makedirs(odir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    cc, defs = hipcc(), (["-DDH_PROFILING_KNOBS"] if knobs else [])
    if os.environ.get("DH_BUILD_DEFS"):            # experiments only, e.g. DH_BUILD_DEFS="-DTRAV_THREADS=512"
        defs += os.environ["DH_BUILD_DEFS"].split()

    def compile_one(src: str) -> str:
        path, obj = os.path.join(CSRC, src), os.path.join(odir, src + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(path)):
            cmd = [cc, *FLAGS, *defs, "-x", "hip", "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, knobs="--knobs" in sys.argv))
