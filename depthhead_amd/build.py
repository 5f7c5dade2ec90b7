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
SOURCES = ["dh_api.hip", "dh_kernels.hip", "dh_biwi.hip"]
HEADERS = ["dh_internal.h", os.path.join("..", "..", "include", "depthhead_hip.h")]

# -ffp-contract=off: no FMA contraction on host or device -- every float expression keeps the
# reference's separate multiply / add rounding (the kernels additionally use explicit *_rn
# intrinsics).  No fast-math: IEEE-correct f32/f64 division.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


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
    cmd = [hipcc(), *FLAGS, *(["-DDH_PROFILING_KNOBS"] if knobs else []), "-o", lib, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, knobs="--knobs" in sys.argv))
