#!/bin/bash
# Per-phase durations of k_emit / k_vote / k_cluster / k_traverse: the profiling twin of the library (built with
# -DDH_PROFILING_KNOBS, `python -m depthhead_amd.build --knobs`) cuts a kernel short after phase N; the HIP-event kernel
# times of the truncated runs are differenced by hand.  Results of truncated runs are INVALID by construction.
# Usage (GPU box, repo root): bash tools/phase_times.sh > gpurun_out/phase_times.txt
export DH_LIB_PATH=$(pwd)/depthhead_amd/libdepthhead_hip_knobs.so
run() { python3 bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith(chr(123)): d=json.loads(l); print('$1', d['kernels_ms'])
"; }
run full
for v in 9 1 3; do DH_TRAV_STOP=$v run "DH_TRAV_STOP=$v"; done
for v in 1 2; do DH_EMIT_STOP=$v run "DH_EMIT_STOP=$v"; done
for v in 1 2 3; do DH_VOTE_STOP=$v run "DH_VOTE_STOP=$v"; done
for v in 1 2 3; do DH_CL_STOP=$v run "DH_CL_STOP=$v"; done
