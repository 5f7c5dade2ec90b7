#!/bin/bash
# How much of a kernel's solo time is EXPOSED in the pipelined throughput (four batches in flight)?  The profiling twin cuts a
# kernel short (results invalid); the change of ms_per_step against the full run is what that kernel costs the pipeline.
# Only cuts whose downstream work stays the same are meaningful: k_cluster after its guess (last kernel), and the traverse /
# whole-pipeline cuts as bounds.  Usage (GPU box, repo root): bash tools/exposed_cost.sh
export DH_LIB_PATH=$(pwd)/depthhead_amd/libdepthhead_hip_knobs.so
run() { python3 bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith(chr(123)): d=json.loads(l); print('$1', 'ms_per_step', d['ms_per_step'], d['repeat_ms_per_step'], 'solo', d['kernels_ms'])
"; }
run full
DH_CL_STOP=1 run "DH_CL_STOP=1(guess only)"
DH_CL_STOP=2 run "DH_CL_STOP=2(+first region)"
DH_VOTE_STOP=1 run "DH_VOTE_STOP=1(no votes: cluster changes too)"
DH_EMIT_STOP=2 run "DH_EMIT_STOP=2(no records: vote+cluster trivial)"
DH_TRAV_STOP=3 run "DH_TRAV_STOP=3(no walks: all after trivial)"
DH_TRAV_STOP=9 run "DH_TRAV_STOP=9(boxsum only)"
