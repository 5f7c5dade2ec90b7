#!/bin/bash
# Vector-memory-path counters of k_traverse (TA = texture addresser, TCP = L1): is the walk phase bound by that path?
# Two passes each for the full kernel and (profiling twin) for the kernel cut after the background gate, so that the walk
# phase is the difference.  Usage (GPU box, repo root): bash tools/pmc_ta.sh > gpurun_out/pmc_ta.txt
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
# (small groups: the *_sum / *_avr metrics expand to one hardware counter per TA / TCP instance, and a pass that asks for
# too many of them does not come back)
G1="TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
G2="TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
G3="TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
i=0
for stop in 0 3; do
  for grp in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    rm -rf /tmp/pmc_ta_$i
    DH_LIB_PATH=$REPO/depthhead_amd/libdepthhead_hip_knobs.so DH_TRAV_STOP=$stop timeout -k 10 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_ta_$i -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > $REPO/gpurun_out/pmc_ta_$i.log 2>&1 || echo "pass $i failed (see gpurun_out/pmc_ta_$i.log)"
    python3 - /tmp/pmc_ta_$i "$stop" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_traverse" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("DH_TRAV_STOP=" + sys.argv[2], {k: round(sum(v) / len(v), 1) for k, v in sorted(acc.items())}, "launches", max((len(v) for v in acc.values()), default=0))
PY
  done
done
