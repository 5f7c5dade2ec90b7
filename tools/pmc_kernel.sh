#!/bin/bash
# Memory-side counters per kernel of the bench (one rocprofv3 pass per counter group).
# Usage (GPU box, repo root): bash tools/pmc_kernel.sh gpurun_out/mem
set -e
OUT=${1:-gpurun_out/mem}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$REPO/$OUT/p$i" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > "$REPO/$OUT/p$i.log" 2>&1 || echo "group $i failed: $grp"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(acc.items()):
        if "k_" not in k or "prepare" in k or "compact" in k: continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:40s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
print(open(out + "/summary.txt").read())
PY
