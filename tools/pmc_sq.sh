#!/bin/bash
# SQ instruction-mix counters for the bench kernels (one rocprofv3 pass per counter group).
# Usage (on the GPU box, from the repo root): bash tools/pmc_sq.sh gpurun_out/sq
set -e
OUT=${1:-gpurun_out/sq}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH"; do
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
        if not k.startswith(("k_", "void k_")): continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
print(open(out + "/summary.txt").read())
PY
