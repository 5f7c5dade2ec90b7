#!/bin/bash
# Per-phase instruction counts of k_traverse: the kernel is cut short after phase N (DH_TRAV_STOP, a
# diagnostic switch) and the SQ counters of the truncated runs are differenced.
# Usage (GPU box, repo root): bash tools/pmc_phases.sh gpurun_out/ph
# Needs the profiling twin of the library (the product library has these switches compiled out):
#   python -m depthhead_amd.build --knobs
set -e
OUT=${1:-gpurun_out/ph}
REPO=$(pwd)
export DH_LIB_PATH=$REPO/depthhead_amd/libdepthhead_hip_knobs.so
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for st in 9 1 3 0; do
  DH_TRAV_STOP=$st timeout -k 10 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$REPO/$OUT/s$st" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > "$REPO/$OUT/s$st.log" 2>&1 || echo "stop $st failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = {}
for st in (9, 1, 3, 0):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/s{st}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_traverse" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[st] = {c: sum(v) / len(v) for c, v in acc.items()}
names = sorted(res[0])
with open(out + "/phases.txt", "w") as fo:
    fo.write("cumulative counters of k_traverse truncated after: 9=entry 1=region build 3=background gate 0=full (walks)\n")
    fo.write("%-24s" % "counter" + "".join("%14s" % f"stop{st}" for st in (9, 1, 3, 0)) + "\n")
    for c in names:
        fo.write("%-24s" % c + "".join("%14.0f" % res[st].get(c, float("nan")) for st in (9, 1, 3, 0)) + "\n")
print(open(out + "/phases.txt").read())
PY
