#!/bin/bash
# Regenerates the profiles/<tag>_* artefacts on the GPU box (run from the repo root):
#   bash tools/profile_round.sh r02_v1 [extra bench.py args]
# writes into gpurun_out/: <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.json (FETCH_SIZE /
# WRITE_SIZE per kernel, separate passes) and <tag>_sq_summary.json (SQ instruction / cycle / LDS counters per
# kernel, two passes).  Copy the ones to be judged into profiles/.  The program follows `--` directly and
# --pmc is never combined with system tracing.
set -e
TAG=${1:-rXX}
shift || true
EXTRA="$@"
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# Per-kernel durations: ONE batch in flight (--pipeline 1), so that a launch's duration is that kernel's own time -- the figure
# bench.py's roofline uses (its HIP-event pass also runs one predictor alone).  With the default four batches in flight
# kernels of consecutive batches share the chip and every launch takes longer while two run at once; that trace is kept
# beside it (*_kernel_stats_inflight.csv) for the record.
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 $EXTRA > "$OUT/${TAG}_stats.log" 2>&1
cp "$(ls "$OUT/${TAG}_stats"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats2" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras $EXTRA > "$OUT/${TAG}_stats2.log" 2>&1
cp "$(ls "$OUT/${TAG}_stats2"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats_inflight.csv"
echo "stats done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc$i" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 $EXTRA > "$OUT/${TAG}_pmc$i.log" 2>&1 || echo "pmc group $i failed: $grp"
  echo "pmc group $i done"
done
cd "$REPO"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, hashlib, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(out)
sys.path.insert(0, root)
from bench import kernel_source_hash as kernel_hash      # the same hash bench.py checks a profile against
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{tag}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
mem, sq = {}, {}
for k, d in acc.items():
    if "k_" not in k:
        continue
    for c, v in d.items():
        tgt = mem if c in ("FETCH_SIZE", "WRITE_SIZE") else sq
        e = tgt.setdefault(k, {})
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e[f"{c}_KB_avg_per_launch"] = round(sum(v) / len(v), 2)
            e[f"launches_{c}"] = len(v)
        else:
            e[c] = round(sum(v) / len(v), 1)
            e["launches"] = len(v)
head_file = os.path.join(root, ".git_head")      # written by tools/gpu.sh before the snapshot leaves (the box has no .git)
meta = {"tag": tag, "kernel_source_sha256_16": kernel_hash(),
        "git_head": open(head_file).read().strip() if os.path.exists(head_file) else None,
        "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1",
        "note": "averages per launch over the bench's launches; FETCH_SIZE is in KB and under-reports wide reads by 2x on gfx950 "
                "(MI355X_MICROARCH.md, HBM); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves"}
mem["_meta"] = meta
sq["_meta"] = meta
json.dump(mem, open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
json.dump(sq, open(f"{out}/{tag}_sq_summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(sq.items()):
    if k != "_meta": print(k, v)
PY
# the bench line once more, now that the counters of THIS build exist: bench.py prints roofline.traffic only from a summary whose
# source hash matches the sources it runs (the copy in profiles/ here is the box's scratch copy; commit the one merged into gpurun_out/)
cp "$OUT/${TAG}_pmc_summary.json" "$OUT/${TAG}_sq_summary.json" "$REPO/profiles/"
cd /tmp
python3 "$REPO/bench.py" --steps 20 --warmup 3 $EXTRA > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench (with traffic) done"
