#!/bin/bash
# Regenerates the profiles/<tag>_* artefacts on the GPU box (run from the repo root):
#   bash tools/make_profiles.sh r01_v6
# writes gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.json; copy them to profiles/.
set -e
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python3 "$REPO/bench.py" --steps 20 --warmup 3 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_stats.log" 2>&1
cp "$(ls "$OUT/${TAG}_stats"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats.csv"
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/${TAG}_$c" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_$c.log" 2>&1
  echo "$c done"
done
cd "$REPO"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][f"{c}_KB_avg_per_launch"] = round(sum(v) / len(v), 2)
        res[k][f"launches_{c}"] = len(v)
json.dump(res, open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1)
for k, v in res.items():
    if "k_" in k: print(k, v)
PY
