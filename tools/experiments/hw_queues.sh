#!/bin/bash
# Does the number of hardware queues (ROCclr maps HIP streams onto GPU_MAX_HW_QUEUES of them, default 4) limit the pipelined bench?
# Every predictor owns 1 + 7 non-blocking streams of its own besides the torch stream its batches run on.  GPU box, repo root.
for q in "" 2 4 8 16; do for pl in 4 6 8; do
  echo -n "GPU_MAX_HW_QUEUES=${q:-default} pipeline=$pl  "
  env ${q:+GPU_MAX_HW_QUEUES=$q} timeout -k 10 200 python bench.py --pipeline $pl --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
