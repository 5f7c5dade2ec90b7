#!/bin/bash
# Is the gap between `bench.py` and `bench.py --force-dist` (one rank through RCCL) a cost per step or per timed region?  GPU box, repo root.
for k in 20 60 200; do
  python bench.py --no-extras --no-cpu-baseline --steps $k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain steps=$k', d['value'], d['ms_per_step'])"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --no-extras --no-cpu-baseline --steps $k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dist  steps=$k', d['value'], d['ms_per_step'])"
done
