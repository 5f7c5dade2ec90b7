#!/bin/bash
# 512 (and 1024) frames per call: forked sub-batches (DH_CHUNKS=2, the automatic choice so far) against one kernel sequence (DH_CHUNKS=1),
# with one and with four calls in flight.  GPU box, repo root.
for nf in 512 1024; do for pl in 1 4; do for ch in 1 2; do
  echo -n "frames=$nf pipeline=$pl DH_CHUNKS=$ch  "
  DH_CHUNKS=$ch timeout -k 10 200 python bench.py --frames $nf --pipeline $pl --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done; done
