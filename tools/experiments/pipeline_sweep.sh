# batches in flight (bench.py --pipeline); GPU box, repo root
for p in 4 6 8 4 6 8; do
python bench.py --no-cpu-baseline --no-extras --steps 64 --warmup 8 --pipeline $p 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pipeline $p', d['value'], d['ms_per_step'])
"; done
