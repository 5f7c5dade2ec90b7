# batches in flight (bench.py --pipeline) x frames per batch; GPU box, repo root
for p in 2 3 4 2 3 4; do for f in 256; do
python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 6 --pipeline $p --frames $f 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pipeline $p frames $f', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'))
"; done; done
