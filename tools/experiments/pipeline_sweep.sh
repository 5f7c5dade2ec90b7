for p in 1 2 3 4; do for f in 256 128; do
python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 --pipeline $p --frames $f 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pipeline $p frames $f', d['value'], d['ms_per_step'])
"; done; done
