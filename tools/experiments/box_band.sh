for b in 32 64 96 128 160 240; do
  echo -n "DH_BOX_BAND=$b  "; DH_BOX_BAND=$b timeout -k 10 120 python tools/kernel_times.py fitted 10 15 4 640 480 256 10 2>/dev/null | grep -o "boxsum [0-9.]*"
done
for b in 64 128; do
  echo -n "bench DH_BOX_BAND=$b  "; DH_BOX_BAND=$b timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith(chr(123)): d=json.loads(l); print(d['value'], d['ms_per_step'], d['kernels_ms'])
"
done
