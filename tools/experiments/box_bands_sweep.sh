#!/bin/bash
# k_boxsum per batch size and per number of bands a frame is cut into (DH_BOX_BANDS; 0 = the library's rule).  GPU box, repo root.
for n in ${NS:-256 320 384 512}; do
  for b in 0 1 2 3 4 5 6 8 15; do
    echo -n "n=$n DH_BOX_BANDS=$b  "; DH_BOX_BANDS=$b timeout -k 10 120 python tools/kernel_times.py fitted 10 15 4 640 480 $n 10 2>/dev/null | grep -o "boxsum [0-9.]*\|total [0-9.]*" | paste -s
  done
done
