#!/bin/bash
# k_boxsum (and the other kernels) per batch size: does the per-frame cost change with the number of frames in a call?  GPU box, repo root.
for n in ${NS:-1 8 32 64 128 192 256 320 384 448 512}; do
  echo -n "n=$n  "; timeout -k 10 120 python tools/kernel_times.py fitted 10 15 4 640 480 $n 10 2>/dev/null | grep -o "boxsum [0-9.]*\|traverse [0-9.]*\|emit [0-9.]*\|vote [0-9.]*\|cluster [0-9.]*\|total [0-9.]*" | paste -s
done
