#!/bin/bash
# k_boxsum's band rule for rectangles other than 24 x 24 (the ring's LDS, hence the workgroups a CU holds, depends on rh): 12 x 12 and
# 8 x 8 rectangles, 256 frames; DH_BOX_BANDS=0 is the rule's choice.  GPU box, repo root.
for rs in 0.15 0.1; do for b in 0 2 3 4 5 8 15; do
  echo -n "rect_scale=$rs DH_BOX_BANDS=$b  "; KT_RECT_SCALE=$rs DH_BOX_BANDS=$b timeout -k 10 120 python tools/kernel_times.py synth 10 12 4 640 480 256 10 2>/dev/null | grep -o "boxsum [0-9.]*"
done; done
