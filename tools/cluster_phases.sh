#!/bin/bash
# Phases of k_cluster per accumulator (profiling twin: DH_CL_STOP low bits 1 guess / 4 region zeroed / 5 leaf list built / 2 first region /
# 3 first weighted sum; + 16 skips the position workgroups, + 32 the rotation workgroups).  GPU box, repo root.
export DH_LIB_PATH=$PWD/depthhead_amd/libdepthhead_hip_knobs.so
for st in 0 16 32 17 20 21 18 19 33 36 34 35; do
  echo -n "DH_CL_STOP=$st  "; DH_CL_STOP=$st timeout -k 10 120 python tools/kernel_times.py ${1:-fitted} 10 15 4 640 480 256 10 2>/dev/null | grep -o "cluster [0-9.]*"
done
