export DH_LIB_PATH=$PWD/depthhead_amd/libdepthhead_hip_knobs.so
export DH_FORCE_GENERAL=1
for st in 0 9 1 3; do
  echo -n "general DH_TRAV_STOP=$st  "; DH_TRAV_STOP=$st timeout -k 10 120 python tools/kernel_times.py fitted 10 15 4 640 480 256 10 2>/dev/null | grep -v amdgpu
done
