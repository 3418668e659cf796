#!/bin/bash
# segments per strip of the last inverse level, forced (ablation build's VC2HIP_STREAM_NSEG_IF) against the planner's choice; several processes per value (the kernel's time is a draw per process)
L=$PWD/vc2-reference_amd/libvc2hip_ablate.so
for r in 1 2 3 4; do for n in $2; do
  echo "$1 NSEG_IF $n $(env VC2HIP_LIB=$L VC2HIP_STREAM_NSEG_IF=$n python tools/time_cfg.py $1 2>&1 | grep -v amdgpu | grep -o "'idwt_level_final': [0-9.]*\|^cfg.*Gpx/s" | tr '\n' ' ')"
done; done
