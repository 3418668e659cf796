#!/bin/bash
# segments per strip of the first two-level forward launch, forced (ablation build's VC2HIP_PAIR_NSEG_EDGE) against the planner's choice
# tools/probe/pair_nseg.sh <cfg> "<counts>"
L=$PWD/vc2-reference_amd/libvc2hip_ablate.so
echo "planner: $(env VC2HIP_LIB=$L VC2HIP_PAIR_DEBUG=1 python tools/time_cfg.py $1 2>&1 | grep '^pair:.*edge 1' | sort | uniq -c | tr '\n' ';')"
for r in 1 2; do for n in $2; do
  echo "$1 NSEG_EDGE $n $(env VC2HIP_LIB=$L VC2HIP_PAIR_NSEG_EDGE=$n python tools/time_cfg.py $1 2>&1 | grep -v amdgpu | grep -o "'dwt_pair_first': [0-9.]*\|^cfg.*Gpx/s" | tr '\n' ' ')"
done; done
