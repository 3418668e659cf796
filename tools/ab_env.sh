#!/bin/bash
# tools/ab_env.sh <VAR> "<values>" <cfgs...>: tools/time_cfg.py on the tools' build (libvc2hip_ablate.so reads the tuning variables) with
# VAR set to each value in turn, alternating, three rounds
var=$1; vals=$2; shift; shift
for r in 1 2 3; do for c in "$@"; do for v in $vals; do
  echo "$c $var=$v $(env $var=$v VC2HIP_LIB=$PWD/vc2-reference_amd/libvc2hip_ablate.so python tools/time_cfg.py $c 2>&1 | grep -v amdgpu | cut -d' ' -f2-)"; done; done; done
