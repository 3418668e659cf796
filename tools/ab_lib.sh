#!/bin/bash
# tools/ab_lib.sh "<tags>" <cfgs...>: tools/time_cfg.py of every configuration with libvc2hip_exp_<tag>.so, tags alternating (twice each)
tags=$1; shift
for c in "$@"; do for r in 1 2; do for t in $tags; do
  L=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so; [ "$t" = release ] && L=$PWD/vc2-reference_amd/libvc2hip.so
  echo "$c $t $(env VC2HIP_LIB=$L python tools/time_cfg.py $c 2>&1 | grep -v amdgpu | sed "s/ {.*hq_unpack/ hq_unpack/")"; done; done; done
