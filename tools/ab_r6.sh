#!/bin/bash
# tools/ab_r6.sh "<tags>" <cfgs...>: tools/time_cfg.py with each build (libvc2hip_exp_<tag>.so; `release` = libvc2hip.so), alternating, three rounds
tags=$1; shift
for r in 1 2 3; do for c in "$@"; do for t in $tags; do
  L=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so; [ "$t" = release ] && L=$PWD/vc2-reference_amd/libvc2hip.so
  echo "$c $t $(env VC2HIP_LIB=$L python tools/time_cfg.py $c 2>&1 | grep -v amdgpu | cut -d' ' -f2-)"; done; done; done
