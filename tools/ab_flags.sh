#!/bin/bash
# tools/ab_flags.sh "<flag sets>" <cfgs...>: tools/time_cfg.py of every configuration with each set of context flags (TIME_CFG_FLAGS;
# `default` = none), alternating, three rounds, ONE build of the library -- A/B of two correct paths on one box
sets=$1; shift
for r in 1 2 3; do for c in "$@"; do for f in $sets; do
  ff=$f; [ "$f" = default ] && ff=
  echo "$c $f $(env TIME_CFG_FLAGS=$ff python tools/time_cfg.py $c 2>&1 | grep -v amdgpu | cut -d' ' -f2-)"; done; done; done
