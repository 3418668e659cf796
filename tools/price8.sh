#!/bin/bash
# tools/price8.sh: what byte records between the first forward launch and the slice coder would be worth (libvc2hip_exp_price8.so:
# wrong results, the traffic and instruction count of the real thing minus its quantiser) -- cfg 2, 128 pictures, alternating
for r in 1 2 3; do for t in release price8; do
  L=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so; [ "$t" = release ] && L=$PWD/vc2-reference_amd/libvc2hip.so
  echo "$t $(env VC2HIP_LIB=$L python tools/time_cfg.py cfg2@128 2>&1 | grep -v amdgpu)"; done; done
