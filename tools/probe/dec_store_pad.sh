#!/bin/bash
# elements of padding between the pictures' decoder stores (ablation build's VC2HIP_DEC_STORE_PAD): does the stride between pictures matter to the decode kernels?
L=$PWD/vc2-reference_amd/libvc2hip_ablate.so
for r in 1 2; do for n in 0 64 1024 2112 8256 33024 131136; do
  echo "PAD $n $(env VC2HIP_LIB=$L VC2HIP_DEC_STORE_PAD=$n python tools/time_cfg.py ${1:-cfg2@128} 2>&1 | grep -v amdgpu | grep -o "'hq_unpack': [0-9.]*\|'idwt_[a-z_]*': [0-9.]*\|^cfg.*Gpx/s" | tr '\n' ' ')"
done; done
