#!/bin/bash
# round-5 A/B runs on one box (ablation / experiment builds through VC2HIP_LIB; tools/time_cfg.py prints the per-kernel table)
mkdir -p gpurun_out/r5
O=gpurun_out/r5/exp_$1.txt; : > $O
run() { echo "## $*" >> $O; env "$@" python tools/time_cfg.py cfg2@32 >> $O 2>&1; }
A=$PWD/vc2-reference_amd/libvc2hip_ablate.so
case $1 in
e12)
  D=$PWD/vc2-reference_amd/libvc2hip_exp_direct.so
  run VC2HIP_LIB=$A
  run VC2HIP_LIB=$D
  run VC2HIP_LIB=$A
  run VC2HIP_LIB=$D
  # occupancy sensitivity: forward first level 3 -> 2 wavefronts per SIMD (LDS 11.5 -> 19 KiB), inverse final 4 -> 3 -> 2
  run VC2HIP_LIB=$A VC2HIP_STREAM_LDSPAD_FF=7000
  run VC2HIP_LIB=$A VC2HIP_STREAM_LDSPAD_IF=13000
  run VC2HIP_LIB=$A VC2HIP_STREAM_LDSPAD_IF=19000
  run VC2HIP_LIB=$A VC2HIP_STREAM_LDSPAD_FL=7000 VC2HIP_STREAM_LDSPAD_IL=13000
  ;;
esac
cat $O
