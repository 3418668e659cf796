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
pair1)
  R=$PWD/vc2-reference_amd/libvc2hip.so
  run VC2HIP_LIB=$R VC2HIP_NO_PAIR=1
  run VC2HIP_LIB=$R
  run VC2HIP_LIB=$R VC2HIP_NO_PAIR=1
  run VC2HIP_LIB=$R
  ;;
pair2)
  run VC2HIP_LIB=$A VC2HIP_NO_PAIR=1
  run VC2HIP_LIB=$A
  run VC2HIP_LIB=$A VC2HIP_PAIR_WHOLE_SIMDS=0
  run VC2HIP_LIB=$A VC2HIP_PAIR_OUT=56
  run VC2HIP_LIB=$A VC2HIP_PAIR_OUT=56 VC2HIP_PAIR_WHOLE_SIMDS=0
  run VC2HIP_LIB=$A VC2HIP_PAIR_NSEG=5
  run VC2HIP_LIB=$A VC2HIP_PAIR_NSEG=6
  run VC2HIP_LIB=$A VC2HIP_PAIR_NSEG=8
  ;;
pair3)
  run VC2HIP_LIB=$A
  for t in w3p1 w2p2 w2p1; do run VC2HIP_LIB=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so; done
  run VC2HIP_LIB=$A
  for t in w3p1 w2p2 w2p1; do run VC2HIP_LIB=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so; done
  ;;
pair4)
  run VC2HIP_LIB=$A VC2HIP_PAIR_DEBUG=1
  for k in 8 16 24 32 56; do run VC2HIP_LIB=$A VC2HIP_DEBUG_SKIP=$k; done
  run VC2HIP_LIB=$A
  ;;
pair5)
  run VC2HIP_LIB=$A VC2HIP_PAIR_DEBUG=1
  run VC2HIP_LIB=$A VC2HIP_NO_PAIR=1
  run VC2HIP_LIB=$A VC2HIP_PAIR_GROUP=1
  run VC2HIP_LIB=$A VC2HIP_PAIR_GROUP=2
  run VC2HIP_LIB=$A
  ;;
pair6)
  for k in 0 64 128 192 256 448 16 8 24; do run VC2HIP_LIB=$A VC2HIP_DEBUG_SKIP=$k; done
  ;;
v)
  shift
  for rep in 1 2; do
    run VC2HIP_LIB=$PWD/vc2-reference_amd/libvc2hip.so
    for t in "$@"; do run VC2HIP_LIB=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so; done
  done
  ;;
inv1)
  L=$PWD/vc2-reference_amd/libvc2hip_exp_i2.so
  for k in 0 8 16 64 80 128 256 464; do run VC2HIP_LIB=$L VC2HIP_DEBUG_SKIP=$k; done
  ;;
prio)
  for rep in 1 2; do for pr in 1 2 3 4 0; do run VC2HIP_LIB=$A VC2HIP_PAIR_PRIO=$pr; done; done
  ;;
esac
cat $O
