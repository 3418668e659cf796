# the ablation build's non-determinism on 2 x 2-sample slices under its tuning overrides (which stage is it?)
L=$PWD/vc2-reference_amd/libvc2hip_ablate.so
for e in "X=1" "VC2HIP_PACK_LANES=64" "VC2HIP_PACK_LANES=32" "VC2HIP_COMPACT_LANES=64" "VC2HIP_COMPACT_LANES=16" "VC2HIP_GENERIC_DWT=1" "VC2HIP_PACK16=0"; do
  echo "$e: $(env $e VC2HIP_LIB=$L python tools/probe/many_small.py 2>&1 | grep -v amdgpu | head -6 | grep -c DIFF) of 6 wrong"
done
