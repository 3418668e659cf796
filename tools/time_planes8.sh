# byte band planes never / always per BASELINE configuration (ablation build's environment names; tools/time_cfg.py)
L=$PWD/vc2-reference_amd/libvc2hip_ablate.so
for c in "$@"; do for e in VC2HIP_PLANES8_NEVER VC2HIP_PLANES8_ALWAYS VC2HIP_PLANES8_NEVER VC2HIP_PLANES8_ALWAYS; do
  echo "$c $e $(env $e=1 VC2HIP_LIB=$L python tools/time_cfg.py $c 2>&1 | grep -v amdgpu | sed "s/ {.*hq_unpack/ hq_unpack/")"; done; done
