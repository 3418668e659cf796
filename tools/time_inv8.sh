# byte band planes on / off: the decoder's kernels (tools/time_inv.py through the ablation build's environment names)
for r in 1 2; do for e in VC2HIP_PLANES8_NEVER VC2HIP_PLANES8_ALWAYS; do echo "$e $(env $e=1 VC2HIP_LIB=$PWD/vc2-reference_amd/libvc2hip_ablate.so python tools/time_inv.py 2>&1 | grep idwt)"; done; done
