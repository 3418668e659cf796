# occupancy sweep of k_fwd_pair (levels 0 + 1) by LDS padding: experiment builds d4 (prefetch 1) / d5 (prefetch 2, 24 bytes of spill)
for t in d4 d5; do L=$PWD/vc2-reference_amd/libvc2hip_exp_$t.so
for pad in 0 600 5000 12000; do echo "$t pad $pad: $(VC2HIP_LIB=$L VC2HIP_PAIR_LDSPAD=$pad python tools/time_fwd.py 2>&1 | grep dwt)"; done; done
echo "release: $(python tools/time_fwd.py 2>&1 | grep dwt)"
