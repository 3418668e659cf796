mkdir -p gpurun_out/r6; ulimit -c 0
o=gpurun_out/r6/fuzz_final2.txt; : > $o
FUZZ_FLAGS=single_pass_vbr FUZZ_BATCH=1 python tools/fuzz_geometry.py 651 200 tall >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr,planes8_always python tools/fuzz_geometry.py 652 200 pair >> $o 2>&1
FUZZ_FLAGS=two_pass_vbr FUZZ_BATCH=1 python tools/fuzz_geometry.py 653 150 pair >> $o 2>&1
python tools/fuzz_cli.py 654 120 >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr python tools/fuzz_pack16.py 655 300 >> $o 2>&1
grep -v amdgpu $o | grep -i "seed\|bad\|mismatch\|cases" | tail -12
