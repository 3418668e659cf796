# round 6's fuzz campaign on the GPU box (gpurun -- bash tools/run_r6_fuzz.sh): the one-pass slice coder forced for every batch size
# (FUZZ_FLAGS=single_pass_vbr) in every geometry mode, single pictures and batches; the default choice; the decoder's mutation fuzz
mkdir -p gpurun_out/r6; ulimit -c 0
o=gpurun_out/r6/fuzz_final.txt; : > $o
FUZZ_FLAGS=single_pass_vbr python tools/fuzz_geometry.py 631 300 >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr python tools/fuzz_geometry.py 632 250 wide >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr python tools/fuzz_geometry.py 633 200 tall >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr python tools/fuzz_geometry.py 634 250 pair >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr FUZZ_BATCH=1 python tools/fuzz_geometry.py 635 150 pair >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr FUZZ_BATCH=1 python tools/fuzz_geometry.py 636 100 wide >> $o 2>&1
FUZZ_FLAGS=single_pass_vbr python tools/fuzz_pack16.py 637 300 >> $o 2>&1
python tools/fuzz_pack16.py 638 100 >> $o 2>&1
python tools/fuzz_geometry.py 639 200 >> $o 2>&1
FUZZ_BATCH=1 python tools/fuzz_geometry.py 640 100 pair >> $o 2>&1
FUZZ_TAIL=1 python tools/fuzz_decode.py 641 20000 >> $o 2>&1
grep -v amdgpu $o | tail -40
