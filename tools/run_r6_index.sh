mkdir -p gpurun_out/r6; ulimit -c 0
bash tools/ab_r6.sh "base release stage1" cfg2@128 cfg1@128 > gpurun_out/r6/ab_index2.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/r6/tests_index2.txt 2>&1
FUZZ_TAIL=1 python tools/fuzz_decode.py 611 3000 > gpurun_out/r6/fuzz_index2.txt 2>&1
FUZZ_TAIL=1 FUZZ_BIG=1 python tools/fuzz_decode.py 612 200 >> gpurun_out/r6/fuzz_index2.txt 2>&1
FUZZ_BATCH=1 python tools/fuzz_geometry.py 621 150 > gpurun_out/r6/fuzz_batch1.txt 2>&1
FUZZ_BATCH=1 python tools/fuzz_geometry.py 622 100 wide >> gpurun_out/r6/fuzz_batch1.txt 2>&1
