#!/bin/bash
# rocprofv3 passes of bench.py on the GPU box: kernel statistics, instruction mix, HBM traffic (run through gpurun).
# bench.py runs its timed region only (no CPU baseline, no end-to-end run, no 32-picture steps: their launches are of
# another size and would mix into the per-kernel averages).
#   tools/prof_kernels.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32 "$@" > $out/bench_under_rocprof.json 2> $out/stats.err
# the same with the batch cut over TWO streams (vc2hip_set_streams: 16 pictures per launch, the kernels of the two halves overlap)
rocprofv3 --kernel-trace --stats -d $out/stats2 -o run --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32 --streams 2 "$@" > $out/bench_under_rocprof_two_streams.json 2> $out/stats2.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $out/mix -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32 "$@" > /dev/null 2> $out/mix.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $out/mix2 -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32 "$@" > /dev/null 2> $out/mix2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32 "$@" > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32 "$@" > /dev/null 2> $out/write.err
find $out -name "*.csv" | head -30
python3 tools/pmc_mix.py $out/mix
