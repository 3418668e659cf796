#!/bin/bash
# instruction mix (rocprofv3 --pmc) of the kernels of tools/time_cfg.py <cfg...>: a quick look at one configuration
#   gpurun -- tools/mix_cfg.sh <tag> cfg2@32
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/mix_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $out/mix -o run --output-format csv -- python3 tools/time_cfg.py "$@" > /dev/null 2> $out/mix.err
python3 tools/pmc_mix.py $out/mix | grep -v "at::native\|rocclr" | head -${MIX_LINES:-14}
