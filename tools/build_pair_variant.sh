#!/bin/bash
# tools/build_pair_variant.sh <tag> <flags...>: libvc2hip_exp_<tag>.so = the ablation build with vc2hip_dwt_pair.hip compiled with extra flags
set -e
cd "$(dirname "$0")/../vc2-reference_amd/csrc"
tag=$1; shift
ABL=-DVC2HIP_ABLATE; if [ "$1" = "noablate" ]; then ABL=; shift; fi   # (the pair file as the release library compiles it)
mkdir -p exp_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $ABL "$@" -c vc2hip_dwt_pair.hip -o exp_$tag/vc2hip_dwt_pair.o
objs=$(ls ablate/*.o | grep -v vc2hip_dwt_pair.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvc2hip_exp_$tag.so $objs exp_$tag/vc2hip_dwt_pair.o
