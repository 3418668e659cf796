#!/bin/bash
# tools/build_stream_variant.sh <tag> <flags...>: libvc2hip_exp_<tag>.so = the ablation build with vc2hip_dwt_stream.hip compiled with extra flags (DD97 only: -DVC2_STREAM_ONE)
set -e
cd "$(dirname "$0")/../vc2-reference_amd/csrc"
tag=$1; shift
mkdir -p exp_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Xarch_host -fvisibility=hidden "$@" -c vc2hip_dwt_stream.hip -o exp_$tag/vc2hip_dwt_stream.o
objs=$(ls ablate/*.o | grep -v vc2hip_dwt_stream.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../libvc2hip_exp_$tag.so $objs exp_$tag/vc2hip_dwt_stream.o
