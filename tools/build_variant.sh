#!/bin/bash
# tools/build_variant.sh <tag> <source.hip> <flags...>: libvc2hip_exp_<tag>.so = the RELEASE objects with one source recompiled with extra flags
set -e
cd "$(dirname "$0")/../vc2-reference_amd/csrc"
tag=$1; src=$2; shift; shift
mkdir -p exp_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Xarch_host -fvisibility=hidden "$@" -c $src -o exp_$tag/${src%.hip}.o
objs=$(ls *.o | grep -v ${src%.hip}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../libvc2hip_exp_$tag.so $objs exp_$tag/${src%.hip}.o
