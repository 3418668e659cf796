#!/bin/bash
# tools/build_variant.sh <tag> <source.hip>[,<source2.hip>...] <flags...>: libvc2hip_exp_<tag>.so = the RELEASE objects with the
# named sources recompiled with extra flags
set -e
cd "$(dirname "$0")/../vc2-reference_amd/csrc"
tag=$1; srcs=${2//,/ }; shift; shift
mkdir -p exp_$tag
objs=$(ls *.o)
for src in $srcs; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Xarch_host -fvisibility=hidden "$@" -c $src -o exp_$tag/${src%.hip}.o &
  objs=$(echo "$objs" | grep -v "^${src%.hip}.o$")
done
for j in $(jobs -p); do wait $j; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../libvc2hip_exp_$tag.so $objs exp_$tag/*.o
