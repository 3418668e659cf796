#!/bin/bash
# tools/p16_regs.sh [extra hipcc flags]: registers / scratch / LDS of the k_hq_pack16 kernels in ~2 s instead of a three-minute
# compile of csrc/vc2hip_slices.hip (no GPU needed): a translation unit of that file's scalar helpers (everything in front of
# k_hq_pack) + vc2hip_pack16.h + explicit instantiations of the three modes -- how the one-pass coder's 86 -> 64 registers were found
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; T=$(mktemp -d)
n=$(grep -n "^template <int W, bool MID, class ST, bool GIMG = false>" $R/vc2-reference_amd/csrc/vc2hip_slices.hip | head -1 | cut -d: -f1)
head -n $((n - 1)) $R/vc2-reference_amd/csrc/vc2hip_slices.hip > $T/p16only.hip
cat >> $T/p16only.hip <<'X'
struct __attribute__((aligned(4))) Dword4 { unsigned x, y, z, w; };
#include "vc2hip_pack16.h"
template __global__ void k_hq_pack16<0>(const PackParams p);
template __global__ void k_hq_pack16<1>(const PackParams p);
template __global__ void k_hq_pack16<2>(const PackParams p);
X
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$R/vc2-reference_amd/csrc -I$R/include "$@" -c $T/p16only.hip -o $T/p16only.o
bash $R/tools/kernel_regs.sh $T/p16only.o pack16I | sed 's/ \+/ /g'
rm -rf $T
