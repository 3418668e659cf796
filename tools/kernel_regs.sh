#!/bin/bash
# registers / scratch / LDS of every kernel in a HIP object or library (no GPU needed): tools/kernel_regs.sh <file.o|.so> [name filter]
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/clang-offload-bundler --unbundle --type=o --input="$1" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co 2>/dev/null || { $B/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin && $B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co; }
$B/llvm-readelf --notes $T/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if flt in name: print(f\"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}\")
" "$2"
rm -rf $T
