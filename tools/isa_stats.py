"""Instruction counts of one kernel in the assembly `hipcc -S --cuda-device-only` writes: static VALU / SALU / LDS / VMEM
instructions, registers, scratch.  python tools/isa_stats.py <file.s> <substring of the mangled kernel name> [dump.s]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if key not in name:
        continue
    c = Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in '.;' or l.endswith(':'):
            continue
        i = l.split()[0]
        c['valu' if i.startswith('v_') else 'salu' if i.startswith('s_') else 'lds' if i.startswith('ds_')
          else 'vmem' if i.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else 'other'] += 1
    regs = dict(re.findall(r'\.set ' + re.escape(name) + r'\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)', s))
    print(name, dict(c), regs)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(body)
