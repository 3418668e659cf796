"""The loops of one kernel's assembly (the body tools/isa_stats.py dumps), largest first: static instruction counts by class.
python tools/isa_loops.py <kernel body .s> [how many]"""
import re
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
loops.sort(reverse=True)
for n, a, b in loops[:int(sys.argv[2]) if len(sys.argv) > 2 else 6]:
    c = Counter()
    for l in lines[a:b + 1]:
        l = l.strip()
        if not l or l[0] in '.;' or l.endswith(':'):
            continue
        t = l.split()[0]
        c['valu' if t.startswith('v_') else 'salu' if t.startswith('s_') else 'lds' if t.startswith('ds_') else 'vmem' if t.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else 'other'] += 1
    print(f"lines {a}-{b}:", dict(c), "total", sum(c.values()))
