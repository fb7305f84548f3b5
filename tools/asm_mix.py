#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S --cuda-device-only listing:
   python tools/asm_mix.py file.s <mangled kernel name> [min instructions per block]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]; minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 12
start = [i for i, l in enumerate(lines) if l.startswith(name + ':')][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].strip() == '.Lfunc_end%s:' % '' or lines[i].startswith('.Lfunc_end'))
blocks = []; cur = ('entry', [])
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: blocks.append(cur); cur = (m.group(1), [])
    elif l.startswith('\t') and not l.strip().startswith(('.', ';')): cur[1].append(l.strip().split()[0])
blocks.append(cur)
tot = Counter()
for nm, ins in blocks:
    c = Counter()
    for i in ins:
        k = 'mfma' if i.startswith('v_mfma') else 'valu' if i.startswith('v_') else 'lds' if i.startswith('ds_') else 'salu' if i.startswith('s_') else 'vmem'
        c[k] += 1; tot[k] += 1
    if len(ins) >= minlen: print('%-12s %5d' % (nm, len(ins)), dict(c))
print('static total', dict(tot))
