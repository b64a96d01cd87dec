#!/usr/bin/env python3
"""Instruction-class counts of one kernel in a hipcc -S listing (k-loop body vs whole kernel).
usage: isa_stats.py file.s <symbol-substring>"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^_ZN.*' + re.escape(pat) + r'.*:', l))
end = next(i for i in range(start, len(txt)) if txt[i].startswith('\t.amdhsa_kernel') or txt[i].startswith('.Lfunc_end'))
body = txt[start:end]
def classify(ins):
    c = Counter()
    for k in ins:
        if 'mfma' in k: c['mfma'] += 1
        elif k.startswith('v_'): c['valu'] += 1
        elif k.startswith('ds_'): c['ds'] += 1
        elif k.startswith(('global_', 'buffer_', 'flat_')): c['vmem'] += 1
        elif k.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif k.startswith('s_barrier'): c['barrier'] += 1
        elif k.startswith('s_'): c['salu'] += 1
    return dict(c)
ins = [l.strip().split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
print('whole kernel', len(ins), classify(ins))
print(Counter(ins).most_common(30))
# basic blocks with mfma: print each block's classes
blocks, cur, name = [], [], 'entry'
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); cur = []; name = l.split(':')[0]
    elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
        cur.append(l.strip().split()[0])
blocks.append((name, cur))
for n, b in blocks:
    if len(b) > 40:
        print(n, len(b), classify(b))
