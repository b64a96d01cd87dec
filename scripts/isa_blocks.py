#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc -S listing that contain MFMA or scratch instructions, with instruction-class counts.
usage: isa_blocks.py file.s <mangled-name-substring> [--dump BLOCK]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^_ZN\S*' + re.escape(pat) + r'\S*:', l))
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
blocks, cur, nm = [], [], 'entry'
for l in txt[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append((nm, cur)); cur = []; nm = m.group(1)
    elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
        cur.append(l.strip())
blocks.append((nm, cur))
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == '--dump' else None
for nm, b in blocks:
    c = lambda p: sum(x.startswith(p) for x in b)
    mf = sum('mfma' in x for x in b)
    if mf or c('scratch_'):
        print(nm, len(b), 'mfma', mf, 'scratch', c('scratch_'), 'ds_read', c('ds_read'), 'ds_write', c('ds_write'), 'gload', c('global_load'),
              'barrier', c('s_barrier'), 'waitcnt', c('s_waitcnt'), 'branch', c('s_cbranch'))
    if dump == nm:
        print('\n'.join(b))
