#!/usr/bin/env python3
"""Per kernel of a `hipcc -S` listing: s_waitcnt vmcnt(0) and loads that sit BETWEEN the first and the last global store (on gfx950
vmcnt counts stores, so such a wait is a store round trip).  usage: isa_store_waits.py file.s [name-substring]"""
import re, sys, subprocess
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
name, ins = None, []
def flush():
    if not name or not ins: return
    st = [i for i, x in enumerate(ins) if x.startswith(("global_store", "buffer_store"))]
    if not st: return
    w = [i for i, x in enumerate(ins) if x.startswith("s_waitcnt vmcnt(0)") and st[0] < i < st[-1]]
    ld = [i for i, x in enumerate(ins) if x.startswith(("global_load", "buffer_load")) and "lds" not in x and st[0] < i < st[-1]]
    n = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat in n and (w or ld):
        print(f"{n[:95]:95s} stores {len(st):3d}  vmcnt(0) between {len(w):3d}  loads between {len(ld):3d}")
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m: flush(); name, ins = m.group(1), []
    elif line.startswith("\t") and not line.strip().startswith((".", ";")): ins.append(" ".join(line.strip().split()[:3]))
flush()
