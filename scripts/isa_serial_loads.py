#!/usr/bin/env python3
"""Finds the hipcc pattern `global_load ... ; s_waitcnt vmcnt(0)` (a load inside a bounds branch whose value has a second definition
at the join: the compiler drains right behind the request, one memory round trip per load) in a `hipcc -S` listing.
usage: isa_serial_loads.py file.s [...]   -> per kernel: loads, loads drained within 3 instructions, vmcnt(0) waits, stores"""
import re, sys
for path in sys.argv[1:]:
    name, ins = None, []
    def flush():
        if not name or not ins: return
        loads = [i for i, x in enumerate(ins) if x.startswith(("global_load", "buffer_load")) and "lds" not in x]
        drained = sum(1 for i in loads if any(y.startswith("s_waitcnt vmcnt(0)") for y in ins[i + 1:i + 4]))
        w0 = sum(1 for x in ins if x.startswith("s_waitcnt vmcnt(0)"))
        st = sum(1 for x in ins if x.startswith(("global_store", "buffer_store")))
        if drained >= 2:
            print(f"{path.split('/')[-1]:14s} {name[:70]:70s} loads {len(loads):3d}  drained-at-once {drained:3d}  vmcnt(0) {w0:3d}  stores {st:3d}")
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush(); name, ins = m.group(1), []
        elif line.startswith("\t") and not line.strip().startswith((".", ";")):
            ins.append(" ".join(line.strip().split()[:2]))
    flush()
