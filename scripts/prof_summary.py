#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) per kernel name: calls, total, avg, min, max.
usage: prof_summary.py results.db [steps]   -> markdown table on stdout"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
span = list(db.execute("select min(start), max(end) from kernels"))[0]
print(f"total kernel time {tot/1e3:.2f} ms over {sum(r[1] for r in rows)} dispatches; trace span {(span[1]-span[0])/1e6:.1f} ms"
      + (f"; {tot/1e3/steps:.3f} ms of kernels per step ({steps} steps incl. warm-up)" if steps else ""))
print("| % | calls | avg us | min us | max us | kernel |")
print("|---|---|---|---|---|---|")
for name, n, total, avg, mn, mx in rows:
    if total / tot < 0.0005:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    print(f"| {total/tot*100:.2f} | {n} | {avg:.1f} | {mn:.1f} | {mx:.1f} | {short[:90]} |")
