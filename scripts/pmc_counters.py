#!/usr/bin/env python3
"""Per-kernel means of whatever counters a rocprofv3 --pmc pass collected (rocpd database).
usage: pmc_counters.py results.db [kernel-name substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"))
per = {}
for k, c, n, v in rows:
    per.setdefault(k, {})[c] = (n, v)
names = sorted({c for d in per.values() for c in d})
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, d in sorted(per.items(), key=lambda kv: -max(v for _, v in kv[1].values())):
    if pat not in k:
        continue
    n = max(n for n, _ in d.values())
    short = k.replace("(anonymous namespace)::", "")
    print(f"| {short[:70]} | {n} | " + " | ".join(f"{d[c][1] / d[c][0]:.4g}" if c in d else "-" for c in names) + " |")
