#!/usr/bin/env python3
"""Idle time around the optimizer kernel of every step in a rocprofv3 kernel trace (rocpd database): the gap between the end of the
last kernel before `sgd_kernel` and its start, the kernels that ran in that gap window on other queues (the collective's), and the
gap between its end and the next kernel's start.  Shows where the DP routes of `bench.py --dp-route` lose their 0.1 ms per step.
usage: r5_dp_gaps.py results.db"""
import sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
out = defaultdict(list)
for i, (name, st, en) in enumerate(rows):
    if "sgd_kernel" not in name or i == 0 or i + 1 >= len(rows):
        continue
    # the last kernel of the backward pass: walk back over collective / copy kernels
    j = i - 1
    between = []
    while j > 0 and ("nccl" in rows[j][0].lower() or "rccl" in rows[j][0].lower() or "copyBuffer" in rows[j][0] or "AllReduce" in rows[j][0]):
        between.append(rows[j][0][:40]); j -= 1
    before = (st - rows[j][2]) / 1e3
    after = (rows[i + 1][1] - en) / 1e3
    out[tuple(sorted(set(between)))].append((before, after, rows[j][0][:30], rows[i + 1][0][:30]))
allv = [x for v in out.values() for x in v]
for nm, idx in (("before sgd", 0), ("after sgd", 1)):
    xs = sorted(x[idx] for x in allv)
    q = lambda f: xs[min(len(xs) - 1, int(f * len(xs)))]
    print(f"gap {nm}: min {xs[0]:.1f}  p10 {q(.1):.1f}  p25 {q(.25):.1f}  p50 {q(.5):.1f}  p75 {q(.75):.1f}  p90 {q(.9):.1f}  max {xs[-1]:.1f} us over {len(xs)} steps")
# the steps in trace order, 16 per line: gap before / after
seq = []
for i, (name, st, en) in enumerate(rows):
    if "sgd_kernel" in name and 0 < i < len(rows) - 1:
        seq.append(f"{(st - rows[i - 1][2]) / 1e3:.0f}/{(rows[i + 1][1] - en) / 1e3:.0f}")
for i in range(0, len(seq), 20):
    print(" ".join(seq[i:i + 20]))
for k, v in out.items():
    v.sort()
    m = v[len(v) // 2]
    print(f"{len(v):4d} steps | kernels between backward and sgd: {list(k) or 'none'} | median gap before sgd {m[0]:.1f} us (after `{m[2]}`), "
          f"after sgd {sorted(x[1] for x in v)[len(v)//2]:.1f} us (before `{m[3]}`)")

# per step (from one sgd_kernel to the next): span, busy time, idle time and the largest gaps
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
print("step | span us | busy us | idle us | largest gaps (us after kernel)")
for a, b in zip(sg[:-1], sg[1:]):
    ks = rows[a + 1:b + 1]
    span = (rows[b][2] - rows[a][2]) / 1e3
    busy = sum(e - s0 for _, s0, e in ks) / 1e3
    gaps = sorted(((ks[i + 1][1] - ks[i][2]) / 1e3, ks[i][0].replace("(anonymous namespace)::", "")[:24]) for i in range(len(ks) - 1))[-3:]
    if span < 20000:
        print(f"{a:6d} | {span:8.1f} | {busy:8.1f} | {span - busy:7.1f} | " + ", ".join(f"{g:.0f} after {n}" for g, n in reversed(gaps)))
