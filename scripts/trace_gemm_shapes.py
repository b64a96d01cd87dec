#!/usr/bin/env python3
"""GEMM launches of a rocprofv3 kernel trace (rocpd database) grouped by kernel and grid size: in-step time per product.
usage: trace_gemm_shapes.py results.db [steps]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows = db.execute("select name, (end - start) / 1000.0, grid_x, workgroup_x from kernels where name like '%gemm_%kernel%'")
agg = collections.defaultdict(list)
for n, d, g, w in rows:
    i = n.index('gemm_'); j = (n.index('>', i) + 1) if '>' in n[i:] else len(n)
    agg[(n[i:j], g // max(w, 1))].append(d)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:62s} workgroups {k[1]:5d}  per step {len(v) / steps:5.1f}  avg {sum(v) / len(v):7.1f} us  us/step {sum(v) / steps:8.1f}")
