#!/usr/bin/env python3
"""Ordered timeline of ONE fine-tune step from a rocprofv3 kernel trace (rocpd database, view `kernels`): every dispatch with its
start offset, duration and the idle gap in front of it, plus per-phase totals (stem | blocks forward | tail forward + head + tail
backward | blocks backward | reduce / chain / SGD) and the idle time between dispatches.

usage: r6_step_timeline.py results.db [step_index_from_the_end=3]  -> markdown on stdout

A step is delimited by its first kernel (zero_fill_kernel of pevit_zero_grads, or the im2col kernel) -- the trace of
`bench.py --steps 30 --warmup 5` holds 55 of them; one of the timed ones is printed."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(db.execute("select name, start, end from kernels order by start"))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    for a, b in (("void ", ""), ("(GemmParams, int)", ""), ("(GemmParams, int, int, int)", "")):
        n = n.replace(a, b)
    return n[:70]


# step boundaries: the SGD kernel ends a step
ends = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
assert len(ends) > back + 1, "not enough steps in the trace"
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = step[0][1]
names = [short(r[0]) for r in step]
# phases by landmark kernels
first_attn_f = next(i for i, n in enumerate(names) if "attn_fwd" in n)
last_attn_f = max(i for i, n in enumerate(names) if "attn_fwd" in n)
first_attn_b = next(i for i, n in enumerate(names) if "attn_bwd" in n)
last_ln_b = max(i for i, n in enumerate(names) if "ln_bwd" in n)
first_block = next(i for i, n in enumerate(names) if "ln_fwd" in n and i > 2) + 1          # ln_pre is the stem's last launch
phases = [("stem (zero, im2col, patch embed, ln_pre)", 0, first_block),
          ("blocks forward up to the last attention", first_block, last_attn_f + 1),
          ("tail: last block on the class rows, ln_post, proj, head + loss, their backward", last_attn_f + 1, first_attn_b),
          ("blocks backward", first_attn_b, last_ln_b + 1),
          ("adapter-gradient reduce / chain / SGD", last_ln_b + 1, len(step))]
span = (step[-1][2] - t0) / 1e3
busy = sum(r[2] - r[1] for r in step) / 1e3
print("(idle gaps are those of the traced run: rocprofv3 serialises dispatches)")
print(f"one step: {len(step)} dispatches, {span:.1f} us from the first kernel's start to the last one's end, {busy:.1f} us of kernels, "
      f"{span - busy:.1f} us idle between dispatches ({(span - busy) / (len(step) - 1):.2f} us per boundary)\n")
print("| phase | dispatches | span us | kernel us | idle us | share of the step |")
print("|---|---|---|---|---|---|")
for name, a, b in phases:
    if b <= a:
        continue
    sp = (step[b - 1][2] - step[a][1]) / 1e3
    ku = sum(r[2] - r[1] for r in step[a:b]) / 1e3
    print(f"| {name} | {b - a} | {sp:.1f} | {ku:.1f} | {sp - ku:.1f} | {sp / span * 100:.1f} % |")
print("\n| # | start us | dur us | gap us | kernel |")
print("|---|---|---|---|---|")
prev = None
for i, (n, s, e) in enumerate(step):
    gap = 0.0 if prev is None else (s - prev) / 1e3
    mark = " **<- tail**" if i == last_attn_f + 1 else (" **<- blocks backward**" if i == first_attn_b else "")
    if last_attn_f - 8 <= i <= first_attn_b + 9 or i < first_block + 8 or i >= last_ln_b - 2:
        print(f"| {i} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.2f} | {short(n)}{mark} |")
    elif i == first_block + 8 or i == first_attn_b + 10:
        print("| ... | | | | |")
    prev = e
