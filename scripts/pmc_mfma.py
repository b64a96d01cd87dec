#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and
GRBM_GUI_ACTIVE (rocpd database).  MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per
v_mfma_f32_32x32x16_bf16 per SIMD) summed over the chip's 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE comes back summed over the
8 XCDs (checked: a 44 us kernel reports ~8 x 44 us x clock), so
    utilisation = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
Cross-check on the vendor library's 4096^3 GEMM in the same pass: 0.58 here vs 1422 / 2500 = 0.57 by its TFLOP/s.
usage: pmc_mfma.py results.db"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"))
per = {}
for k, c, n, v in rows:
    per.setdefault(k, {})[c] = (n, v)
print("| kernel | dispatches | MFMA busy cycles / dispatch (all SIMDs) | GRBM_GUI_ACTIVE / dispatch (sum of 8 XCDs) | MFMA utilisation |")
print("|---|---|---|---|---|")
tot_busy = tot_act = 0.0
fam_busy = fam_act = 0.0
for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    n, busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]; _, act = d["GRBM_GUI_ACTIVE"]
    tot_busy += busy; tot_act += act
    if any(x in k for x in ("gemm_kernel<", "gemm8_kernel<", "gemm_streamk_kernel<", "gemm_ksplit_kernel<", "gemm_kphase_kernel<")):
        fam_busy += busy; fam_act += act
    if busy <= 0:
        continue
    short = k.replace("(anonymous namespace)::", "")
    print(f"| {short[:80]} | {n} | {busy / n:.3e} | {act / n:.3e} | {busy / (128.0 * act):.3f} |")
if fam_act:
    print(f"\nGEMM family: {fam_busy / (128.0 * fam_act):.3f} of the matrix-core cycles busy while its kernels run")
if tot_act:
    print(f"all kernels: {tot_busy / (128.0 * tot_act):.3f}")
