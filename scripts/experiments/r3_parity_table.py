#!/usr/bin/env python3
"""Error table of the production bf16 path against the rounding-point emulation (oracle/emul_bf16.py): every case of
tests/test_gpu_emulation.py, without asserting.  -> markdown on stdout (profiles/r03_parity_errors.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_emulation as T

CASES = [("tiny-128", m, 4, 4, 10) for m in ("kadaptation", "lora", "adapter", "compacter")] + \
        [("ViT-B/32", "kadaptation", 4, 8, 10), ("ViT-B/32", "lora", 8, 8, 10), ("ViT-B/32", "adapter", 4, 8, 10),
         ("ViT-B/32", "compacter", 4, 8, 10), ("ViT-B/16", "compacter", 4, 8, 10), ("ViT-L/14", "kadaptation", 4, 8, 10),
         ("ViT-B/32-2L", "kadaptation", 4, 128, 100), ("ViT-B/32-2L", "lora", 8, 128, 100)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if any(a in f"{c[0]}|{c[1]}" for a in sys.argv[1:])]
print("| case | logits (max / max) | loss (abs) | worst gradient (rel L2) | tensor | median gradient |")
print("|---|---|---|---|---|---|")
for arch, method, r, B, C in CASES:
    le, lo, errs = T._run(arch, method, r, B, C)
    worst = max(errs.items(), key=lambda kv: kv[1])
    med = sorted(errs.values())[len(errs) // 2]
    print(f"| {arch} + {method} (r={r}) bs {B} | {le:.2e} | {lo:.2e} | {worst[1]:.2e} | {worst[0][-48:]} | {med:.2e} |", flush=True)
    if os.environ.get("PARITY_VERBOSE"):
        for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:8]:
            print(f"    {e:.3e}  {k}")

print()
print("| single block (seam, teacher-forced) | y (rel L2) | dx (rel L2) | worst gradient | tensor | median gradient |")
print("|---|---|---|---|---|---|")
BLOCKS = [(768, 32, 224, 512, "kadaptation", 4, 128, 11), (768, 32, 224, 512, "kadaptation", 4, 128, 12), (768, 32, 224, 512, "lora", 8, 128, 13),
          (768, 32, 224, 512, "adapter", 4, 128, 14), (768, 32, 224, 512, "compacter", 4, 128, 15), (768, 32, 224, 512, "kadaptation", 4, 64, 16),
          (768, 16, 224, 512, "compacter", 4, 16, 17), (1024, 14, 224, 768, "kadaptation", 4, 8, 18), (128, 16, 48, 64, "kadaptation", 4, 4, 19)]
for width, patch, res, embed, method, r, B, seed in BLOCKS:
    if len(sys.argv) > 1 and not any(a in f"block|{method}" for a in sys.argv[1:]):
        continue
    ye, dxe, errs = T._run_block(width, patch, res, embed, method, r, B, seed)
    worst = max(errs.items(), key=lambda kv: kv[1])
    med = sorted(errs.values())[len(errs) // 2]
    print(f"| E={width} patch {patch} {method} (r={r}) bs {B} | {ye:.2e} | {dxe:.2e} | {worst[1]:.2e} | {worst[0][-48:]} | {med:.2e} |", flush=True)
    if os.environ.get("PARITY_VERBOSE"):
        for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:8]:
            print(f"    {e:.3e}  {k}")
