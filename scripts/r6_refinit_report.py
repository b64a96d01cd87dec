#!/usr/bin/env python3
"""Round 6: the production kernels against the *_refinit fixtures AND against the bf16 floor the reference itself records in them
(tests/golden/make_golden.py --refinit --bf16-weights), per tensor kind (the block index replaced by *: the floor of a kind is the
largest floor over its blocks and legs -- a single tensor's floor is one draw of an ill-conditioned sum).

    python scripts/r6_refinit_report.py [--ab gstream_bf16=0] [--cases tag:weights,...]  -> gpurun_out/r6_refinit/{report.md,*.json}

--ab KEY=VAL: every bf16 case is measured a second time with that library knob (pevit_tune) and both columns are printed."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def kind_of(n):
    return re.sub(r"resblocks\.\d+\.", "resblocks.*.", n)


def kind_floors(fl, legs, kind):
    out = {}
    for leg in legs:
        for n, v in fl[leg][kind].items():
            out[kind_of(n)] = max(out.get(kind_of(n), 0.0), v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ab", default="")
    ap.add_argument("--cases", default="")
    args = ap.parse_args()
    import test_gpu_refinit as T
    cases = [tuple(c.split(":")) for c in args.cases.split(",")] if args.cases else [c for c in T.CASES if c[1] != "f32-verify"]
    tune = None
    if args.ab:
        k, v = args.ab.split("=")
        tune = {k: int(v)}
    os.makedirs("gpurun_out/r6_refinit", exist_ok=True)
    lines = ["| fixture | weights | quantity | engine | " + (f"engine, {args.ab} | " if tune else "") + "floor (weights leg) | floor (operands leg) | engine / floor |", "|---|---|---|---|" + ("---|" if tune else "") + "---|---|---|"]
    kind_lines = ["| fixture | weights | step | tensor kind (worst block) | engine | " + (f"engine, {args.ab} | " if tune else "") + "floor of the kind | engine / floor |", "|---|---|---|---|---|" + ("---|" if tune else "") + "---|---|"]
    for tag, w in cases:
        meta, t, m = T.measure(tag, w)
        m2 = T.measure(tag, w, tune)[2] if (tune and w == "bf16") else None
        fl = meta.get("floor")
        legs = ["weights", "operands"] + (["fp8"] if (w == "fp8" and fl and "fp8" in fl) else [])
        rec = {"tag": tag, "weights": w, "logits": m["logits"], "loss0": m["loss0"], "loss_traj": max(m["loss_traj"]),
               **{k + "_all": m[k + "_all"] for k in ("grad", "grad_last", "delta")}, "kinds": {}}
        if m2:
            rec["ab"] = {"knob": args.ab, "logits": m2["logits"], "loss0": m2["loss0"], "loss_traj": max(m2["loss_traj"]),
                         **{k + "_all": m2[k + "_all"] for k in ("grad", "grad_last", "delta")}}

        def f(leg, q):
            if not fl or leg not in fl:
                return float("nan")
            v = fl[leg][q]
            return max(v) if isinstance(v, list) else v
        for q, val in (("logits", m["logits"]), ("loss0", m["loss0"]), ("loss_traj", max(m["loss_traj"])),
                       ("grad_all", m["grad_all"]), ("grad_last_all", m["grad_last_all"]), ("delta_all", m["delta_all"])):
            fw, fo = f("weights", q), f(legs[-1], q)
            v2 = (max(m2[q]) if q == "loss_traj" else m2[q]) if m2 else None
            lines.append(f"| {tag} | {w} | {q} | {val:.3g} | " + (("- | " if v2 is None else f"{v2:.3g} | ") if tune else "") + f"{fw:.3g} | {fo:.3g} | {val / max(fw, fo, 1e-12):.2f} |")
        for kind in ("grad", "grad_last", "delta"):
            kf = kind_floors(fl, legs, kind) if fl else {}
            worst, worst2 = {}, {}
            for n, r in m[kind].items():
                k = kind_of(T.key_of(n)); worst[k] = max(worst.get(k, 0.0), r)
            if m2:
                for n, r in m2[kind].items():
                    k = kind_of(T.key_of(n)); worst2[k] = max(worst2.get(k, 0.0), r)
            rec["kinds"][kind] = {k: {"engine": v, "floor": kf.get(k), "ab": worst2.get(k)} for k, v in worst.items()}
            for k, v in sorted(worst.items(), key=lambda kv: -kv[1] / max(kf.get(kv[0], 1e-12), 1e-12))[:6]:
                kind_lines.append(f"| {tag} | {w} | {kind} | {k.split('*.')[-1].replace('backbone.visual.transformer.', '')} | {v:.3g} | "
                                  + (("- | " if not m2 else f"{worst2.get(k, float('nan')):.3g} | ") if tune else "") + f"{kf.get(k, float('nan')):.3g} | {v / max(kf.get(k, 1e-12), 1e-12):.2f} |")
        with open(f"gpurun_out/r6_refinit/{tag}_{w}.json", "w") as fjson:
            json.dump(rec, fjson, indent=1)
        print(tag, w, "done", flush=True)
    with open("gpurun_out/r6_refinit/report.md", "w") as fmd:
        fmd.write("\n".join(lines) + "\n\n" + "\n".join(kind_lines) + "\n")
    print("\n".join(lines)); print(); print("\n".join(kind_lines))


if __name__ == "__main__":
    main()
