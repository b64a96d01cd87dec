"""Parity in the regime every reference run is actually in: full-size towers at the REFERENCE INITIALISATION, bs 8, five SGD steps,
on the PRODUCTION kernels.  Fixtures tests/golden/*_refinit.{json,npz}, recorded from the imported reference by
tests/golden/make_golden.py --refinit:

    full_b32_kadaptation_refinit   model.py:533-539,554: both Kronecker factors zero -- only attn.b and the head ever move
    full_b32_lora_r8_refinit       lora_model.py:466-475: A ~ N(0, 0.02), B = 0 -- B moves from step 1, A from step 2 (config 3)
    full_b32_adapter_refinit       adapter_model.py:285-295: N(0, 0.02) weights, zero biases -- everything trains
    full_b16_compacter_refinit     compacter_model.py:254-288,511-519: glorot W_left / W_right, frozen rule (config 4)
    full_l14_kadaptation_refinit   ViT-L/14, bf16 and fp8-e4m3 frozen weights (config 5)

Recorded per fixture: logits / loss / features of step 0, every non-zero gradient of step 0 AND of the last step (by then the
low-rank factors have moved, so every adapter-gradient kernel carries signal), what the five steps changed (final - initial), the
loss trajectory, the BatchNorm statistics.  Tensors are stored in full; for the two large adapters (LoRA r=8, bottleneck Adapter)
in full on the blocks listed in ``full_layers`` and as 32 seeded sign projections + norm on the others (an unbiased estimate of the
relative L2 error that, unlike a norm, sees permutations; asserted at 1.4 x the gate = its 3-sigma sampling width).

The gates (round 6): every quantity is asserted at

    max( STATED gate of BASELINE.md section 3,  FLOOR_C x the bf16 floor the REFERENCE ITSELF records for that quantity )

with ONE constant FLOOR_C = 3.  The floor is part of each fixture (``floor`` in the .json, written by ``make_golden.py --refinit
--bf16-weights``): the imported reference run again from the same initial state with (leg "weights") the frozen weights rounded to
bf16 -- what BASELINE configs 2-4 prescribe -- and (leg "operands") additionally both operands of every contraction and the
gradient arriving at its output rounded to bf16, i.e. what ANY engine that feeds a bf16 matrix core does; recorded per tensor as
the deviation from the reference's own f32 run (ViT-L/14 also leg "fp8": e4m3 block weights, BASELINE config 5).  No line of this
repository's engine, oracle or emulation enters the floor.  For a gradient tensor the floor is that of its KIND -- the name with
the block index replaced by *, the largest value over the blocks and legs: one tensor's floor is a single draw of an
ill-conditioned sum (measured engine / own-tensor floor 0.3-6.6, engine / kind floor 0.35-1.8).

Why 3: measured over every tensor kind of the fixtures (profiles/r06_parity_refinit.md) the worst tensor of a kind sits at
0.8-1.8 x the kind's floor, median 1.0-1.4 per fixture, whole-step vectors at 0.8-1.35, the five-step loss trajectory of Compacter at
2.4 (the engine also STORES activations -- q / k / v, the attention output, h, LayerNorm outputs, the residual gradient stream -- in
bf16 between its kernels, which the floor legs do not, and a trajectory accumulates it); both sides of a per-kind ratio are maxima
over 12-24 draws of a heavy-tailed error.  The MEDIAN ratio over the kinds of a fixture is asserted at <= 2 next to it: a
systematic loss of accuracy fails even when every kind stays under its own gate.
The attention-site methods on bf16 weights -- KAdaptation B/32 and L/14, LoRA r = 8 -- take no floor at all: they are held to the
STATED gates (STATED_ONLY_METHODS); the floor gates the two post-MLP adapters (config 4 among them) and the fp8 weights.

Per-tensor gates stay below 1 (a zero tensor scores 1.0, a permutation 1.4, a sign flip 2.0): a kind whose gate FLOOR_C x floor
would reach 0.95 (floor >= 0.317) is CHAOTIC at this batch -- the reference with bf16 operands is a third or more away from its own
f32 run on it (the down-path column sums of the two post-MLP adapters: over the 400 / 1576 tokens of the batch with almost complete
cancellation at step 0, and after five steps, with the loss down from 4.6 to 0.2-0.3, most of their tensors) -- so that no per-tensor
L2 statement about it means anything; such tensors are listed in the report and counted (KAdaptation, LoRA and ViT-L/14 have none at any step; at step 0 the
up-projection and the head are never among them), and covered by the whole-step gate
(all gradient tensors of a step as ONE vector: relative L2 <= max(stated, FLOOR_C x the reference's whole-step floor) -- 0.03-0.2
measured -- and < 1) and by tests/test_gpu_emulation.py.  profiles/r06_parity_refinit.md has engine-vs-floor per quantity and the
gstream_bf16 0 / 1 A/B on the ill-conditioned kinds."""
import json
import math
import os

import pytest
import torch

from conftest import FLOOR_C, kind_of, load_golden, max_rel, proj_rel_err, reference_floors, rel_err

pytestmark = pytest.mark.gpu
STATED_LOGITS, STATED_GRADS, LOSS_ABS = 2e-2, 5e-2, 2e-2          # BASELINE.md section 3 (+ the 2-layer loss gate of test_gpu_tower.py)
PROJ_WIDTH = 1.4                                                  # 32 projections: the estimate of a relative error has sigma ~ 12.5 %
PER_TENSOR_CAP = 0.95                                             # no per-tensor gate at or above the score of a zero tensor (1.0)
CHAOTIC_FLOOR = PER_TENSOR_CAP / FLOOR_C                          # 0.317: FLOOR_C x floor would reach the cap -- no per-tensor gate (docstring)
STATED_ONLY_METHODS = ("kadaptation", "lora")                     # bf16 weights: held to the STATED gates, no floor (B/32 and L/14 KAdaptation, LoRA r = 8)
MEDIAN_RATIO = 2.0                                                # median over a fixture's tensor kinds of (worst engine error / kind floor)

CASES = [("full_b32_kadaptation_refinit", "bf16"), ("full_b32_lora_r8_refinit", "bf16"), ("full_b32_adapter_refinit", "bf16"),
         ("full_b16_compacter_refinit", "bf16"), ("full_l14_kadaptation_refinit", "bf16"), ("full_l14_kadaptation_refinit", "fp8"),
         # the same launch sequences with f32 storage and contractions (PEVIT_W_F32_VERIFY): separates arithmetic from layout
         ("full_b32_lora_r8_refinit", "f32-verify"), ("full_b32_adapter_refinit", "f32-verify"), ("full_b16_compacter_refinit", "f32-verify")]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def engine_at_reference_init(meta, t, weight_format="bf16", tune=None):
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, reference_init_, synth_state_dict
    arch = ARCHS[meta["arch"]]
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    for k, (s1, s2) in meta["sd_checksum"].items():          # the regenerated checkpoint is the one the fixture was recorded on
        v = sd[k].double()
        assert math.isclose(float(v.sum()), s1, rel_tol=1e-9, abs_tol=1e-9) and math.isclose(float((v ** 2).sum()), s2, rel_tol=1e-9), k
    spec = adapter_param_spec(meta["method"], arch.width, arch.layers, meta["lora_r"])
    if "init_checksum" in meta:                              # initial values re-drawn from the reference's law by the seeded helper
        named = [(n, torch.zeros(s)) for n, s, tr in spec if tr]
        reference_init_(named, meta["method"], seed=7)
        for n, v in named:
            s1, s2 = meta["init_checksum"][n]
            assert math.isclose(float(v.double().sum()), s1, rel_tol=1e-9, abs_tol=1e-9) and math.isclose(float((v.double() ** 2).sum()), s2, rel_tol=1e-9, abs_tol=1e-12), n
            sd[n] = v
    for n, s, _ in spec:                                     # the reference's own draws where it drew; zero where it left zero
        if n not in sd:
            sd[n] = t["adapter/" + n].float().view(s) if "adapter/" + n in t else torch.zeros(s)
    eng = HipEngine(arch, meta["method"], meta["classes"], meta["batch"], lora_rank=meta["lora_r"], weight_format=weight_format)
    eng.load_state_dict(sd)
    for key, val in (tune or {}).items():                    # A/B knobs of the library (scripts/r6_refinit_report.py)
        assert eng.tune(key, val) == 0, key
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(t["head_w"]); v["layers.0.bias"].copy_(t["head_b"])
    return arch, eng


def key_of(name):
    return name if name.startswith("layers.") else "backbone." + name


def compare(kind, name, value, meta, t):
    """relative L2 error of one tensor against the fixture: exact where it is stored in full, estimated from the projections where
    not; None when the fixture has no entry (the reference left it exactly zero / unchanged)."""
    k = key_of(name)
    v = value.detach().float().cpu()
    if f"{kind}/{k}" in t:
        return rel_err(v, t[f"{kind}/{k}"].view_as(v)), False
    if f"{kind}_proj/{k}" in t:
        return proj_rel_err(v, meta["proj_index"][k], t[f"{kind}_proj/{k}"], t[f"{kind}_norm/{k}"]), True
    return None, False


def measure(tag, weight_format="bf16", tune=None):
    from pevit_amd.synth import synth_batch
    meta, t = load_golden(tag)
    arch, eng = engine_at_reference_init(meta, t, weight_format, tune)
    images, labels = synth_batch(meta["batch"], arch.resolution, meta["classes"])
    images, labels = images.cuda(), labels.cuda()
    init = {n: p.detach().clone() for n, p in eng.param_views().items()}
    out = {"losses": [], "tag": tag, "weights": weight_format}
    for step in range(meta["steps"]):
        logits, loss = eng.forward_backward(images, labels)
        torch.cuda.synchronize()
        out["losses"].append(float(loss))
        if step == 0:
            out["logits"] = max_rel(logits.cpu(), t["logits0"])
            out["loss0"] = abs(float(loss) - float(t["loss0"]))
        if step in (0, meta["steps"] - 1):
            kind = "grad" if step == 0 else "grad_last"
            out[kind + "_all"] = whole_step_error(kind, eng.grad_views(), meta, t)
            res, est, zero_bad = {}, set(), []
            for name, g in eng.grad_views().items():
                r, e = compare(kind, name, g, meta, t)
                if r is not None:
                    res[name] = r
                    if e:
                        est.add(name)
                elif float(g.abs().max()) != 0.0:
                    zero_bad.append(name)
            out[kind], out[kind + "_estimated"], out[kind + "_nonzero_where_reference_has_zero"] = res, sorted(est), zero_bad
        eng.sgd_step(meta["lr"], 0.9, meta["wd"])
    torch.cuda.synchronize()
    delta, est, moved_bad = {}, set(), []
    for name, p in eng.param_views().items():
        d = p.detach() - init[name]
        r, e = compare("delta", name, d, meta, t)
        if r is not None:
            delta[name] = r
            if e:
                est.add(name)
        elif float(d.abs().max()) != 0.0:
            moved_bad.append(name)
    out["delta"], out["delta_estimated"], out["moved_where_reference_did_not"] = delta, sorted(est), moved_bad
    out["delta_all"] = whole_step_error("delta", {n: p.detach() - init[n] for n, p in eng.param_views().items()}, meta, t)
    out["loss_traj"] = [abs(a - b) for a, b in zip(out["losses"], meta["losses"])]
    out["bn_var"] = rel_err(eng.running_var.cpu(), t["bn_var"]); out["bn_mean"] = rel_err(eng.running_mean.cpu(), t["bn_mean"])
    return meta, t, out


def worst(d):
    return max(d.items(), key=lambda kv: kv[1]) if d else ("-", 0.0)


def floors(meta, weights):
    """conftest.reference_floors, except that the f32 verification mode and the attention-site methods on bf16 weights take none
    (they are held to the stated gates)."""
    if weights == "f32-verify" or (meta["method"] in STATED_ONLY_METHODS and weights == "bf16"):
        return None
    return reference_floors(meta, fp8=(weights == "fp8"))


def whole_step_error(kind, values, meta, t):
    """All tensors of one kind as ONE vector: sqrt(sum |a - ref|^2 / sum |ref|^2) over everything the fixture recorded (exact on
    the tensors stored in full, the projection estimate on the others)."""
    num = den = 0.0
    for name, v in values.items():
        k = key_of(name)
        v = v.detach().double().cpu().flatten()
        if f"{kind}/{k}" in t:
            r = t[f"{kind}/{k}"].double().flatten()
            num += float((v - r).pow(2).sum()); den += float(r.pow(2).sum())
        elif f"{kind}_proj/{k}" in t:
            nrm = float(t[f"{kind}_norm/{k}"])
            num += (proj_rel_err(v.float(), meta["proj_index"][k], t[f"{kind}_proj/{k}"], nrm) * nrm) ** 2; den += nrm ** 2
    return (num / (den + 1e-300)) ** 0.5


@pytest.mark.parametrize("tag,weights", CASES, ids=[f"{t}-{w}" for t, w in CASES])
def test_production_path_meets_the_stated_gates_at_reference_init(tag, weights):
    meta, t, m = measure(tag, weights)
    report = {"fixture": tag, "weights": weights, "logits_max_rel": m["logits"], "loss0_abs": m["loss0"],
              "worst_grad_step0": worst(m["grad"]), "n_grads_step0": len(m["grad"]),
              "worst_grad_last_step": worst(m["grad_last"]), "n_grads_last_step": len(m["grad_last"]),
              "worst_delta": worst(m["delta"]), "n_moved": len(m["delta"]),
              "loss_trajectory_abs": m["loss_traj"], "losses": m["losses"], "reference_losses": meta["losses"],
              "bn_var": m["bn_var"], "bn_mean": m["bn_mean"]}
    print("refinit parity:", json.dumps(report))
    os.makedirs("gpurun_out/refinit", exist_ok=True)
    with open(f"gpurun_out/refinit/{tag}_{weights}.json", "w") as f:
        json.dump({"summary": report, "grad": m["grad"], "grad_last": m["grad_last"], "delta": m["delta"],
                   "estimated_from_projections": {"grad": m["grad_estimated"], "grad_last": m["grad_last_estimated"], "delta": m["delta_estimated"]}}, f, indent=1)
    fl = floors(meta, weights)

    def gate(q, stated, name=None):
        if fl is None:
            return stated
        f = fl[q].get(kind_of(key_of(name)), 0.0) if name is not None else fl[q]
        return max(stated, FLOOR_C * f)

    # exact zeros stay exact zeros, what the reference never moves never moves
    assert not m["grad_nonzero_where_reference_has_zero"], m["grad_nonzero_where_reference_has_zero"][:4]
    assert not m["grad_last_nonzero_where_reference_has_zero"], m["grad_last_nonzero_where_reference_has_zero"][:4]
    assert not m["moved_where_reference_did_not"], m["moved_where_reference_did_not"][:4]
    n0 = sum(k.startswith(("grad/", "grad_proj/")) for k in t); n1 = sum(k.startswith(("grad_last/", "grad_last_proj/")) for k in t)
    assert len(m["grad"]) == n0 and len(m["grad_last"]) == n1, (len(m["grad"]), n0, len(m["grad_last"]), n1)   # every recorded tensor was compared
    assert m["logits"] <= gate("logits", STATED_LOGITS), (m["logits"], gate("logits", STATED_LOGITS))
    assert m["loss0"] <= gate("loss0", LOSS_ABS), (m["loss0"], gate("loss0", LOSS_ABS))
    chaotic, worst_of_kind = {}, {}
    for kind in ("grad", "grad_last", "delta"):
        for name, r in m[kind].items():
            k = kind_of(key_of(name))
            if fl is not None and fl[kind].get(k, 0.0) >= CHAOTIC_FLOOR:      # the reference's own bf16 legs are a third or more away from its f32 run
                chaotic.setdefault(kind, []).append(name)
                continue
            g = min(gate(kind, STATED_GRADS, name), PER_TENSOR_CAP) * (PROJ_WIDTH if name in m[kind + "_estimated"] else 1.0)
            worst_of_kind[(kind, k)] = max(worst_of_kind.get((kind, k), 0.0), r / (PROJ_WIDTH if name in m[kind + "_estimated"] else 1.0))
            assert r <= g, (kind, name, r, g)
        ga = gate(kind + "_all", STATED_GRADS)
        assert m[kind + "_all"] <= ga and ga < 1.0, (kind, "all tensors as one vector", m[kind + "_all"], ga)
    # the median over the fixture's tensor kinds of (worst engine error of the kind / the kind's floor), where the floor is what
    # sets the gate (C x floor above the stated gate): a systematic loss of accuracy shows here before any single kind fails
    ratios = sorted(v / fl[kd][k] for (kd, k), v in worst_of_kind.items() if fl is not None and FLOOR_C * fl[kd].get(k, 0.0) > STATED_GRADS)
    median_ratio = ratios[len(ratios) // 2] if ratios else 0.0
    assert median_ratio <= MEDIAN_RATIO, (median_ratio, ratios)
    assert max(m["loss_traj"]) <= gate("loss_traj", LOSS_ABS), (m["loss_traj"], gate("loss_traj", LOSS_ABS))
    assert m["bn_var"] <= STATED_GRADS and m["bn_mean"] <= STATED_GRADS
    # chaotic tensors are what the two post-MLP fixtures hold (the down path: LayerNorm affine + down projection; Compacter: at
    # the last step only); the attention-site methods have none at any step
    n_all = sum(len(m[k]) for k in ("grad", "grad_last", "delta"))
    n_chaotic = sum(len(v) for v in chaotic.values())
    if meta["method"] in ("kadaptation", "lora"):
        assert n_chaotic == 0, chaotic
    else:
        assert all("adapter_up" not in n and not n.startswith("layers.") for n in chaotic.get("grad", [])), chaotic.get("grad")
    with open(f"gpurun_out/refinit/{tag}_{weights}.json") as f:
        rec = json.load(f)
    rec["gates"] = {"floor_c": FLOOR_C, "logits": gate("logits", STATED_LOGITS), "loss0": gate("loss0", LOSS_ABS),
                    "loss_traj": gate("loss_traj", LOSS_ABS), **{k + "_all": gate(k + "_all", STATED_GRADS) for k in ("grad", "grad_last", "delta")},
                    "floor": None if fl is None else {k: v for k, v in fl.items() if not isinstance(v, dict)},
                    "worst_floor": None if fl is None else {k: max(fl[k].values()) for k in ("grad", "grad_last", "delta")},
                    "whole_step": {k: m[k + "_all"] for k in ("grad", "grad_last", "delta")},
                    "kind_ratio_to_floor": {"median": median_ratio, "max": max(ratios) if ratios else 0.0, "n_kinds": len(ratios)},
                    "worst_per_kind": {f"{kd}/{k}": [v, None if fl is None else fl[kd].get(k)] for (kd, k), v in sorted(worst_of_kind.items())},
                    "chaotic_tensors_not_gated_per_tensor": chaotic, "n_chaotic": n_chaotic, "n_tensors": n_all}
    with open(f"gpurun_out/refinit/{tag}_{weights}.json", "w") as f:
        json.dump(rec, f, indent=1)
    print("refinit gates:", json.dumps(rec["gates"]))
