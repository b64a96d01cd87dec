"""End-to-end parity of the HIP fine-tune step against (a) the golden fixtures recorded from
the reference's own model code and (b) the CPU oracle run live on the same seeded inputs.

Stated tolerances (bf16 operands, f32 accumulation):
  trainable-parameter count              : bit-exact
  2-layer cases (tiny-128/256)           : logits/features <= 3e-2 of the largest reference
                                           magnitude, loss <= 2e-2 absolute, gradients <= 1.5e-1
                                           relative L2 per tensor (worst measured: 2.4e-2 / 1.02e-1)
  12-layer ViT-B/32 (bs=8 fixture)       : logits <= 1e-1, loss <= 2e-2, gradient norms <= 1.5e-1
These are NOT slack for kernel bugs: every kernel is separately held to 2e-4 (f32 outputs) /
1e-2 (bf16 outputs) against PyTorch on identical operands (tests/test_gpu_ops.py).  They are the
measured size of bf16 operand rounding itself: the f32 oracle with nothing changed except its
GEMM operands rounded to bf16 (oracle.ref_cpu.operand_rounding) deviates from the f32 reference
by the same amount as the HIP path does (scripts/diag_precision.py, DESIGN.md "Numerics":
e.g. ViT-B/32 bs=8 logits 5.7e-2 emulated vs 6.5e-2 HIP; worst gradient 1.20e-1 vs 1.14e-1).
test_error_is_bf16_operand_rounding keeps that calibration live: the HIP deviation must stay
within 2.5x of the emulated one.
"""
import pytest
import torch

from conftest import floor_gate, golden_param_dict, kind_of, load_golden, load_tiny_sd, max_rel, reference_floors, rel_err

pytestmark = pytest.mark.gpu

# sign-projection estimate of a gradient tensor's relative L2 error on the full-size random-adapter fixtures (round 5).  Measured on
# the production bf16 path: worst per fixture 0.06 (LoRA r=8) / 0.15 (KAdaptation, Compacter B/32) / 0.18 (LoRA) / 0.29 (Compacter B/16)
# / 0.37 (Adapter: adapter_down.1.bias, the ill-conditioned column sums of profiles/r05_parity_refinit.md); a permuted tensor reads
# 1.4, a sign-flipped one 2.0.  The f32 verification mode holds the same projections to the STATED 5e-2 (tests/test_gpu_verify.py).
PROJ_GRAD_TOL = 0.5
LOGIT_TOL, LOSS_TOL, GRAD_TOL = 3e-2, 2e-2, 1.5e-1
DEEP_LOGIT_TOL, DEEP_GRAD_TOL = 1e-1, 1.5e-1
TRAJ_LOSS_TOL, TRAJ_NORM_TOL = 1.5e-1, 6e-2     # second SGD step of the full-size fixtures (random x160 adapters): calibrated, see profiles/r04_parity_gates.md
BUILT = ("kadaptation", "lora")


def make_engine(meta, tensors, max_batch=None):
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import ARCHS
    arch = ARCHS[meta["arch"]]
    eng = HipEngine(arch, meta["method"], meta["classes"], max_batch or meta["batch"], lora_rank=meta["lora_r"])
    sd = golden_param_dict(meta, tensors)
    eng.load_state_dict(sd)
    views = eng.param_views()
    with torch.no_grad():
        views["layers.0.weight"].copy_(tensors["head_w"]); views["layers.0.bias"].copy_(tensors["head_b"])
    return eng, sd


def bf16_noise(sd, method, classes, images, labels, head_w, head_b):
    """Per-tensor deviation that bf16 operand rounding ALONE causes in the f32 oracle on this case:
    returns (oracle_f32, logits_f32, loss_f32, logit_err, {name: grad rel-L2 err})."""
    from oracle import ref_cpu

    def run(emulate):
        tr = ref_cpu.OracleTrainer(sd, method, classes)
        with torch.no_grad():
            tr.head_w.copy_(head_w); tr.head_b.copy_(head_b)
        if emulate:
            with ref_cpu.operand_rounding(torch.bfloat16):
                lg, ls = tr.loss_and_grads(images, labels)
        else:
            lg, ls = tr.loss_and_grads(images, labels)
        return tr, lg, ls
    f32, l32, loss32 = run(False)
    emu, lemu, lemu_loss = run(True)
    errs = {n: rel_err(emu.p[n].grad, f32.p[n].grad) for n in f32.names if f32.p[n].grad is not None}
    errs["layers.0.weight"] = rel_err(emu.head_w.grad, f32.head_w.grad)
    errs["layers.0.bias"] = rel_err(emu.head_b.grad, f32.head_b.grad)
    if method in ("adapter", "compacter"):
        # the bottleneck Adapter has a hard non-linearity on the trainable path (ReLU, adapter_model.py:271): WHICH gradient
        # tensor absorbs the mask flips of near-zero pre-activations depends on where the rounding happens, so the noise of a
        # tensor is taken as the larger of the two emulations of it.  (Compacter since round 4: its LayerNorm-affine gradients are
        # cancellation-heavy column sums; two valid launch sequences of the engine -- fused and separate post-MLP kernels, equal to
        # 2e-2 -- sit 0.18 and 0.21 from f32 on the same tensor, on either side of 2.5 x the operand-rounding figure alone.)
        from oracle import emul_bf16
        rp = emul_bf16.EmulTrainer(sd, method, classes)
        with torch.no_grad():
            rp.head_w.copy_(head_w); rp.head_b.copy_(head_b)
        rp.loss_and_grads(images, labels)
        for n in list(errs):
            g = rp.head_w.grad if n == "layers.0.weight" else rp.head_b.grad if n == "layers.0.bias" else rp.p[n].grad
            errs[n] = max(errs[n], rel_err(g, f32.head_w.grad if n == "layers.0.weight" else f32.head_b.grad if n == "layers.0.bias" else f32.p[n].grad))
    bf16_noise.last_loss_err = abs(float(lemu_loss) - float(loss32))      # same measure for the scalar loss
    return f32, l32, loss32, max_rel(lemu, l32), errs


def tol(base, emulated):
    """Gate of the bf16 production path against the f32 fixtures = the calibrated tolerance, widened per tensor only where
    bf16 rounding ALONE (bf16_noise: the f32 oracle with its contraction operands rounded to bf16 and, for the bottleneck
    Adapter, also the rounding-point emulation oracle/emul_bf16.py -- same inputs, same tensor) already exceeds it.  The
    STATED gates of BASELINE.md (2e-2 / 5e-2) are asserted without any widening in the f32-class verification mode
    (tests/test_gpu_verify.py); the production kernels themselves are held per block to 2e-3 ... 1.2e-2 against the
    rounding-point emulation (tests/test_gpu_emulation.py)."""
    return max(base, 2.5 * emulated + 1e-2)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter"])
def test_train_step_matches_reference_fixture(case):
    meta, t = load_golden(case)
    eng, sd = make_engine(meta, t)
    assert eng.n_params == meta["n_trainable_params"]                    # bit-exact count
    assert [n for n in eng.param_views() if not n.startswith("layers.")] == meta["trainable_names"]
    images, labels = t["images"].cuda(), t["labels"].cuda()
    feat = eng.visual_forward(images, save=False)
    assert max_rel(feat.cpu(), t["feat"]) < LOGIT_TOL
    logits, loss = eng.forward_backward(images, labels, bn_training=True)
    torch.cuda.synchronize()
    assert max_rel(logits.cpu(), t["logits0"]) < LOGIT_TOL
    assert abs(float(loss) - float(t["loss0"])) < LOSS_TOL
    none = {n[len("backbone."):] for n in meta["grad_is_none"]}
    _, _, _, _, noise = bf16_noise(sd, meta["method"], meta["classes"], t["images"], t["labels"], t["head_w"], t["head_b"])
    for name, g in eng.grad_views().items():
        key = "grad/" + (name if name.startswith("layers.") else "backbone." + name)
        if name in none:
            assert float(g.abs().max()) == 0.0, name              # reference .grad is None
            continue
        err = rel_err(g.cpu(), t[key])
        assert err < tol(GRAD_TOL, noise[name]), (name, err, noise[name])


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora", "tiny_adapter", "tiny_compacter"])
def test_sgd_trajectory_matches_reference_fixture(case):
    meta, t = load_golden(case)
    eng, sd = make_engine(meta, t)
    images, labels = t["images"].cuda(), t["labels"].cuda()
    losses = []
    for _ in range(meta["steps"]):
        _, loss = eng.train_step(images, labels, lr=meta["lr"], momentum=0.9, weight_decay=meta["wd"])
        losses.append(float(loss))
    for a, b in zip(losses, meta["losses"]):
        assert abs(a - b) < 3.5e-2, (losses, meta["losses"])          # measured <= 1.7e-2 (profiles/r03_parity_errors.md)
    none = {n[len("backbone."):] for n in meta["grad_is_none"]}
    for name, p in eng.param_views().items():
        key = "final/" + (name if name.startswith("layers.") else "backbone." + name)
        if name in none:      # never updated: torch skips params without grad, even with weight decay
            assert torch.equal(p.cpu(), t["adapter/" + name]), name
            continue
        assert rel_err(p.cpu(), t[key]) < 8e-2, (name, rel_err(p.cpu(), t[key]))      # measured <= 6.4e-2
    assert rel_err(eng.running_mean.cpu(), t["bn_mean"]) < 3e-2       # measured <= 1.4e-2
    assert rel_err(eng.running_var.cpu(), t["bn_var"]) < 1.2e-2       # measured <= 5.2e-3


@pytest.mark.parametrize("method,arch_name,B", [("kadaptation", "tiny-128", 6), ("kadaptation", "tiny-256", 5),
                                                 ("lora", "tiny-256", 3), ("adapter", "tiny-256", 5),
                                                 ("compacter", "tiny-128", 6), ("compacter", "tiny-256", 3)])
def test_transformer_seam_vs_oracle(method, arch_name, B):
    """Transformer.forward / backward at the (N,B,E) operator seam against the live oracle."""
    from oracle import ref_cpu
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_state_dict
    arch = ARCHS[arch_name]
    sd = {k: v for k, v in synth_state_dict(arch, seed=11, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=4)
    for n, v in ad:
        if n.endswith("phm_rule"):       # Compacter's frozen rule ~ U(-1,1) (compacter_model.py:513)
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    eng = HipEngine(arch, method, 10, B)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(arch.tokens, B, arch.width, generator=g)
    dy = torch.randn(arch.tokens, B, arch.width, generator=g)
    y = eng.transformer_forward(x.cuda())
    eng.zero_grad()
    dx = eng.transformer_backward(dy.cuda(), need_dx=True)
    torch.cuda.synchronize()
    p = {k: v.clone() for k, v in sd.items()}
    names = ref_cpu.trainable_names(p, method)
    for k in names:
        p[k].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = ref_cpu.transformer_forward(xr, p, arch.layers, arch.heads, method)
    yr.backward(dy)
    assert max_rel(y.cpu(), yr.detach()) < LOGIT_TOL
    assert rel_err(dx.cpu(), xr.grad) < GRAD_TOL
    gv = eng.grad_views()
    for k in names:
        if p[k].grad is None:
            assert float(gv[k].abs().max()) == 0.0
        else:
            assert rel_err(gv[k].cpu(), p[k].grad) < GRAD_TOL, (k, rel_err(gv[k].cpu(), p[k].grad))


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora", "tiny_adapter", "tiny_compacter"])
def test_error_is_bf16_operand_rounding(case):
    """Calibration: the HIP path's deviation from the f32 oracle must be of the size that bf16
    operand rounding ALONE causes in the oracle (same inputs), not larger."""
    from oracle import ref_cpu
    meta, t = load_golden(case)
    eng, sd = make_engine(meta, t)

    def run(emulate):
        tr = ref_cpu.OracleTrainer(sd, meta["method"], meta["classes"])
        with torch.no_grad():
            tr.head_w.copy_(t["head_w"]); tr.head_b.copy_(t["head_b"])
        if emulate:
            with ref_cpu.operand_rounding(torch.bfloat16):
                lg, _ = tr.loss_and_grads(t["images"], t["labels"])
        else:
            lg, _ = tr.loss_and_grads(t["images"], t["labels"])
        return tr, lg
    f32, l32 = run(False)
    emu, lemu = run(True)
    logits, _ = eng.forward_backward(t["images"].cuda(), t["labels"].cuda())
    torch.cuda.synchronize()
    assert max_rel(logits.cpu(), l32) <= 2.5 * max_rel(lemu, l32) + 5e-3
    gv = eng.grad_views()
    hip_w = max(rel_err(gv[n].cpu(), f32.p[n].grad) for n in f32.names if f32.p[n].grad is not None)
    emu_w = max(rel_err(emu.p[n].grad, f32.p[n].grad) for n in f32.names if f32.p[n].grad is not None)
    assert hip_w <= 2.5 * emu_w + 1e-2, (hip_w, emu_w)


def test_reference_init_gives_bias_only_gradients():
    """At the reference initialisation (Kronecker factors all zero) only attn.b and
    the head receive non-zero gradients -- and the delta reduces to the scrambled bias."""
    meta, t = load_golden("tiny_kadaptation")
    eng, sd = make_engine(meta, t)
    views = eng.param_views()
    with torch.no_grad():
        for name, v in views.items():
            if name.startswith("layers."):
                continue
            v.copy_(t["init/" + name])
            if name.endswith("attn.b"):
                v.add_(0.1)
    eng.forward_backward(t["images"].cuda(), t["labels"].cuda())
    torch.cuda.synchronize()
    for name, g in eng.grad_views().items():
        nz = float(g.abs().max()) > 0
        if "adapter1" in name or "phm_rule" in name:
            assert not nz, name
        else:
            assert nz, name


@pytest.mark.parametrize("case", ["full_b32_kadaptation", "full_b32_lora", "full_b32_adapter", "full_b32_compacter",
                                  "full_b32_lora_r8", "full_b16_compacter"])
def test_full_size_bs8_matches_reference_fixture(case):
    """Real width/depth (ViT-B/32 all four methods, LoRA r=8, ViT-B/16 + Compacter), bs=8: logits, loss and
    per-tensor gradient norms recorded from the reference itself (fixtures hold summaries only; the 88M-parameter
    checkpoint is regenerated from its seed).  The 24-layer ViT-L/14 fixture (full_l14_kadaptation) pins the ORACLE
    on CPU (tests/test_oracle_golden.py); on this random-weight 24-layer network bf16 operand rounding alone moves
    the logits by ~13 % (emulated), so the engine is held to the live oracle with the emulation-calibrated gate
    instead (test_baseline_config_architectures_vs_oracle)."""
    import math
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
    meta, t = load_golden(case)
    arch, method = ARCHS[meta["arch"]], meta["method"]
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    spec = {n: s for n, s, _ in adapter_param_spec(method, arch.width, arch.layers, meta["lora_r"])}
    ordered = [(n, torch.zeros(spec[n])) for n in meta["trainable_names"]]
    randomize_adapters(ordered, seed=3)
    sd.update(dict(ordered))
    for k, v in t.items():                       # tensors the reference adds but never trains (Compacter's phm_rule)
        if k.startswith("adapter/"):
            sd[k[len("adapter/"):]] = v.float()
    eng = HipEngine(arch, method, meta["classes"], meta["batch"], lora_rank=meta["lora_r"])
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    bound = 1.0 / math.sqrt(arch.embed_dim)
    views = eng.param_views()
    with torch.no_grad():
        views["layers.0.weight"].copy_((torch.rand((meta["classes"], arch.embed_dim), generator=g) * 2 - 1) * bound)
        views["layers.0.bias"].copy_((torch.rand((meta["classes"],), generator=g) * 2 - 1) * bound)
    images, labels = synth_batch(meta["batch"], arch.resolution, meta["classes"])
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    deep = 1.0
    # Round 6: the PARITY gate of every quantity is max(stated gate of BASELINE.md, FLOOR_C x the bf16 floor the reference itself
    # records in the fixture) -- conftest.reference_floors: the imported reference re-run with bf16 weights / bf16 contraction
    # operands, no line of this repository in it.  The calibrated figures of rounds 3-5 (DEEP_LOGIT_TOL, PROJ_GRAD_TOL, ...) stay
    # next to it as regression bounds: whichever is tighter binds.
    fl = reference_floors(meta)
    assert fl is not None, "fixture without a reference-recorded floor: tests/golden/make_golden.py --floor-random"
    logit_err = max_rel(logits.cpu(), t["logits0"])
    assert logit_err < min(DEEP_LOGIT_TOL * deep, floor_gate(2e-2, fl["logits"])), (logit_err, fl["logits"])
    assert abs(float(loss) - float(t["loss0"])) < min(LOSS_TOL * deep, floor_gate(2e-2, fl["loss0"]))
    bad, proj_errs = [], {}
    for name, gten in eng.grad_views().items():
        key = name if name.startswith("layers.") else "backbone." + name
        ref = meta["grad_norms"][key]
        if ref is None:
            assert float(gten.abs().max()) == 0.0
        else:
            got = float(gten.double().norm())
            if abs(got - ref) > DEEP_GRAD_TOL * deep * max(ref, 1e-8):
                bad.append((name, got, ref))
            # the tensor itself through its 32 recorded sign projections (round 5): an unbiased estimate of the relative L2 error that
            # sees what a norm cannot -- a permuted or sign-flipped tensor reads 1.4 / 2.0.  Gate: the calibrated bf16 gradient gate
            # of the random-adapter towers (GRAD_TOL) widened by the 3-sigma sampling width of 32 projections
            if "grad_proj/" + key in t:
                from conftest import proj_rel_err
                e = proj_rel_err(gten.cpu(), meta["proj_index"][key], t["grad_proj/" + key], ref)
                proj_errs[name] = e
    assert not bad, bad[:5]
    worst_p = max(proj_errs.items(), key=lambda kv: kv[1]) if proj_errs else ("-", 0.0)
    print(f"projection-estimated gradient errors {case}: worst {worst_p[1]:.3e} ({worst_p[0]}), mean {sum(proj_errs.values()) / max(len(proj_errs), 1):.3e}")
    assert worst_p[1] < PROJ_GRAD_TOL, worst_p
    # ... and per tensor against the floor of its KIND (x 1.4: the 3-sigma sampling width of 32 projections), capped below the score of
    # a zero tensor; the ratio to the floor is printed for profiles/r06_parity_refinit.md
    ratios = {}
    for name, e in proj_errs.items():
        k = kind_of(name if name.startswith("layers.") else "backbone." + name)
        f = fl["grad"].get(k, 0.0)
        assert e < 1.4 * floor_gate(5e-2, f, cap=0.68), (name, e, f)            # 1.4 x 0.68 = 0.95
        if f > 0:
            ratios[k] = max(ratios.get(k, 0.0), e / f)
    rs = sorted(ratios.values())
    print(f"floor ratios {case}: logits {logit_err:.3e} / floor {fl['logits']:.3e} = {logit_err / max(fl['logits'], 1e-12):.2f}; "
          f"gradient kinds: median {rs[len(rs) // 2] if rs else 0:.2f}, max {rs[-1] if rs else 0:.2f} (estimated from projections)")
    # the recorded SGD trajectory (full_b32_*: two steps): loss of the second step and the norm of every trained tensor after it
    if len(meta["losses"]) > 1:
        eng.sgd_step(meta["lr"], 0.9, meta["wd"])
        for _ in range(len(meta["losses"]) - 1):
            _, loss = eng.forward_backward(images.cuda(), labels.cuda())
            eng.sgd_step(meta["lr"], 0.9, meta["wd"])
        torch.cuda.synchronize()
        dl = abs(float(loss) - meta["losses"][-1])
        worst = ("", 0.0)
        for name, p in eng.param_views().items():
            key = name if name.startswith("layers.") else "backbone." + name
            ref = meta["final_norms"][key]
            e = abs(float(p.double().norm()) - ref) / max(ref, 1e-8)
            if e > worst[1]:
                worst = (name, e)
        print(f"trajectory {case}: |loss_{len(meta['losses']) - 1} - ref| = {dl:.3e}, worst final-norm deviation {worst[1]:.3e} ({worst[0]})")
        assert dl < min(TRAJ_LOSS_TOL, floor_gate(2e-2, fl["loss_traj_steps"][-1])), (dl, fl["loss_traj_steps"])
        assert worst[1] < TRAJ_NORM_TOL, worst


def test_bs128_properties_full_size():
    """BASELINE config 2 size (B=128): size-independent properties instead of an oracle run:
    determinism of a step, and linearity of the backward pass in the upstream gradient."""
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_state_dict
    arch = ARCHS["ViT-B/32"]
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec("kadaptation", 768, 12)]
    randomize_adapters(ad, seed=3)
    sd.update(dict(ad))
    B = 128
    eng = HipEngine(arch, "kadaptation", 100, B)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(arch.tokens, B, 768, generator=g).cuda()
    dy = torch.randn(arch.tokens, B, 768, generator=g).cuda()
    y1 = eng.transformer_forward(x)
    eng.zero_grad(); dx1 = eng.transformer_backward(dy); g1 = eng.grads.clone()
    y2 = eng.transformer_forward(x)
    eng.zero_grad(); dx2 = eng.transformer_backward(dy); g2 = eng.grads.clone()
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and torch.equal(g1, g2)      # deterministic
    eng.zero_grad(); dx3 = eng.transformer_backward(2.0 * dy); g3 = eng.grads.clone()
    assert rel_err(dx3.cpu(), (2.0 * dx1).cpu()) < 1e-2                               # linear in dy
    assert rel_err(g3.cpu(), (2.0 * g1).cpu()) < 1e-2
    assert torch.isfinite(y1).all() and torch.isfinite(g1).all()


# B >= 8: BatchNorm over 2-3 samples is ill-conditioned (the f32 oracle with bf16-rounded operands
# alone then moves gradients by 40-150 %, scripts/diag_precision.py long)
@pytest.mark.parametrize("method,arch_name,B", [("kadaptation", "tiny-n197", 8), ("lora", "tiny-n257", 8),
                                                 ("compacter", "tiny-n197", 9), ("adapter", "tiny-n257", 8)])
def test_long_sequences_full_step_vs_oracle(method, arch_name, B):
    """Token counts of ViT-B/16 (N=197) and ViT-L/14 (N=257; patch 14 needs a zero-padded im2col K):
    whole step (images -> loss -> gradients) against the live oracle."""
    from oracle import ref_cpu
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
    arch = ARCHS[arch_name]
    sd = {k: v for k, v in synth_state_dict(arch, seed=21, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=6)
    for n, v in ad:
        if n.endswith("phm_rule"):
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    images, labels = synth_batch(B, arch.resolution, 10, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5)
    D = arch.embed_dim
    head_w = (torch.rand((10, D), generator=g) * 2 - 1) / D ** 0.5
    head_b = (torch.rand((10,), generator=g) * 2 - 1) / D ** 0.5
    tr, ref_logits, ref_loss, logit_noise, noise = bf16_noise(sd, method, 10, images, labels, head_w, head_b)
    eng = HipEngine(arch, method, 10, B)
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(head_w); v["layers.0.bias"].copy_(head_b)
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    assert max_rel(logits.cpu(), ref_logits) < tol(LOGIT_TOL, logit_noise)
    assert abs(float(loss) - float(ref_loss)) < 3e-2
    gv = eng.grad_views()
    for k in tr.names:
        if tr.p[k].grad is None:
            assert float(gv[k].abs().max()) == 0.0
        else:
            err = rel_err(gv[k].cpu(), tr.p[k].grad)
            assert err < tol(GRAD_TOL, noise[k]), (k, err, noise[k])


# ---- the architectures of BASELINE configs 3-5 at full width/depth ---------------------------------------
def _full_size_case(arch_name, method, lora_r, seed=2):
    from pevit_amd.engine import adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_state_dict
    arch = ARCHS[arch_name]
    sd = {k: v for k, v in synth_state_dict(arch, seed=seed, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers, lora_r)]
    randomize_adapters(ad, seed=3)
    for n, v in ad:
        if n.endswith("phm_rule"):
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    return arch, sd


@pytest.mark.parametrize("arch_name,method,lora_r", [("ViT-B/32", "kadaptation", 4), ("ViT-B/32", "lora", 8),
                                                      ("ViT-B/16", "compacter", 4), ("ViT-L/14", "kadaptation", 4)])
def test_baseline_config_architectures_vs_oracle(arch_name, method, lora_r):
    """BASELINE configs 2 (the headline: ViT-B/32 + KAdaptation), 3 (ViT-B/32 + LoRA r=8), 4 (ViT-B/16 + Compacter, N=197) and 5 (ViT-L/14 + KAdaptation,
    width 1024, 24 layers, N=257, patch 14) at their real width and depth, batch 8 (what the CPU oracle finishes
    in seconds): whole step against the live oracle.  Same gates as the 12-layer fixture test, widened per
    tensor only where bf16 operand rounding alone exceeds them (24 layers accumulate more of it)."""
    from oracle import ref_cpu
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import synth_batch
    arch, sd = _full_size_case(arch_name, method, lora_r)
    B, C = 8, 10
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5)
    D = arch.embed_dim
    head_w = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5
    head_b = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
    if method == "lora" and lora_r != 4:
        pytest.importorskip("oracle")      # the oracle infers r from the tensor shapes
    tr, ref_logits, ref_loss, logit_noise, noise = bf16_noise(sd, method, C, images, labels, head_w, head_b)
    eng = HipEngine(arch, method, C, B, lora_rank=lora_r)
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(head_w); v["layers.0.bias"].copy_(head_b)
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and torch.isfinite(eng.grads).all()
    assert max_rel(logits.cpu(), ref_logits) < tol(DEEP_LOGIT_TOL, logit_noise)
    # absolute loss gate at ~2x the measured error (12 blocks <= 2.1e-2 (ViT-B/16 + Compacter; ViT-B/32 7e-3), ViT-L/14's 24 blocks 3.1e-2 -- profiles/r03_parity_errors.md)
    assert abs(float(loss) - float(ref_loss)) <= (4e-2 if arch.layers <= 12 else 6e-2)
    gv = eng.grad_views()
    for k in tr.names:
        if tr.p[k].grad is None:
            assert float(gv[k].abs().max()) == 0.0
        else:
            err = rel_err(gv[k].cpu(), tr.p[k].grad)
            assert err < tol(DEEP_GRAD_TOL, noise[k]), (k, err, noise[k])
    for k, ref in (("layers.0.weight", tr.head_w.grad), ("layers.0.bias", tr.head_b.grad)):
        assert rel_err(gv[k].cpu(), ref) < tol(DEEP_GRAD_TOL, noise[k]), k


@pytest.mark.parametrize("method", ["kadaptation", "lora"])
def test_whole_train_step_at_b128_vs_oracle(method):
    """BASELINE config 2's batch (B = 128, 6400 token rows, the tile shapes and grid sizes bench.py runs) through the WHOLE
    step -- images -> loss -> every gradient -> SGD update -- against the CPU oracle, on a tower with ViT-B/32's width,
    patch and token count cut to two blocks so that the oracle finishes in seconds.  Full tensors, not norms."""
    from oracle import ref_cpu
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import synth_batch
    arch, sd = _full_size_case("ViT-B/32-2L", method, 8 if method == "lora" else 4)
    B, C = 128, 100
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5)
    D = arch.embed_dim
    head_w = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5
    head_b = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
    tr, ref_logits, ref_loss, logit_noise, noise = bf16_noise(sd, method, C, images, labels, head_w, head_b)
    eng = HipEngine(arch, method, C, B, lora_rank=8 if method == "lora" else 4)
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(head_w); v["layers.0.bias"].copy_(head_b)
    p_before = eng.params.clone()
    logits, loss = eng.train_step(images.cuda(), labels.cuda(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    torch.cuda.synchronize()
    assert max_rel(logits.cpu(), ref_logits) < LOGIT_TOL
    assert abs(float(loss) - float(ref_loss)) < LOSS_TOL
    gv = eng.grad_views()
    for k in tr.names:
        if tr.p[k].grad is None:
            assert float(gv[k].abs().max()) == 0.0
        else:
            err = rel_err(gv[k].cpu(), tr.p[k].grad)
            assert err < tol(GRAD_TOL, noise[k]), (k, err, noise[k])
    assert rel_err(gv["layers.0.weight"].cpu(), tr.head_w.grad) < GRAD_TOL
    # the update itself: first SGD step = p - lr * (g + wd * p) on every tensor that has a gradient
    upd = p_before - 0.01 * (eng.grads + 1e-4 * p_before)
    mask = eng.grad_mask.bool()
    assert max_rel(eng.params[mask].cpu(), upd[mask].cpu()) < 1e-6
    assert torch.equal(eng.params[~mask], p_before[~mask])


@pytest.mark.parametrize("arch_name,method,lora_r,B", [("ViT-B/32", "lora", 8, 128), ("ViT-B/16", "compacter", 4, 64),
                                                        ("ViT-L/14", "kadaptation", 4, 32)])
def test_baseline_config_per_gpu_sizes_train(arch_name, method, lora_r, B):
    """The per-GPU shard sizes of BASELINE configs 3-5 (1024/8, 512/8, 256/8): the step is deterministic and
    a few SGD steps on one batch reduce the loss (size-independent properties; no oracle at this size)."""
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import reference_init_, synth_batch
    arch, sd = _full_size_case(arch_name, method, lora_r)
    eng = HipEngine(arch, method, 100, B, lora_rank=lora_r)
    eng.load_state_dict(sd)
    views = eng.param_views()
    reference_init_(views.items(), method)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        views["layers.0.weight"].copy_((torch.rand(views["layers.0.weight"].shape, generator=g) * 2 - 1) * arch.embed_dim ** -0.5)
        views["layers.0.bias"].zero_()
    images, labels = synth_batch(B, arch.resolution, 100)
    images, labels = images.cuda(), labels.cuda()
    l1, loss1 = eng.forward_backward(images, labels); g1 = eng.grads.clone(); l1 = l1.clone(); loss1 = float(loss1)
    l2, loss2 = eng.forward_backward(images, labels)
    assert torch.equal(l1, l2) and torch.equal(g1, eng.grads) and loss1 == float(loss2)
    losses = [float(eng.train_step(images, labels, lr=0.05, momentum=0.9, weight_decay=0.0)[1]) for _ in range(6)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < 0.7 * losses[0], losses


def test_dp_bucketed_step_equals_fused_step_on_one_rank():
    """The DP route (head-gradient all-reduce overlapped with the tower backward, adapter bucket after it) on a
    one-rank RCCL group must reproduce the fused single-GPU step bit for bit; covers the code bench.py runs for
    --gpus N > 1 (the 2-rank arithmetic is covered on CPU by tests/test_dp_gloo.py)."""
    import os
    import torch.distributed as dist
    meta, t = load_golden("tiny_kadaptation")
    images, labels = t["images"].cuda(), t["labels"].cuda()
    eng_a, _ = make_engine(meta, t)
    eng_b, _ = make_engine(meta, t)
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for _ in range(2):
            la, lossa = eng_a.train_step(images, labels, lr=0.01, momentum=0.9, weight_decay=1e-4)
            la, lossa = la.clone(), float(lossa)
            lb, _ = eng_b.forward_backward_dp(images, labels)
            eng_b.sgd_step(0.01, 0.9, 1e-4, 1.0)
            assert torch.equal(la, lb) and lossa == float(eng_b._loss)
        assert torch.equal(eng_a.params, eng_b.params) and torch.equal(eng_a.running_mean, eng_b.running_mean)
    finally:
        if own_group:
            dist.destroy_process_group()


def test_state_reads_right_behind_a_pipelined_step_see_the_finished_update():
    """in the pipelined DP schedule the all-reduce and the SGD kernel of step N still run on a second stream when
    train_step returns.  Everything that reads or writes the trainable state afterwards -- param_views() (state export /
    checkpoint), grad_views(), reset_optimizer(), a direct sgd_step(), transformer_forward, head_forward_backward, load_trainable --
    first makes the current stream wait for that update (dp_flush): read right behind a pipelined step, without a synchronize, the
    parameters equal those of the single-exchange schedule bit for bit."""
    import os
    import torch.distributed as dist
    meta, t = load_golden("tiny_kadaptation")
    images, labels = t["images"].cuda(), t["labels"].cuda()
    eng_a, _ = make_engine(meta, t)
    eng_b, _ = make_engine(meta, t)
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for step in range(4):
            eng_a.forward_backward_dp(images, labels, mode="single"); eng_a.sgd_step(0.01, 0.9, 1e-4, 1.0)
            eng_b._train_step_pipelined(images, labels, 0.01, 0.9, 1e-4, True, None, 1, False, None, None)
            got = {n: v.clone() for n, v in eng_b.param_views().items()}          # no synchronize, no explicit flush
            want = eng_a.param_views()
            assert all(torch.equal(got[n], want[n]) for n in want), step
        assert eng_b._pipe is not None and not eng_b._pipe["open"]                 # the read flushed
        eng_b._train_step_pipelined(images, labels, 0.01, 0.9, 1e-4, True, None, 1, False, None, None)
        eng_a.forward_backward_dp(images, labels, mode="single"); eng_a.sgd_step(0.01, 0.9, 1e-4, 1.0)
        eng_b.sgd_step(0.01, 0.9, 1e-4, 1.0); eng_a.sgd_step(0.01, 0.9, 1e-4, 1.0)   # a direct step behind a pipelined one waits for it
        assert torch.equal(eng_a.params, eng_b.params) and torch.equal(eng_a.momentum, eng_b.momentum)
        eng_b._train_step_pipelined(images, labels, 0.01, 0.9, 1e-4, True, None, 1, False, None, None)
        eng_b.reset_optimizer()                                                    # ... and so does clearing the momentum
        torch.cuda.synchronize()
        assert float(eng_b.momentum.abs().max()) == 0.0
        # capture leaves the pipelined schedule altogether (the step gate must not be baked into the graph)
        eng_b._train_step_pipelined(images, labels, 0.01, 0.9, 1e-4, True, None, 1, False, None, None)
        replay = eng_b.capture_train_step(images, labels, lr=0.01, momentum=0.9, weight_decay=1e-4)
        assert eng_b._pipe is None
        replay(); torch.cuda.synchronize()
    finally:
        if own_group:
            dist.destroy_process_group()


def test_single_sample_batches_fail_like_batchnorm():
    """BatchNorm1d in training mode rejects a 1-sample batch in PyTorch (ValueError); the engine does the same
    instead of normalising by a zero variance.  Evaluation mode (running statistics) accepts it."""
    from pevit_amd import _lib
    meta, t = load_golden("tiny_lora")
    eng, _ = make_engine(meta, t)
    img, lab = t["images"][:1].cuda(), t["labels"][:1].cuda()
    with pytest.raises(_lib.PevitError, match="more than 1 sample"):
        eng.forward_backward(img, lab, bn_training=True)
    logits, loss = eng.forward_backward(img, lab, bn_training=False)
    torch.cuda.synchronize()
    assert logits.shape == (1, meta["classes"]) and torch.isfinite(logits).all() and torch.isfinite(loss).all()


def test_bf16_ln_input_gradient_hand_over_costs_what_it_is_said_to():
    """The dX GEMMs hand the LayerNorm-input gradient to LayerNorm backward in bf16 (`dx_stored=1`, the production default:
    one more rounding on the gradient path, half the bytes on both sides of an HBM-bound kernel).  Its price is pinned here:
    same step, same inputs, `dx_stored` 1 vs 0 (f32 hand-over) -- logits identical (forward untouched), every gradient tensor
    within 1.5e-2 relative L2 (measured <= 6e-3 on the 2-block B = 128 tower), so a precision regression of the production
    path cannot hide behind the calibrated gates of the fixture tests."""
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import synth_batch
    arch, sd = _full_size_case("ViT-B/32-2L", "kadaptation", 4)
    B, C = 128, 100
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    outs = []
    for stored in (1, 0):
        eng = HipEngine(arch, "kadaptation", C, B)
        eng.load_state_dict(sd)
        assert eng.tune("dx_stored", stored) == 0
        logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
        torch.cuda.synchronize()
        outs.append((logits.clone().cpu(), {k: v.clone().cpu() for k, v in eng.grad_views().items()}))
    assert torch.equal(outs[0][0], outs[1][0])
    worst = max((rel_err(outs[0][1][k], outs[1][1][k]), k) for k in outs[1][1] if float(outs[1][1][k].abs().max()) > 0)
    assert worst[0] < 1.5e-2, worst


@pytest.mark.parametrize("arch_name,B", [("tiny-256", 7), ("ViT-B/32-2L", 24)])
@pytest.mark.parametrize("method", ["adapter", "compacter"])
def test_fused_post_mlp_adapter_equals_the_separate_launches(method, arch_name, B):
    """adapter_fused.hip (LayerNorm -> down -> activation -> up -> residual as one launch, and its backward with the affine
    LayerNorm gradients) against the separate LayerNorm / GEMM launches it replaces (`adapter_fused` = 0): the same bf16 rounding
    points (z, activation, d pre), so logits and every gradient agree to f32 summation order -- ragged last row block included."""
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
    arch, C = ARCHS[arch_name], 10
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=3)
    for n, v in ad:
        if n.endswith("phm_rule"):
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    res = []
    for fused in (1, 0):
        eng = HipEngine(arch, method, C, B)
        eng.load_state_dict(sd)
        assert eng.tune("adapter_fused", fused) == 0
        logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
        torch.cuda.synchronize()
        res.append((logits.cpu().clone(), float(loss), {k: v.cpu().clone() for k, v in eng.grad_views().items()}))
    (l1, s1, g1), (l0, s0, g0) = res
    assert max_rel(l1, l0) < 2e-3 and abs(s1 - s0) < 1e-4
    for k in g0:
        if float(g0[k].norm()) > 0:
            assert rel_err(g1[k], g0[k]) < 2e-2, (k, rel_err(g1[k], g0[k]))


@pytest.mark.parametrize("arch_name,B", [("tiny-256", 7), ("ViT-B/32-2L", 24), ("ViT-B/32", 13)])
@pytest.mark.parametrize("method", ["adapter", "compacter"])
def test_weight_gradient_products_inside_the_adapter_backward_launch_are_bit_identical(method, arch_name, B):
    """`adapter_tn_fold` (the d W_up product of a layer and the d W_down product of the layer walked before it as extra workgroups
    of adapter_bwd_kernel; the last one in a launch of its own) against one tn_gemm64 launch per product: the same arithmetic
    in the same order, so every gradient -- and the loss -- has the same bits.  Ragged last chunk (B * N not a multiple of 256) and
    an odd unit count (tiny-256: 4 slabs x 1 chunk) included."""
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
    arch, C = ARCHS[arch_name], 10
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=3)
    for n, v in ad:
        if n.endswith("phm_rule"):
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    res = []
    for fold in (1, 0):
        eng = HipEngine(arch, method, C, B)
        eng.load_state_dict(sd)
        assert eng.tune("adapter_tn_fold", fold) == 0
        for _ in range(2):      # twice: the d pre buffers alternate, the second pass starts on the other one
            logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
        torch.cuda.synchronize()
        res.append((logits.cpu().clone(), float(loss), {k: v.cpu().clone() for k, v in eng.grad_views().items()}))
    (l1, s1, g1), (l0, s0, g0) = res
    assert torch.equal(l1, l0) and s1 == s0
    for k in g0:
        assert torch.equal(g1[k], g0[k]), (k, rel_err(g1[k], g0[k]))


# (ViT-B/32-2L, ViT-B/16, ViT-L/14: the instances of lowrank_combo_kernel with heads / tokens as compile-time constants, <12, 50>,
# <12, 197>, <16, 257> (round 5), against kernels that take them at run time; the tiny towers: its run-time instance)
@pytest.mark.parametrize("method,arch_name,B", [("kadaptation", "ViT-B/32-2L", 24), ("lora", "tiny-256", 7), ("kadaptation", "tiny-n197", 3),
                                                 ("kadaptation", "ViT-B/16", 3), ("lora", "ViT-L/14", 2)])
def test_combined_lowrank_backward_equals_the_two_launches(method, arch_name, B):
    """lowrank_combo_kernel (u + dQ + d bias of a layer and the dP of the layer walked before it in one launch, the last layer's dP
    in a launch of its own) against lowrank_u + lowrank_grad per layer (`lowrank_combo` = 0): same products on the same bf16
    operands, u summed over E in a different split -> every gradient agrees to f32 summation order; logits are untouched."""
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
    arch, C = ARCHS[arch_name], 10
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers, 8)]
    randomize_adapters(ad, seed=3); sd.update(dict(ad))
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    res = []
    for combo in (1, 0):
        eng = HipEngine(arch, method, C, B, lora_rank=8)
        eng.load_state_dict(sd)
        assert eng.tune("lowrank_combo", combo) == 0
        logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
        torch.cuda.synchronize()
        res.append((logits.cpu().clone(), {k: v.cpu().clone() for k, v in eng.grad_views().items()}))
    (l1, g1), (l0, g0) = res
    assert torch.equal(l1, l0)
    for k in g0:
        if float(g0[k].norm()) > 0:
            assert rel_err(g1[k], g0[k]) < 2e-3, (k, rel_err(g1[k], g0[k]))
        else:
            assert float(g1[k].abs().max()) == 0.0, k
