"""The STATED parity gates of BASELINE.md section 3 -- logits within 2e-2 of the largest magnitude, every gradient within
5e-2 relative L2 -- asserted against the fixtures recorded from the reference, with no calibration and no widening.

They are asserted in the engine's f32-class verification mode (weight_format "f32-verify", include/pevit_hip.h
PEVIT_W_F32_VERIFY; csrc/verify.hip): the same C entry points, launch sequences, memory layouts and index arithmetic
(head layout, raw-reshape scramble, class-token pruning, per-chunk partials, chain rules, flat parameter buffer, fused
head and SGD) as the production path, with activations and weights kept in f32 and the matrix-core contractions run as
plain f32 kernels.  On the fixtures' random-weight towers bf16 operand rounding ALONE moves logits by 1-10 % and single
gradients by 5-50 % (tests/test_gpu_tower.py measures this on the f32 oracle), which is why the production bf16 path is
held to calibrated gates there; this file shows that everything except that rounding agrees with the reference to
f32 round-off, i.e. far inside the stated gates.  Kernel-level parity of the bf16 kernels themselves (against PyTorch on
identical operands) is tests/test_gpu_ops.py / test_gpu_ops2.py.
"""
import math

import pytest
import torch

from conftest import golden_param_dict, load_golden, max_rel, rel_err

pytestmark = pytest.mark.gpu

STATED_LOGITS, STATED_GRADS = 2e-2, 5e-2
F32_LOGITS, F32_GRADS = 2e-4, 2e-3                # what the verification mode actually delivers (asserted as well)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _engine(meta, sd, t=None, batch=None, head=None):
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import ARCHS
    eng = HipEngine(ARCHS[meta["arch"]], meta["method"], meta["classes"], batch or meta["batch"], lora_rank=meta["lora_r"],
                    weight_format="f32-verify")
    eng.load_state_dict(sd)
    v = eng.param_views()
    hw, hb = head if head is not None else (t["head_w"], t["head_b"])
    with torch.no_grad():
        v["layers.0.weight"].copy_(hw); v["layers.0.bias"].copy_(hb)
    return eng


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter"])
def test_stated_gates_on_tiny_fixtures_full_tensors(case):
    meta, t = load_golden(case)
    eng = _engine(meta, golden_param_dict(meta, t), t)
    images, labels = t["images"].cuda(), t["labels"].cuda()
    feat = eng.visual_forward(images, save=False)
    assert max_rel(feat.cpu(), t["feat"]) < F32_LOGITS
    logits, loss = eng.forward_backward(images, labels)
    torch.cuda.synchronize()
    err = max_rel(logits.cpu(), t["logits0"])
    assert err < STATED_LOGITS and err < F32_LOGITS, err
    assert abs(float(loss) - float(t["loss0"])) < 1e-4
    none = {n[len("backbone."):] for n in meta["grad_is_none"]}
    for name, g in eng.grad_views().items():
        key = "grad/" + (name if name.startswith("layers.") else "backbone." + name)
        if name in none:
            assert float(g.abs().max()) == 0.0, name
            continue
        e = rel_err(g.cpu(), t[key])
        assert e < STATED_GRADS and e < F32_GRADS, (name, e)
    # the recorded 3-step SGD trajectory (losses and every final parameter)
    eng2 = _engine(meta, golden_param_dict(meta, t), t)
    losses = [float(eng2.train_step(images, labels, lr=meta["lr"], momentum=0.9, weight_decay=meta["wd"])[1]) for _ in range(meta["steps"])]
    assert max(abs(a - b) for a, b in zip(losses, meta["losses"])) < 1e-3, (losses, meta["losses"])
    for name, p in eng2.param_views().items():
        key = "final/" + (name if name.startswith("layers.") else "backbone." + name)
        if name in none:
            assert torch.equal(p.cpu(), t["adapter/" + name])
        else:
            assert rel_err(p.cpu(), t[key]) < 1e-3, name


def _full_case(meta, t):
    from pevit_amd.engine import adapter_param_spec
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_state_dict
    arch, method = ARCHS[meta["arch"]], meta["method"]
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    spec = {n: s for n, s, _ in adapter_param_spec(method, arch.width, arch.layers, meta["lora_r"])}
    ordered = [(n, torch.zeros(spec[n])) for n in meta["trainable_names"]]
    randomize_adapters(ordered, seed=3)
    sd.update(dict(ordered))
    for k, v in t.items():                       # tensors the reference adds but never trains (Compacter's phm_rule)
        if k.startswith("adapter/"):
            sd[k[len("adapter/"):]] = v.float()
    g = torch.Generator().manual_seed(5)
    bound = 1.0 / math.sqrt(arch.embed_dim)
    hw = (torch.rand((meta["classes"], arch.embed_dim), generator=g) * 2 - 1) * bound
    hb = (torch.rand((meta["classes"],), generator=g) * 2 - 1) * bound
    return arch, sd, hw, hb


@pytest.mark.parametrize("case", ["full_b32_kadaptation", "full_b32_lora", "full_b32_lora_r8", "full_b32_adapter",
                                  "full_b32_compacter", "full_b16_compacter", "full_l14_kadaptation"])
def test_stated_gates_on_full_size_fixtures(case):
    """ViT-B/32 (12 layers) and ViT-L/14 (24 layers, N = 257) at full width, bs = 8, as recorded from the reference: logits,
    loss and the norm of every gradient tensor."""
    from pevit_amd.synth import synth_batch
    meta, t = load_golden(case)
    arch, sd, hw, hb = _full_case(meta, t)
    eng = _engine(meta, sd, head=(hw, hb))
    images, labels = synth_batch(meta["batch"], arch.resolution, meta["classes"])
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    err = max_rel(logits.cpu(), t["logits0"])
    assert err < STATED_LOGITS and err < 10 * F32_LOGITS, err
    assert abs(float(loss) - float(t["loss0"])) < 1e-3
    for name, g in eng.grad_views().items():
        ref = meta["grad_norms"][name if name.startswith("layers.") else "backbone." + name]
        if ref is None:
            assert float(g.abs().max()) == 0.0
        else:
            assert abs(float(g.double().norm()) - ref) < 5e-3 * max(ref, 1e-8), (name, float(g.double().norm()), ref)
            key = name if name.startswith("layers.") else "backbone." + name
            if "grad_proj/" + key in t:               # the tensor itself through its recorded sign projections (round 5)
                from conftest import proj_rel_err
                e = proj_rel_err(g.cpu(), meta["proj_index"][key], t["grad_proj/" + key], ref)
                assert e < STATED_GRADS, (name, e)


@pytest.mark.parametrize("arch_name,method,lora_r,B", [("ViT-B/32", "kadaptation", 4, 8), ("ViT-B/32", "lora", 8, 8),
                                                        ("ViT-B/32", "adapter", 4, 8), ("ViT-B/32", "compacter", 4, 8),
                                                        ("ViT-B/32-2L", "kadaptation", 4, 128)])
def test_stated_gates_full_tensors_vs_live_oracle(arch_name, method, lora_r, B):
    """Every gradient TENSOR (not its norm) of the headline architecture, and the whole step at the headline batch of 128
    on the two-block tower, against the oracle on the same inputs."""
    from oracle import ref_cpu
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import synth_batch
    from test_gpu_tower import _full_size_case
    arch, sd = _full_size_case(arch_name, method, lora_r)
    C = 10 if B == 8 else 100
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    tr = ref_cpu.OracleTrainer(sd, method, C)
    ref_logits, ref_loss = tr.loss_and_grads(images, labels)
    eng = HipEngine(arch, method, C, B, lora_rank=lora_r, weight_format="f32-verify")
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(tr.head_w.detach()); v["layers.0.bias"].copy_(tr.head_b.detach())
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    err = max_rel(logits.cpu(), ref_logits)
    assert err < STATED_LOGITS and err < 10 * F32_LOGITS, err
    assert abs(float(loss) - float(ref_loss)) < 1e-3
    gv = eng.grad_views()
    for k in tr.names:
        if tr.p[k].grad is None:
            assert float(gv[k].abs().max()) == 0.0
        else:
            e = rel_err(gv[k].cpu(), tr.p[k].grad)
            assert e < STATED_GRADS and e < 5 * F32_GRADS, (k, e)
    assert rel_err(gv["layers.0.weight"].cpu(), tr.head_w.grad) < F32_GRADS


def test_verification_mode_and_production_mode_share_the_layout():
    """Same flat parameter buffer, same mask, same parameter count; only the arena / workspace element size differs."""
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import ARCHS
    a = HipEngine(ARCHS["tiny-128"], "kadaptation", 10, 4)
    b = HipEngine(ARCHS["tiny-128"], "kadaptation", 10, 4, weight_format="f32-verify")
    assert a.n_params == b.n_params and torch.equal(a.grad_mask_host, b.grad_mask_host)
    assert list(a.param_views()) == list(b.param_views())
    # (the production workspace also carries the stream-K slabs, 64 KiB per residency slot, which verification mode lacks)
    sk_bytes = 2 * torch.cuda.get_device_properties(0).multi_processor_count * 128 * 128 * 4
    assert b.arena.numel() > a.arena.numel() and b.workspace.numel() > a.workspace.numel() - sk_bytes
