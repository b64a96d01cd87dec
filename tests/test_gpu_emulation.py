"""The PRODUCTION bf16 path (MFMA GEMMs, attention, low-rank products -- the kernels bench.py times) against the
rounding-point-faithful CPU emulation ``oracle/emul_bf16.py``: both sides round at the SAME places, so
what is left is the f32 summation order inside the contractions (and 1-ulp transcendental differences).

What that buys, measured (profiles/r03_parity_errors.md):
  * ONE block with identical inputs on both sides (teacher forcing, the (N,B,E) seam, forward AND backward, at the width,
    token count and batch of the benchmark -- every production tile shape): y 2-7e-4, dx 1-3e-3, adapter gradients
    2-5e-3 relative L2.  Gates here: 2e-3 / 8e-3 / 1.2e-2 -- 5-30x sharper than anything the f32 fixtures allow.
    (Bottleneck Adapter: the tensors behind the ReLU mask 2.5e-2 -- a pre-activation within f32 round-off of zero flips
    its mask bit and moves the gradient by the whole upstream value; gate 6e-2 for those four tensors only.)
  * Whole towers: NOT ~1e-3.  A difference eps between two implementations meets the next bf16 storage point, where a
    fraction eps/ulp of the elements rounds the other way, each by a full ulp: eps' ~ sqrt(eps * ulp) -- every storage point
    pulls a 1e-6 summation-order difference up towards the rounding noise itself (ulp = 2^-8): 1e-6 -> 6e-5 -> 5e-4 -> ...
    After two blocks the engine is 2x closer to the emulation than the f32 oracle is (B = 128: logits 7.6e-3 vs 1.4e-2),
    after 12-24 blocks 1.3-1.7x (ViT-B/32: 2.9e-2 vs 4.8e-2; ViT-L/14: 1.5e-1 vs 2.0e-1).  So the whole-tower tests below
    assert (a) the 2-block B = 128 step at 2x its measured errors and (b) at full depth that the engine is as close to the
    emulation as plain f32 arithmetic is (within 1.5x) -- the sharp statement about the kernels is the per-block one.
PEVIT_RECORD_PARITY=<file> appends every measured error as a JSON line.
"""
import json
import os

import pytest
import torch

from conftest import max_rel, rel_err

pytestmark = pytest.mark.gpu

BLOCK_Y_TOL, BLOCK_DX_TOL, BLOCK_GRAD_TOL, RELU_PATH_TOL = 2e-3, 8e-3, 1.2e-2, 6e-2
STEP_LOGIT_TOL, STEP_LOSS_TOL, STEP_GRAD_TOL = 1.5e-2, 2e-3, 2.5e-2          # 2-block tower, B = 128: 2x the measured errors


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _case(arch_name, method, lora_r, seed=2):
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


def _record(tag, logit_err, loss_err, grad_errs):
    path = os.environ.get("PEVIT_RECORD_PARITY")
    if path:
        worst = max(grad_errs.items(), key=lambda kv: kv[1])
        with open(path, "a") as f:
            f.write(json.dumps({"case": tag, "logits": logit_err, "loss": loss_err, "worst_grad": worst[1],
                                "worst_grad_name": worst[0], "n_grads": len(grad_errs),
                                "median_grad": sorted(grad_errs.values())[len(grad_errs) // 2]}) + "\n")


def _run(arch_name, method, lora_r, B, C, weights="bf16"):
    from oracle import emul_bf16
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import synth_batch
    arch, sd = _case(arch_name, method, lora_r)
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5)
    D = arch.embed_dim
    head_w = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5
    head_b = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
    em = emul_bf16.EmulTrainer(sd, method, C)
    with torch.no_grad():
        em.head_w.copy_(head_w); em.head_b.copy_(head_b)
    ref_logits, ref_loss = em.loss_and_grads(images, labels)
    eng = HipEngine(arch, method, C, B, lora_rank=lora_r, weight_format=weights)
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(head_w); v["layers.0.bias"].copy_(head_b)
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and torch.isfinite(eng.grads).all()
    gv = eng.grad_views()
    errs = {}
    for k in em.names:
        if em.p[k].grad is None or float(em.p[k].grad.abs().max()) == 0.0:
            assert float(gv[k].abs().max()) == 0.0, k            # reference .grad is None / structurally zero
        else:
            errs[k] = rel_err(gv[k].cpu(), em.p[k].grad)
    errs["layers.0.weight"] = rel_err(gv["layers.0.weight"].cpu(), em.head_w.grad)
    errs["layers.0.bias"] = rel_err(gv["layers.0.bias"].cpu(), em.head_b.grad)
    logit_err, loss_err = max_rel(logits.cpu(), ref_logits), abs(float(loss) - float(ref_loss))
    _record(f"{arch_name}|{method}|r{lora_r}|bs{B}|{weights}", logit_err, loss_err, errs)
    return logit_err, loss_err, errs


def _run_block(width, patch, res, embed, method, lora_r, B, seed):
    """ONE residual block through the (N,B,E) seam, forward and backward, at the given width / token count / batch: inputs are
    identical on both sides (teacher forcing), so no rounding difference can re-amplify from block to block.
    Returns (y_err, dx_err, {name: grad err}), relative L2."""
    from oracle import emul_bf16
    from pevit_amd.engine import HipEngine, adapter_param_spec
    from pevit_amd.synth import VitArch, randomize_adapters, synth_state_dict
    arch = VitArch(f"block-{width}", width, 1, patch, res, embed)
    sd = {k: v for k, v in synth_state_dict(arch, seed=seed, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers, lora_r)]
    randomize_adapters(ad, seed=seed + 1)
    for n, v in ad:
        if n.endswith("phm_rule"):
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(arch.tokens, B, width, generator=g)
    dy = torch.randn(arch.tokens, B, width, generator=g)
    names = [n for n, _, _ in adapter_param_spec(method, arch.width, arch.layers, lora_r)]
    p = {k: v.clone() for k, v in sd.items()}
    from oracle import ref_cpu
    for k in names:
        if ref_cpu.is_trainable(method, k):
            p[k].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y_em = emul_bf16.transformer_forward(xr, p, 1, arch.heads, method)
    y_em.backward(dy)
    eng = HipEngine(arch, method, 10, B, lora_rank=lora_r)
    eng.load_state_dict(sd)
    y = eng.transformer_forward(x.cuda())
    eng.zero_grad()
    dx = eng.transformer_backward(dy.cuda())
    torch.cuda.synchronize()
    gv = eng.grad_views()
    errs = {}
    for k in names:
        if not ref_cpu.is_trainable(method, k):
            continue
        ge = p[k].grad
        if ge is None or float(ge.abs().max()) == 0.0:
            assert float(gv[k].abs().max()) == 0.0, k
        else:
            errs[k] = rel_err(gv[k].cpu(), ge)
    y_err, dx_err = rel_err(y.cpu(), y_em.detach()), rel_err(dx.cpu(), xr.grad)
    _record(f"block|E{width}|N{arch.tokens}|{method}|r{lora_r}|bs{B}|seed{seed}", y_err, dx_err, errs)
    return y_err, dx_err, errs


BLOCKS = [  # width, patch, resolution, embed, method, lora_r, batch, seed
    (768, 32, 224, 512, "kadaptation", 4, 128, 11), (768, 32, 224, 512, "kadaptation", 4, 128, 12),
    (768, 32, 224, 512, "lora", 8, 128, 13), (768, 32, 224, 512, "adapter", 4, 128, 14),
    (768, 32, 224, 512, "compacter", 4, 128, 15), (768, 32, 224, 512, "kadaptation", 4, 64, 16),
    (768, 16, 224, 512, "compacter", 4, 16, 17), (1024, 14, 224, 768, "kadaptation", 4, 8, 18),
    (128, 16, 48, 64, "kadaptation", 4, 4, 19)]


@pytest.mark.parametrize("width,patch,res,embed,method,lora_r,B,seed", BLOCKS)
def test_single_block_vs_emulation(width, patch, res, embed, method, lora_r, B, seed):
    """One residual block of ViT-B/32 (B = 128 and 64: the bench's GEMM tiles, k-split and staggered kernels), ViT-B/16
    (N = 197), ViT-L/14 (E = 1024, N = 257) and the tiny test width, every PEFT method: output, input gradient and every
    adapter gradient against the emulation on identical inputs."""
    y_err, dx_err, errs = _run_block(width, patch, res, embed, method, lora_r, B, seed)
    assert y_err < BLOCK_Y_TOL, y_err
    assert dx_err < BLOCK_DX_TOL, dx_err
    relu_path = ("adapter_down", "adapter_norm_before") if method == "adapter" else ()
    bad = {k: e for k, e in errs.items() if not e < (RELU_PATH_TOL if any(t in k for t in relu_path) else BLOCK_GRAD_TOL)}
    assert not bad, bad


@pytest.mark.parametrize("method,lora_r", [("kadaptation", 4), ("lora", 8)])
def test_b128_step_vs_emulation(method, lora_r):
    """BASELINE config 2's batch (B = 128) through the WHOLE step (stem, two blocks, head, loss, every gradient) on a
    ViT-B/32-width tower cut to two blocks."""
    logit_err, loss_err, errs = _run("ViT-B/32-2L", method, lora_r, 128, 100)
    assert logit_err < STEP_LOGIT_TOL, logit_err
    assert loss_err < STEP_LOSS_TOL, loss_err
    bad = {k: e for k, e in errs.items() if not e < STEP_GRAD_TOL}
    assert not bad, bad


def _f32_vs_emulation(arch_name, method, lora_r, B, C):
    """How far plain f32 arithmetic (oracle/ref_cpu.py) is from the emulation on the same case: (logits, median gradient)."""
    from oracle import emul_bf16, ref_cpu
    from pevit_amd.synth import synth_batch
    arch, sd = _case(arch_name, method, lora_r)
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5)
    D = arch.embed_dim
    head_w = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5
    head_b = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
    out = []
    for cls in (emul_bf16.EmulTrainer, ref_cpu.OracleTrainer):
        tr = cls(sd, method, C)
        with torch.no_grad():
            tr.head_w.copy_(head_w); tr.head_b.copy_(head_b)
        out.append((tr, tr.loss_and_grads(images, labels)[0]))
    (em, le), (orc, lo) = out
    errs = [rel_err(orc.p[k].grad, em.p[k].grad) for k in em.names
            if em.p[k].grad is not None and float(em.p[k].grad.abs().max()) > 0.0]
    return max_rel(lo, le), sorted(errs)[len(errs) // 2]


@pytest.mark.parametrize("arch_name,method,lora_r", [("tiny-128", "kadaptation", 4), ("tiny-128", "lora", 4), ("tiny-128", "compacter", 4),
                                                      ("ViT-B/32", "kadaptation", 4), ("ViT-B/32", "lora", 8),
                                                      ("ViT-B/32", "compacter", 4), ("ViT-B/16", "compacter", 4),
                                                      ("ViT-L/14", "kadaptation", 4)])
def test_whole_tower_as_close_to_emulation_as_f32(arch_name, method, lora_r):
    """Full depth (2, 12 and 24 blocks, N = 50 / 197 / 257), whole step: the engine's logits and its median gradient tensor
    are as far from the emulation as the f32 oracle's are, within 1.5x (measured: 0.55-0.9x for the attention-site methods,
    1.0-1.3x for Compacter; single small tensors -- a bias of 768 numbers -- scatter more than that on either side, hence
    the median).  See the module docstring for why the absolute numbers cannot be small here."""
    B = 4 if arch_name.startswith("tiny") else 8
    logit_err, loss_err, errs = _run(arch_name, method, lora_r, B, 10)
    f32_logits, f32_median = _f32_vs_emulation(arch_name, method, lora_r, B, 10)
    median = sorted(errs.values())[len(errs) // 2]
    assert logit_err <= 1.5 * f32_logits + 1e-3, (logit_err, f32_logits)
    assert median <= 1.5 * f32_median + 1e-3, (median, f32_median)
