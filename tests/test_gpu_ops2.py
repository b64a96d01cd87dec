"""Kernel-level parity, part 2: the kernels that round 1 only exercised inside whole-step tests -- the post-MLP adapter
kernels (adapter.hip), GEMM epilogues 9-12, the stem (im2col), the head (BatchNorm1d + Linear + cross-entropy, forward
and backward) and the fused SGD update -- each through the C ABI against plain PyTorch f32 on identical operands.

Tolerances as in test_gpu_ops.py: f32 results of bf16 MFMA contractions 2e-4 of the largest magnitude, bf16 outputs 1e-2,
pure f32 kernels 1e-5.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import max_rel, rel_err

pytestmark = pytest.mark.gpu

EPI = dict(BIAS_RESID_KEEP=9, BIAS_GELUNEW=10, DRELU=11, DGELUNEW=12)
ADAPTER, COMPACTER = 2, 3


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pevit_amd import _lib
    return _lib.load()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ok(lib, rc):
    assert rc == 0, lib.pevit_last_error().decode()


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


def test_gemm_epilogues_9_to_12(lib):
    M, N, K = 300, 256, 192
    A = rnd(M, K, seed=3, dtype=torch.bfloat16)
    B = rnd(N, K, seed=4, scale=0.08, dtype=torch.bfloat16)
    bias = rnd(N, seed=5, scale=0.1)
    resid = rnd(M, N, seed=6)
    acc = A.float() @ B.float().T

    def run(epi, **kw):
        ok(lib, lib.pevit_op_gemm(S(), epi, P(A), K, P(B), K, N, M, N, K, P(bias), P(kw.get("resid")), N, P(kw.get("outf")), N,
                                  P(kw.get("outb")), N, P(kw.get("outb2")), N, P(kw.get("aux")), N, 0, 0, 0, 0))
        torch.cuda.synchronize()
    # 9: out = acc + bias + resid ; out2 = acc + bias.  pevit_op_gemm has no out2_f32 slot: the engine path is checked in
    # test_gpu_tower; here the residual form through epilogue 1 must agree with 9's first output by construction.
    z16 = lambda: torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    a, g = z16(), z16()
    run(EPI["BIAS_GELUNEW"], outb=a, outb2=g)
    assert max_rel(a.float().cpu(), (acc + bias).cpu()) < 1e-2
    assert max_rel(g.float().cpu(), gelu_new(a.float()).cpu()) < 1e-2
    aux = rnd(M, N, seed=8, dtype=torch.bfloat16)
    o = z16()
    run(EPI["DRELU"], outb=o, aux=aux)
    assert max_rel(o.float().cpu(), (acc * (aux.float() > 0)).cpu()) < 1e-2
    o = z16()
    run(EPI["DGELUNEW"], outb=o, aux=aux)
    x = aux.float().clone().requires_grad_(True)
    gelu_new(x).backward(torch.ones_like(x))
    assert max_rel(o.float().cpu(), (acc * x.grad).cpu()) < 1e-2


@pytest.mark.parametrize("T,E", [(60, 128), (1000, 256), (6400, 768)])
def test_tn_gemm64_and_column_sums(lib, T, E):
    X = rnd(T, E, seed=1, dtype=torch.bfloat16)
    Y = rnd(T, 64, seed=2, dtype=torch.bfloat16)
    chunks = lib.pevit_op_tn_chunks(T)
    partial = torch.full((chunks, E, 64), float("nan"), device="cuda")
    csx = torch.zeros((chunks, E), device="cuda"); csy = torch.zeros((chunks, 64), device="cuda")
    ok(lib, lib.pevit_op_tn_gemm64(S(), P(X), E, P(Y), 64, P(partial), P(csx), P(csy), T, E))
    torch.cuda.synchronize()
    assert max_rel(partial.sum(0).cpu(), (X.float().T @ Y.float()).cpu()) < 2e-4
    assert max_rel(csx.sum(0).cpu(), X.float().sum(0).cpu()) < 1e-4
    assert max_rel(csy.sum(0).cpu(), Y.float().sum(0).cpu()) < 1e-4
    out = torch.ones(E * 64, device="cuda")
    ok(lib, lib.pevit_op_colsum_reduce(S(), P(partial), chunks, E * 64, P(out), None, None))
    torch.cuda.synchronize()
    assert max_rel((out - 1).cpu(), partial.sum(0).flatten().cpu()) < 1e-5         # accumulates, fixed order
    out2 = torch.ones(E * 64, device="cuda")
    ok(lib, lib.pevit_op_colsum_reduce(S(), P(partial), chunks, E * 64, P(out2), None, None))
    torch.cuda.synchronize()
    assert torch.equal(out, out2)                                                  # deterministic


@pytest.mark.parametrize("rows,E", [(37, 128), (1000, 256), (6400, 768)])
def test_ln_bwd_affine(lib, rows, E):
    x = rnd(rows, E, seed=1, scale=2.0) + 0.5
    g = 1 + rnd(E, seed=2, scale=0.2); b = rnd(E, seed=3, scale=0.2)
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y = F.layer_norm(xr, (E,), gr, br, 1e-5)
    dy = rnd(rows, E, seed=4); dres = rnd(rows, E, seed=5)
    y.backward(dy)
    mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    blocks = lib.pevit_op_lna_blocks(rows)
    partial = torch.full((blocks, 3, E), float("nan"), device="cuda")
    dx = torch.zeros((rows, E), device="cuda"); dxb = torch.zeros((rows, E), dtype=torch.bfloat16, device="cuda")
    ok(lib, lib.pevit_op_ln_bwd_affine(S(), P(dy), P(x), P(mean), P(rstd), P(g), P(dres), P(dx), P(dxb), P(partial), rows, E))
    torch.cuda.synchronize()
    assert max_rel(dx.cpu(), (xr.grad + dres).cpu()) < 2e-5
    assert max_rel(dxb.float().cpu(), dx.cpu()) < 1e-2
    dg, db, dr = (torch.zeros(E, device="cuda") for _ in range(3))
    ok(lib, lib.pevit_op_colsum_reduce(S(), P(partial), blocks, E, P(dg), P(db), P(dr)))
    torch.cuda.synchronize()
    assert max_rel(dg.cpu(), gr.grad.cpu()) < 1e-4
    assert max_rel(db.cpu(), br.grad.cpu()) < 1e-4
    assert max_rel(dr.cpu(), dres.sum(0).cpu()) < 1e-4


def _kron_sum(A, B):
    return torch.einsum("bac,bkp->bakcp", A, B).reshape(A.size(0), A.size(1) * B.size(1), A.size(2) * B.size(2)).sum(0)


@pytest.mark.parametrize("E", [128, 768])
def test_prep_and_chain_adapter(lib, E):
    wd_f = rnd(64, E, seed=1, scale=0.1); wu_f = rnd(E, 64, seed=2, scale=0.1)
    z = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device="cuda")
    wd, wdT, wu, wuT = z(64, E), z(E, 64), z(E, 64), z(64, E)
    ok(lib, lib.pevit_op_prep_bottleneck(S(), ADAPTER, None, P(wd_f), P(wu_f), None, None, P(wd), P(wdT), P(wu), P(wuT), E))
    torch.cuda.synchronize()
    assert torch.equal(wd, wd_f.to(torch.bfloat16)) and torch.equal(wdT, wd_f.T.contiguous().to(torch.bfloat16))
    assert torch.equal(wu, wu_f.to(torch.bfloat16)) and torch.equal(wuT, wu_f.T.contiguous().to(torch.bfloat16))
    Gd = rnd(E, 64, seed=3); Gu = rnd(E, 64, seed=4)          # dL/dH_down[e][j], dL/dW_up[e][j]
    grads = torch.ones(2 * 64 * E, device="cuda")
    ok(lib, lib.pevit_op_chain_bottleneck(S(), ADAPTER, P(Gd), P(Gu), None, None, P(grads), E, 0, 64 * E, 0, 0))
    torch.cuda.synchronize()
    assert torch.equal(grads[:64 * E].view(64, E) - 1, (Gd.T + 1) - 1)
    assert torch.equal(grads[64 * E:].view(E, 64) - 1, (Gu + 1) - 1)


@pytest.mark.parametrize("E", [128, 768])
def test_prep_and_chain_compacter(lib, E):
    """PHM expansion H = sum_i kron(rule_i, W_left_i W_right_i) (compacter_model.py:302-308) and its chain rule."""
    Fi = E // 4
    rule = (torch.rand((4, 4, 4), generator=torch.Generator().manual_seed(8)) * 2 - 1).cuda()
    dWl = rnd(4, Fi, 1, seed=1, scale=0.2); dWr = rnd(4, 1, 16, seed=2, scale=0.2)
    uWl = rnd(4, 16, 1, seed=3, scale=0.2); uWr = rnd(4, 1, Fi, seed=4, scale=0.2)
    leaves = [t.clone().requires_grad_(True) for t in (dWl, dWr, uWl, uWr)]
    Hd = _kron_sum(rule, torch.bmm(leaves[0], leaves[1]))      # (E, 64):  y = x @ Hd
    Hu = _kron_sum(rule, torch.bmm(leaves[2], leaves[3]))      # (64, E)
    z = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device="cuda")
    wd, wdT, wu, wuT = z(64, E), z(E, 64), z(E, 64), z(64, E)
    ok(lib, lib.pevit_op_prep_bottleneck(S(), COMPACTER, P(rule), P(dWl), P(dWr), P(uWl), P(uWr), P(wd), P(wdT), P(wu), P(wuT), E))
    torch.cuda.synchronize()
    assert max_rel(wdT.float().cpu(), Hd.detach().cpu()) < 1e-2 and max_rel(wd.float().cpu(), Hd.detach().T.cpu()) < 1e-2
    assert max_rel(wuT.float().cpu(), Hu.detach().cpu()) < 1e-2 and max_rel(wu.float().cpu(), Hu.detach().T.cpu()) < 1e-2
    Gd = rnd(E, 64, seed=5); Gu = rnd(E, 64, seed=6)           # Gd = dL/dHd [e][j] ; Gu[e][j] = dL/dHu[j][e]
    (Hd * Gd).sum().backward(retain_graph=True)
    (Hu * Gu.T).sum().backward()
    params = torch.cat([t.flatten() for t in (dWl, dWr, uWl, uWr)])
    offs = [0, 4 * Fi, 4 * Fi + 64, 4 * Fi + 128]
    grads = torch.zeros_like(params)
    ok(lib, lib.pevit_op_chain_bottleneck(S(), COMPACTER, P(Gd), P(Gu), P(rule), P(params), P(grads), E, *offs))
    torch.cuda.synchronize()
    sizes = [4 * Fi, 64, 64, 4 * Fi]
    for o, n, leaf in zip(offs, sizes, leaves):
        assert rel_err(grads[o:o + n].cpu(), leaf.grad.flatten().cpu()) < 1e-5


@pytest.mark.parametrize("B,R,Pp", [(3, 48, 16), (2, 224, 32), (2, 224, 14)])
def test_im2col(lib, B, R, Pp):
    img = rnd(B, 3, R, R, seed=1)
    K = 3 * Pp * Pp
    Kp = (K + 63) // 64 * 64
    G = R // Pp
    out = torch.full((B * G * G, Kp), float("nan"), dtype=torch.bfloat16, device="cuda")
    ok(lib, lib.pevit_op_im2col(S(), P(img), P(out), B, R, Pp, Kp))
    torch.cuda.synchronize()
    ref = F.unfold(img, kernel_size=Pp, stride=Pp).transpose(1, 2).reshape(B * G * G, K)    # k = c*P*P + py*P + px
    assert torch.equal(out[:, :K], ref.to(torch.bfloat16))
    assert float(out[:, K:].float().abs().sum()) == 0.0


def _engine(method="kadaptation", arch="tiny-128", B=16, C=10):
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import ARCHS
    return HipEngine(ARCHS[arch], method, C, B)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,C", [(16, 10), (128, 100), (7, 3)])
def test_head_batchnorm_linear_cross_entropy(training, B, C):
    """Classifier tail (kadaptation_clip.py:128-132,176-185) + CrossEntropyLoss, forward and backward, vs torch f32."""
    eng = _engine(B=B, C=C)
    D = eng.arch.embed_dim
    feat = rnd(B, D, seed=1, scale=1.5) + 0.3
    labels = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(2)).cuda()
    W = rnd(C, D, seed=3, scale=0.2); bias = rnd(C, seed=4, scale=0.2)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(W); v["layers.0.bias"].copy_(bias)
    rm0 = rnd(D, seed=5, scale=0.1); rv0 = rnd(D, seed=6, scale=0.1).abs() + 0.5
    eng.running_mean.copy_(rm0); eng.running_var.copy_(rv0)
    eng.zero_grad()
    logits, loss, dfeat = eng.head_forward_backward(feat, labels, bn_training=training)
    torch.cuda.synchronize()
    f = feat.clone().requires_grad_(True); Wr = W.clone().requires_grad_(True); br = bias.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    ref_logits = F.linear(F.batch_norm(f, rm, rv, None, None, training, 0.1, 1e-5), Wr, br)
    ref_loss = F.cross_entropy(ref_logits, labels)
    ref_loss.backward()
    assert max_rel(logits.cpu(), ref_logits.detach().cpu()) < 1e-5
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    assert max_rel(dfeat.cpu(), f.grad.cpu()) < 2e-5
    gv = eng.grad_views()
    assert max_rel(gv["layers.0.weight"].cpu(), Wr.grad.cpu()) < 2e-5
    assert max_rel(gv["layers.0.bias"].cpu(), br.grad.cpu()) < 2e-5
    assert max_rel(eng.running_mean.cpu(), rm.cpu()) < 1e-6 and max_rel(eng.running_var.cpu(), rv.cpu()) < 1e-6


@pytest.mark.parametrize("nesterov", [False, True])
@pytest.mark.parametrize("world", [1, 4])
def test_fused_sgd_matches_torch_optim(nesterov, world):
    """sgd_kernel (optim/build.py:120-127 -> torch.optim.SGD): momentum, weight decay, Nesterov, the first-step rule, the
    1/world gradient scale of data parallelism and the mask of parameters whose .grad is None (never touched, not even
    by weight decay)."""
    eng = _engine()
    n = eng.n_params
    p0 = rnd(n, seed=1)
    mask = eng.grad_mask.bool()
    assert int((~mask).sum()) > 0                                   # the dead v_proj_adapter1_* slots of KAdaptation
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.05, momentum=0.9, weight_decay=1e-2, nesterov=nesterov)
    eng.params.copy_(p0)
    for step in range(3):
        g = rnd(n, seed=10 + step)
        eng.grads.copy_(g * world)                                   # what the sum-all-reduce leaves behind
        eng.sgd_step(0.05, 0.9, 1e-2, 1.0 / world, nesterov)
        ref.grad = g.clone()
        opt.step()
    torch.cuda.synchronize()
    assert max_rel(eng.params[mask].cpu(), ref.detach()[mask].cpu()) < 1e-6
    assert torch.equal(eng.params[~mask], p0[~mask])
    assert float(eng.momentum[~mask].abs().max()) == 0.0


def test_a_failed_launch_is_reported_not_swallowed(lib):
    """A launch the runtime rejects (grid of 2^31 workgroups) must come back as a negative return code with a message
    -- it used to return 0 (round-1 advisor finding)."""
    x = torch.zeros(64, device="cuda")
    rc = lib.pevit_op_attn_fwd(S(), P(x), P(x), P(x), P(x), 64, P(x), 1 << 20, 1 << 11, 8)
    assert rc != 0
    assert b"launch failed" in lib.pevit_last_error()
    y = torch.zeros((4, 128), device="cuda"); yb = torch.zeros((4, 128), dtype=torch.bfloat16, device="cuda")
    g = torch.ones(128, device="cuda")
    ok(lib, lib.pevit_op_ln_fwd(S(), P(y), P(g), P(g), 4, 128, P(yb), None, None, None))      # the error does not stick
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,R,Pp", [(3, 48, 16), (2, 224, 32), (2, 224, 14)])
def test_im2col_from_uint8_pixels_equals_the_host_preprocessing(lib, B, R, Pp):
    """pevit_op_im2col_u8: ToTensor + Normalize of the reference's dataset transforms (feature.py:537-542, INPUT.MEAN / STD of
    vitb32_CLIP.yaml) inside the patch gather.  Bit for bit the patches of the f32 images `(x.float() / 255 - mean) / std`
    computed on the host, for the 8-pixel kernel (patch 16 / 32) and the generic one (patch 14)."""
    import ctypes as C
    mean, std = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    u8 = torch.randint(0, 256, (B, 3, R, R), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    host = (u8.float() / 255.0 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    K = 3 * Pp * Pp
    Kp = (K + 63) // 64 * 64
    G = R // Pp
    a = torch.full((B * G * G, Kp), float("nan"), dtype=torch.bfloat16, device="cuda"); b = a.clone()
    ok(lib, lib.pevit_op_im2col(S(), P(host.cuda().contiguous()), P(a), B, R, Pp, Kp))
    ok(lib, lib.pevit_op_im2col_u8(S(), P(u8.cuda()), (C.c_float * 3)(*mean), (C.c_float * 3)(*std), P(b), B, R, Pp, Kp))
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_a_step_on_uint8_pixels_equals_the_step_on_the_preprocessed_batch():
    """Engine level: train_step(uint8 images) after set_input_normalization == train_step(host-normalised f32 images), bit for
    bit (logits, loss, every gradient); uint8 without the constants is refused."""
    from pevit_amd._lib import PevitError
    from pevit_amd.synth import ARCHS, randomize_adapters, synth_state_dict
    from pevit_amd.engine import adapter_param_spec
    arch, C_ = ARCHS["tiny-128"], 10
    mean, std = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec("kadaptation", arch.width, arch.layers)]
    randomize_adapters(ad, seed=3); sd.update(dict(ad))
    u8 = torch.randint(0, 256, (6, 3, arch.resolution, arch.resolution), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
    host = (u8.float() / 255.0 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    labels = torch.arange(6) % C_
    res = []
    for images in (host, u8):
        eng = _engine(B=6, C=C_); eng.load_state_dict(sd)
        if images.dtype == torch.uint8:
            with pytest.raises(PevitError, match="uint8"):
                eng.forward_backward(images.cuda(), labels.cuda())
            eng.set_input_normalization(mean, std)
        logits, loss = eng.forward_backward(images.cuda().contiguous(), labels.cuda())
        torch.cuda.synchronize()
        res.append((logits.clone(), loss.clone(), eng.grads.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert float(res[0][2].abs().max()) > 0
