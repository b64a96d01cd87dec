"""ViT-B/16 + Compacter at bs 8 against the f32 oracle with the fused and with the separate post-MLP kernels: the worst gradient
tensors relative to the bf16-operand noise (how the LayerNorm-affine gradients straddle the 2.5x gate, tests/test_gpu_tower.py)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_gpu_tower import _full_size_case, bf16_noise
from conftest import rel_err, max_rel
from pevit_amd.engine import HipEngine
from pevit_amd.synth import synth_batch
arch, sd = _full_size_case("ViT-B/16", "compacter", 4)
B, C = 8, 10
images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
g = torch.Generator().manual_seed(5); D = arch.embed_dim
head_w = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5; head_b = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
tr, ref_logits, ref_loss, logit_noise, noise = bf16_noise(sd, "compacter", C, images, labels, head_w, head_b)
for fused in (0, 1):
    eng = HipEngine(arch, "compacter", C, B); eng.load_state_dict(sd); eng.tune("adapter_fused", fused)
    v = eng.param_views()
    with torch.no_grad(): v["layers.0.weight"].copy_(head_w); v["layers.0.bias"].copy_(head_b)
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda()); torch.cuda.synchronize()
    gv = eng.grad_views()
    errs = {k: rel_err(gv[k].cpu(), tr.p[k].grad) for k in tr.names if tr.p[k].grad is not None}
    worst = sorted(errs.items(), key=lambda kv: -kv[1] / (2.5 * noise[kv[0]] + 1e-2))[:4]
    print("fused", fused, "logits", max_rel(logits.cpu(), ref_logits), [(k.split("resblocks.")[-1], round(e, 4), round(noise[k], 4)) for k, e in worst])
