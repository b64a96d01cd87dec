import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import max_rel, rel_err
from test_gpu_tower import bf16_noise
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import VitArch, randomize_adapters, synth_batch, synth_state_dict
for layers in (2, 6, 12, 24):
    arch = VitArch(f"L14-{layers}", 1024, layers, 14, 224, 768)
    method = "kadaptation"
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=3); sd.update(dict(ad))
    B, C = 8, 10
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5); D = arch.embed_dim
    head_w = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5; head_b = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
    tr, ref_logits, ref_loss, logit_noise, noise = bf16_noise(sd, method, C, images, labels, head_w, head_b)
    eng = HipEngine(arch, method, C, B); eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(head_w); v["layers.0.bias"].copy_(head_b)
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda()); torch.cuda.synchronize()
    gv = eng.grad_views()
    worst = max((rel_err(gv[k].cpu(), tr.p[k].grad), k) for k in tr.names if tr.p[k].grad is not None)
    print(f"layers {layers}: logits HIP {max_rel(logits.cpu(), ref_logits):.4f} vs emulated {logit_noise:.4f}; worst grad HIP {worst[0]:.4f} vs emulated {max(noise.values()):.4f}", flush=True)
    del eng
