import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import max_rel, rel_err
from oracle import ref_cpu
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict

def run(method, arch_name, B, seed):
    arch = ARCHS[arch_name]
    sd = {k: v for k, v in synth_state_dict(arch, seed=seed, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=6)
    for n, v in ad:
        if n.endswith("phm_rule"):
            v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
    sd.update(dict(ad))
    images, labels = synth_batch(B, arch.resolution, 10, seed_img=3, seed_lbl=4)
    tr = ref_cpu.OracleTrainer(sd, method, 10)
    tr.loss_and_grads(images, labels)
    tre = ref_cpu.OracleTrainer(sd, method, 10)
    with ref_cpu.operand_rounding(torch.bfloat16):
        tre.loss_and_grads(images, labels)
    eng = HipEngine(arch, method, 10, B)
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(tr.head_w.detach()); v["layers.0.bias"].copy_(tr.head_b.detach())
    eng.forward_backward(images.cuda(), labels.cuda()); torch.cuda.synchronize()
    gv = eng.grad_views()
    print(f"== {method} {arch_name} B={B} seed={seed}")
    for n in tr.names:
        if tr.p[n].grad is None: continue
        print(f"  {n[len('visual.transformer.resblocks.'):]:45s} hip {rel_err(gv[n].cpu(), tr.p[n].grad):.4f} emul {rel_err(tre.p[n].grad, tr.p[n].grad):.4f} |g| {float(tr.p[n].grad.norm()):.3e}")

run("adapter", "tiny-n257", 8, 21)
run("adapter", "tiny-n257", 8, 33)
run("adapter", "tiny-n197", 8, 21)
run("adapter", "tiny-256", 64, 21)
