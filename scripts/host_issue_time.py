import sys, time, torch
sys.path.insert(0, '/root/repo')
from pevit_amd.engine import HipEngine
from pevit_amd.synth import ARCHS, reference_init_, synth_batch, synth_state_dict
arch = ARCHS["ViT-B/32"]
sd = synth_state_dict(arch, seed=2, text_tower=False)
for B in (128, 64, 16, 4):
    eng = HipEngine(arch, "kadaptation", 100, B); eng.load_state_dict(sd)
    reference_init_(eng.param_views().items(), "kadaptation")
    images, labels = synth_batch(B, 224, 100); images, labels = images.cuda(), labels.cuda()
    for _ in range(5): eng.train_step(images, labels, lr=0.01)
    torch.cuda.synchronize()
    n = 30; t0 = time.perf_counter()
    for _ in range(n): eng.train_step(images, labels, lr=0.01)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
    print(f"B={B}: host issue time {t_host*1e3:.3f} ms/step, wall {t_all*1e3:.3f} ms/step")
    del eng
