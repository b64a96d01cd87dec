#!/usr/bin/env python3
"""Fine-tune throughput of CLIP ViT-B/32 + KAdaptation on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run); a "step" is one
pass of the reference's train_one body (kadaptation_clip.py:347-353): zero_grad -> forward ->
cross-entropy -> backward -> [all-reduce of the flat adapter-gradient buffer over RCCL] -> SGD,
on a synthetic CIFAR100-shaped batch (B=128 per GPU, 3x224x224, C=100) that is resident in HBM
before the timed region.  Weak scaling: every rank processes its own 128-image shard.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- algorithmic FLOPs per image of one training step (MAC x 2) --------------
def train_gflop_per_image(E, L, P, R, D, C, rank, attention_site=True):
    N = (R // P) ** 2 + 1
    lin = 12 * E * E
    attn = 2 * N * E
    conv = (N - 1) * 3 * P * P * E
    tail = E * D + D * C
    tail_b = E * D + 2 * D * C
    if attention_site:
        a_f = 2 * (2 * E * rank)
        fwd = 2 * (N * L * (lin + attn + a_f) + conv + tail)
        bwd = 2 * (N * (L * lin - 3 * E * E) + 2 * N * L * attn + 2 * N * L * a_f + tail_b)
    else:
        a_f = 2 * E * 64
        fwd = 2 * (N * L * (lin + attn + a_f) + conv + tail)
        bwd = 2 * (N * (L - 1) * (lin + 2 * attn) + 2 * N * (L - 1) * a_f + N * a_f + tail_b)
    return (fwd + bwd) / 1e9


PEAK_TFLOPS_BF16 = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_TFLOPS_FP8 = 5000.0      # dense MX-fp8 MFMA (the forward frozen products of --weights fp8-act run on it)
PEAK_HBM_TBS = 8.0            # HBM3E, spec (6.3 TB/s is what a streaming copy reaches on this chip)
CPU_BASELINE_THREADS = 16     # best of the sweep 8/16/32/64/128 on the GPU box's host (profiles/r03_cpu_thread_sweep.md)
EPI_NAMES = {0: "qkv(+t) -> head layout", 1: "bias+residual f32", 2: "bias+QuickGELU", 3: "dQuickGELU", 4: "f32", 5: "bf16",
             6: "bias bf16", 7: "patch embed", 8: "bias+ReLU", 9: "bias+residual (keep h)", 10: "bias+gelu_new",
             11: "dReLU", 12: "dgelu_new"}


def cpu_baseline(seconds_budget=25.0):
    """The oracle (op-for-op restatement of the reference algorithm, dense-H einsum) timed on the
    host cores: BASELINE config 1 (ViT-B/32 + KAdaptation, fp32, bs=32), bounded sample."""
    from oracle import ref_cpu
    from pevit_amd.synth import ARCHS, synth_batch, synth_state_dict
    # Thread count: profiles/r03_cpu_thread_sweep.md (python bench.py --cpu-sweep on the GPU box's host) -- the eager fp32
    # step of the reference scales to a few dozen threads and then falls apart (one intra-op thread per logical core of a
    # 256-thread host: minutes per step); PEVIT_CPU_THREADS overrides.
    cores = int(os.environ.get("PEVIT_CPU_THREADS", "0")) or min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    arch = ARCHS["ViT-B/32"]
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    sd.update(ref_cpu.init_adapter_params("kadaptation", arch.width, arch.layers))
    tr = ref_cpu.OracleTrainer(sd, "kadaptation", 100, lr=0.01, wd=0.0)
    bs = 32
    images, labels = synth_batch(bs, 224, 100)
    t0 = time.time(); tr.step(images, labels); warm = time.time() - t0        # warm-up
    times = []
    t_end = time.time() + seconds_budget
    while len(times) < 10 and (time.time() < t_end or len(times) < 1):
        t0 = time.time(); tr.step(images, labels); times.append(time.time() - t0)
        if times[-1] > seconds_budget:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": bs / med, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{len(times)} fine-tune steps of ViT-B/32+KAdaptation fp32 bs={bs} (BASELINE config 1), "
                      f"median {med * 1e3:.0f} ms/step, torch {torch.__version__} CPU, {cores} of {os.cpu_count()} "
                      f"logical cores of {cpu_model()}"}


def harness_throughput(dev, steps=20):
    """images/s of the reference-API path: ``train_one`` (kadaptation_clip.py:321-361 -> pevit_amd/evaluation/_harness.py) over a
    synthetic TensorLoader of ``steps`` full batches, epoch-end read-back and train metric included, at bs 128 and bs 64, with
    the set resident in HBM as f32 and with uint8 pixels on the HOST (pinned staging, uploaded one batch ahead on a side stream,
    ToTensor + Normalize inside the engine) -- next to the bare engine step on a resident batch of the same size."""
    import dataclasses
    import tempfile
    from pevit_amd.config import vitb32_clip_config
    from pevit_amd.evaluation import _harness, kadaptation_clip as mod
    from pevit_amd.evaluation.dataloader import TensorLoader, _Tensors
    from pevit_amd.optim import build_optimizer
    from pevit_amd.synth import ARCHS, synth_state_dict
    arch = dataclasses.replace(ARCHS["ViT-B/32"], text_layers=1)          # the text tower is not on this path: keep the file small
    out = {"how": f"train_one over {steps} full batches of a synthetic TensorLoader (shuffle on, fused engine step, one read-back per "
                  "epoch, train accuracy computed), second epoch timed; engine_step = eng.train_step on one resident batch"}
    with tempfile.TemporaryDirectory() as tmp:
        ckpt = os.path.join(tmp, "vitb32_synth.pt")
        torch.save(synth_state_dict(arch, seed=2, text_tower=True), ckpt)
        cfg = vitb32_clip_config()
        cfg.MODEL.NAME = ckpt
        cfg.DATASET.NUM_CLASSES = 100
        cfg.TRAIN.LR, cfg.TRAIN.WD = 0.01, 1e-6
        cfg.TRAIN.BATCH_SIZE_PER_GPU = cfg.TEST.BATCH_SIZE_PER_GPU = 128
        cfg.GPUS = (dev.index or 0,)
        model = mod.Classifier(cfg, 0).cuda(dev)
        crit = torch.nn.CrossEntropyLoss()
        opt = build_optimizer(cfg, model)
        assert model.can_fuse(crit, opt)
        g = torch.Generator().manual_seed(0)
        for bs in (128, 64):
            n = steps * bs
            u8 = torch.randint(0, 256, (n, 3, 224, 224), dtype=torch.uint8, generator=g)
            labels = torch.randint(0, 100, (n,), generator=g)
            mean = torch.tensor(cfg.INPUT.MEAN, device=dev).view(1, 3, 1, 1); std = torch.tensor(cfg.INPUT.STD, device=dev).view(1, 3, 1, 1)
            f32_dev = (u8.to(dev).float() / 255.0 - mean) / std
            res = {}
            for name, imgs, lbl in (("resident_f32", f32_dev, labels.to(dev)), ("host_uint8_prefetched", u8, labels)):
                loader = TensorLoader(_Tensors(imgs, lbl), batch_size=bs, shuffle=True)
                mod.train_one(loader, model, crit, opt, 0, cfg)                       # warm-up epoch (staging buffers, workspace)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                mod.train_one(loader, model, crit, opt, 1, cfg)
                torch.cuda.synchronize(dev)
                res[name] = n / (time.perf_counter() - t0)
            eng = model.engine()
            xb, yb = f32_dev[:bs].contiguous(), labels[:bs].to(dev)
            for _ in range(5):
                eng.train_step(xb, yb, lr=0.01, momentum=0.9, weight_decay=1e-6)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2 * steps):
                eng.train_step(xb, yb, lr=0.01, momentum=0.9, weight_decay=1e-6)
            torch.cuda.synchronize(dev)
            res["engine_step_resident"] = 2 * steps * bs / (time.perf_counter() - t0)
            res["harness_over_engine"] = {k: res[k] / res["engine_step_resident"] for k in ("resident_f32", "host_uint8_prefetched")}
            out[f"bs{bs}"] = res
            del f32_dev, u8
        del model, opt
        _harness._BACKBONES.clear()
    return out


def concurrent_runs_throughput(dev, sd, arch, steps=40):
    """K independent fine-tune runs (K engine contexts: own parameters, workspace and optimizer state, the same frozen checkpoint)
    on K streams, stepped in lockstep by one host thread -- the reference's real workload is ~90 short sweep runs at batch 64 on one
    frozen backbone (kadaptation_clip.py:188-243,446-466; feature.py:101), and a batch-64 step fills 120 of the 256 CUs in its
    large GEMMs.  Aggregate images/s over the runs; every concurrent run's parameters are compared bit for bit with the same run
    stepped alone.  A side measurement: `value` is ONE run at batch 128."""
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import reference_init_, synth_batch
    out = {"how": f"K engine contexts on K streams, {steps} lockstep train_steps after 8 warm-up steps, lr differs per run; "
                  "aggregate images/s of the K runs; parameters of every run bit-identical to the same run alone"}
    for bs, ks in ((64, (1, 2, 3)), (128, (1, 2))):
        engines, batches, streams = [], [], []
        for r in range(max(ks)):
            e = HipEngine(arch, "kadaptation", 100, bs, device=dev)
            e.load_state_dict(sd)
            reference_init_(e.param_views().items(), "kadaptation", seed=7 + r)
            g = torch.Generator().manual_seed(5 + r)
            with torch.no_grad():
                v = e.param_views(); bound = arch.embed_dim ** -0.5
                v["layers.0.weight"].copy_(((torch.rand(v["layers.0.weight"].shape, generator=g) * 2 - 1) * bound).to(dev))
            im, lb = synth_batch(bs, arch.resolution, 100, seed_img=10 + 2 * r, seed_lbl=11 + 2 * r)
            engines.append(e); batches.append((im.to(dev), lb.to(dev))); streams.append(torch.cuda.Stream(dev))
        init = [e.params.clone() for e in engines]

        def reset():
            for e, p in zip(engines, init):
                e.reset_run(); e.params.copy_(p)
            torch.cuda.synchronize(dev)

        def run(k, n):
            for _ in range(n):
                for r in range(k):
                    with torch.cuda.stream(streams[r]):
                        engines[r].train_step(*batches[r], lr=0.01 * (r + 1), momentum=0.9, weight_decay=1e-6)
        solo = []
        for r in range(max(ks)):
            reset()
            with torch.cuda.stream(streams[r]):
                for _ in range(steps + 8):
                    engines[r].train_step(*batches[r], lr=0.01 * (r + 1), momentum=0.9, weight_decay=1e-6)
            torch.cuda.synchronize(dev)
            solo.append(engines[r].params.clone())
        res, same = {}, True
        for k in ks:
            reset(); run(k, 8); torch.cuda.synchronize(dev)
            t0 = time.perf_counter(); run(k, steps); torch.cuda.synchronize(dev)
            res[str(k)] = k * bs * steps / (time.perf_counter() - t0)
            same = same and all(torch.equal(engines[r].params, solo[r]) for r in range(k))
        for e in engines:
            e.check_streamk()
        out[f"bs{bs}"] = {"aggregate_images_per_sec_by_runs": res, "bit_identical_to_solo": same,
                          "gain_over_one_run": {k: v / res["1"] for k, v in res.items()}}
        del engines, batches, init, solo
        torch.cuda.empty_cache()
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def kernels_hash():
    """sha256 over the kernel sources the library is built from: a PMC pass is only valid for the kernels it profiled.
    (allreduce.hip -- the DP exchange -- and verify.hip -- the f32 verification mode -- hold no kernel of the profiled single-GPU
    bf16 step and are left out.)"""
    import hashlib
    d = os.path.join(ROOT, "pevit_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name in ("allreduce.hip", "verify.hip"):
            continue
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def measured_attainable_peak(dev):
    """What a vendor-library bf16 GEMM reaches on THIS device right now (4096^3, torch.matmul -> hipBLASLt): the practical
    ceiling next to the 2.5 PFLOP/s nominal peak.  Measurement only -- the product path never calls a BLAS library."""
    n = 4096
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        a @ b
    e1.record(); torch.cuda.synchronize(dev)
    return 10 * 2.0 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def pmc_traffic(arch, method, batch):
    """HBM bytes per GEMM launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE cannot share a pass and cannot be read from inside this process; scripts/pmc_traffic.py
    produced the file from this very command line).  None when no pass exists for this workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(f"{arch}|{method}|bs{batch}")
    except (OSError, ValueError):
        entry = None
    if not entry:
        return None, "no PMC pass committed for this workload", None
    if entry.get("kernels_hash") != kernels_hash():
        return None, ("the committed PMC pass (profiles/hbm_traffic.json, kernels %s) was taken with different kernel sources "
                      "than this build (%s): re-run scripts/run_pmc_passes.sh" % (entry.get("kernels_hash"), kernels_hash())), None
    return entry["gemm"]["hbm_bytes_per_launch"], entry["how"], entry.get("all_kernels", {}).get("hbm_bytes_per_step")


def pin_rank_to_cores(local_rank, local_world):
    """One contiguous slice of the host's cores per rank (so that N launch threads do not migrate over each other) and a bounded
    intra-op pool: the step is one C call per batch, its host side needs one core.  PEVIT_NO_PIN=1 leaves the affinity alone."""
    if os.environ.get("PEVIT_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // max(local_world, 1))
        mine = cores[local_rank * per:(local_rank + 1) * per] or cores
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(8, len(mine))))
        return mine
    except OSError:
        return None


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (BASELINE config 2: 128)")
    ap.add_argument("--method", default="kadaptation")
    ap.add_argument("--arch", default="ViT-B/32")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8", "fp8-act"],
                    help="frozen block weights: bf16, or e4m3 codes + per-channel scales (BASELINE config 5 with --arch ViT-L/14)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-exchange", choices=["auto", "single", "staged", "pipelined"], default="auto",
                    help="(N > 1) auto (default): time BOTH single and staged, W warm-up + K timed steps each, report the faster one "
                         "as `value` and both under config.exchange_schedules -- which of the two wins on N real GPUs has never been "
                         "measured; single: the fused forward/backward call, then ONE all-reduce of the flat gradient buffer (NOT "
                         "overlapped with the backward); staged: the backward in two halves with three buckets all-reduced while it "
                         "continues; pipelined: single with the exchange + SGD on a second stream under the next step's stem "
                         "(pevit_set_step_gate)")
    ap.add_argument("--dp-route", action="store_true",
                    help="(N = 1) also time the DP step on a 1-rank RCCL group in its three exchange schedules (single / pipelined / "
                         "staged) and report them beside the fused step (dp_route)")
    ap.add_argument("--no-harness", action="store_true", help="skip the reference-API (train_one) throughput measurement")
    ap.add_argument("--graph", action="store_true",
                    help="(N = 1, measurement) time the step as ONE captured HIP graph replay (engine.capture_train_step) instead of the "
                         "~200 launches of the C call; the line then carries step_launch = 'hip-graph'")
    ap.add_argument("--cpu-sweep", action="store_true", help="only time the CPU baseline at 8/16/32/64/128 threads and exit")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="library tuning knob for A/B runs, e.g. gemm_big=0 (see pevit_tune)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "flat"],
                    help="N > 1: gradient exchange through the process group's all-reduce (RCCL) or through pevit_allreduce_flat "
                         "(IPC-mapped peer mailboxes + copy engines + deterministic local reduction; csrc/allreduce.hip)")
    ap.add_argument("--dist-backend", default="nccl", help="(tests only) process-group backend; RCCL refuses two ranks on one device, "
                                                            "gloo carries device tensors")
    ap.add_argument("--share-device", action="store_true", help="(tests only) every rank uses cuda:0")
    ap.add_argument("--strict-traffic", action="store_true",
                    help="exit with status 3 when the committed PMC pass (profiles/hbm_traffic.json) was not taken with this "
                         "build's kernel sources, instead of printing the line with traffic null and traffic_stale true "
                         "(tests/test_zz_profiles_current.py enforces the same on the committed tree)")
    ap.add_argument("--pmc-calib", action="store_true",
                    help="(profiling runs only) first move a known byte count through HBM so that the FETCH_SIZE / "
                         "WRITE_SIZE counters of the same rocprofv3 pass can be calibrated (scripts/pmc_traffic.py)")
    args = ap.parse_args()

    if args.cpu_sweep:
        for n in (8, 16, 32, 64, 128):
            if n <= (os.cpu_count() or 1):
                os.environ["PEVIT_CPU_THREADS"] = str(n)
                r = cpu_baseline(seconds_budget=20.0)
                print(json.dumps({"threads": n, "images_per_sec": r["value"], "sample": r["sample"]}), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run (one rank per GPU of this node,
        # rendezvous on 127.0.0.1) instead of printing an error line where the driver expects a number
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import ARCHS, reference_init_, synth_batch, synth_state_dict
    arch = ARCHS[args.arch]
    classes = 100
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    if args.method == "compacter":      # frozen shared rule ~ U(-1,1) (compacter_model.py:511-519)
        sd["visual.transformer.phm_rule"] = torch.rand((4, 4, 4), generator=torch.Generator().manual_seed(4)) * 2 - 1
    eng = HipEngine(arch, args.method, classes, args.batch, lora_rank=8 if args.method == "lora" else 4, device=dev,
                    weight_format=args.weights)
    eng.load_state_dict(sd)
    for kv in args.tune:
        key, val = kv.split("=")
        if eng.tune(key, int(val)) < 0:
            raise SystemExit(f"unknown tuning key {key}")
    # adapters at the reference initialisation; head ~ nn.Linear default
    views = eng.param_views()
    reference_init_(views.items(), args.method)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        bound = arch.embed_dim ** -0.5
        views["layers.0.weight"].copy_(((torch.rand(views["layers.0.weight"].shape, generator=g) * 2 - 1) * bound).to(dev))
        views["layers.0.bias"].copy_(((torch.rand(views["layers.0.bias"].shape, generator=g) * 2 - 1) * bound).to(dev))
    images, labels = synth_batch(args.batch, arch.resolution, classes, seed_img=rank * 2, seed_lbl=rank * 2 + 1)
    images, labels = images.to(dev), labels.to(dev)

    if args.pmc_calib:
        import ctypes as C
        n = 256 << 20                                   # 1 GiB of f32: 4x the 256 MiB Infinity Cache
        src = torch.ones(n, dtype=torch.float32, device=dev)
        dst16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):                              # pevit cast kernel: reads 4n bytes (16 B/lane), writes 2n
            eng.lib.pevit_op_cast_bf16(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(src.data_ptr()),
                                       C.c_void_p(dst16.data_ptr()), n, 1.0)
        torch.cuda.synchronize()
        del src, dst16

    schedules = ["single", "staged"] if args.dp_exchange == "auto" else [args.dp_exchange]
    if world == 1:
        schedules = schedules[:1]
    eng.dp_exchange_mode = schedules[0]
    if world > 1 and args.exchange == "flat":
        eng.use_flat_allreduce()
    if world > 1:
        eng.sync_replicas()      # parameters, momentum and BatchNorm buffers of rank 0 on every rank, as a real run starts

    def step():
        return eng.train_step(images, labels, lr=0.01, momentum=0.9, weight_decay=1e-6, world_size=world)
    eager_step = step
    if args.graph and world == 1:
        eager_step()                                   # the first step (no momentum yet) cannot be the captured one
        step = eng.capture_train_step(images, labels, lr=0.01, momentum=0.9, weight_decay=1e-6)

    def timed_region(schedule):
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides; returns this rank's wall
        time, the median of its per-step event intervals, the last loss and (N > 1) the device time of every exchange."""
        if world > 1:
            eng.dp_pipeline_off()
            eng.dp_exchange_mode = schedule
        for _ in range(args.warmup):
            step()
        if world > 1:
            eng.dp_flush()
            _flush_c_stdio()                            # (the collective library's banner, printed when its communicator came up)
            torch.distributed.barrier()
            eng.time_exchange(True)
        torch.cuda.synchronize()
        # one event per step on the launch stream (the engine launches on torch's current stream): the median step time
        # next to the wall-clock mean
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            logits, loss = step()
            marks[i + 1].record()
        eng.dp_flush()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        v = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
        xus = sorted(eng.exchange_times_us()) if world > 1 else []
        eng.time_exchange(False)
        return dt, med, loss, (xus[len(xus) // 2] if xus else None)

    per_schedule = {}
    for sch in schedules:
        dt, median_ms, loss, x_us = timed_region(sch)
        print(f"[bench rank {rank}/{world}] {sch if world > 1 else 'step'}: {dt / args.steps * 1e3:.3f} ms/step (median {median_ms:.3f})"
              + (f", exchange {x_us:.1f} us" if x_us is not None else ""), file=sys.stderr, flush=True)   # stragglers show here
        rec = {"dt": dt, "median_ms": median_ms, "loss": float(loss)}
        if world > 1:
            # every rank's own clock: the whole-job time is the MAX (a straggler shows as one large entry), and the exchange time per
            # rank says whether a slow step is the collective or the rank's kernels
            mine = torch.tensor([dt, median_ms, -1.0 if x_us is None else x_us], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
            allr = torch.stack(allr).cpu()
            rec.update(dt=float(allr[:, 0].max()), median_ms=float(allr[:, 1].max()),
                       per_rank_ms_per_step=[float(x) / args.steps * 1e3 for x in allr[:, 0]],
                       per_rank_median_ms=[float(x) for x in allr[:, 1]],
                       per_rank_exchange_us=[None if float(x) < 0 else float(x) for x in allr[:, 2]])
            xs = [x for x in rec["per_rank_exchange_us"] if x is not None]
            rec["exchange_us_per_step"] = max(xs) if xs else None
        per_schedule[sch] = rec
    best = min(per_schedule, key=lambda k: per_schedule[k]["dt"])
    if world > 1:
        eng.dp_pipeline_off()
        eng.dp_exchange_mode = best
    dt, median_ms = per_schedule[best]["dt"], per_schedule[best]["median_ms"]
    args.dp_exchange = best
    final_loss = per_schedule[best]["loss"]
    eng.check_streamk()          # a stream-K hand-off that timed out would make the timed steps invalid: fail loudly

    # ---- dominant kernel (MFMA GEMM family): HIP events around every GEMM launch, measured over
    # extra steps right after the timed region so that event recording does not perturb `value`.
    prof_steps = max(1, min(args.steps, 10))
    step = eager_step                                  # the per-launch event passes below bracket individual launches: eager
    gemm_ms, gemm_flops, gemm_launches = eng.profile_gemms(lambda: [step() for _ in range(prof_steps)])
    gemm_by_shape = dict(eng.last_profile_by_shape)
    algo_bytes_total = eng.last_profile_bytes
    # ... and a second pass that also brackets the HBM-bound kernels (their own pass: more events between the kernels)
    eng.profile_gemms(lambda: [step() for _ in range(prof_steps)], all_kernels=True)
    hbm_kernels = {k: {"launches_per_step": c / prof_steps, "avg_us": m * 1e3 / max(c, 1), "algorithmic_bytes_per_launch": b / max(c, 1),
                       "achieved_TBps": b / max(m * 1e-3, 1e-12) / 1e12, "frac_of_hbm_peak": b / max(m * 1e-3, 1e-12) / 1e12 / PEAK_HBM_TBS}
                   for k, (c, m, b) in sorted(eng.last_profile_hbm.items(), key=lambda kv: -kv[1][1])}

    dp_route = None
    if args.dp_route and world == 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

        def dp_single():
            eng.forward_backward_dp(images, labels, mode="single")
            eng.sgd_step(0.01, 0.9, 1e-6, 1.0)

        def dp_staged():
            eng.forward_backward_dp(images, labels, mode="staged")
            eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
        def dp_pipelined():
            eng._train_step_pipelined(images, labels, 0.01, 0.9, 1e-6, True, None, 1, False, None, None)
        res = {}
        # (the staged route switches stream-K off; "fused_streamk_off" is the fused step in that state)
        for name, fn in (("dp_single", dp_single), ("dp_pipelined", dp_pipelined), ("dp_staged", dp_staged), ("fused_streamk_off", step)):
            if name == "dp_staged":
                eng.dp_pipeline_off()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            mk = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            mk[0].record()
            for i in range(args.steps):
                fn(); mk[i + 1].record()
            torch.cuda.synchronize()
            v = sorted(mk[i].elapsed_time(mk[i + 1]) for i in range(args.steps))
            res[name] = v[len(v) // 2]
        torch.distributed.destroy_process_group()
        dp_route = {"median_ms_per_step": res["dp_" + args.dp_exchange], "single_exchange_median_ms": res["dp_single"],
                    "staged_exchange_median_ms": res["dp_staged"], "pipelined_exchange_median_ms": res["dp_pipelined"],
                    "pipelined_over_fused": res["dp_pipelined"] / median_ms, "fused_step_streamk_off_median_ms": res["fused_streamk_off"],
                    "fused_step_median_ms": median_ms, "dp_route_over_fused": res["dp_" + args.dp_exchange] / median_ms,
                    "single_over_fused": res["dp_single"] / median_ms, "staged_over_fused": res["dp_staged"] / median_ms,
                    "how": "engine.forward_backward_dp + SGD on a 1-rank RCCL process group: single = the fused call + one all-reduce of "
                           "the flat gradient buffer; pipelined = single with the exchange + SGD on a second stream under the next "
                           "step's stem; staged = backward in two halves, stream-K off, head / upper-half / lower-half "
                           "buckets all-reduced asynchronously.  The host-side and launch-structure cost of the DP path; N > 1 "
                           "remains unmeasured"}
        eng.tune("gemm_streamk", 1); eng._dp_streamk_off = False
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        site = args.method in ("kadaptation", "lora")
        r = 32 if args.method == "kadaptation" else 8
        gflop = train_gflop_per_image(arch.width, arch.layers, arch.patch, arch.resolution, arch.embed_dim, classes, r, site)
        step_tflops = value / world * gflop / 1e3      # whole-step algorithmic TFLOP/s per GPU
        gemm_tflops = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        algo_bytes = algo_bytes_total / max(gemm_launches, 1)
        traffic, traffic_how, traffic_all = pmc_traffic(args.arch, args.method, args.batch) if args.weights == "bf16" else (None, "no PMC pass for fp8 weights", None)
        traffic_stale = traffic is None and "different kernel sources" in traffic_how
        if traffic_stale:
            print("[bench] " + traffic_how, file=sys.stderr, flush=True)
            if args.strict_traffic:
                raise SystemExit(3)
        attainable = measured_attainable_peak(dev)
        # the matrix peak the family is priced against: bf16 MFMA, except for the fp8 x fp8 forward path
        peak = PEAK_TFLOPS_FP8 if args.weights == "fp8-act" else PEAK_TFLOPS_BF16
        # the GEMM launches of a step by (epilogue, M, N, K), largest share of the time first: which product is furthest
        # from the peak
        per_kernel = []
        ridge = peak * 1e12 / (PEAK_HBM_TBS * 1e12)            # flop per byte where the two roofs meet (312 for bf16)
        # both floors of the WHOLE step: algorithmic flops at the matrix peak, algorithmic bytes of every launch at 8 TB/s
        hbm_bytes_step = algo_bytes_total / prof_steps + sum(v["launches_per_step"] * v["algorithmic_bytes_per_launch"] for v in hbm_kernels.values())
        mfma_floor_ms = gflop * args.batch / peak               # GFLOP per step / (TFLOP/s) = ms
        hbm_floor_ms = hbm_bytes_step / (PEAK_HBM_TBS * 1e12) * 1e3
        for v in hbm_kernels.values():
            v["bound"] = "hbm"
        for (epi, M, N, K), (cnt, ms_k, fl_k, by_k) in sorted(gemm_by_shape.items(), key=lambda kv: -kv[1][1])[:8]:
            sec = max(ms_k, 1e-9) * 1e-3
            per_kernel.append({"epilogue": EPI_NAMES.get(epi, str(epi)), "M": M, "N": N, "K": K,
                               "launches_per_step": cnt / prof_steps, "avg_us": ms_k * 1e3 / cnt,
                               "share_of_gemm_time": ms_k / max(gemm_ms, 1e-9), "tflops": fl_k / sec / 1e12,
                               "frac": fl_k / sec / 1e12 / peak, "algorithmic_TBps": by_k / sec / 1e12,
                               # which roof is lower for THIS product: flop per algorithmic byte against the ridge peak / 8 TB/s
                               "flop_per_byte": fl_k / max(by_k, 1.0), "bound": "mfma" if fl_k / max(by_k, 1.0) >= ridge else "hbm",
                               "frac_of_hbm_peak": by_k / sec / 1e12 / PEAK_HBM_TBS})
        if os.environ.get("PEVIT_BENCH_ALL_SHAPES"):      # measurement: every GEMM shape of the step on stderr
            for (epi, M, N, K), (cnt, ms_k, fl_k, by_k) in sorted(gemm_by_shape.items(), key=lambda kv: -kv[1][1]):
                print(f"[shape] epi={EPI_NAMES.get(epi, epi)} M={M} N={N} K={K} n/step={cnt / prof_steps:.1f} avg_us={ms_k * 1e3 / cnt:.1f}",
                      file=sys.stderr)
        headline = (args.arch, args.method, args.batch, args.weights) == ("ViT-B/32", "kadaptation", 128, "bf16")
        metric = "images/sec fine-tune, CLIP ViT-B/32 + KAdaptation, bs=128, 1/2/4/8 GPU" if headline else \
            f"images/sec fine-tune, CLIP {args.arch} + {args.method}, bs={args.batch}/GPU, {args.weights} weights"
        out = {
            "metric": metric,
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "median_ms_per_step": median_ms, "value_at_median": args.batch * world / (median_ms * 1e-3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic type of the path: bf16 operands on the bf16 MFMA, f32 accumulate; with --weights fp8 the frozen
            # weights are e4m3 codes converted to bf16 in the GEMM (the activation side stays bf16)
            "dtype": {"bf16": "bf16", "fp8": "bf16 x fp8-e4m3 weights", "fp8-act": "fp8-e4m3 x fp8-e4m3 (forward frozen products), bf16 x fp8 (backward)"}[args.weights], "data": "synthetic",
            "config": {"workload": f"CLIP {args.arch} + {args.method} fine-tune step (fwd+CE+bwd+SGD), "
                                   f"{args.batch} images/GPU 3x{arch.resolution}x{arch.resolution}, C={classes}, "
                                   f"synthetic OpenAI-layout checkpoint, adapters at reference init, frozen block "
                                   f"weights {args.weights}" + (" (e4m3 codes + per-channel scales, bf16 activations, f32 accumulate)"
                                                                 if args.weights == "fp8" else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "step_launch": "hip-graph replay" if (args.graph and world == 1) else "eager (one C call, ~200 kernel launches)",
                       "gradient_exchange": ("none" if world == 1 else args.exchange), "rccl_ranks": world if (world > 1 and args.dist_backend == "nccl") else 0,
                       "gradient_buckets": 0 if world == 1 else (3 if args.dp_exchange == "staged" else 1),
                       "exchange_us_per_step": per_schedule[best].get("exchange_us_per_step"),
                       "per_rank_ms_per_step": per_schedule[best].get("per_rank_ms_per_step"),
                       "exchange_schedules": None if world == 1 else {
                           k: {"images_per_sec": args.batch * world * args.steps / v["dt"], "ms_per_step": v["dt"] / args.steps * 1e3,
                               "median_ms_per_step": v["median_ms"], "exchange_us_per_step": v.get("exchange_us_per_step"),
                               "per_rank_ms_per_step": v.get("per_rank_ms_per_step"), "per_rank_exchange_us": v.get("per_rank_exchange_us"),
                               "overlapped_with_backward": k == "staged"}
                           for k, v in per_schedule.items()},
                       "exchange_schedule_chosen": None if world == 1 else ("the faster of " + " / ".join(per_schedule) + f" in THIS run: {best}" if len(per_schedule) > 1 else best),
                       "exchange_us_how": None if world == 1 else "median over the timed steps of an event pair on the stream the exchange is enqueued on, max over ranks (single: around the in-stream all-reduce; staged: around the waits behind the backward = what the overlap did not hide; pipelined: on the second stream)",
                       "gradient_exchange_schedule": "none" if world == 1 else {"single": "one all-reduce of the flat buffer behind the fused forward/backward call", "pipelined": "one all-reduce of the flat buffer + SGD on a second stream under the next step's stem", "staged": "three buckets overlapped with the staged backward"}[args.dp_exchange], "exchanged_floats_per_step": 0 if world == 1 else int(eng.n_params),
                       "train_gflop_per_image": gflop, "final_loss": final_loss},
            "roofline": {"bound": "mfma", "kernel": "gemm8_kernel<...> + gemm_kphase_kernel<...> + gemm_kernel<...> + gemm_streamk_kernel<...> (pevit_amd/csrc/gemm.hip: all epilogues / tile shapes)",
                         "achieved": gemm_tflops, "peak": peak, "unit": "TFLOP/s",
                         "frac": gemm_tflops / peak, "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_stale": traffic_stale,
                         # the whole step against the same peak: images/s x algorithmic GFLOP/image, and the matrix-core work
                         # actually executed (class-token pruning of the last block) over the step time
                         "whole_step_frac": step_tflops / peak,
                         # round 5: BOTH floors of the step.  Its arithmetic intensity (flop per algorithmic byte of every launch)
                         # lies below the ridge, so the step as a whole is HBM-side; `bound` above names the roof of the dominant
                         # KERNEL FAMILY (the GEMMs, each well above the ridge), per_kernel[*].bound that of every product
                         "mfma_floor_ms": mfma_floor_ms, "hbm_floor_ms": hbm_floor_ms,
                         "algorithmic_bytes_per_step": hbm_bytes_step,
                         "whole_step_flop_per_byte": gflop * args.batch * 1e9 / max(hbm_bytes_step, 1.0), "ridge_flop_per_byte": ridge,
                         "whole_step_bound": "hbm" if gflop * args.batch * 1e9 / max(hbm_bytes_step, 1.0) < ridge else "mfma",
                         "whole_step_hbm_frac": hbm_bytes_step / (ms * 1e-3) / (PEAK_HBM_TBS * 1e12),
                         # HBM-side bytes of EVERY dispatch of a step from the same two PMC passes as `traffic` (per step)
                         "traffic_all": traffic_all, "traffic_all_unit": "bytes/step",
                         "traffic_all_over_algorithmic": (traffic_all / hbm_bytes_step) if traffic_all else None,
                         "executed_gemm_frac_of_peak": gemm_flops / prof_steps / 1e12 / (ms * 1e-3) / peak,
                         "non_gemm_ms_per_step": ms - gemm_ms / prof_steps,
                         "hbm_kernels": hbm_kernels,
                         "peak_attainable_measured": {"value": attainable, "unit": "TFLOP/s",
                                                      "how": "vendor-library bf16 GEMM 4096^3 (torch.matmul), measured in this run "
                                                             "after the timed region; not used by the product path"},
                         "kernels_hash": kernels_hash(),
                         "traffic_how": traffic_how, "algorithmic_bytes_per_launch": algo_bytes,
                         "launches_per_step": gemm_launches / prof_steps,
                         "per_kernel": per_kernel,
                         "avg_launch_us": gemm_ms * 1e3 / max(gemm_launches, 1),
                         "gemm_ms_per_step": gemm_ms / prof_steps,
                         "flops_per_launch": gemm_flops / max(gemm_launches, 1),
                         "how": "2*M*N*K of every GEMM launch / HIP-event duration of that launch (events on the "
                                "launch stream), summed over %d steps" % prof_steps,
                         "whole_step": {"achieved": step_tflops, "frac": step_tflops / peak,
                                        "how": "images/s x algorithmic GFLOP/image (SURVEY 8d) / peak",
                                        # the last block runs its post-attention products on the class-token rows only
                                        # (identical results): the matrix-core work actually executed per step is lower
                                        "executed_gemm_tflop_per_step": gemm_flops / prof_steps / 1e12,
                                        "executed_gemm_frac_of_peak": gemm_flops / prof_steps / 1e12 / (ms * 1e-3) / peak}},
        }
        if dp_route is not None:
            out["dp_route"] = dp_route
        if world == 1 and not args.no_harness and headline:
            del eng, images, labels
            torch.cuda.empty_cache()
            try:
                out["concurrent_runs"] = concurrent_runs_throughput(dev, sd, arch)
            except Exception as e:           # a side measurement must never cost the headline line
                out["concurrent_runs"] = {"error": f"{type(e).__name__}: {e}"}
                print(f"[bench] concurrent-runs measurement failed: {e}", file=sys.stderr, flush=True)
            try:
                out["harness_images_per_sec"] = harness_throughput(dev)
            except Exception as e:           # a side measurement must never cost the headline line
                out["harness_images_per_sec"] = {"error": f"{type(e).__name__}: {e}"}
                print(f"[bench] harness measurement failed: {e}", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}", "kind": "port"}
                print(f"[bench] CPU baseline failed: {e}", file=sys.stderr, flush=True)
    # The JSON line is the LAST thing on stdout: the collective library writes a version banner through C stdio, which sits in the
    # process's buffer until exit when stdout is a pipe -- tear the process group down first, flush C stdio, then print.
    _flush_c_stdio()
    if world > 1:
        torch.distributed.barrier()                     # every rank has emptied its C stdio buffer
        torch.distributed.destroy_process_group()
        _flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
