"""CPU checks of the host mirror of the reference's Python API: module / parameter names,
trainability rules, counts, loader errors, text tower and zero-shot head, optimizer groups, sweep logic.
Expected values come from the fixtures written by tests/golden/make_golden.py (the imported reference)."""
import hashlib
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, load_tiny_sd

from pevit_amd import _lib
from pevit_amd.config import CfgNode, default_config
from pevit_amd.evaluation import _harness, clip_load, metric
from pevit_amd.evaluation.feature import extract_text_features
from pevit_amd.evaluation.model import build_model, build_peft_model
from pevit_amd.optim import build_optimizer

METHODS = ["kadaptation", "lora", "adapter", "compacter"]
HARNESS = {"kadaptation": "kadaptation_clip", "lora": "lora_clip", "adapter": "adapter_tuning_clip",
           "compacter": "compacter_clip"}


def tiny_config(path, classes=10):
    c = default_config()
    c.MODEL.NAME = str(path)
    c.MODEL.SPEC.EMBED_DIM = 64
    c.MODEL.SPEC.TEXT.CONTEXT_LENGTH = 8
    c.DATASET.NUM_CLASSES = classes
    c.TRAIN.NESTEROV = False
    c.TRAIN.BATCH_SIZE_PER_GPU = 4
    c.TEST.BATCH_SIZE_PER_GPU = 4
    return c


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    p = tmp_path_factory.mktemp("ckpt") / "tiny.pt"
    torch.save(load_tiny_sd(), p)
    return p


@pytest.mark.parametrize("fixture", ["tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter"])
def test_named_parameters_match_reference(fixture):
    meta, _ = load_golden(fixture)
    model = build_peft_model(load_tiny_sd(), meta["method"], meta["lora_r"])
    assert [n for n, _ in model.named_parameters()] == meta["all_names"]
    assert not model.training                                     # build_model(...).eval(), model.py:1250
    assert all(p.dtype == torch.float32 for p in model.parameters())
    assert sum(p.numel() for p in model.parameters()) == meta["n_backbone_params"]
    assert sum(p.numel() for p in model.visual.parameters()) == meta["n_visual_params"]
    assert model.visual.input_resolution == 48 and model.visual.proj.shape == (128, 64)


@pytest.mark.parametrize("method", METHODS)
def test_classifier_trainability_rule(method, ckpt):
    import importlib
    mod = importlib.import_module("pevit_amd.evaluation." + HARNESS[method])
    meta, _ = load_golden("tiny_" + method)
    clf = mod.Classifier(tiny_config(ckpt), 0)
    names = [n for n, p in clf.named_parameters() if p.requires_grad]
    assert names == ["backbone." + n for n in meta["trainable_names"]] + ["layers.0.weight", "layers.0.bias"]
    assert sum(p.numel() for p in clf.parameters() if p.requires_grad) == meta["n_trainable_params"]
    assert isinstance(clf.channel_bn, torch.nn.BatchNorm1d) and not clf.channel_bn.affine
    assert float(clf.logit_scale) == 0.0 and not clf.logit_scale.requires_grad     # LOGIT_SCALE_INIT 'none'


@pytest.mark.parametrize("method", METHODS)
def test_adapter_initialisation_matches_reference_structure(method):
    """Same zero / one / random pattern as the reference's initialisers (values are RNG-dependent)."""
    meta, t = load_golden("tiny_" + method)
    model = build_peft_model(load_tiny_sd(), method, meta["lora_r"])
    named = dict(model.named_parameters())
    for k, ref in t.items():
        if not k.startswith("init/"):
            continue
        p = named[k[5:]].detach()
        assert tuple(p.shape) == tuple(ref.shape), k
        assert bool((ref == 0).all()) == bool((p == 0).all()), k
        if bool((ref == 1).all()):
            assert bool((p == 1).all()), k
        if ref.abs().max() > 0:
            assert 0.3 < float(p.abs().max() / ref.abs().max()) < 3.0, k       # same scale of the distribution


def test_checkpoint_adapter_keys_overlay_initial_values():
    meta, t = load_golden("tiny_lora")
    sd = load_tiny_sd()
    key = "visual.transformer.resblocks.1.attn.q_proj_adapter2.weight"
    sd[key] = t["adapter/" + key].clone()
    sd["input_resolution"] = torch.tensor(48)                     # deleted like model.py:1241-1243
    model = build_peft_model(sd, "lora")
    assert torch.equal(dict(model.named_parameters())[key], t["adapter/" + key])


def test_vision_tower_has_no_cpu_fallback():
    model = build_model(load_tiny_sd())
    with pytest.raises(_lib.PevitError):
        model.encode_image(torch.zeros(2, 3, 48, 48))


def test_resnet_checkpoints_are_rejected():
    sd = {k: v for k, v in load_tiny_sd().items() if k != "visual.proj"}
    with pytest.raises(RuntimeError):
        build_model(sd)


def test_text_tower_and_zero_shot_head_match_reference():
    z = np.load(os.path.join(GOLDEN, "tiny_text.npz"))
    tokens = torch.from_numpy(z["tokens"])
    model = build_model(load_tiny_sd())
    with torch.no_grad():
        feats = torch.stack([model.encode_text(tokens[c]) for c in range(tokens.shape[0])])
    assert torch.allclose(feats, torch.from_numpy(z["text_features"]), atol=1e-6, rtol=1e-6)
    cfg = default_config()
    cfg.MODEL.SPEC.TEXT.CONTEXT_LENGTH = 8
    w = extract_text_features(cfg, None, model=model, return_numpy=True, class_names=[tokens[c] for c in range(10)])
    assert w.shape == (64, 10)
    np.testing.assert_allclose(w, z["zeroshot_weights"], atol=1e-6, rtol=1e-6)


def test_init_head_with_text_encoder(ckpt):
    from pevit_amd.evaluation.kadaptation_clip import Classifier
    z = np.load(os.path.join(GOLDEN, "tiny_text.npz"))
    cfg = tiny_config(ckpt)
    cfg.TRAIN.INIT_HEAD_WITH_TEXT_ENCODER = True
    cfg.DATASET.CLASS_NAMES = [torch.from_numpy(z["tokens"][c]) for c in range(10)]
    clf = Classifier(cfg, 0)
    np.testing.assert_allclose(clf.layers[0].weight.detach().numpy(), z["zeroshot_weights"].T, atol=1e-6)
    assert float(clf.layers[0].bias.abs().max()) == 0.0


# ---------------------------------------------------------------------------------- clip_load
def test_load_errors_like_reference(tmp_path, ckpt):
    with pytest.raises(RuntimeError, match="not found; available models"):
        clip_load.load("no-such-model", device="cpu")
    with pytest.raises(RuntimeError):                              # known name, nothing on disk, no network
        clip_load.load("ViT-B/32", device="cpu", download_root=str(tmp_path))
    (tmp_path / "ViT-B-32.pt").write_bytes(b"not the release")
    with pytest.raises(RuntimeError, match="SHA256"):
        clip_load.load("ViT-B/32", device="cpu", download_root=str(tmp_path))
    with pytest.raises(RuntimeError):
        clip_load.load(str(ckpt), device="cpu", jit=True)
    assert clip_load.available_models()[-3:] == ["ViT-B/32", "ViT-B/16", "ViT-L/14"]


def test_load_by_path_and_by_verified_name(tmp_path, ckpt, monkeypatch):
    model, preprocess = clip_load.lora_load(str(ckpt), device="cpu")
    assert any("q_proj_adapter2.weight" in n for n, _ in model.named_parameters())
    assert preprocess.n_px == model.visual.input_resolution == 48
    data = open(ckpt, "rb").read()
    monkeypatch.setitem(clip_load._MODELS, "tiny", ("tiny.pt", hashlib.sha256(data).hexdigest()))
    (tmp_path / "tiny.pt").write_bytes(data)
    model, _ = clip_load.load("tiny", device="cpu", download_root=str(tmp_path))
    assert "visual.transformer.phm_rule1_left" in dict(model.named_parameters())


def test_preprocess_matches_clip_transform():
    from PIL import Image
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 255, (60, 90, 3), dtype=np.uint8))
    x = clip_load._transform(48)(img)
    assert x.shape == (3, 48, 48) and x.dtype == torch.float32
    ref = img.resize((72, 48), Image.BICUBIC).crop((12, 0, 60, 48))
    r = torch.from_numpy(np.asarray(ref)).permute(2, 0, 1).float() / 255
    r = (r - torch.tensor(clip_load._Preprocess.MEAN).view(3, 1, 1)) / torch.tensor(clip_load._Preprocess.STD).view(3, 1, 1)
    assert torch.allclose(x, r)


# ---------------------------------------------------------------------------------- optimiser / schedule
def test_build_optimizer_groups_and_fusability(ckpt):
    from pevit_amd.evaluation.kadaptation_clip import Classifier, adjust_learning_rate
    cfg = tiny_config(ckpt)
    cfg.TRAIN.LR, cfg.TRAIN.WD, cfg.TRAIN.SCHEDULE = 0.1, 1e-3, [2, 4]
    clf = Classifier(cfg, 0)
    opt = build_optimizer(cfg, clf)
    assert type(opt) is torch.optim.SGD and len(opt.param_groups) == 2
    assert len(opt.param_groups[1]["params"]) == 0 and opt.param_groups[1]["weight_decay"] == 0.0
    assert len(opt.param_groups[0]["params"]) == sum(1 for p in clf.parameters() if p.requires_grad)
    crit = torch.nn.CrossEntropyLoss()
    assert clf.can_fuse(crit, opt)
    for epoch, lr in [(0, 0.1), (2, 0.01), (5, 0.001)]:
        adjust_learning_rate(opt, epoch, cfg)
        assert all(abs(g["lr"] - lr) < 1e-12 for g in opt.param_groups)
    cfg.TRAIN.NESTEROV = True                                       # the fused kernel implements Nesterov momentum too
    assert clf.can_fuse(crit, build_optimizer(cfg, clf))
    cfg.TRAIN.NESTEROV = False
    # anything the fused kernel does not implement falls back to the optimizer's own step()
    cfg.TRAIN.WITHOUT_WD_LIST = ["bias"]
    opt = build_optimizer(cfg, clf)
    assert [n for n, p in clf.named_parameters() if any(p is q for q in opt.param_groups[1]["params"])] == \
        [n for n, p in clf.named_parameters() if p.requires_grad and n.endswith(".bias")]
    assert not clf.can_fuse(crit, opt)                             # two different weight decays
    cfg.TRAIN.WITHOUT_WD_LIST = []
    cfg.TRAIN.OPTIMIZER = "adam"
    assert not clf.can_fuse(crit, build_optimizer(cfg, clf))
    assert not clf.can_fuse(torch.nn.BCEWithLogitsLoss(), opt)
    cfg.TRAIN.OPTIMIZER = "timm"
    with pytest.raises(RuntimeError):
        build_optimizer(cfg, clf)


def test_head_merge_folds_the_projection_into_the_head(ckpt):
    """TRAIN.MERGE_ENCODER_AND_HEAD_PROJ (kadaptation_clip.py:146-158): head weight = head_proj @ proj^T over the tower's width,
    BatchNorm over that width, the tower's projection leaves the computation (here: becomes the identity)."""
    from pevit_amd.evaluation.kadaptation_clip import Classifier
    from pevit_amd.evaluation import _harness
    cfg = tiny_config(ckpt)
    _harness._BACKBONES.clear()                 # both builds load from the file, so both draw the same numbers before the head
    torch.manual_seed(0)
    plain = Classifier(cfg, 0)
    proj = plain.backbone.visual.proj.data.clone()
    w, b = plain.layers[0].weight.data.clone(), plain.layers[0].bias.data.clone()
    cfg.TRAIN.MERGE_ENCODER_AND_HEAD_PROJ = True
    _harness._BACKBONES.clear()
    torch.manual_seed(0)
    merged = Classifier(cfg, 0)
    E = proj.shape[0]
    assert merged.layers[0].weight.shape == (w.shape[0], E) and merged.channel_bn.num_features == E
    assert torch.allclose(merged.layers[0].weight.data, w @ proj.T) and torch.equal(merged.layers[0].bias.data, b)
    assert torch.equal(merged.backbone.visual.proj.data, torch.eye(E)) and merged.backbone.visual.arch.embed_dim == E
    del merged                                  # a merged tower is never recycled: the next Classifier gets the file's projection
    import gc; gc.collect()
    cfg.TRAIN.MERGE_ENCODER_AND_HEAD_PROJ = False
    again = Classifier(cfg, 0)
    assert torch.equal(again.backbone.visual.proj.data, proj)


# ---------------------------------------------------------------------------------- sweep logic
def test_weight_decay_sweep_visits_reference_grid():
    cfg = default_config()
    cfg.TRAIN.SWEEP_CONCURRENCY = 1                                # the visiting ORDER is asserted below: one run at a time
    seen = []

    def fake_train_task(tr, va, config, sweep_run=False):
        assert sweep_run
        seen.append(config.TRAIN.WD)
        if abs(np.log10(config.TRAIN.WD) - 4.0) < 1e-9:
            raise RuntimeError("diverged")                         # swallowed, scores 0 (reference :202-205)
        return 100.0 - abs(np.log10(config.TRAIN.WD) - 1.4)        # peak between grid points

    wd, score = _harness.hyperparameter_sweep(fake_train_task, None, None, cfg)
    grid = np.logspace(-6, 6, num=97)
    assert np.allclose(seen[:7], np.logspace(-6, 6, num=7))        # 7 coarse points first
    assert len(seen) == 7 + 2 * 4                                  # then spans 8,4,2,1, two probes each
    assert all(np.isclose(grid, s).any() for s in seen)
    assert abs(np.log10(wd) - 1.4) <= 0.125 / 2 + 1e-9 and score > 99.9


def test_sweep_runs_k_at_a_time_in_ordered_sections():
    """run_tasks (round 6): k train_task calls at a time in worker threads; scores come back in the order of the weight decays, a
    failing run scores None without blocking its group, every run sees its OWN config, and what the runs draw from the process-wide
    generator inside ordered sections is the same every time (section n of run r after section n of the runs before it and section
    n - 1 of the runs behind it)."""
    import threading
    cfg = default_config()
    wds = [10.0 ** e for e in range(-3, 4)]

    def fake_train_task(tr, va, config, sweep_run=False):
        draws = []
        for _ in range(3):                                        # construction + two epochs
            with _harness.ordered_section():
                draws.append(float(torch.rand(())))
            time.sleep(0.002 * (1 + hash(config.TRAIN.WD) % 3))   # uneven run lengths
        if config.TRAIN.WD == 1.0:
            raise RuntimeError("diverged")
        assert threading.current_thread().name.startswith("sweep-run-")
        return (config.TRAIN.WD, tuple(draws))

    import time
    res = []
    for _ in range(2):
        torch.manual_seed(0)
        res.append(_harness.run_tasks(fake_train_task, None, None, cfg, wds, 3))
    assert res[0] == res[1]                                        # deterministic
    assert [r[0] if r else None for r in res[0]] == [w if w != 1.0 else None for w in wds]
    # the order of the draws inside a group of three: section 0 of runs 0, 1, 2, then section 1 of runs 0, 1, 2, ...
    torch.manual_seed(0)
    expect = [float(torch.rand(())) for _ in range(9)]
    group = res[0][:3]
    assert [group[r][1][n] for n in range(3) for r in range(3)] == expect
    # one at a time is the plain loop in the caller's thread
    seen = []
    _harness.run_tasks(lambda tr, va, c, sweep_run=False: seen.append((c.TRAIN.WD, threading.current_thread().name)) or 1.0, None, None, cfg, wds, 1)
    assert [w for w, _ in seen] == wds and all(not n.startswith("sweep-run-") for _, n in seen)


def test_lr_sweep_and_final_run_contract():
    cfg = default_config()
    cfg.TRAIN.END_EPOCH, cfg.TRAIN.EXTRA_FINAL_TRAIN_EPOCH = 10, 40
    cfg.DATASET.MERGE_TRAIN_VAL_FINAL_RUN = False
    lrs = []

    def sweep(tr, va, config):
        lrs.append(config.TRAIN.LR)
        return 0.5, 50.0 - abs(np.log10(config.TRAIN.LR) + 3)

    best_lr, best_l2 = _harness.hyperparameter_sweep_lr(sweep, None, None, cfg)
    assert np.allclose(lrs, np.logspace(-6, -1, 6)) and np.isclose(best_lr, 1e-3) and best_l2 == 0.5
    calls = []
    loader = types.SimpleNamespace(dataset=list(range(5)))
    out = _harness.final_run(lambda tr, te, c: calls.append((tr, te, c.TRAIN.LR, c.TRAIN.WD, c.TRAIN.END_EPOCH)) or (1.0, {}),
                             None, loader, None, "test", True, 0.01, 0.25, cfg)
    assert out == (1.0, {}) and calls == [(loader, "test", 0.01, 0.25, 50)] and cfg.is_frozen()


# ---------------------------------------------------------------------------------- metrics / config / alias
def test_metrics_known_answers():
    probs = np.array([[0.7, 0.2, 0.1], [0.1, 0.8, 0.1], [0.3, 0.3, 0.4], [0.6, 0.3, 0.1]])
    labels = np.array([0, 1, 1, 2])
    assert metric.get_metric("accuracy")(labels, probs) == 0.5
    assert abs(metric.get_metric("mean-per-class")(labels, probs) - (1.0 + 0.5 + 0.0) / 3) < 1e-12
    # one class, ranking + - + : precision at recall>=0 .. 0.5 is 1.0, above 0.5 is 2/3
    ap = metric._ap_11_points(np.array([1, 0, 1]), np.array([0.9, 0.8, 0.7]))
    assert abs(ap - (6 * 1.0 + 5 * (2 / 3)) / 11) < 1e-12
    assert abs(metric.get_metric("roc_auc")(np.array([0, 1, 1, 0]), np.array([[.6, .4], [.3, .7], [.2, .8], [.4, .6]])) - 1.0) < 1e-12
    assert metric.get_metric("accuracy").__name__ == "accuracy"


def test_cfgnode_semantics(tmp_path):
    c = default_config()
    c.freeze()
    with pytest.raises(AttributeError):
        c.TRAIN.LR = 1.0
    c.defrost()
    c.merge_from_list(["TRAIN.LR", "0.5", "DATASET.DATASET", "cifar10", "TRAIN.SCHEDULE", "[3, 6]"])
    assert c.TRAIN.LR == 0.5 and c.DATASET.DATASET == "cifar10" and c.TRAIN.SCHEDULE == [3, 6]
    y = tmp_path / "m.yaml"
    y.write_text("TRAIN:\n  WD: 0.\n  NESTEROV: false\nMODEL:\n  SPEC:\n    EMBED_DIM: 768\n")
    c.merge_from_file(str(y))
    assert c.TRAIN.WD == 0.0 and c.TRAIN.NESTEROV is False and c.MODEL.SPEC.EMBED_DIM == 768 and c.TRAIN.LR == 0.5
    d = c.clone(); d.TRAIN.LR = 9
    assert c.TRAIN.LR == 0.5 and isinstance(d.TRAIN, CfgNode) and c.MODEL.SPEC.TEXT.get("SKIP_TOKENIZE", False) is False


def test_reference_import_names_resolve():
    from vision_benchmark.evaluation.kadaptation_clip import Classifier, kadapt_clip, train_one, validate  # noqa: F401
    from vision_benchmark.evaluation.lora_clip import lora_tuning_clip  # noqa: F401
    from vision_benchmark.evaluation.adapter_tuning_clip import adapt_clip  # noqa: F401
    from vision_benchmark.evaluation.compacter_clip import compacter_clip  # noqa: F401
    from vision_benchmark.evaluation.clip_load import adapter_load, compacter_load, load, lora_load  # noqa: F401
    from vision_benchmark.evaluation.model import build_model as bm
    from vision_benchmark.evaluation.lora_model import build_lora_model  # noqa: F401
    from vision_benchmark.evaluation.adapter_model import build_adapter_model  # noqa: F401
    from vision_benchmark.evaluation.compacter_model import build_compacter_model  # noqa: F401
    from vision_benchmark.optim import build_optimizer as bo
    assert bm is build_model and bo is build_optimizer
    with pytest.raises(ModuleNotFoundError):
        import vision_benchmark.datasets  # noqa: F401  (out of scope: not mirrored)


def test_consecutive_classifiers_reuse_the_backbone(ckpt):
    """Sweep-level reuse: the second Classifier of a sweep gets the first one's (garbage-collected) backbone back,
    re-initialised like a fresh build_model(); a live Classifier is never robbed of its backbone."""
    import gc
    from pevit_amd.evaluation.kadaptation_clip import Classifier
    cfg = tiny_config(ckpt)
    _harness._BACKBONES.clear()
    a = Classifier(cfg, 0)
    backbone_id = id(a.backbone)
    b = Classifier(cfg, 0)                                   # a is alive: b must get its own model
    assert id(b.backbone) != backbone_id
    name = "visual.transformer.resblocks.0.attn.q_proj_adapter1_left"
    with torch.no_grad():
        dict(b.backbone.named_parameters())[name].fill_(3.0)  # "training"
        dict(b.backbone.named_parameters())["visual.transformer.phm_rule1_left"].fill_(3.0)
    kept = id(b.backbone)
    del a, b
    gc.collect()
    c = Classifier(cfg, 0)
    assert id(c.backbone) == kept                            # re-used ...
    named = dict(c.backbone.named_parameters())
    assert float(named[name].abs().max()) == 0.0             # ... with the reference initialisation restored
    assert 0.0 < float(named["visual.transformer.phm_rule1_left"].abs().max()) <= 0.01
    assert [n for n, p in c.named_parameters() if p.requires_grad][-2:] == ["layers.0.weight", "layers.0.bias"]
    _harness._BACKBONES.clear()


def test_script_path_launchers_exist_and_parse(tmp_path):
    """reference scripts/*.sh do `cd ../vision_benchmark; python commands/<name>.py --ds resources/datasets/<d>.yaml
    --model resources/model/<m>.yaml ...` (scripts/kadapter_clip.sh:65-70): the launchers and the two yaml files must be
    at those relative paths and accept the reference's flags."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vb = os.path.join(root, "vision_benchmark")
    for name in ("kronecker_adaptation_clip", "lora_clip", "compacter_clip", "adapter_clip"):
        assert os.path.isfile(os.path.join(vb, "commands", name + ".py"))
    out = subprocess.run([sys.executable, "commands/kronecker_adaptation_clip.py", "--help"], cwd=vb, capture_output=True, text=True)
    assert out.returncode == 0 and "--no-tuning" in out.stdout and "--ds" in out.stdout
    from pevit_amd.config import update_config
    cfg = default_config()
    for rel in ("resources/datasets/cifar100.yaml", "resources/model/vitb32_CLIP.yaml"):
        update_config(cfg, types.SimpleNamespace(cfg=os.path.join(vb, rel), opts=[]))
    assert cfg.MODEL.NAME == "ViT-B/32" and cfg.DATASET.NUM_CLASSES == 100 and cfg.TRAIN.END_EPOCH == 10


def test_transformer_seam_is_callable_and_train_mode_is_refused():
    """model.visual.transformer is the reference's operator seam (model.py:1013): a module with forward(x: (N,B,E)); the
    KAdaptation tower carries kdropout (model.py:516,582), which only acts in train mode -- the reference never enters it
    and the engine refuses it instead of silently differing."""
    model = build_model(load_tiny_sd())
    tr = model.visual.transformer
    assert callable(tr) and type(tr).forward is not torch.nn.Module.forward and tr.kdropout == 0.5
    assert not model.training and not model.visual.training
    with pytest.raises(_lib.PevitError, match="kdropout"):
        model.train()
    model.eval()
    with pytest.raises(_lib.PevitError, match="no PyTorch/CPU fallback"):
        tr(torch.zeros(10, 2, 128))                       # CPU tensors: the seam runs only in the HIP engine
    lora = build_peft_model(load_tiny_sd(), "lora")
    lora.train(); lora.eval()                             # no dropout on the other methods' paths


def test_tensor_loader_fetch_into_staging_equals_plain_iteration():
    """TensorLoader.iter_indices / fetch (what _harness.DeviceFeeder drives from its helper thread): gathering a batch straight
    into caller-owned staging buffers gives the batches plain iteration gives, for views (train / val splits of one set), uint8
    and f32 images, and a ragged last batch."""
    from pevit_amd.evaluation.dataloader import TensorLoader, _Tensors, _View
    g = torch.Generator().manual_seed(0)
    for dtype in (torch.uint8, torch.float32):
        imgs = torch.randint(0, 256, (23, 3, 8, 8), generator=g).to(dtype)
        full = _Tensors(imgs, torch.arange(23) % 5)
        for ds in (full, _View(full, [1, 3, 4, 9, 10, 11, 17, 20, 22])):
            loader = TensorLoader(ds, batch_size=4, shuffle=False)
            plain = list(loader)
            assert len(plain) == len(loader)
            out_i, out_l = torch.empty((4, 3, 8, 8), dtype=dtype), torch.empty((4,), dtype=torch.long)
            for sel, (a, b) in zip(loader.iter_indices(), plain):
                x, y = loader.fetch(sel, out_i, out_l)
                assert x.shape == a.shape and torch.equal(x, a) and torch.equal(y, b)
            spec = loader.sample_spec()
            assert spec[0] == ((3, 8, 8), dtype) and spec[1] == ((), torch.long) and spec[2].type == "cpu"
