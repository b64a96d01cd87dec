"""CLI / output-file contract of the fine-tune commands and the loader construction behind them."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_tiny_sd

from pevit_amd.commands import _finetune
from pevit_amd.config import default_config
from pevit_amd.evaluation import _harness
from pevit_amd.evaluation.dataloader import class_balanced_split, construct_dataloader, few_shot_subset


def test_class_balanced_split_is_the_reference_rule():
    labels = np.array([2, 0, 0, 1, 2, 2, 0, 1, 2, 2, 0])          # counts: 0->4, 1->2, 2->5
    train, val = class_balanced_split(labels, 0.2)
    # per class the FIRST ceil(20 %) samples: class 2 -> ceil(1.0)=1 -> idx 0; class 0 -> 1 -> idx 1; class 1 -> 1 -> idx 3
    assert val == [0, 1, 3] and train == [2, 4, 5, 6, 7, 8, 9, 10]
    keep = few_shot_subset(labels, 2, seed=0)
    assert sorted(np.bincount(labels[keep]).tolist()) == [2, 2, 2]
    assert np.array_equal(keep, few_shot_subset(labels, 2, seed=0)) and not np.array_equal(keep, few_shot_subset(labels, 2, seed=1))


def test_loaders_from_tensor_archive(tmp_path):
    rng = np.random.default_rng(0)
    np.savez(tmp_path / "toy.npz", train_images=rng.integers(0, 255, (40, 3, 8, 8), dtype=np.uint8),
             train_labels=np.arange(40) % 4, test_images=rng.standard_normal((6, 3, 8, 8)).astype(np.float32),
             test_labels=np.arange(6) % 4)
    cfg = default_config()
    cfg.DATASET.DATASET, cfg.DATASET.ROOT, cfg.DATASET.NUM_CLASSES = "toy", str(tmp_path), 4
    cfg.DATASET.NUM_SAMPLES_PER_CLASS = 5
    train, val, test = construct_dataloader(cfg)
    assert len(train.dataset) + len(val.dataset) == 20 and len(val.dataset) == 4 and len(test.dataset) == 6
    assert train.batch_size == 64 and train.dataset.dataset is val.dataset.dataset
    x, y = next(iter(val))
    # uint8 archives stay uint8 (round 4): the Classifier hands INPUT.MEAN / STD to the engine, which applies ToTensor + Normalize in
    # its patch gather; DATASET.NORMALIZE_ON_HOST restores the float tensors of the reference's loaders
    assert x.dtype == torch.uint8 and y.dtype == torch.int64 and x.shape == (4, 3, 8, 8)
    assert next(iter(test))[0].dtype == torch.float32
    cfg.DATASET.NORMALIZE_ON_HOST = True
    _, val_f, _ = construct_dataloader(cfg)
    xf, yf = next(iter(val_f))
    assert xf.dtype == torch.float32 and torch.equal(yf, y)
    assert float(xf.max()) < 3.0 and float(xf.min()) > -2.5          # uint8 -> CLIP mean/std normalisation
    mean, std = torch.tensor(cfg.INPUT.MEAN).view(1, 3, 1, 1), torch.tensor(cfg.INPUT.STD).view(1, 3, 1, 1)
    assert torch.equal(xf, (x.float() / 255.0 - mean) / std)
    cfg.DATASET.NORMALIZE_ON_HOST = False
    merged = _harness.merge_trainval_loader(train, val)
    assert len(merged.dataset) == 20
    cfg.DATASET.DATASET = "missing"
    with pytest.raises(RuntimeError):
        construct_dataloader(cfg)


def test_result_files_parse_like_the_reference_readers(tmp_path):
    cfg = default_config()
    cfg.OUTPUT_DIR, cfg.DATASET.DATASET, cfg.DATASET.NUM_SAMPLES_PER_CLASS = str(tmp_path), "cifar-100", 5
    info = {"best_logits": np.full((3, 4), 0.123456789, dtype=np.float32), "n_trainable_params": 101476, "n_params": 151378790,
            "n_visual_params": 87899648, "n_backbone_params": 151327589}
    stem = _finetune.write_results(cfg, "finetuning_5", 61.25, info)
    assert stem == os.path.join(str(tmp_path), "predictions", "finetuning_5", "seed0_cifar-100")
    j = json.load(open(stem + ".json"))
    assert set(j) == {"model_name", "dataset_name", "num_trainable_params", "num_params", "num_visual_params",
                      "num_backbone_params", "n_shot", "rnd_seeds", "predictions"}
    assert j["predictions"][0][0][0] == 0.123457 and j["n_shot"] == 5 and j["rnd_seeds"] == [0]
    text = open(stem + ".txt").read()
    # the two extractions read_txt.py performs (read_txt.py:62-67)
    assert text.strip().split("n_trainable_params is:")[-1].split(",")[0] == "0.101476"
    assert text.strip().split("best acc is:")[-1].split(",")[0] == "61.25"


def test_cli_flags_and_config_layering(tmp_path, monkeypatch):
    ds = tmp_path / "ds.yaml"
    ds.write_text("DATASET:\n  DATASET: 'synthetic'\n  NUM_CLASSES: 7\nTEST:\n  METRIC: 'accuracy'\n")
    model = tmp_path / "model.yaml"
    model.write_text("MODEL:\n  NAME: 'ViT-B/32'\n  SPEC:\n    EMBED_DIM: 512\nTRAIN:\n  END_EPOCH: 10\n  EXTRA_FINAL_TRAIN_EPOCH: 40\n"
                     "  NESTEROV: false\n  WD: 0.\n")
    seen = {}

    def fake_entry(tr, va, te, no_tuning, lr, l2, cfg):
        seen.update(no_tuning=no_tuning, lr=lr, l2=l2, classes=cfg.DATASET.NUM_CLASSES, shots=cfg.DATASET.NUM_SAMPLES_PER_CLASS,
                    end=cfg.TRAIN.END_EPOCH, n=(len(tr.dataset), len(va.dataset), len(te.dataset)), frozen=cfg.is_frozen())
        return 12.5, {"best_logits": np.zeros((2, 7)), "n_trainable_params": 10, "n_params": 20, "n_backbone_params": 5}

    monkeypatch.setattr(_finetune, "config", default_config())
    acc, _ = _finetune.run(fake_entry, ["--ds", str(ds), "--model", str(model), "--no-tuning", "True", "--lr", "0.1", "--l2", "1e-6",
                                        "DATASET.NUM_SAMPLES_PER_CLASS", "5", "OUTPUT_DIR", str(tmp_path / "out"),
                                        "TRAIN.IMAGE_SIZE", "[8, 8]", "DATASET.SYNTHETIC_SIZES", "(70, 14)"])
    assert acc == 12.5 and seen["no_tuning"] is True and seen["lr"] == 0.1 and seen["l2"] == 1e-6
    assert seen["classes"] == 7 and seen["shots"] == 5 and seen["end"] == 10 and seen["frozen"]
    assert seen["n"] == (28, 7, 14)                                    # 5 shots x 7 classes = 35 -> 28 train + 7 val
    assert os.path.isfile(tmp_path / "out" / "predictions" / "finetuning_5" / "seed0_synthetic.txt")


@pytest.mark.gpu
def test_kadaptation_command_end_to_end(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pevit_amd.commands import kronecker_adaptation_clip as cmd
    ckpt = tmp_path / "tiny.pt"
    torch.save(load_tiny_sd(), ckpt)
    model = tmp_path / "model.yaml"
    model.write_text(f"MODEL:\n  NAME: '{ckpt}'\n  SPEC:\n    EMBED_DIM: 64\nTRAIN:\n  END_EPOCH: 1\n  EXTRA_FINAL_TRAIN_EPOCH: 1\n"
                     "  NESTEROV: false\n  WD: 0.\nTEST:\n  METRIC: 'accuracy'\n")
    monkeypatch.setattr(_finetune, "config", default_config())
    acc, info = cmd.main(["--model", str(model), "--no-tuning", "True", "--lr", "0.01", "--l2", "1e-6", "DATASET.DATASET", "synthetic",
                          "DATASET.NUM_CLASSES", "5", "OUTPUT_DIR", str(tmp_path / "out"), "TRAIN.IMAGE_SIZE", "[48, 48]",
                          "DATASET.SYNTHETIC_SIZES", "(40, 10)"])
    assert 0.0 <= acc <= 100.0 and info["best_logits"].shape == (10, 5)
    j = json.load(open(tmp_path / "out" / "predictions" / "finetuning_full" / "seed0_synthetic.json"))
    assert j["num_trainable_params"] == info["n_trainable_params"] and len(j["predictions"][0]) == 10


@pytest.mark.gpu
def test_full_sweep_then_final_run_on_the_engine(tmp_path, monkeypatch):
    """The reference's whole job: 6 learning rates x (7 coarse + 8 bisection) weight decays on (train, val), then
    the final run on train+val evaluated on test -- up to 91 train_task() runs through the fused engine step, the backbone
    built once."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pevit_amd.commands import lora_clip as cmd
    from pevit_amd.evaluation import _harness, clip_load
    ckpt = tmp_path / "tiny.pt"
    torch.save(load_tiny_sd(), ckpt)
    model = tmp_path / "model.yaml"
    model.write_text(f"MODEL:\n  NAME: '{ckpt}'\n  SPEC:\n    EMBED_DIM: 64\nTRAIN:\n  END_EPOCH: 1\n  EXTRA_FINAL_TRAIN_EPOCH: 1\n"
                     "  NESTEROV: false\nTEST:\n  METRIC: 'accuracy'\n")
    monkeypatch.setattr(_finetune, "config", default_config())
    runs = []
    real = _harness.train_task
    monkeypatch.setattr(_harness, "train_task", lambda *a, **k: runs.append(1) or real(*a, **k))
    loads = []
    real_build = clip_load.build_lora_model
    monkeypatch.setattr(clip_load, "build_lora_model", lambda sd: loads.append(1) or real_build(sd))
    _harness._BACKBONES.clear()
    acc, info = cmd.main(["--model", str(model), "--no-tuning", "False", "DATASET.DATASET", "synthetic", "DATASET.NUM_CLASSES", "4",
                          "OUTPUT_DIR", str(tmp_path / "out"), "TRAIN.IMAGE_SIZE", "[48, 48]", "DATASET.SYNTHETIC_SIZES", "(40, 12)"])
    assert 6 * 11 + 1 <= len(runs) <= 6 * 15 + 1   # 7 coarse + 4 x (1 or 2) bisection probes per learning rate (one when the
                                                    # peak sits on the edge of the grid), + the final run
    # the module tree is built from the checkpoint once per concurrently live run (TRAIN.SWEEP_CONCURRENCY, default 2), not 91 times
    assert len(loads) == _harness.sweep_concurrency(_finetune.config)
    assert 0.0 <= acc <= 100.0 and info["best_logits"].shape == (12, 4)
    assert os.path.isfile(tmp_path / "out" / "predictions" / "finetuning_full" / "seed0_synthetic.json")
    _harness._BACKBONES.clear()


def test_learning_rate_scales_with_the_world_size(tmp_path, monkeypatch):
    """config/default.py:257,260: ``config.TRAIN.LR *= comm.world_size`` and ``config.RANK = comm.rank`` in update_config -- 1 and 0
    in every reference run (no process group is ever initialised); under data parallelism the engine averages the ranks' gradients,
    so the line is the reference's own linear scaling rule for a global batch of world x B."""
    import argparse
    from pevit_amd import config as cfgmod
    y = tmp_path / "m.yaml"
    y.write_text("TRAIN:\n  LR: 0.01\n")
    for world, rank in ((1, 0), (8, 3)):
        monkeypatch.setattr(cfgmod, "distributed_world", lambda w=world, r=rank: (w, r))
        c = default_config()
        cfgmod.update_config(c, argparse.Namespace(cfg=str(y), opts=["TRAIN.WD", "0.5"]))
        assert abs(c.TRAIN.LR - 0.01 * world) < 1e-12 and c.RANK == rank and c.TRAIN.WD == 0.5 and c.is_frozen()
    assert cfgmod.distributed_world.__module__                         # (the real one: (1, 0) without a process group)
    monkeypatch.undo()
    assert cfgmod.distributed_world() == (1, 0)
