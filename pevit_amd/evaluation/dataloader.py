"""Loader construction for the fine-tune CLI (reference: evaluation/feature.py:101-190,533-608).

The reference builds its datasets through the ``vision_datasets`` hub (download, manifests, PIL decoding,
torchvision transforms); that dataset layer is outside this build.  What the harness needs from it is kept:
three loaders over ``(image fp32 (3,R,R), target)`` pairs with the reference's batch size (64), the
class-balanced deterministic validation split -- per class the first ceil(20 %) samples go to validation,
feature.py:137-149 -- and train / val being ``Subset`` views of ONE dataset so that
``merge_trainval_loader`` can put them back together for the final run.

Sources: ``<DATASET.ROOT>/<DATASET.DATASET>.npz`` holding already preprocessed tensors
(``train_images``, ``train_labels``, ``test_images``, ``test_labels``; images float32 (N,3,R,R) or uint8, in
which case INPUT.MEAN / INPUT.STD are applied), or ``DATASET.DATASET == 'synthetic'`` (seeded random tensors,
for smoke runs and benchmarks).
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from torch.utils.data import Subset, TensorDataset

from .feature import create_dataloader


def class_balanced_split(labels: np.ndarray, val_split: float = 0.2):
    val = []
    for c in dict.fromkeys(labels.tolist()):                 # classes in order of first appearance, like Counter
        idx = np.where(labels == c)[0]
        val.append(idx[:math.ceil(len(idx) * val_split)])
    val_idx = set(np.concatenate(val).tolist()) if val else set()
    train_idx = [i for i in range(len(labels)) if i not in val_idx]
    return train_idx, sorted(val_idx)


def few_shot_subset(labels: np.ndarray, shots: int, seed: int):
    """``shots`` samples per class, drawn with a seeded generator (the hub's sample_few_shot_subset uses its own
    RNG stream, so the *selection* differs from the reference's for the same seed; the sizes do not)."""
    rng = np.random.default_rng(seed)
    keep = []
    for c in np.unique(labels):
        idx = np.where(labels == c)[0]
        keep.append(rng.permutation(idx)[:shots])
    return np.sort(np.concatenate(keep))


def _tensors(images, labels, config):
    x = torch.as_tensor(np.asarray(images))
    if x.dtype == torch.uint8:
        mean = torch.tensor(config.INPUT.MEAN).view(1, 3, 1, 1)
        std = torch.tensor(config.INPUT.STD).view(1, 3, 1, 1)
        x = (x.float() / 255.0 - mean) / std
    return x.float(), torch.as_tensor(np.asarray(labels)).long()


def _synthetic(config):
    sizes = config.DATASET.get("SYNTHETIC_SIZES", (256, 128))
    R = int(config.TRAIN.get("IMAGE_SIZE", [224, 224])[0])
    C = config.DATASET.NUM_CLASSES
    g = torch.Generator().manual_seed(int(config.DATASET.RANDOM_SEED_SAMPLING))
    out = []
    for n in sizes:
        out.append((torch.randn((n, 3, R, R), generator=g), torch.arange(n) % C))
    return out


def construct_dataloader(config, feature_type="image", test_split_only=False):
    if feature_type != "image":
        raise RuntimeError("only image loaders are built here")
    name = config.DATASET.DATASET
    if name == "synthetic":
        (trx, try_), (tex, tey) = _synthetic(config)
    else:
        path = config.DATASET.ROOT if str(config.DATASET.ROOT).endswith(".npz") else os.path.join(config.DATASET.ROOT, name + ".npz")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} not found: this build reads preprocessed tensor archives; the reference's "
                               "vision_datasets hub / ImageFolder pipeline (feature.py:533-583) is not part of it")
        z = np.load(path)
        trx, try_ = _tensors(z["train_images"], z["train_labels"], config)
        tex, tey = _tensors(z["test_images"], z["test_labels"], config)
    bs, workers, pin = 64, 0, False                      # tensors are already in memory: no worker processes needed
    test_loader = create_dataloader(TensorDataset(tex, tey), batch_size=bs, shuffle=False, num_workers=workers, pin_memory=pin)
    if test_split_only:
        return test_loader
    labels = try_.numpy()
    shots = int(config.DATASET.NUM_SAMPLES_PER_CLASS)
    if shots > 0:
        keep = few_shot_subset(labels, shots, int(config.DATASET.RANDOM_SEED_SAMPLING))
        trx, try_, labels = trx[keep], try_[keep], labels[keep]
    full = TensorDataset(trx, try_)
    train_idx, val_idx = class_balanced_split(labels, 0.2)
    train_loader = create_dataloader(Subset(full, train_idx), batch_size=bs, shuffle=True, num_workers=workers, pin_memory=pin)
    val_loader = create_dataloader(Subset(full, val_idx), batch_size=bs, shuffle=False, num_workers=workers, pin_memory=pin)
    return train_loader, val_loader, test_loader
