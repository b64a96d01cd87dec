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

from .feature import create_dataloader


class _Tensors:
    """The whole (images, labels) set, resident on one device."""

    def __init__(self, images, labels):
        self.images, self.labels = images, labels

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        return self.images[i], self.labels[i]


class _View:
    """Subset-like view (``.dataset`` is the full set, like torch.utils.data.Subset, which is what
    merge_trainval_loader checks)."""

    def __init__(self, full, indices):
        self.dataset, self.indices = full, indices

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.dataset[int(self.indices[i])]


class TensorLoader:
    """DataLoader stand-in for in-memory tensor sets: one index-gather per batch instead of stacking 64 samples on
    the host (the stock DataLoader spent 4.6 s of an 8.7 s few-shot run in torch.stack), and -- when the set fits --
    resident on the GPU, so a step does no host-to-device copy at all.  Same iteration contract as the loaders of
    the reference's get_dataloader: no drop_last, reshuffled every epoch when ``shuffle``."""

    def __init__(self, dataset, batch_size=64, shuffle=False, num_workers=0, pin_memory=False):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.num_workers, self.pin_memory = num_workers, pin_memory

    def _base(self):
        if isinstance(self.dataset, _View):
            full = self.dataset.dataset
            key = (id(self.dataset.indices), len(self.dataset.indices), full.images.device)
            if getattr(self, "_idx_key", None) != key:           # the index list of a view is uploaded once, not per batch
                self._idx = torch.as_tensor(self.dataset.indices, dtype=torch.long, device=full.images.device)
                self._idx_key = key
            return full, self._idx
        return self.dataset, None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def iter_indices(self):
        """The index tensors of one epoch's batches (on the device the set lives on)."""
        full, idx = self._base()
        n = len(self.dataset)
        dev = full.images.device
        order = torch.randperm(n).to(dev) if self.shuffle else torch.arange(n, device=dev)
        if idx is not None:
            order = idx[order]
        for lo in range(0, n, self.batch_size):
            yield order[lo:lo + self.batch_size]

    def fetch(self, sel, out_images=None, out_labels=None):
        """Gather one batch; with ``out_*`` (pinned staging buffers of _harness.DeviceFeeder) straight into them."""
        full, _ = self._base()
        if out_images is None:
            return full.images[sel], full.labels[sel]
        n = sel.shape[0]
        if full.images.device.type == "cpu":
            # one memcpy per row, single-threaded (numpy): torch.index_select fans a 19 MB gather out over every core of the
            # host -- 69 ms per 128-image batch on the 256-thread GPU box against 1.4 ms this way (scripts/r4_feeder_timing.py)
            idx = sel.numpy()
            np.take(full.images.numpy().reshape(len(full), -1), idx, axis=0, out=out_images[:n].numpy().reshape(n, -1), mode="clip")
            np.take(full.labels.numpy(), idx, axis=0, out=out_labels[:n].numpy(), mode="clip")
        else:
            torch.index_select(full.images, 0, sel, out=out_images[:n])
            torch.index_select(full.labels, 0, sel, out=out_labels[:n])
        return out_images[:n], out_labels[:n]

    def sample_spec(self):
        full, _ = self._base()
        return (tuple(full.images.shape[1:]), full.images.dtype), (tuple(full.labels.shape[1:]), full.labels.dtype), full.images.device

    def __iter__(self):
        for sel in self.iter_indices():
            yield self.fetch(sel)


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
    """uint8 archives stay uint8: the Classifier hands INPUT.MEAN / INPUT.STD to the engine, which applies the reference's
    ToTensor + Normalize inside its patch gather (bit for bit the host arithmetic; a quarter of the memory and of the bytes to
    upload).  DATASET.NORMALIZE_ON_HOST = True restores the float path."""
    x = torch.as_tensor(np.asarray(images))
    if x.dtype == torch.uint8:
        if config.DATASET.get("NORMALIZE_ON_HOST", False):
            mean = torch.tensor(config.INPUT.MEAN).view(1, 3, 1, 1)
            std = torch.tensor(config.INPUT.STD).view(1, 3, 1, 1)
            x = (x.float() / 255.0 - mean) / std
    else:
        x = x.float()
    return x.contiguous(), torch.as_tensor(np.asarray(labels)).long()


def _synthetic(config):
    sizes = config.DATASET.get("SYNTHETIC_SIZES", (256, 128))
    R = int(config.TRAIN.get("IMAGE_SIZE", [224, 224])[0])
    C = config.DATASET.NUM_CLASSES
    g = torch.Generator().manual_seed(int(config.DATASET.RANDOM_SEED_SAMPLING))
    out = []
    for n in sizes:
        if config.DATASET.get("SYNTHETIC_UINT8", False):      # raw pixels, as an image archive would hold them
            out.append((torch.randint(0, 256, (n, 3, R, R), generator=g, dtype=torch.uint8), torch.arange(n) % C))
        else:
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
    bs = int(config.DATASET.get("LOADER_BATCH_SIZE", 64))    # get_dataloader's batch_size_per_gpu default is 64 (feature.py:101,597)
    dev = _resident_device(config, trx.numel() * trx.element_size() + tex.numel() * tex.element_size())
    tex, tey = tex.to(dev), tey.to(dev)
    test_loader = TensorLoader(_Tensors(tex, tey), batch_size=bs, shuffle=False)
    if test_split_only:
        return test_loader
    labels = try_.numpy()
    shots = int(config.DATASET.NUM_SAMPLES_PER_CLASS)
    if shots > 0:
        keep = few_shot_subset(labels, shots, int(config.DATASET.RANDOM_SEED_SAMPLING))
        trx, try_, labels = trx[keep], try_[keep], labels[keep]
    full = _Tensors(trx.to(dev), try_.to(dev))
    train_idx, val_idx = class_balanced_split(labels, 0.2)
    train_loader = TensorLoader(_View(full, train_idx), batch_size=bs, shuffle=True)
    val_loader = TensorLoader(_View(full, val_idx), batch_size=bs, shuffle=False)
    return train_loader, val_loader, test_loader


def _resident_device(config, nbytes, limit=64 << 30):
    """Keep the tensor set on the training GPU when it fits comfortably (288 GB of HBM per MI355X); DATASET.RESIDENT = False
    keeps it on the host (then _harness.DeviceFeeder uploads batch i+1 while batch i is trained on)."""
    if not config.DATASET.get("RESIDENT", True):
        return torch.device("cpu")
    if torch.cuda.is_available() and nbytes < limit and len(config.GPUS) == 1:
        return torch.device("cuda", int(config.GPUS[0]))
    return torch.device("cpu")
