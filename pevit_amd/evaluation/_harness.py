"""Shared implementation of the four fine-tune harness modules (kadaptation_clip / lora_clip /
adapter_tuning_clip / compacter_clip), which in the reference are four near-identical copies
(e.g. evaluation/kadaptation_clip.py:78-521).  Behaviour kept from the reference:

* trainability by substring of the parameter name (kadaptation_clip.py:104-122, lora_clip.py:120,
  adapter_tuning_clip.py:116, compacter_clip.py:122);
* ``Classifier.forward`` = backbone -> BatchNorm1d(affine=False) -> [normalize] -> Linear (:176-185);
* ``train_one`` / ``validate`` / ``adjust_learning_rate`` / the two-level sweep (:188-243, :446-466) --
  including that ``validate`` leaves the module in eval mode, so from the second epoch on the reference
  trains with the BatchNorm *running* statistics (:385, no ``model.train()`` anywhere);
* ``train_task`` return contract: best score for sweep runs, ``(best, model_info)`` otherwise (:257-317).

What differs is where the work runs: one ``train_one`` iteration is a single fused call into the HIP engine
(forward, loss, backward, SGD with or without Nesterov momentum) whenever the optimizer is SGD with one weight
decay for every trainable tensor, and
an autograd step over the engine's forward/backward otherwise.  Per-step ``loss.item()`` is replaced by one
read-back per epoch.  Consecutive ``train_task`` calls (the ~90 runs of a sweep) re-use the resident frozen
backbone.
"""
from __future__ import annotations

import contextlib
import gc
import itertools
import logging
import os
import threading
import time
import weakref

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..optim import build_optimizer
from . import clip_load
from .feature import create_dataloader, extract_text_features
from .metric import get_metric

MULTILABEL_DATASETS = {"voc-2007-classification", "chestx-ray8"}

_LOADERS = {"kadaptation": "load", "lora": "lora_load", "adapter": "adapter_load", "compacter": "compacter_load"}


def trainable_by_name(method: str, name: str) -> bool:
    if method == "kadaptation":
        return "adapter" in name or "phm_rule" in name or "attn.b" in name
    if method == "compacter":
        return "compacter" in name
    return "adapter" in name


def gpu_gc(full=True):
    """kadaptation_clip.py:72-75.  ``full=False`` (between the ~90 runs of a sweep): a young-generation collection only --
    the Classifier of a run holds no reference cycle and is freed by its ``del``; a full collection walks the whole CLIP module
    tree kept alive by the backbone cache and costs 0.07 s of a 0.3 s run."""
    gc.collect() if full else gc.collect(0)
    torch.cuda.empty_cache()


class AverageMeter(object):
    """Computes and stores the average and current value"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class _BackboneCache:
    """Sweep-level reuse.  ``train_task`` builds a Classifier per run (~90 per dataset); rebuilding
    the 151 M-parameter CLIP module tree and uploading it costs far more than the few dozen steps of a few-shot run.
    The backbone of a Classifier that has been garbage-collected is handed to the next one after being put back
    into the state a fresh ``load()`` would give: adapters re-initialised from the torch RNG exactly like
    ``build_model`` does, adapter keys of the checkpoint overlaid, frozen tensors untouched (nothing on this path
    ever writes them), engine optimiser state cleared when the new Classifier binds."""

    MAX_PER_KEY = 4                 # each model carries a HIP context (arena + workspace, ~2 GB at ViT-B/32 batch 64)

    def __init__(self):
        self._items = {}            # key -> [(weakref to the owning Classifier, model), ...]: one entry per concurrently live run
        self._lock = threading.Lock()

    def take(self, key):
        with self._lock:
            for owner, model in reversed(self._items.get(key, ())):
                if owner() is None:                                  # its Classifier is gone: idle (the most recently used first)
                    return model
        return None

    def give(self, key, owner, model):
        with self._lock:
            for k in [k for k in self._items if k != key]:           # drop the idle models of other checkpoints / methods
                self._items[k] = [e for e in self._items[k] if e[0]() is not None]
                if not self._items[k]:
                    del self._items[k]
            entries = [e for e in self._items.get(key, ()) if e[1] is not model]
            entries.append((weakref.ref(owner), model))
            while len(entries) > self.MAX_PER_KEY and any(e[0]() is None for e in entries[:-1]):
                entries.remove(next(e for e in entries[:-1] if e[0]() is None))      # oldest idle model (and its engine) goes
            self._items[key] = entries

    def evict(self, model):
        """A backbone whose frozen tensors were changed (head merge) must never be handed to another Classifier."""
        with self._lock:
            self._items = {k: [e for e in v if e[1] is not model] for k, v in self._items.items()}
            self._items = {k: v for k, v in self._items.items() if v}

    def clear(self):
        with self._lock:
            self._items.clear()

    def __len__(self):
        with self._lock:
            return sum(len(v) for v in self._items.values())


_BACKBONES = _BackboneCache()


# ---- K sweep runs at a time (round 6) ---------------------------------------------------------------------------------------------
# The reference's sweep is ~90 independent short runs at batch 64 on ONE frozen backbone (kadaptation_clip.py:188-243,446-466;
# feature.py:101).  A batch-64 step fills 120 of the 256 CUs in its large GEMMs, and a second run -- its own engine context, its
# own stream, no dependency on the first -- fills the rest: measured 21.2 k -> 28.4 k (two runs) -> 30.2 k images/s (three) in
# aggregate, every run bit-identical to the same run stepped alone (scripts/r6_dual_stream.py, bench.py `concurrent_runs`).
# TRAIN.SWEEP_CONCURRENCY (or PEVIT_SWEEP_CONCURRENCY; default 2, 1 = the reference's strictly sequential order) runs that many
# train_task calls of a sweep at a time, each in a worker thread under its own stream.  What stays deterministic: everything that
# draws from the process-wide torch generator (adapter / head initialisation of a new Classifier, the epoch's shuffle) is an
# ORDERED SECTION -- section n of run r starts when section n of the runs before it and section n - 1 of the runs behind it
# are done -- so that a seeded sweep gives the same scores every time (they differ from the sequential sweep's, whose runs draw in
# another order; each run equals the same run alone from the same draws).
_RUN = threading.local()


class _Turns:
    TIMEOUT_S = 900

    def __init__(self, k):
        self.cv, self.done, self.finished = threading.Condition(), [0] * k, [False] * k

    @contextlib.contextmanager
    def ordered(self, r):
        with self.cv:
            n = self.done[r]
            ok = self.cv.wait_for(lambda: all(self.finished[j] or self.done[j] > n for j in range(r)) and
                                  all(self.finished[j] or self.done[j] >= n for j in range(r + 1, len(self.done))), timeout=self.TIMEOUT_S)
            if not ok:               # a peer is stuck: fail this run loudly (it scores None) instead of hanging the sweep
                raise RuntimeError(f"concurrent sweep: run {r} waited {self.TIMEOUT_S} s for its turn at ordered section {n}")
        try:
            yield
        finally:
            with self.cv:
                self.done[r] += 1
                self.cv.notify_all()

    def finish(self, r):
        with self.cv:
            self.finished[r] = True
            self.cv.notify_all()


def ordered_section():
    """Context of a block that consumes the process-wide torch generator; a no-op outside a concurrent sweep."""
    turns = getattr(_RUN, "turns", None)
    return turns.ordered(_RUN.index) if turns is not None else contextlib.nullcontext()


def sweep_concurrency(config) -> int:
    k = os.environ.get("PEVIT_SWEEP_CONCURRENCY")
    if k is None:
        k = config.TRAIN.get("SWEEP_CONCURRENCY", 2) if hasattr(config.TRAIN, "get") else getattr(config.TRAIN, "SWEEP_CONCURRENCY", 2)
    k = max(1, int(k))
    return k if (torch.cuda.is_available() and len(config.GPUS) == 1) else 1


def run_tasks(train_task_fn, train_dataloader, val_dataloader, config, wds, k):
    """train_task_fn(..., sweep_run=True) for every weight decay of ``wds``, ``k`` at a time; a failed run scores None (the
    reference's bare ``except: score = 0``, kadaptation_clip.py:200-205).  Returns the scores in the order of ``wds``."""
    def one(cfg, wd):
        cfg.defrost()
        cfg.TRAIN.WD = wd
        try:
            return train_task_fn(train_dataloader, val_dataloader, cfg, sweep_run=True)
        except Exception:
            gpu_gc()
            return None

    if any(getattr(dl, "persistent_workers", False) for dl in (train_dataloader, val_dataloader)):
        k = 1                                    # such a loader has ONE live iterator: two runs cannot walk it at the same time
    if k <= 1 or len(wds) <= 1:
        return [one(config, wd) for wd in wds]
    scores = [None] * len(wds)
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", config.GPUS[0]) if cuda else None
    streams = _RUN_STREAMS.setdefault(str(dev), [])
    while cuda and len(streams) < k:
        streams.append(torch.cuda.Stream(dev))
    for lo in range(0, len(wds), k):
        group = list(range(lo, min(lo + k, len(wds))))
        turns = _Turns(len(group))
        if cuda:
            torch.cuda.synchronize(dev)          # models handed over from the pool were last used on another stream

        def worker(r, i):
            _RUN.turns, _RUN.index = turns, r
            try:
                if cuda:
                    torch.cuda.set_device(dev)
                with (torch.cuda.stream(streams[r]) if cuda else contextlib.nullcontext()):
                    scores[i] = one(config.clone(), wds[i])
                    if cuda:
                        streams[r].synchronize()
            finally:
                _RUN.turns = None
                turns.finish(r)
        threads = [threading.Thread(target=worker, args=(r, i), name=f"sweep-run-{r}") for r, i in enumerate(group)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    config.defrost()
    config.TRAIN.WD = wds[-1]                    # (the sequential sweep leaves the last weight decay in the shared config)
    return scores


_RUN_STREAMS = {}
_ZEROSHOT = {}


def _reinitialise(model, method, name):
    from .model import _init_adapters
    with torch.no_grad():
        for n, p in model.visual.named_parameters():
            if n in model.visual._trainable_names or "phm_rule" in n:
                p.zero_()
    _init_adapters(model, method)
    if method == "compacter" and model.visual._engine is not None:       # the engine keeps its own copy of the frozen rule
        model.visual._engine.load_phm_rule(dict(model.visual.named_parameters())["transformer.phm_rule"])
    if os.path.isfile(name):
        sd = clip_load._read_state_dict(name)
        own = dict(model.named_parameters())
        with torch.no_grad():
            for k in ["visual." + n for n in model.visual._trainable_names]:
                if k in sd:
                    own[k].copy_(sd[k].to(own[k].device, own[k].dtype))
    return model.eval()


def get_cls_model(method, config, feature_type="image", owner=None):
    key = (config.MODEL.NAME, method, clip_load._DEFAULT_DEVICE)
    model = _BACKBONES.take(key) if owner is not None else None
    if model is not None:
        _reinitialise(model, method, config.MODEL.NAME)
    else:
        model, _ = getattr(clip_load, _LOADERS[method])(config.MODEL.NAME, jit=False)
    if owner is not None:
        _BACKBONES.give(key, owner, model)
    if feature_type == "image":
        model.forward = model.encode_image
    elif feature_type == "text":
        model.forward = model.encode_text
    else:
        raise Exception("Incorrect model type.")
    return model


def _zeroshot_key(config):
    names = config.DATASET.get("CLASS_NAMES", None)
    if not names:
        return None
    parts = []
    for n in names:
        n = n[0] if type(n) == list else n
        parts.append(n.cpu().numpy().tobytes() if torch.is_tensor(n) else str(n))
    return (config.MODEL.NAME, tuple(parts), tuple(config.DATASET.get("TEMPLATES", None) or ()))


class ClassifierBase(nn.Module):
    """Linear classifier on the adapted CLIP tower."""
    METHOD = "kadaptation"
    tokenizer = None        # set to a callable(texts, context_length=...) to use INIT_HEAD_WITH_TEXT_ENCODER

    def __init__(self, config, l2_lambda):
        super().__init__()
        self.backbone = get_cls_model(self.METHOD, config, feature_type="image", owner=self)
        for name, param in self.backbone.named_parameters():
            param.requires_grad = trainable_by_name(self.METHOD, name)
        input_dim, output_dim = config.MODEL.SPEC.EMBED_DIM, config.DATASET.NUM_CLASSES
        self.optim = None
        self.l2_lambda = l2_lambda
        self.channel_bn = nn.BatchNorm1d(input_dim, affine=False)
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_bn_counter())
        self.layers = nn.Sequential(nn.Linear(input_dim, output_dim))
        dev = self.backbone.logit_scale.device
        self.channel_bn.to(dev); self.layers.to(dev)

        if config.TRAIN.INIT_HEAD_WITH_TEXT_ENCODER:
            zkey = _zeroshot_key(config)
            zeroshot_weights = _ZEROSHOT.get(zkey) if zkey is not None else None
            if zeroshot_weights is None:            # the text tower is frozen: one evaluation per (checkpoint, prompts)
                zeroshot_weights = extract_text_features(config, self.tokenizer, model=self.backbone, return_numpy=False)
                if zkey is not None:
                    _ZEROSHOT.clear(); _ZEROSHOT[zkey] = zeroshot_weights.detach().clone()
            w = self.layers[0].weight
            w.data = zeroshot_weights.T.to(w.dtype).to(w.device).contiguous()
            self.layers[0].bias.data.fill_(0.0)

        if config.TRAIN.MERGE_ENCODER_AND_HEAD_PROJ and getattr(self.backbone.visual, "proj", None) is not None:
            # kadaptation_clip.py:146-158: visual.proj leaves the tower and is multiplied into the head; BatchNorm then runs
            # over the tower's width.  (The reference's scripts pass False; here the tower keeps an identity projection.)
            head_proj, head_bias = self.layers[0].weight.data, self.layers[0].bias.data
            _BACKBONES.evict(self.backbone)                                                # its proj is about to change
            encoder_proj = self.backbone.visual.merge_proj_into_head()                     # (E, D)
            encoder_ic = encoder_proj.shape[0]
            self.channel_bn = nn.BatchNorm1d(encoder_ic, affine=False).to(dev)
            self.layers = nn.Sequential(nn.Linear(encoder_ic, output_dim)).to(dev)
            self.layers[0].weight.data = head_proj @ encoder_proj.T.to(head_proj.dtype).to(head_proj.device)
            self.layers[0].bias.data = head_bias

        self.logit_scale = nn.Parameter(torch.ones([], device=dev))
        self.logit_scale.requires_grad = config.TRAIN.TRAINABLE_LOGIT_SCALE
        if config.TRAIN.LOGIT_SCALE_INIT == "pretrained":
            self.logit_scale.data = self.backbone.logit_scale.data.to(self.logit_scale.dtype).to(self.logit_scale.device)
        elif config.TRAIN.LOGIT_SCALE_INIT == "ln_cls":
            self.logit_scale.data *= np.log(np.log(config.DATASET.NUM_CLASSES))
        elif config.TRAIN.LOGIT_SCALE_INIT == "clip":
            self.logit_scale.data *= np.log(1 / 0.07)
        else:
            self.logit_scale.data *= 0

        self.normalize_visual_output = config.TRAIN.NORMALIZE_VISUAL_FEATURE
        if not config.TRAIN.USE_CHANNEL_BN:
            self.channel_bn = nn.Identity()

        visual = self.backbone.visual
        # MODEL.WEIGHT_FORMAT, when the config has the key, is authoritative (an explicit "bf16" is not overridden by the
        # PEVIT_WEIGHT_FORMAT environment default the tower was constructed with); a change of format drops a stale engine
        wf = config.MODEL.get("WEIGHT_FORMAT", None) if hasattr(config.MODEL, "get") else None
        if wf is not None:
            visual.weight_format = str(wf)
        if hasattr(config, "INPUT") and getattr(config.INPUT, "MEAN", None) is not None:
            visual.set_input_normalization(config.INPUT.MEAN, config.INPUT.STD)      # uint8 batches: ToTensor + Normalize in the engine
        visual._num_classes = output_dim
        visual._max_batch = max(int(config.TRAIN.BATCH_SIZE_PER_GPU), int(config.TEST.BATCH_SIZE_PER_GPU))
        if visual._engine is not None and (visual._engine.num_classes != output_dim or
                                           visual._engine.weight_format != visual.weight_format):
            visual._engine = None
        self._bound = None

    # ---- engine binding ------------------------------------------------------------------
    def engine(self):
        """HIP context of the tower with this module's head / BatchNorm buffers seated in it."""
        eng = self.backbone.visual.engine()
        if self._bound is not eng:
            views = eng.param_views()
            lin = self.layers[0]
            with torch.no_grad():
                views["layers.0.weight"].copy_(lin.weight)
                views["layers.0.bias"].copy_(lin.bias)
            lin.weight.data, lin.bias.data = views["layers.0.weight"], views["layers.0.bias"]
            eng.reset_optimizer()                 # a new Classifier is a new run: no momentum carried over
            if isinstance(self.channel_bn, nn.BatchNorm1d):
                eng.running_mean.copy_(self.channel_bn.running_mean)
                eng.running_var.copy_(self.channel_bn.running_var)
                self.channel_bn.running_mean = eng.running_mean
                self.channel_bn.running_var = eng.running_var
            self._bound = eng
        return eng

    def forward(self, img):
        pdtype = torch.float32 if img.dtype == torch.uint8 else img.dtype
        if img.is_cuda:
            self.engine()
        feature = self.backbone(img).to(pdtype)
        outputs = self.channel_bn(feature)
        if self.normalize_visual_output:
            outputs = F.normalize(outputs)
        return self.layers(outputs)

    # ---- fused step ----------------------------------------------------------------------
    def can_fuse(self, criterion, optimizer) -> bool:
        """True when one ``pevit_train_forward_backward`` + ``pevit_sgd_step`` is exactly the reference step."""
        if not isinstance(criterion, nn.CrossEntropyLoss) or criterion.weight is not None:
            return False
        if criterion.reduction != "mean" or getattr(criterion, "label_smoothing", 0.0) != 0.0 or criterion.ignore_index >= 0:
            return False
        if type(optimizer) is not torch.optim.SGD or not isinstance(self.channel_bn, nn.BatchNorm1d):
            return False
        if self.normalize_visual_output or self.logit_scale.requires_grad:
            return False
        if self.channel_bn.momentum != 0.1 or self.channel_bn.eps != 1e-5:
            return False
        live = [g for g in optimizer.param_groups if len(g["params"]) > 0]
        if not live:
            return False
        g0 = live[0]
        for g in live:
            if g["dampening"] != 0 or g.get("maximize", False):
                return False
            if (g["lr"], g["momentum"], g["weight_decay"], g["nesterov"]) != (g0["lr"], g0["momentum"], g0["weight_decay"],
                                                                             g0["nesterov"]):
                return False
        mine = {id(p) for p in self.parameters() if p.requires_grad}
        theirs = {id(p) for g in live for p in g["params"]}
        return mine == theirs

    def fused_train_step(self, images, target, optimizer, logits_out=None, loss_out=None):
        """One reference train_one iteration as one engine call.  ``logits_out`` (B, C) / ``loss_out`` (1,): rows of the caller's
        epoch buffers the engine writes straight into; without them the results are copies of the engine's per-step buffers."""
        eng = self.engine()
        eng.ensure_batch(images.shape[0])
        g = next(g for g in optimizer.param_groups if len(g["params"]) > 0)
        img = images.contiguous() if images.dtype == torch.uint8 else images.contiguous().float()
        logits, loss = eng.train_step(img, target.contiguous(), lr=g["lr"], momentum=g["momentum"],
                                      weight_decay=g["weight_decay"], bn_training=self.channel_bn.training,
                                      nesterov=bool(g["nesterov"]), logits_out=logits_out, loss_out=loss_out)
        if self.channel_bn.training:
            self._bn_steps = getattr(self, "_bn_steps", 0) + 1       # folded into num_batches_tracked once per epoch
        if logits_out is None:
            logits = logits.clone()
        if loss_out is None:
            loss = loss.clone()
        return logits, loss

    def flush_bn_counter(self, *_):
        """Fold the fused steps counted since the last flush into ``channel_bn.num_batches_tracked`` (train_one does it at the end
        of an epoch, also when the epoch raises; ``state_dict()`` does it first, so a checkpoint taken mid-epoch -- or after direct
        calls of ``fused_train_step`` -- carries the reference BatchNorm1d's counter)."""
        n = getattr(self, "_bn_steps", 0)
        if n and isinstance(self.channel_bn, nn.BatchNorm1d):
            self.channel_bn.num_batches_tracked += n
        self._bn_steps = 0


def adjust_learning_rate(optimizer, epoch, config):
    """Decay the learning rate based on schedule"""
    lr = config.TRAIN.LR
    for milestone in config.TRAIN.SCHEDULE:
        lr *= 0.1 if epoch >= milestone else 1.0
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr


def accuracy(output, target, topk=(1,)):
    """Computes the accuracy over the k top predictions for the specified values of k"""
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.view(1, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def _score(metric, outputs, targets):
    logits = torch.cat(outputs, dim=0).softmax(-1).data.cpu().numpy()
    labels = torch.cat(targets, dim=0).data.cpu().numpy()
    try:                                    # the reference guards NaNs of mAP-like metrics the same way
        return 100.0 * metric(labels, logits), logits
    except Exception:
        return 0.0, logits


_FEED_STATE = {}        # "device/run" -> {"stream": copy stream, "slots": staging ring}; see DeviceFeeder (run = index within a concurrent sweep group)


class DeviceFeeder:
    """Batches of any loader on the training GPU, one batch ahead of the consumer.

    The reference feeds ``train_one`` from a 6-worker DataLoader with pinned memory and ``images.cuda(non_blocking=True)``
    (feature.py:101, kadaptation_clip.py:340): the upload of batch i+1 overlaps the step of batch i.  Here a helper thread pulls
    the next batch from the loader (for host-resident tensor sets that is the index gather, straight into a pinned staging
    buffer), issues the host-to-device copies on a side stream into a device staging buffer and hands over
    (images, target, event); the consumer makes its stream wait for the event.  SLOTS staging pairs (pinned host + device),
    allocated once per epoch and re-used in a ring: a pair is rewritten only after the consumer has asked for the batch behind
    it (its step is enqueued) -- the copy stream then waits for the event the consumer recorded at that moment.  Batches that
    already live on the device pass through untouched."""
    DEPTH = 2
    SLOTS = DEPTH + 2

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device("cuda", device) if isinstance(device, int) else torch.device(device)

    def __iter__(self):
        import queue
        import threading
        direct = hasattr(self.loader, "iter_indices") and hasattr(self.loader, "fetch")    # evaluation/dataloader.TensorLoader
        if direct and self.loader.sample_spec()[2] == self.device:
            for batch in self.loader:                        # resident set: nothing to move
                yield batch[0], batch[1]
            return
        it = self.loader.iter_indices() if direct else iter(self.loader)
        try:
            first = next(it)
        except StopIteration:
            return
        if not direct and all(not torch.is_tensor(t) or t.device == self.device for t in first[:2]):
            yield first[0], first[1]
            for batch in it:
                yield batch[0], batch[1]
            return
        q = queue.Queue(maxsize=self.DEPTH)
        # the copy stream and the staging ring outlive the epoch (pinning 4 x 19 MB of host memory costs tens of ms: more than
        # the uploads of a 20-step epoch); every epoch ends with a stream synchronisation, so the next one finds them idle
        cache = _FEED_STATE.setdefault(f"{self.device}/{getattr(_RUN, 'index', 0) if getattr(_RUN, 'turns', None) is not None else 0}", {})
        if "stream" not in cache:
            cache["stream"] = torch.cuda.Stream(self.device)
            cache["slots"] = [dict(h_img=None, h_tgt=None, d_img=None, d_tgt=None, used=None) for _ in range(self.SLOTS)]
        copy_stream, slots = cache["stream"], cache["slots"]
        for sl in slots:
            sl["used"] = None
        stop = threading.Event()

        def staging(slot, n, ishape, idt, tshape, tdt):
            cap = max(n, getattr(self.loader, "batch_size", 1))
            if slot["h_img"] is None or slot["h_img"].shape[0] < n or slot["h_img"].dtype != idt or tuple(slot["h_img"].shape[1:]) != ishape \
                    or slot["h_tgt"].dtype != tdt or tuple(slot["h_tgt"].shape[1:]) != tshape:
                slot["h_img"] = torch.empty((cap,) + ishape, dtype=idt).pin_memory()
                slot["h_tgt"] = torch.empty((cap,) + tshape, dtype=tdt).pin_memory()
                slot["d_img"] = torch.empty((cap,) + ishape, dtype=idt, device=self.device)
                slot["d_tgt"] = torch.empty((cap,) + tshape, dtype=tdt, device=self.device)

        def produce():
            try:
                torch.cuda.set_device(self.device)
                i, batch = 0, first
                while batch is not None and not stop.is_set():
                    slot = slots[i % self.SLOTS]
                    if direct:                               # gather straight into the pinned staging buffer
                        (ishape, idt), (tshape, tdt), _ = self.loader.sample_spec()
                        n = batch.shape[0]
                        staging(slot, n, ishape, idt, tshape, tdt)
                        if slot["used"] is not None:
                            slot["used"].synchronize()       # (already complete: the consumer is SLOTS - DEPTH batches further on)
                        self.loader.fetch(batch, slot["h_img"], slot["h_tgt"])
                    else:
                        images, target = batch[0], batch[1]
                        n = images.shape[0]
                        staging(slot, n, tuple(images.shape[1:]), images.dtype, tuple(target.shape[1:]), target.dtype)
                        if slot["used"] is not None:
                            slot["used"].synchronize()
                        slot["h_img"][:n].copy_(images); slot["h_tgt"][:n].copy_(target)
                    with torch.cuda.stream(copy_stream):
                        slot["d_img"][:n].copy_(slot["h_img"][:n], non_blocking=True)
                        slot["d_tgt"][:n].copy_(slot["h_tgt"][:n], non_blocking=True)
                        ev = torch.cuda.Event(); ev.record(copy_stream)
                    q.put((i % self.SLOTS, n, ev))
                    i += 1
                    batch = next(it, None)
                q.put(None)
            except BaseException as e:                       # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                k, n, ev = item
                # the upload was enqueued a batch or more ago and has normally finished: the HOST seeing the event complete orders the
                # copy before everything launched from here on, and spares the consumer's stream a cross-queue dependency (round 5:
                # such a dependency costs tens of microseconds of dispatch latency on this platform: profiles/r05_dp_single_exchange.md)
                if not ev.query():
                    torch.cuda.current_stream(self.device).wait_event(ev)
                yield slots[k]["d_img"][:n], slots[k]["d_tgt"][:n].clone()    # targets outlive the ring (epoch metric): a copy of their own
                # back here the consumer has enqueued everything that reads this pair: the ring may come round to it after that
                used = torch.cuda.Event(); used.record(torch.cuda.current_stream(self.device))
                slots[k]["used"] = used
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except Exception:
                    pass
                th.join(timeout=0.01)
            torch.cuda.current_stream(self.device).synchronize()      # the staging buffers die with this generator


def train_one(train_loader, model, criterion, optimizer, epoch, config):
    batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter()
    metric = get_metric(config.TEST.METRIC)
    outputs, targets, step_losses, step_sizes = [], [], [], []
    fused = model.can_fuse(criterion, optimizer)
    dev = config.GPUS[0]
    single = len(config.GPUS) == 1
    # epoch buffers the fused step writes its logits / loss straight into (no per-step copies of the engine's buffers)
    n_total = len(getattr(train_loader, "dataset", ())) if (fused and hasattr(train_loader, "__len__")) else 0
    out_buf = loss_buf = None
    if fused and n_total > 0:
        out_buf = torch.empty((n_total, config.DATASET.NUM_CLASSES), dtype=torch.float32, device=torch.device("cuda", dev))
        loss_buf = torch.empty((len(train_loader) + 1,), dtype=torch.float32, device=out_buf.device)
    row = step = 0
    end = time.time()
    try:     # the fused step counts its BatchNorm batches host-side: folded into num_batches_tracked whatever ends the epoch
        batches = iter(DeviceFeeder(train_loader, dev) if single else ((b[0], b[1]) for b in train_loader))
        with ordered_section():                               # the epoch's shuffle is drawn on the way to the first batch
            first = next(batches, None)
        for images, target in (() if first is None else itertools.chain((first,), batches)):
            data_time.update(time.time() - end)
            if images.shape[0] == 1:
                continue                                      # BatchNorm cannot take a single-sample batch (reference :341)
            if target.shape[-1] == 1:
                target = target[:, 0]
            if not target.is_cuda:
                target = target.cuda(dev, non_blocking=True)

            if fused and target.dim() == 1 and target.dtype == torch.int64:
                n = images.shape[0]
                direct = out_buf is not None and row + n <= out_buf.shape[0] and step < loss_buf.shape[0]
                output, loss = model.fused_train_step(images, target, optimizer,
                                                      out_buf[row:row + n] if direct else None,
                                                      loss_buf[step:step + 1] if direct else None)
                row += n if direct else 0
                step += 1 if direct else 0
            else:
                optimizer.zero_grad()
                output = model.forward(images)
                loss = criterion(output, target)
                loss.backward()
                optimizer.step()
            step_losses.append(loss.detach().reshape(1)); step_sizes.append(images.size(0))
            outputs.append(output.detach())
            targets.append(target)
            batch_time.update(time.time() - end)
            end = time.time()
    finally:
        if fused:
            model.flush_bn_counter()

    if not outputs:
        return
    for v, n in zip(torch.cat(step_losses).cpu().tolist(), step_sizes):       # one sync per epoch, not per step
        losses.update(v, n)
    metric_result, _ = _score(metric, outputs, targets)
    logging.info(f"[Epoch {epoch}] Train: {metric.__name__} {metric_result:.3f}")
    return losses.avg


@torch.no_grad()
def validate(val_loader, model, criterion, epoch, config, return_logits=False):
    metric = get_metric(config.TEST.METRIC)
    outputs, targets = [], []
    # a stream-K hand-off error raised during the preceding training steps (whose SGD updates were skipped on device) must
    # not pass silently into a validation score or a checkpoint
    eng = getattr(getattr(getattr(model, "backbone", None), "visual", None), "_engine", None)
    if eng is not None:
        eng.check_streamk()
    model.eval()
    dev = config.GPUS[0]
    single = len(config.GPUS) == 1
    for images, target in (DeviceFeeder(val_loader, dev) if single else ((b[0], b[1]) for b in val_loader)):
        if not target.is_cuda:
            target = target.cuda(dev, non_blocking=True)
        if target.shape[-1] == 1:
            target = target[:, 0]
        outputs.append(model(images))
        targets.append(target)
    metric_result, logits = _score(metric, outputs, targets)
    logging.info(f"[Epoch {epoch}] Val: {metric.__name__} {metric_result:.3f}")
    return (metric_result, logits) if return_logits else metric_result


def train_task(classifier_cls, train_dataloader, test_dataloader, config, sweep_run=False):
    best_acc1 = 0
    with ordered_section():                                   # adapter / head initialisation draws from the process-wide generator
        model = classifier_cls(config, 0)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    logging.info(f"Number of trainable params: {n_train / 1000000}M.")

    gpu = config.GPUS
    if len(gpu) == 1:
        torch.cuda.set_device(gpu[0])
        model = model.cuda(gpu[0])

    if config.DATASET.DATASET in MULTILABEL_DATASETS:
        criterion = nn.BCEWithLogitsLoss().cuda(gpu[0])
    else:
        criterion = nn.CrossEntropyLoss().cuda(gpu[0])
    optimizer = build_optimizer(config, model)

    model_info = {}
    visual = model.backbone.visual if getattr(model.backbone, "visual", None) is not None else model.backbone
    model_info["n_trainable_params"] = sum(p.numel() for p in model.parameters() if p.requires_grad)
    model_info["n_visual_params"] = sum(p.numel() for p in visual.parameters())
    model_info["n_backbone_params"] = sum(p.numel() for p in model.backbone.parameters())
    model_info["n_params"] = sum(p.numel() for p in model.parameters())

    acc1 = 0.0
    for epoch in range(config.TRAIN.BEGIN_EPOCH, config.TRAIN.END_EPOCH):
        adjust_learning_rate(optimizer, epoch, config)
        if not config.TRAIN.EMULATE_ZERO_SHOT:
            train_one(train_dataloader, model, criterion, optimizer, epoch, config)
        acc1, logits = validate(test_dataloader, model, criterion, epoch, config, return_logits=True)
        if acc1 > best_acc1:
            model_info["best_logits"] = logits
        best_acc1 = max(acc1, best_acc1)

    logging.info(f"=> Learning rate {config.TRAIN.LR}, L2 lambda {config.TRAIN.WD}: Best score: Acc@1 {best_acc1:.3f}")
    if sweep_run and config.TRAIN.SEARCH_RESULT_ON_LAST_EPOCH:
        return acc1

    del model, criterion, optimizer
    gpu_gc(full=not sweep_run)
    return best_acc1 if sweep_run else (best_acc1, model_info)


def hyperparameter_sweep(train_task_fn, train_dataloader, val_dataloader, config):
    """Weight-decay search at a fixed LR: 7 coarse points of a 97-point log grid, then bisection with spans
    8,4,2,1 around the running peak (reference :188-243).  Failed runs score 0, as in the reference."""
    logging.info(f"=> Learning rate {config.TRAIN.LR}: tuning l2 regularization strength.")
    start = time.time()
    lo, hi = config.TRAIN.SEARCH_WD_LOG_LOWER, config.TRAIN.SEARCH_WD_LOG_UPPER
    grid = np.logspace(lo, hi, num=97).tolist()
    coarse = set(np.logspace(lo, hi, num=7))
    init_idx = [i for i, v in enumerate(grid) if v in coarse]
    peak_idx, peak_score, score = -1, 0, 0.0

    k = sweep_concurrency(config)
    # the runs of one search stage are independent of each other (the reference evaluates them one after the other and only
    # then moves the peak): k at a time, scores consumed in the reference's order
    for idx, score in zip(init_idx, run_tasks(train_task_fn, train_dataloader, val_dataloader, config, [grid[i] for i in init_idx], k)):
        if score is None:
            score = 0.0
            continue
        if score > peak_score:
            peak_idx, peak_score = idx, score
    logging.info(f"Iteration 0: l2_lambda: {grid[peak_idx]}, best score {score}")

    step_span, it = 8, 0
    while step_span > 0:
        left, right = max(peak_idx - step_span, 0), min(peak_idx + step_span, len(grid) - 1)
        cand = [i for i in (left, right) if i != peak_idx]
        # WD_SEARCH_LEFT reproduces the reference's initial release, which always probed the left neighbour
        wds = [grid[left] if config.TRAIN.WD_SEARCH_LEFT else grid[i] for i in cand]
        for idx, score in zip(cand, run_tasks(train_task_fn, train_dataloader, val_dataloader, config, wds, k)):
            if score is None:
                score = 0.0
                continue
            if score > peak_score:
                peak_idx, peak_score = idx, score
        it += 1
        logging.info(f"Iteration {it}: l2_lambda: {grid[peak_idx]}, best score {score}")
        step_span //= 2

    logging.info(f"=> Learning rate {config.TRAIN.LR}: The best l2 lambda is {grid[peak_idx]}")
    logging.info("=> Learning rate {}: l2 regularization strength tuning duration time: {:.2f}s".format(
        config.TRAIN.LR, time.time() - start))
    return grid[peak_idx], peak_score


def hyperparameter_sweep_lr(sweep_fn, train_dataloader, val_dataloader, config):
    logging.info("=> Start hyperparameter tuning.")
    start = time.time()
    best_score, best_lr, best_l2 = 0, 0, 0
    for lr_one in np.logspace(-6, -1, num=6).tolist():
        config.defrost()
        config.TRAIN.LR = lr_one
        config.freeze()
        l2, score = sweep_fn(train_dataloader, val_dataloader, config)
        logging.info(f"=> Learning rate: {lr_one}, best_score {score}")
        if best_score < score:
            best_score, best_lr, best_l2 = score, lr_one, l2
    logging.info(f"Hyper parameter tuning result: learning rate {best_lr}, l2_lambda {best_l2}")
    logging.info("=> Hyperparameter tuning duration time: {:.2f}s".format(time.time() - start))
    return best_lr, best_l2


def clone_loader(loader, shuffle=True):
    from .dataloader import TensorLoader
    if isinstance(loader, TensorLoader):
        return TensorLoader(loader.dataset, batch_size=loader.batch_size, shuffle=shuffle)
    return create_dataloader(loader.dataset, batch_size=loader.batch_size, shuffle=shuffle,
                             num_workers=loader.num_workers, pin_memory=loader.pin_memory)


def merge_trainval_loader(train_loader, val_loader):
    trainset, valset = train_loader.dataset, val_loader.dataset
    fullset = trainset.dataset
    assert trainset.dataset is valset.dataset
    assert len(fullset) == len(trainset) + len(valset)
    from .dataloader import TensorLoader
    if isinstance(train_loader, TensorLoader):
        return TensorLoader(fullset, batch_size=train_loader.batch_size, shuffle=True)
    return torch.utils.data.DataLoader(fullset, batch_size=train_loader.batch_size, shuffle=True,
                                       num_workers=train_loader.num_workers, pin_memory=train_loader.pin_memory,
                                       sampler=None, drop_last=False)


def final_run(train_task_fn, sweep_lr_fn, train_dataloader, val_dataloader, test_dataloader, no_hyperparameter_tuning,
              lr, l2, config):
    """Entry point of a fine-tune job (reference kadapt_clip, :488-521): optional LR x WD search on (train, val),
    then the final run on train(+val) evaluated on test with END_EPOCH += EXTRA_FINAL_TRAIN_EPOCH."""
    if no_hyperparameter_tuning:
        best_lr, best_l2 = lr, l2
    else:
        best_lr, best_l2 = sweep_lr_fn(train_dataloader, val_dataloader, config)
    logging.info("=> The final classifier is on training ...")
    logging.info(f"Hyperparameters: learning_rate = {best_lr}, l2_lambda = {best_l2}")
    config.defrost()
    config.TRAIN.LR = best_lr
    config.TRAIN.WD = best_l2
    config.TRAIN.END_EPOCH += config.TRAIN.EXTRA_FINAL_TRAIN_EPOCH
    config.freeze()
    if config.DATASET.DATASET == "patch-camelyon" and config.DATASET.NUM_SAMPLES_PER_CLASS == 10000:
        raise RuntimeError("the patch-camelyon full-set regeneration (:505-513) needs the reference's dataset layer")
    if config.DATASET.MERGE_TRAIN_VAL_FINAL_RUN:
        trainval = merge_trainval_loader(train_dataloader, val_dataloader)
        logging.info(f"Using the full trainval set to train final model. len(dataset)={len(trainval.dataset)}")
    else:
        trainval = train_dataloader
        logging.info(f"Using the train set only to train final model. len(dataset)={len(trainval.dataset)}")
    return train_task_fn(trainval, test_dataloader, config)
