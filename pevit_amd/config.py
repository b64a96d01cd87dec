"""Attribute-style configuration for the fine-tune harness.

The reference drives everything from a yacs ``CfgNode`` (config/default.py:7-234 <- dataset yaml <- model yaml
<- CLI ``opts``).  yacs is not a dependency of this build; ``CfgNode`` below implements the small part of
that interface the harness touches (attribute access, ``defrost``/``freeze``, ``merge_from_file``,
``merge_from_list``, ``clone``, ``get``), and ``default_config()`` holds the keys the hot path reads (SURVEY
section 5) at the reference's default values.  A real yacs node works just as well: the harness only uses
attribute access.
"""
from __future__ import annotations

import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def defrost(self):
        self._set_frozen(False)

    def freeze(self):
        self._set_frozen(True)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, "_frozen", self.is_frozen())
        return out

    def _merge(self, other: dict):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k]._merge(v)
            else:
                dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) else v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        """['TRAIN.LR', '0.01', 'DATASET.NUM_SAMPLES_PER_CLASS', '5', ...] like yacs."""
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError("opts must be KEY VALUE pairs")
        for key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    dict.__setitem__(node, p, CfgNode())
                node = node[p]
            if isinstance(raw, str):
                try:
                    raw = ast.literal_eval(raw)
                except (ValueError, SyntaxError):
                    pass
            dict.__setitem__(node, parts[-1], raw)


def default_config() -> CfgNode:
    c = CfgNode(dict(
        NAME="", OUTPUT_DIR="", GPUS=(0,), RANK=0, VERBOSE=True, WORKERS=4, PIN_MEMORY=True,
        CUDNN=dict(BENCHMARK=True, DETERMINISTIC=False, ENABLED=True),
        MODEL=dict(NAME="ViT-B/32", NUM_PARAMS_IN_M=0.0, AUTHOR="", PRETRAINED_DATA="", CREATION_TIME="", CLIP_FP32=False,
                   WEIGHT_FORMAT="bf16",     # engine storage of the frozen block weights: "bf16" | "fp8" (not a reference key)
                   SPEC=dict(EMBED_DIM=512, TEXT=dict(TOKENIZER="clip", CONTEXT_LENGTH=77))),
        DATASET=dict(DATASET="cifar100", ROOT="", NUM_CLASSES=100, NUM_SAMPLES_PER_CLASS=-1, RANDOM_SEED_SAMPLING=0,
                     MERGE_TRAIN_VAL_FINAL_RUN=True, CENTER_CROP=True, IMAGE_SIZE=(224,)),
        KNOWLEDGE=dict(WORDNET=dict(USE_HIERARCHY=False, USE_DEFINITION=False), WIKITIONARY=dict(USE_DEFINITION=False),
                       GPT3=dict(USE_GPT3=False)),
        INPUT=dict(MEAN=[0.48145466, 0.4578275, 0.40821073], STD=[0.26862954, 0.26130258, 0.27577711]),
        TRAIN=dict(LR=0.001, SCHEDULE=[], SEARCH_WD_LOG_LOWER=-6, SEARCH_WD_LOG_UPPER=6, FREEZE_IMAGE_BACKBONE=False,
                   TWO_LR=False, USE_CHANNEL_BN=True, INIT_HEAD_WITH_TEXT_ENCODER=False, LOGIT_SCALE_INIT="none",
                   TRAINABLE_LOGIT_SCALE=False, MERGE_ENCODER_AND_HEAD_PROJ=False, NORMALIZE_VISUAL_FEATURE=False,
                   SEARCH_RESULT_ON_LAST_EPOCH=False, OPTIMIZER="sgd", MOMENTUM=0.9, WD=0.0001, WD_SEARCH_LEFT=False,
                   SWEEP_CONCURRENCY=2,      # sweep runs at a time, each on its own stream (not a reference key; 1 = sequential)
                   WITHOUT_WD_LIST=[], NESTEROV=True, BEGIN_EPOCH=0, END_EPOCH=100, EXTRA_FINAL_TRAIN_EPOCH=0,
                   EMULATE_ZERO_SHOT=False, BATCH_SIZE_PER_GPU=32, SHUFFLE=True, RMSPROP_ALPHA=0.99, RMSPROP_CENTERED=False),
        TEST=dict(BATCH_SIZE_PER_GPU=32, METRIC="accuracy", MODEL_FILE=""),
    ))
    return c


def vitb32_clip_config() -> CfgNode:
    """default_config() + resources/model/vitb32_CLIP.yaml + resources/datasets/cifar100.yaml of the reference."""
    c = default_config()
    c.MODEL.NAME = "ViT-B/32"
    c.MODEL.SPEC.EMBED_DIM = 512
    c.TRAIN.BATCH_SIZE_PER_GPU = 64
    c.TRAIN.END_EPOCH = 10
    c.TRAIN.EXTRA_FINAL_TRAIN_EPOCH = 40
    c.TRAIN.WD = 0.0
    c.TRAIN.NESTEROV = False
    c.TEST.BATCH_SIZE_PER_GPU = 128
    return c


# module-level node + update_config(config, args), as ``from vision_benchmark.config import config, update_config``
config = default_config()


def update_config(config, args):
    """args.cfg (yaml, with optional BASE includes) then args.opts (KEY VALUE ...), config/default.py:236-272."""
    import os.path as op

    def from_file(path):
        with open(path) as f:
            y = yaml.safe_load(f) or {}
        for base in y.get("BASE", [""]) or [""]:
            if base:
                from_file(op.join(op.dirname(path), base))
        y.pop("BASE", None)
        config._merge(y)

    config.defrost()
    from_file(args.cfg)
    config.merge_from_list(getattr(args, "opts", None))
    # config/default.py:257,260: the learning rate is scaled by the world size and the rank recorded (utils/comm.py:11-33: both read
    # torch.distributed, 1 / 0 in every run of the reference, which never initialises a process group).  Under this build's data
    # parallelism the gradients of the ranks are AVERAGED (1 / world folded into the SGD kernel), i.e. the global batch is world x B
    # at world x LR: the reference's own linear scaling rule, applied by the same line.
    world, rank = distributed_world()
    config.TRAIN.LR = config.TRAIN.LR * world
    config.RANK = rank
    config.NAME = op.splitext(op.basename(args.cfg))[0] + config.get("NAME", "")
    config.freeze()


def distributed_world():
    """(world size, rank) of the default process group; (1, 0) without one (utils/comm.py:11-33)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except Exception:
        pass
    return 1, 0
