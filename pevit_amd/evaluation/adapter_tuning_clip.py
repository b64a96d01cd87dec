"""adapter fine-tune harness with the reference's module surface (evaluation/adapter_tuning_clip.py:85-502).

Trainable tensors: 'adapter' in the name.  Implementation shared with the sibling harnesses in
:mod:`._harness`; every step of ``train_one`` executes in the HIP engine.
"""
from . import _harness as _h
from ._harness import (MULTILABEL_DATASETS, AverageMeter, accuracy, adjust_learning_rate, clone_loader, gpu_gc,  # noqa: F401
                       merge_trainval_loader, train_one, validate)
from .clip_load import *  # noqa: F401,F403  (the reference re-exports load() this way)
from .metric import get_metric  # noqa: F401
from ..optim import build_optimizer  # noqa: F401

METHOD = "adapter"


def get_cls_model(config, feature_type="image"):
    return _h.get_cls_model(METHOD, config, feature_type)


class Classifier(_h.ClassifierBase):
    """Linear classifier."""
    METHOD = METHOD


def train_task(train_dataloader, test_dataloader, config, sweep_run=False):
    return _h.train_task(Classifier, train_dataloader, test_dataloader, config, sweep_run=sweep_run)


def hyperparameter_sweep(train_dataloader, val_dataloader, config):
    return _h.hyperparameter_sweep(train_task, train_dataloader, val_dataloader, config)


def hyperparameter_sweep_lr(train_dataloader, val_dataloader, config):
    return _h.hyperparameter_sweep_lr(hyperparameter_sweep, train_dataloader, val_dataloader, config)


def adapt_clip(train_dataloader, val_dataloader, test_dataloader, no_hyperparameter_tuning, lr, l2, config):
    return _h.final_run(train_task, hyperparameter_sweep_lr, train_dataloader, val_dataloader, test_dataloader,
                        no_hyperparameter_tuning, lr, l2, config)
