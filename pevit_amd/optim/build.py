"""Optimizer construction with the reference's grouping rules (optim/build.py:18-85,107-170).

The object returned is a stock ``torch.optim`` optimizer over the *same Parameter objects* the model
exposes, so schedulers / ``adjust_learning_rate`` keep working on ``param_groups``.  When the
configuration is the plain one the reference's yaml files use (SGD, momentum, no Nesterov, one weight
decay for every trainable tensor) the harness executes the update with the fused HIP kernel
(``pevit_sgd_step``) and reads ``lr`` / ``momentum`` / ``weight_decay`` from ``param_groups`` each step;
any other configuration runs this optimizer's own ``step()`` on the gradients the engine produced.
"""
import torch.nn as nn
import torch.optim as optim


def _is_depthwise(m):
    return isinstance(m, nn.Conv2d) and m.groups == m.in_channels and m.groups == m.out_channels


def _set_wd(cfg, model):
    """Two groups: [with decay, without decay]; membership by cfg.TRAIN.WITHOUT_WD_LIST
    ('depthwise' | 'bn' | 'gn' | 'ln' | 'bias'), frozen parameters skipped (build.py:18-85)."""
    no_wd = list(cfg.TRAIN.WITHOUT_WD_LIST)
    exempt = []
    for m in model.modules():
        if _is_depthwise(m) and "depthwise" in no_wd:
            exempt.append(m.weight)
        elif isinstance(m, nn.BatchNorm2d) and "bn" in no_wd:
            exempt += [m.weight, m.bias]
        elif isinstance(m, nn.GroupNorm) and "gn" in no_wd:
            exempt += [m.weight, m.bias]
        elif isinstance(m, nn.LayerNorm) and "ln" in no_wd:
            exempt += [m.weight, m.bias]
    skip = model.no_weight_decay() if hasattr(model, "no_weight_decay") else {}
    with_decay, without_decay = [], []
    for n, p in model.named_parameters():
        if p.requires_grad is False:
            continue
        if n in skip or any(p is q for q in exempt) or ("bias" in no_wd and n.endswith(".bias")):
            without_decay.append(p)
        else:
            with_decay.append(p)
    return [{"params": with_decay}, {"params": without_decay, "weight_decay": 0.0}]


def _trunk_head(model):
    trunk = [p for n, p in model.named_parameters() if "backbone" in n]
    head = [p for n, p in model.named_parameters() if "backbone" not in n]
    return trunk, head


def build_optimizer(cfg, model):
    name = cfg.TRAIN.OPTIMIZER
    if name == "timm":
        raise RuntimeError("TRAIN.OPTIMIZER == 'timm' needs the timm package, which this build does not depend on")
    params = _set_wd(cfg, model)
    two_lr = bool(getattr(cfg.TRAIN, "TWO_LR", False))
    if name == "sgd":
        if two_lr:
            trunk, head = _trunk_head(model)
            return optim.SGD([{"params": trunk}, {"params": head, "lr": cfg.TRAIN.LR}], lr=cfg.TRAIN.LR * 0.1,
                             momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD, nesterov=cfg.TRAIN.NESTEROV)
        return optim.SGD(params, lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD,
                         nesterov=cfg.TRAIN.NESTEROV)
    if name == "adam":
        if two_lr:
            trunk, head = _trunk_head(model)
            return optim.Adam([{"params": trunk}, {"params": head, "lr": cfg.TRAIN.LR}], lr=cfg.TRAIN.LR * 0.1,
                              weight_decay=cfg.TRAIN.WD)
        return optim.Adam(params, lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    if name == "adamW":
        return optim.AdamW(params, lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    if name == "rmsprop":
        return optim.RMSprop(params, lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD,
                             alpha=cfg.TRAIN.RMSPROP_ALPHA, centered=cfg.TRAIN.RMSPROP_CENTERED)
    return None
