"""Metrics named by ``config.TEST.METRIC`` (reference: evaluation/metric.py:7-50).

The reference delegates to the third-party ``vision_evaluation`` package (not vendored in the reference
tree, not installed here); the three evaluators it uses are restated from their published definitions:
top-1 accuracy, balanced accuracy (mean of per-class recall) and the 11-point interpolated mean average
precision of PASCAL VOC 2007.  ``roc_auc`` uses scikit-learn exactly as the reference does.
"""
import logging

import numpy as np


def accuracy(y_label, y_pred):
    """Top-1 accuracy.  y_label (N,) ints; y_pred (N, C) scores (or (N,) class ids)."""
    y_label, y_pred = np.asarray(y_label), np.asarray(y_pred)
    if y_label.ndim == 2:                       # one-hot / (N,1)
        y_label = y_label[:, 0] if y_label.shape[1] == 1 else y_label.argmax(1)
    top1 = y_pred if y_pred.ndim == 1 else y_pred.argmax(axis=1)
    return float((top1 == y_label).mean()) if y_label.size else 0.0


def balanced_accuracy_score(y_label, y_pred):
    """Mean per-class recall over the classes that occur in the targets."""
    y_label, y_pred = np.asarray(y_label), np.asarray(y_pred)
    top1 = y_pred if y_pred.ndim == 1 else y_pred.argmax(axis=1)
    recalls = [float((top1[y_label == c] == c).mean()) for c in np.unique(y_label)]
    return float(np.mean(recalls)) if recalls else 0.0


def _ap_11_points(target, score):
    order = np.argsort(-score, kind="stable")
    t = target[order] > 0
    npos = int(t.sum())
    if npos == 0:
        return 0.0
    tp = np.cumsum(t)
    prec = tp / np.arange(1, len(t) + 1)
    rec = tp / npos
    ap = 0.0
    for thr in np.linspace(0.0, 1.0, 11):
        m = rec >= thr
        ap += (prec[m].max() if m.any() else 0.0) / 11.0
    return float(ap)


def map_11_points(y_label, y_pred_proba):
    """PASCAL VOC-2007 11-point interpolated mAP over classes; y_label multi-hot (N, C) or ids (N,)."""
    y_label, y_pred_proba = np.asarray(y_label), np.asarray(y_pred_proba)
    if y_label.ndim == 1:
        y_label = np.eye(y_pred_proba.shape[1], dtype=np.int64)[y_label]
    return float(np.mean([_ap_11_points(y_label[:, c], y_pred_proba[:, c]) for c in range(y_label.shape[1])]))


def roc_auc(y_true, y_score):
    from sklearn.metrics import roc_auc_score
    if y_score.shape[1] == 2:
        return roc_auc_score(y_true, y_score[:, 1])
    return roc_auc_score(y_true, y_score)


def get_metric(metric_name):
    if metric_name == "accuracy":
        return accuracy
    if metric_name == "mean-per-class":
        return balanced_accuracy_score
    if metric_name == "11point_mAP":
        return map_11_points
    if metric_name == "roc_auc":
        return roc_auc
    logging.error("Undefined metric.")
