"""Metrics with the reference's names and definitions (utils/get_metrics.py:6-31): they define "identical mIoU".
get_iou_obj = TP / (GT_pos + FP) for one label over a flat batch, as a float32 quotient like the reference's
(an int64 torch tensor over a numpy integer); accuracy = mean(pred == target), float32 quotient as well."""
import numpy as np
import torch


def get_iou_obj(pc_preds, targets, label=1):
    preds = np.asarray(torch.as_tensor(pc_preds).cpu()).reshape(-1)
    tg = np.asarray(torch.as_tensor(targets).cpu()).reshape(-1)
    detected = preds == label
    tp = np.logical_and(detected, preds == tg).sum()
    fp = detected.sum() - tp
    gt_positive = (tg == label).sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float32(tp) / np.float32(gt_positive + fp))


def get_accuracy(preds, targets, metrics, task, c_weights=None):
    if task == 'classification':
        raise NotImplementedError("only the segmentation task is on the AMP-Net hot path")
    preds = np.asarray(torch.as_tensor(preds).cpu()).reshape(-1)
    tg = np.asarray(torch.as_tensor(targets).cpu()).reshape(-1)
    metrics['accuracy'] = float(np.float32((preds == tg).sum()) / np.float32(len(preds)))
    metrics['accuracy_w'] = None
    return metrics
