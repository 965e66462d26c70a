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


# ---- the same quotients from confusion counts taken on the device (include/ampnet_hip.h: ampnet_confusion_i64) ----------------------
def confusion_device(preds, targets, n_classes=5):
    """preds / targets: int64 GPU tensors of equal numel (-1 targets = padding) -> int64 GPU tensor [C*C + 1]:
    counts[t * C + p] and, last, the number of ignored points.  No synchronisation."""
    import ctypes
    from .. import _lib
    _lib.require_gpu(preds, "preds")
    _lib.require_gpu(targets, "targets")
    p = preds.reshape(-1).contiguous()
    t = targets.reshape(-1).contiguous()
    if p.dtype != torch.int64 or t.dtype != torch.int64 or p.numel() != t.numel():
        raise _lib.AmpnetError(f"confusion_device: int64 tensors of equal size expected, got {p.dtype} {tuple(p.shape)} / {t.dtype} {tuple(t.shape)}")
    out = torch.empty(n_classes * n_classes + 1, dtype=torch.int64, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.lib().ampnet_confusion_i64(_lib.ptr(p), _lib.ptr(t), ctypes.c_longlong(p.numel()), n_classes, _lib.ptr(out), _lib.stream_ptr(p.device))
    _lib.check(rc, "ampnet_confusion_i64")
    return out


def metrics_from_confusion(counts, n_classes=5):
    """counts: [C*C + 1] integers (numpy / CPU) -> (accuracy, [IoU per label]) with the float32 quotients of get_accuracy / get_iou_obj
    applied after rm_padding."""
    c = np.asarray(counts, dtype=np.int64)[: n_classes * n_classes].reshape(n_classes, n_classes)
    kept = c.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = float(np.float32(np.trace(c)) / np.float32(kept))
        ious = [float(np.float32(c[k, k]) / np.float32(c[k, :].sum() + (c[:, k].sum() - c[k, k]))) for k in range(n_classes)]
    return acc, ious
