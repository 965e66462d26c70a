"""The per-batch step of the GRU variant on the HIP path.

`train_loop` keeps the name, argument list and return value of the reference's pointNet/rnn/train_pointnetGRU.py:335-441.  That loop
has no augmentation and no key-padding mask: W encoder calls (BatchNorm statistics per cluster slot), the window tokens as a
[B, W, 256] sequence through SegmentationWithGRU, an UNWEIGHTED CrossEntropyLoss(ignore_index=-1), the regularisation term on the
feature transforms of the last encoder call, two Adam optimisers.  Here all B*W windows run through one encoder launch sequence
(n_slots = W), the GRU head through ampnet_gru_head_fwd_f32 / _bwd_f32, and loss + argmax are fused as in amp_step.py."""
import ctypes

import numpy as np
import torch

from .. import _lib, ops
from .amp_step import _class_weights, _download

GLOBAL_FEAT_SIZE = 256
HIDDEN_SIZE = 64


def relayout_batch(pc_w, targets, device):
    """[B, N, 9, W] / [B, N, W] as collate_seq_padd returns them -> device x [B, W, N, 9] f32, t [B, W, N] i64 (ampnet_augment_f32 with
    the identity permutations and no rotation: the GRU loop feeds the windows as they are, train_pointnetGRU.py:386-388)."""
    pc = torch.as_tensor(pc_w)
    tg = torch.as_tensor(targets)
    B, N, D, W = pc.shape
    if D != 9 or pc.dtype != torch.float32 or tg.dtype != torch.int64 or tuple(tg.shape) != (B, N, W):
        raise _lib.AmpnetError(f"relayout_batch: expected pc [B, N, 9, W] f32 and targets [B, N, W] i64, got {tuple(pc.shape)} {pc.dtype} / "
                               f"{tuple(tg.shape)} {tg.dtype}")
    dev = torch.device(device)
    pcd = pc.to(dev, non_blocking=True).contiguous()
    tgd = tg.to(dev, non_blocking=True).contiguous()
    ident = torch.arange(W, dtype=torch.int32, device=dev)
    x = torch.empty((B, W, N, 9), dtype=torch.float32, device=dev)
    t = torch.empty((B, W, N), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_augment_f32(_lib.ptr(pcd), _lib.ptr(tgd), _lib.ptr(ident), None, ctypes.c_double(1.0), ctypes.c_double(0.0), 0,
                                           B, N, W, _lib.ptr(x), _lib.ptr(t), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_augment_f32")
    return x, t


def forward_batch(pointnet, gru_model, x, t, class_w=None, want_loss=True, want_preds=True):
    """x [B, W, N, 9] f32, t [B, W, N] i64 on the device -> dict(logits [B, C, W*N], preds, ce [2] or None, feat_last, targets_pc)."""
    B, W, N, _ = x.shape
    targets_pc = t.reshape(B, W * N)
    n_slots = W if pointnet.training else 1
    local, glob, feat_T = pointnet.forward_windows(x.reshape(B * W, N, 9), n_slots=n_slots)
    logits, preds, loss = gru_model.forward_rows(glob, local, [N] * W, B, targets=targets_pc if want_loss else None, class_w=class_w,
                                                 want_preds=want_preds)
    # reg loss on the transforms of the LAST encoder call = cluster slot W-1 (train_pointnetGRU.py:428-429)
    feat_last = feat_T[-B:] if pointnet.training else feat_T.view(B, W, 64, 64)[:, W - 1].contiguous()
    return dict(logits=logits, preds=preds, ce=loss, feat_last=feat_last, targets_pc=targets_pc, B=B)


def train_loop(data, optimizer_rnn, optimizer_pred, ce_loss, pointnet, gru_model, w_tensorboard=None, task='classification', train=True,
               c_weights=None, epoch=0, last_epoch=0, first_batch_val=False, device_outputs=False):
    """Drop-in for the reference's train_loop (segmentation task; its classification branch never assigns `logits`,
    train_pointnetGRU.py:405-407, i.e. it cannot run there either).
    Returns (metrics {'ce_loss', 'reg_loss', 'loss'}, targets_pc [B, W*N] cpu, preds [B, W*N] cpu, last_epoch)."""
    if task != 'segmentation':
        raise NotImplementedError("the reference's GRU train_loop only reaches a model call for task='segmentation'")
    pc_w, targets, filenames = data[0], data[1], data[2]
    optimizer_rnn.zero_grad()
    optimizer_pred.zero_grad()
    pointnet.train(train)
    gru_model.train(train)
    dev = next(pointnet.parameters()).device
    x, t = relayout_batch(pc_w, targets, dev)
    cw = _class_weights(ce_loss, dev)
    metrics = {}
    if train:
        from ..trainer import fused_train_step
        out = fused_train_step(pointnet, gru_model, optimizer_rnn, optimizer_pred, x, t, None, cw)
    else:
        with torch.no_grad():
            out = forward_batch(pointnet, gru_model, x, t, cw)
        out["reg"] = ops.reg_loss(out["feat_last"])
    metrics['ce_loss'] = out["ce"][0].view(-1, 1)
    metrics['reg_loss'] = out["reg"]
    metrics['loss'] = metrics['ce_loss'] + 0.001 * metrics['reg_loss'] if train else metrics['ce_loss']
    if device_outputs:                        # not in the reference: nothing leaves the GPU, nothing synchronises (amp_step.train_loop)
        return metrics, out["targets_pc"], out["preds"], last_epoch
    return metrics, _download(out["targets_pc"], 0), _download(out["preds"], 1), last_epoch
