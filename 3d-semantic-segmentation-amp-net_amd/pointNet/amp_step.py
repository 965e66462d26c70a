"""The per-batch step of AMP-Net training / validation on the HIP path.

`train_loop` keeps the name, argument list and return value of the reference's
pointNet/self-attention/train_pointnet-attention.py:337-475 and draws from numpy's global RNG in the same
order (cluster permutation, angle, one point permutation per window), so a seeded run sees the same
augmented batch as the reference.  What changes is the execution: the reference runs the encoder W times
in a Python loop with 9 H2D copies, a repeat/cat loop and a CPU argmax; here the augmented windows go up
in one copy, all B*W windows run through one encoder launch sequence with per-slot BatchNorm statistics,
the head never materialises the 320-channel tensor, and loss + argmax are fused into the last kernel.
"""
import os

import numpy as np
import torch

from .. import _lib
from ..utils.utils import shuffle_clusters

GLOBAL_FEAT_SIZE = 256


def augment_batch(pc_clusters, targets, train):
    """Host-side augmentation with the reference's draws in the reference's order
    (train_pointnet-attention.py:390-405; utils/utils.py:582-632).
    pc_clusters [B, N, 9, W] tensor, targets [B, N, W] tensor -> numpy x [B, W, N, 9] f32, t [B, W, N] i64."""
    pc_clusters, targets = shuffle_clusters(pc_clusters, targets)
    r_angle = np.random.uniform() * 2 * np.pi
    pc = pc_clusters.numpy() if isinstance(pc_clusters, torch.Tensor) else np.asarray(pc_clusters)
    tg = targets.numpy() if isinstance(targets, torch.Tensor) else np.asarray(targets)
    B, N, D, W = pc.shape
    x = np.ascontiguousarray(np.transpose(pc, (0, 3, 1, 2)), dtype=np.float32)      # [B, W, N, 9]
    t = np.ascontiguousarray(np.transpose(tg, (0, 2, 1)))                           # [B, W, N]
    if train:
        c, s = np.cos(r_angle), np.sin(r_angle)
        rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
        xyz = x[..., :3]
        x[..., :3] = np.dot(xyz.reshape(-1, 3), rot).astype(np.float32).reshape(xyz.shape)   # float64 product, as the reference
        for w in range(W):                         # one permutation per window, shared by the batch
            idx = np.arange(N)
            np.random.shuffle(idx)
            x[:, w] = x[:, w][:, idx]
            t[:, w] = t[:, w][:, idx]
    return x, t


def augment_batch_device(pc_clusters, targets, train, device):
    """augment_batch on the GPU (include/ampnet_hip.h: ampnet_augment_f32): the collated batch goes up in one copy and one
    kernel applies cluster permutation, z-rotation (float64, as numpy computes it), the per-window point permutations and the
    re-layout.  The draws come from numpy's global RNG in the reference's order -- cluster permutation, angle, then (train
    only) one point permutation per window -- so a seeded run sees the same batch as augment_batch / the reference.
    pc_clusters [B, N, 9, W], targets [B, N, W] (host or device tensors) -> device x [B, W, N, 9] f32, t [B, W, N] i64."""
    import ctypes
    pc = torch.as_tensor(pc_clusters)
    tg = torch.as_tensor(targets)
    B, N, D, W = pc.shape
    if D != 9 or pc.dtype != torch.float32 or tg.dtype != torch.int64 or tuple(tg.shape) != (B, N, W):
        raise _lib.AmpnetError(f"augment_batch_device: expected pc [B, N, 9, W] f32 and targets [B, N, W] i64, got {tuple(pc.shape)} {pc.dtype} / {tuple(tg.shape)} {tg.dtype}")
    cperm = np.arange(W)
    np.random.shuffle(cperm)                                   # shuffle_clusters
    r_angle = np.random.uniform() * 2 * np.pi
    pperm = None
    if train:
        pperm = np.empty((W, N), dtype=np.int32)
        for w in range(W):                                     # shuffle_data: one permutation per window
            idx = np.arange(N)
            np.random.shuffle(idx)
            pperm[w] = idx
    dev = torch.device(device)
    pcd = pc.to(dev, non_blocking=True).contiguous()
    tgd = tg.to(dev, non_blocking=True).contiguous()
    cpd = torch.from_numpy(cperm.astype(np.int32)).to(dev, non_blocking=True)
    ppd = torch.from_numpy(pperm).to(dev, non_blocking=True) if pperm is not None else None
    x = torch.empty((B, W, N, 9), dtype=torch.float32, device=dev)
    t = torch.empty((B, W, N), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_augment_f32(_lib.ptr(pcd), _lib.ptr(tgd), _lib.ptr(cpd), _lib.ptr(ppd),
                                           ctypes.c_double(float(np.cos(r_angle))), ctypes.c_double(float(np.sin(r_angle))),
                                           1 if train else 0, B, N, W, _lib.ptr(x), _lib.ptr(t), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_augment_f32")
    return x, t


def augment_ragged_device(rb, train, device):
    """augment_batch_device for a collate_fns.RaggedBatch: the same draws from numpy's RNG in the same order, ONE upload of the ragged
    samples (less than half the bytes of the padded batch) and one kernel that resamples, pads, permutes, rotates and re-lays-out
    (include/ampnet_hip.h: ampnet_collate_augment_f32) -> device x [B, W, N, 9] f32, t [B, W, N] i64, bit-identical to
    augment_batch_device(*rb.to_padded()) (tests/test_augment_gpu.py)."""
    import ctypes
    B, N, W = len(rb), rb.n_points, rb.n_windows
    cperm = np.arange(W)
    np.random.shuffle(cperm)                                   # shuffle_clusters
    r_angle = np.random.uniform() * 2 * np.pi
    pperm = None
    if train:
        pperm = np.empty((W, N), dtype=np.int32)
        for w in range(W):                                     # shuffle_data: one permutation per window
            idx = np.arange(N)
            np.random.shuffle(idx)
            pperm[w] = idx
    dev = torch.device(device)
    rbd = rb if rb.is_cuda else rb.to(dev, non_blocking=True)
    cpd = torch.from_numpy(cperm.astype(np.int32)).to(dev, non_blocking=True)
    ppd = torch.from_numpy(pperm).to(dev, non_blocking=True) if pperm is not None else None
    x = torch.empty((B, W, N, 9), dtype=torch.float32, device=dev)
    t = torch.empty((B, W, N), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_collate_augment_f32(_lib.ptr(rbd.pts), _lib.ptr(rbd.lab), _lib.ptr(rbd.idx), _lib.ptr(rbd.meta), _lib.ptr(cpd), _lib.ptr(ppd),
                                                   ctypes.c_double(float(np.cos(r_angle))), ctypes.c_double(float(np.sin(r_angle))),
                                                   1 if train else 0, B, N, W, _lib.ptr(x), _lib.ptr(t), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_collate_augment_f32")
    return x, t


def forward_batch(pointnet, att_net, x, t, centroids, class_w=None, want_loss=True, want_preds=True):
    """x [B, W, N, 9] f32 (host or device), t [B, W, N] i64, centroids [B, W, 2].
    Returns dict(logits [B, C, W*N], preds [B, W*N], ce (device scalar tensor [2] or None), feat_T, targets_pc)."""
    dev = next(pointnet.parameters()).device
    if dev.type != "cuda":
        raise _lib.AmpnetError("the AMP-Net HIP path needs the model on the GPU")
    xd = torch.as_tensor(x).to(dev, non_blocking=True)
    B, W, N, _ = xd.shape
    targets_pc = torch.as_tensor(t).reshape(B, W * N)
    tgd = targets_pc.to(dev, non_blocking=True)
    cent = torch.as_tensor(centroids).to(dev).float()
    n_slots = W if pointnet.training else 1
    local, glob, feat_T = pointnet.forward_windows(xd.reshape(B * W, N, 9), n_slots=n_slots)
    # key_padding_mask exactly as the reference builds it (train_pointnet-attention.py:428-431): a view(B, -1, W)
    # of the cluster-concatenated targets, NOT a per-cluster test (SURVEY.md F-notes; restated literally).
    from .. import ops
    mask = ops.pad_mask(tgd.long(), W)                    # == (tgd.view(B, -1, W) == -1).all(dim=1), one launch (ampnet_pad_mask_i64)
    logits, preds, loss = att_net.forward_rows(glob, local, cent, [N] * W, mask,
                                               targets=tgd if want_loss else None, class_w=class_w, want_preds=want_preds)
    # the reference's reg loss uses the transforms of the LAST encoder call = cluster slot W-1
    # (train_pointnet-attention.py:463-464): train mode returns them slot-major, eval mode window-major
    feat_last = feat_T[-B:] if pointnet.training else feat_T.view(B, W, 64, 64)[:, W - 1].contiguous()
    return dict(logits=logits, preds=preds, ce=loss, feat_T=feat_T, feat_last=feat_last, targets_pc=targets_pc,
                n_slots=n_slots, B=B)


def _class_weights(ce_loss, device):
    w = getattr(ce_loss, "weight", None)
    if getattr(ce_loss, "ignore_index", -1) != -1 or getattr(ce_loss, "reduction", "mean") != "mean":
        raise _lib.AmpnetError("train_loop: the fused loss implements CrossEntropyLoss(weight, reduction='mean', ignore_index=-1)")
    return None if w is None else w.to(device)


def train_loop(data, optimizer_pointnet, optimizer_att, ce_loss, pointnet, att_net,
               w_tensorboard=None, task='classification', train=True, epoch=0, last_epoch=0, first_batch_val=False, device_outputs=False):
    """Drop-in for the reference's train_loop (segmentation task).
    Returns (metrics {'ce_loss', 'reg_loss', 'loss'}, targets_pc [B, W*N] cpu, preds [B, W*N] cpu, last_epoch).
    device_outputs=True (not in the reference): targets and predictions stay on the GPU and nothing synchronises -- for drivers that
    take their metrics on the device (utils.get_metrics.confusion_device)."""
    if task != 'segmentation':
        raise NotImplementedError("only the segmentation task is on the AMP-Net hot path")
    from .. import ops
    pc_clusters, targets, filenames, centroids = data
    optimizer_pointnet.zero_grad()
    optimizer_att.zero_grad()
    pointnet.train(train)
    att_net.train(train)
    dev = next(pointnet.parameters()).device
    from .collate_fns import RaggedBatch
    ragged = isinstance(pc_clusters, RaggedBatch)                    # collate_seq_ragged: resampling / padding happen in the kernel
    if os.environ.get("AMPNET_HOST_AUG") == "1":
        if ragged:
            pc_clusters, targets = pc_clusters.to_padded()
        x, t = augment_batch(pc_clusters, targets, train)            # the numpy path (same draws, same batch)
    elif ragged:
        x, t = augment_ragged_device(pc_clusters, train, dev)
    else:
        x, t = augment_batch_device(pc_clusters, targets, train, dev)
    cw = _class_weights(ce_loss, dev)
    metrics = {}
    if train:
        from ..trainer import fused_train_step
        out = fused_train_step(pointnet, att_net, optimizer_pointnet, optimizer_att, x, t, centroids, cw)
    else:
        with torch.no_grad():
            out = forward_batch(pointnet, att_net, x, t, centroids, cw)
        out["reg"] = ops.reg_loss(out["feat_last"])
    metrics['ce_loss'] = out["ce"][0].view(-1, 1)
    metrics['reg_loss'] = out["reg"]
    metrics['loss'] = metrics['ce_loss'] + 0.001 * metrics['reg_loss'] if train else metrics['ce_loss']
    if device_outputs:
        return metrics, out["targets_pc"].to(dev), out["preds"], last_epoch
    return metrics, _download(out["targets_pc"], 0), _download(out["preds"], 1), last_epoch


_PINNED = {}


def _download(t, slot):
    """Device -> host through page-locked buffers (2 x 9.4 MB of int64 per step at B = 64: a pageable .cpu() costs several ms).
    Two generations per slot alternate, so the tensors of one train_loop call stay valid while the next call runs; a caller that keeps
    them longer must clone them (the reference consumes them within the step, train_pointnet-attention.py:217-240)."""
    if not t.is_cuda:
        return t
    key = (slot, tuple(t.shape), t.dtype)
    ring = _PINNED.get(key)
    if ring is None:
        if len(_PINNED) > 16:
            _PINNED.clear()
        ring = _PINNED[key] = [[torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for _ in range(2)], 0]
    buf = ring[0][ring[1]]
    ring[1] ^= 1
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf
