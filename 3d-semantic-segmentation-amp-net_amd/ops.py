"""Thin Python layer over the C ABI: pointer tables, workspaces, argument checks.

Nothing here computes: every function validates shapes/dtypes/devices, allocates outputs with torch
(device memory plumbing) and calls one entry point of libampnet_hip.so on the current HIP stream.
"""
import ctypes

import torch

from . import _lib
from . import params as P


class PointerTable:
    """Host array of device pointers in the ABI's fixed order, built from {state_dict key: GPU tensor}."""

    def __init__(self, table, tensors, what):
        self.names = list(table.keys())
        missing = [n for n in self.names if n not in tensors]
        if missing:
            raise _lib.AmpnetError(f"{what}: missing tensors {missing[:4]}...")
        self.keep = []
        arr = (ctypes.c_void_p * len(self.names))()
        for i, n in enumerate(self.names):
            t = tensors[n]
            _lib.require_gpu(t, f"{what}[{n}]")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.AmpnetError(f"{what}[{n}] must be contiguous float32")
            if t.numel() != P.numel(table[n]):
                raise _lib.AmpnetError(f"{what}[{n}] has {t.numel()} elements, expected {P.numel(table[n])}")
            arr[i] = t.data_ptr()
            self.keep.append(t)
        self.arr = arr


_OFF_CACHE = {}


def window_offsets(np_cluster, device):
    """[Q+1] int32 device prefix offsets from a list of window sizes (cached: the upload is a blocking copy)."""
    key = (tuple(int(n) for n in np_cluster), str(device))
    hit = _OFF_CACHE.get(key)
    if hit is None:
        off = [0]
        for n in key[0]:
            off.append(off[-1] + n)
        if len(_OFF_CACHE) > 64:
            _OFF_CACHE.clear()
        hit = (torch.tensor(off, dtype=torch.int32, device=device), off[-1], max(key[0]))
        _OFF_CACHE[key] = hit
    return hit


class Workspace:
    """Grow-only device scratch buffer (one per module instance)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


def encoder_forward(enc_params, enc_buffers, x, win_off, n_windows, total_rows, max_rows, n_slots, train, ws,
                    want_in_T=False):
    """x [total_rows, 9] f32 GPU -> (local [total_rows, 64], global [Q, 256], feat_T [Q, 64, 64], in_T or None).
    enc_params / enc_buffers: PointerTable.  See include/ampnet_hip.h: ampnet_encoder_fwd_f32."""
    _lib.require_gpu(x, "x")
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != P.N_FEATS or x.shape[0] != total_rows:
        raise _lib.AmpnetError(f"encoder_forward: x must be [total_rows={total_rows}, 9] float32, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    dev = x.device
    L = _lib.lib()
    L.ampnet_encoder_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_encoder_workspace_bytes(n_windows, n_slots, total_rows, max_rows, int(train))
    buf = ws.get(need, dev)
    local = torch.empty((total_rows, 64), dtype=torch.float32, device=dev)
    glob = torch.empty((n_windows, P.GLOBAL_DIM), dtype=torch.float32, device=dev)
    feat_T = torch.empty((n_windows, 64, 64), dtype=torch.float32, device=dev)
    in_T = torch.empty((n_windows, 3, 3), dtype=torch.float32, device=dev) if want_in_T else None
    with torch.cuda.device(dev):
        rc = L.ampnet_encoder_fwd_f32(enc_params.arr, enc_buffers.arr, _lib.ptr(x), _lib.ptr(win_off),
                                      n_windows, n_slots, total_rows, max_rows, int(train),
                                      _lib.ptr(local), _lib.ptr(glob), _lib.ptr(feat_T), _lib.ptr(in_T),
                                      _lib.ptr(buf), ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_encoder_fwd_f32")
    return local, glob, feat_T, in_T


def head_forward(head_params, head_buffers, gl, lo, centroids, win_off, mask, B, W, total_rows, max_rows,
                 n_classes, train, drop_p, seed, ws, targets=None, class_w=None, want_preds=False):
    """gl [B*W, 256] (row b*W+w), lo [total_rows, 64], centroids [B, W, 2], mask [B, W] bool/uint8 or None
    -> (logits [B, C, P], preds [B, P] int64 or None, loss [2] (ce, sum of weights) or None).
    See include/ampnet_hip.h: ampnet_head_fwd_f32."""
    for t, name in ((gl, "gl"), (lo, "lo"), (centroids, "centroids")):
        _lib.require_gpu(t, name)
        if t.dtype != torch.float32:
            raise _lib.AmpnetError(f"head_forward: {name} must be float32")
    Q = B * W
    if tuple(gl.shape) != (Q, P.GLOBAL_DIM) or tuple(lo.shape) != (total_rows, P.LOCAL_DIM) or tuple(centroids.shape) != (B, W, 2):
        raise _lib.AmpnetError(f"head_forward: shapes gl {tuple(gl.shape)} lo {tuple(lo.shape)} centroids {tuple(centroids.shape)} "
                               f"do not match B={B} W={W} rows={total_rows}")
    dev = gl.device
    gl, lo, centroids = gl.contiguous(), lo.contiguous(), centroids.contiguous()
    m8 = None
    if mask is not None:
        m8 = mask.to(device=dev, dtype=torch.uint8).contiguous()
        if tuple(m8.shape) != (B, W):
            raise _lib.AmpnetError(f"head_forward: mask {tuple(m8.shape)} != [B={B}, W={W}]")
    Pp = total_rows // B
    L = _lib.lib()
    L.ampnet_head_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_head_workspace_bytes(B, W, total_rows, max_rows, n_classes, int(train))
    buf = ws.get(need, dev)
    logits = torch.empty((B, n_classes, Pp), dtype=torch.float32, device=dev)
    preds = torch.empty((B, Pp), dtype=torch.int64, device=dev) if want_preds else None
    loss = None
    tg = None
    if targets is not None:
        tg = targets.to(device=dev, dtype=torch.int64).contiguous()
        if tuple(tg.shape) != (B, Pp):
            raise _lib.AmpnetError(f"head_forward: targets {tuple(tg.shape)} != [B={B}, P={Pp}]")
        loss = torch.empty(2, dtype=torch.float32, device=dev)
    cw = class_w.to(device=dev, dtype=torch.float32).contiguous() if class_w is not None else None
    with torch.cuda.device(dev):
        rc = L.ampnet_head_fwd_f32(head_params.arr, head_buffers.arr, _lib.ptr(gl), _lib.ptr(lo), _lib.ptr(centroids),
                                   _lib.ptr(win_off), _lib.ptr(m8), B, W, total_rows, max_rows, n_classes, int(train),
                                   ctypes.c_float(drop_p), ctypes.c_uint32(seed & 0xFFFFFFFF), _lib.ptr(logits),
                                   _lib.ptr(tg), _lib.ptr(cw), _lib.ptr(preds), _lib.ptr(loss), _lib.ptr(buf),
                                   ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_head_fwd_f32")
    return logits, preds, loss


def head_forward_files(head_params, head_buffers, gl, lo, centroids, win_off, mask, n_files, W, total_rows, max_rows, n_classes, ws):
    """Eval forward of the attention head for several files at once (include/ampnet_hip.h: ampnet_head_fwd_files_f32): gl [n_files*W, 256],
    lo [total_rows, 64], centroids [n_files, W, 2], win_off [n_files*W + 1] (unused slots = zero-row windows), mask [n_files, W] uint8
    -> (logits [n_classes, total_rows], preds [total_rows] int64)."""
    for t, name in ((gl, "gl"), (lo, "lo"), (centroids, "centroids"), (mask, "mask")):
        _lib.require_gpu(t, name)
    if tuple(gl.shape) != (n_files * W, P.GLOBAL_DIM) or tuple(lo.shape) != (total_rows, P.LOCAL_DIM) or tuple(centroids.shape) != (n_files, W, 2) \
            or tuple(mask.shape) != (n_files, W) or mask.dtype != torch.uint8 or win_off.numel() != n_files * W + 1:
        raise _lib.AmpnetError(f"head_forward_files: shapes gl {tuple(gl.shape)} lo {tuple(lo.shape)} centroids {tuple(centroids.shape)} "
                               f"mask {tuple(mask.shape)} do not match n_files={n_files} W={W} rows={total_rows}")
    dev = gl.device
    L = _lib.lib()
    L.ampnet_head_workspace_bytes.restype = ctypes.c_size_t
    buf = ws.get(L.ampnet_head_workspace_bytes(n_files, W, total_rows, max_rows, n_classes, 0), dev)
    logits = torch.empty((n_classes, total_rows), dtype=torch.float32, device=dev)
    preds = torch.empty(total_rows, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = L.ampnet_head_fwd_files_f32(head_params.arr, head_buffers.arr, _lib.ptr(gl.contiguous().float()), _lib.ptr(lo.contiguous().float()),
                                         _lib.ptr(centroids.contiguous().float()), _lib.ptr(win_off), _lib.ptr(mask.contiguous()), n_files, W,
                                         total_rows, max_rows, n_classes, _lib.ptr(logits), _lib.ptr(preds), _lib.ptr(buf),
                                         ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_head_fwd_files_f32")
    return logits, preds


def gru_head_forward(head_params, head_buffers, gl, lo, win_off, B, W, total_rows, max_rows, n_classes, train, drop_p, seed, ws,
                     targets=None, class_w=None, want_preds=False):
    """SegmentationWithGRU on the HIP path: gl [B*W, 256] (row b*W+w = global_seq[b, w]), lo [total_rows, 64]
    -> (logits [B, C, P], preds [B, P] int64 or None, loss [2] (ce, sum of weights) or None).
    See include/ampnet_hip.h: ampnet_gru_head_fwd_f32."""
    for t, name in ((gl, "gl"), (lo, "lo")):
        _lib.require_gpu(t, name)
        if t.dtype != torch.float32:
            raise _lib.AmpnetError(f"gru_head_forward: {name} must be float32")
    if tuple(gl.shape) != (B * W, P.GLOBAL_DIM) or tuple(lo.shape) != (total_rows, P.LOCAL_DIM):
        raise _lib.AmpnetError(f"gru_head_forward: shapes gl {tuple(gl.shape)} lo {tuple(lo.shape)} do not match B={B} W={W} rows={total_rows}")
    dev = gl.device
    gl, lo = gl.contiguous(), lo.contiguous()
    Pp = total_rows // B
    L = _lib.lib()
    L.ampnet_gru_head_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_gru_head_workspace_bytes(B, W, total_rows, max_rows, n_classes, int(train))
    buf = ws.get(need, dev)
    logits = torch.empty((B, n_classes, Pp), dtype=torch.float32, device=dev)
    preds = torch.empty((B, Pp), dtype=torch.int64, device=dev) if want_preds else None
    loss, tg, cw = None, None, None
    if targets is not None:
        tg = targets.to(device=dev, dtype=torch.int64).contiguous()
        if tuple(tg.shape) != (B, Pp):
            raise _lib.AmpnetError(f"gru_head_forward: targets {tuple(tg.shape)} != [B={B}, P={Pp}]")
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        cw = (class_w if class_w is not None else torch.ones(n_classes)).to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        rc = L.ampnet_gru_head_fwd_f32(head_params.arr, head_buffers.arr, _lib.ptr(gl), _lib.ptr(lo), _lib.ptr(win_off), B, W,
                                       total_rows, max_rows, n_classes, int(train), ctypes.c_float(drop_p),
                                       ctypes.c_uint32(seed & 0xFFFFFFFF), _lib.ptr(logits), _lib.ptr(tg), _lib.ptr(cw), _lib.ptr(preds),
                                       _lib.ptr(loss), _lib.ptr(buf), ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_gru_head_fwd_f32")
    return logits, preds, loss


def gru_head_backward(head_params, grad_table, gl, lo, win_off, B, W, total_rows, max_rows, n_classes, drop_p, seed, dlogits, fwd_ws, bwd_ws):
    """Backward of a train-mode gru_head_forward (same arguments, fwd_ws untouched since) -> (d_lo [rows, 64], d_gl [B*W, 256])."""
    dev = lo.device
    L = _lib.lib()
    L.ampnet_gru_head_bwd_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_gru_head_bwd_workspace_bytes(B, W, total_rows, max_rows, n_classes)
    buf = bwd_ws.get(need, dev)
    fbuf = fwd_ws.buf
    Pp = total_rows // B
    if tuple(dlogits.shape) != (B, n_classes, Pp) or dlogits.dtype != torch.float32 or not dlogits.is_contiguous():
        raise _lib.AmpnetError(f"gru_head_backward: dlogits must be contiguous float32 [B, C, P], got {tuple(dlogits.shape)}")
    d_lo = torch.empty((total_rows, 64), dtype=torch.float32, device=dev)
    d_gl = torch.empty((B * W, P.GLOBAL_DIM), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.ampnet_gru_head_bwd_f32(head_params.arr, grad_table.arr, _lib.ptr(gl.contiguous()), _lib.ptr(lo.contiguous()), _lib.ptr(win_off),
                                       B, W, total_rows, max_rows, n_classes, ctypes.c_float(drop_p), ctypes.c_uint32(seed & 0xFFFFFFFF),
                                       _lib.ptr(dlogits), _lib.ptr(d_lo), _lib.ptr(d_gl), _lib.ptr(fbuf), ctypes.c_size_t(fbuf.numel()),
                                       _lib.ptr(buf), ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_gru_head_bwd_f32")
    return d_lo, d_gl


def cls_head_forward(head_params, head_buffers, gl, mask, B, W, n_classes, train, drop_p, seed, ws, want_weights=True):
    """ClassificationWithAttention on the HIP path: gl [B*W, 256] (row b*W+w), mask [B, W] or None -> (out [B, C], attention weights
    [B, W, W] or None).  See include/ampnet_hip.h: ampnet_cls_head_fwd_f32."""
    _lib.require_gpu(gl, "gl")
    if gl.dtype != torch.float32 or tuple(gl.shape) != (B * W, P.GLOBAL_DIM):
        raise _lib.AmpnetError(f"cls_head_forward: gl must be float32 [B*W={B * W}, 256], got {tuple(gl.shape)} {gl.dtype}")
    dev = gl.device
    gl = gl.contiguous()
    m8 = None
    if mask is not None:
        m8 = mask.to(device=dev, dtype=torch.uint8).contiguous()
        if tuple(m8.shape) != (B, W):
            raise _lib.AmpnetError(f"cls_head_forward: mask {tuple(m8.shape)} != [B={B}, W={W}]")
    L = _lib.lib()
    L.ampnet_cls_head_workspace_bytes.restype = ctypes.c_size_t
    buf = ws.get(L.ampnet_cls_head_workspace_bytes(B, W), dev)
    out = torch.empty((B, n_classes), dtype=torch.float32, device=dev)
    aw = torch.empty((B, W, W), dtype=torch.float32, device=dev) if want_weights else None
    with torch.cuda.device(dev):
        rc = L.ampnet_cls_head_fwd_f32(head_params.arr, head_buffers.arr, _lib.ptr(gl), _lib.ptr(m8), B, W, n_classes, int(train),
                                       ctypes.c_float(drop_p), ctypes.c_uint32(seed & 0xFFFFFFFF), _lib.ptr(out), _lib.ptr(aw), _lib.ptr(buf),
                                       ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_cls_head_fwd_f32")
    return out, aw


def cls_head_backward(head_params, grad_table, gl, B, W, n_classes, drop_p, seed, d_out, fwd_ws, bwd_ws):
    """Backward of a train-mode cls_head_forward (same arguments, fwd_ws untouched since) -> d_gl [B*W, 256]."""
    dev = gl.device
    L = _lib.lib()
    L.ampnet_cls_head_bwd_workspace_bytes.restype = ctypes.c_size_t
    buf = bwd_ws.get(L.ampnet_cls_head_bwd_workspace_bytes(B, W), dev)
    fbuf = fwd_ws.buf
    if tuple(d_out.shape) != (B, n_classes) or d_out.dtype != torch.float32 or not d_out.is_contiguous():
        raise _lib.AmpnetError(f"cls_head_backward: d_out must be contiguous float32 [B, C], got {tuple(d_out.shape)}")
    d_gl = torch.empty((B * W, P.GLOBAL_DIM), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.ampnet_cls_head_bwd_f32(head_params.arr, grad_table.arr, _lib.ptr(gl.contiguous()), B, W, n_classes, ctypes.c_float(drop_p),
                                       ctypes.c_uint32(seed & 0xFFFFFFFF), _lib.ptr(d_out), _lib.ptr(d_gl), _lib.ptr(fbuf),
                                       ctypes.c_size_t(fbuf.numel()), _lib.ptr(buf), ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_cls_head_bwd_f32")
    return d_gl


def pad_mask(targets, W):
    """targets [B, P] int64 GPU (cluster-major, -1 = padded) -> uint8 [B, W]: the reference's `(targets.view(B, -1, W) == -1).all(dim=1)`
    (train_pointnet-attention.py:428-431) in one launch (include/ampnet_hip.h: ampnet_pad_mask_i64)."""
    _lib.require_gpu(targets, "targets")
    if targets.dtype != torch.int64 or targets.dim() != 2 or targets.shape[1] % W:
        raise _lib.AmpnetError(f"pad_mask: targets must be int64 [B, P] with P % W == 0, got {tuple(targets.shape)} {targets.dtype}, W={W}")
    t = targets.contiguous()
    B, Pp = t.shape
    mask = torch.empty((B, W), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        rc = _lib.lib().ampnet_pad_mask_i64(_lib.ptr(t), B, Pp, W, _lib.ptr(mask), _lib.stream_ptr(t.device))
    _lib.check(rc, "ampnet_pad_mask_i64")
    return mask


def small_gemm(A, B, trans_a=False, trans_b=False, out=None, accumulate=False, want_row_sums=False, k_scale=None):
    """C (+)= op(A) op(B) on the kernel the backward uses for its token-level products (include/ampnet_hip.h: ampnet_small_gemm_f32).
    A, B: 2-D float32 GPU tensors whose rows are contiguous (a column slice of a wider matrix is fine: the leading dimension is its stride)."""
    _lib.require_gpu(A, "A")
    _lib.require_gpu(B, "B")
    for name, x in (("A", A), ("B", B)):
        if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
            raise _lib.AmpnetError(f"small_gemm: {name} must be a 2-D float32 tensor with contiguous rows, got {tuple(x.shape)} {x.dtype} strides {x.stride()}")
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N, Kb = (B.shape[0], B.shape[1]) if trans_b else (B.shape[1], B.shape[0])
    if K != Kb:
        raise _lib.AmpnetError(f"small_gemm: inner dimensions differ ({K} vs {Kb})")
    if B.device != A.device:
        raise _lib.AmpnetError(f"small_gemm: A is on {A.device}, B on {B.device}")
    if out is None:
        if accumulate:
            raise _lib.AmpnetError("small_gemm: accumulate=True adds into `out`, which was not given")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    elif (out.dtype != torch.float32 or out.dim() != 2 or out.device != A.device or out.shape[0] < M or out.shape[1] < N or out.stride(1) != 1
          or out.stride(0) < N):
        raise _lib.AmpnetError(f"small_gemm: out must be a float32 matrix on {A.device} with contiguous rows and at least {M} x {N} elements, "
                               f"got {tuple(out.shape)} {out.dtype} on {out.device} strides {out.stride()}")
    if k_scale is not None:
        if not k_scale.is_cuda or k_scale.device != A.device or k_scale.numel() != K:
            raise _lib.AmpnetError(f"small_gemm: k_scale must hold K = {K} values on {A.device}, got {k_scale.numel()} on {k_scale.device}")
        k_scale = k_scale.contiguous().float()
    rs = torch.empty((M,), dtype=torch.float32, device=A.device) if want_row_sums else None
    with torch.cuda.device(A.device):
        vp = ctypes.c_void_p
        rc = _lib.lib().ampnet_small_gemm_f32(int(trans_a), int(trans_b), M, N, K, vp(A.data_ptr()), A.stride(0), vp(B.data_ptr()), B.stride(0),
                                              vp(out.data_ptr()), out.stride(0), int(accumulate), _lib.ptr(rs),
                                              _lib.ptr(k_scale), _lib.stream_ptr(A.device))
    _lib.check(rc, "ampnet_small_gemm_f32")
    return (out, rs) if want_row_sums else out


def reg_loss(feat_T, keep_G=False):
    """|| I - F F^T ||_F over the stack feat_T [n, 64, 64] -> device scalar tensor [1] (and G when keep_G)."""
    _lib.require_gpu(feat_T, "feat_T")
    f = feat_T.contiguous().float()
    n = f.shape[0]
    dev = f.device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    part = torch.empty(n, dtype=torch.float32, device=dev)
    G = torch.empty_like(f) if keep_G else None
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_reg_loss_fwd_f32(_lib.ptr(f), n, _lib.ptr(out), _lib.ptr(G), _lib.ptr(part), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_reg_loss_fwd_f32")
    return (out, G) if keep_G else out


def encoder_backward(enc_params, grad_table, x, win_off, n_windows, total_rows, max_rows, n_slots, local, feat_T,
                     d_local, d_global, d_feat_T, fwd_ws, bwd_ws):
    """Backward of a train-mode encoder_forward (same x / windows / n_slots; fwd_ws untouched since).
    grad_table: PointerTable over the gradient tensors (same order as the parameters); gradients are overwritten."""
    dev = x.device
    L = _lib.lib()
    L.ampnet_encoder_bwd_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_encoder_bwd_workspace_bytes(n_windows, n_slots, total_rows, max_rows)
    buf = bwd_ws.get(need, dev)
    fbuf = fwd_ws.buf
    for t, name, shape in ((d_local, "d_local", (total_rows, 64)), (d_global, "d_global", (n_windows, 256)),
                           (d_feat_T, "d_feat_T", (n_windows, 64, 64))):
        if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda):
            raise _lib.AmpnetError(f"encoder_backward: {name} must be contiguous float32 GPU {shape}, got {tuple(t.shape)} {t.dtype}")
    with torch.cuda.device(dev):
        rc = L.ampnet_encoder_bwd_f32(enc_params.arr, grad_table.arr, _lib.ptr(x), _lib.ptr(win_off), n_windows, n_slots,
                                      total_rows, max_rows, _lib.ptr(local), _lib.ptr(d_local), _lib.ptr(d_global),
                                      _lib.ptr(d_feat_T), _lib.ptr(feat_T), _lib.ptr(fbuf), ctypes.c_size_t(fbuf.numel()),
                                      _lib.ptr(buf), ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_encoder_bwd_f32")


def head_backward(head_params, grad_table, lo, centroids, win_off, B, W, total_rows, max_rows, n_classes, drop_p, seed,
                  dlogits, fwd_ws, bwd_ws):
    """Backward of a train-mode head_forward (same arguments, fwd_ws untouched since) -> (d_lo [rows, 64], d_gl [B*W, 256])."""
    dev = lo.device
    L = _lib.lib()
    L.ampnet_head_bwd_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_head_bwd_workspace_bytes(B, W, total_rows, max_rows, n_classes)
    buf = bwd_ws.get(need, dev)
    fbuf = fwd_ws.buf
    Pp = total_rows // B
    if tuple(dlogits.shape) != (B, n_classes, Pp) or dlogits.dtype != torch.float32 or not dlogits.is_contiguous():
        raise _lib.AmpnetError(f"head_backward: dlogits must be contiguous float32 [B, C, P], got {tuple(dlogits.shape)}")
    d_lo = torch.empty((total_rows, 64), dtype=torch.float32, device=dev)
    d_gl = torch.empty((B * W, 256), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.ampnet_head_bwd_f32(head_params.arr, grad_table.arr, _lib.ptr(lo.contiguous()), _lib.ptr(centroids.contiguous()),
                                   _lib.ptr(win_off), B, W, total_rows, max_rows, n_classes, ctypes.c_float(drop_p),
                                   ctypes.c_uint32(seed & 0xFFFFFFFF), _lib.ptr(dlogits), _lib.ptr(d_lo), _lib.ptr(d_gl),
                                   _lib.ptr(fbuf), ctypes.c_size_t(fbuf.numel()), _lib.ptr(buf), ctypes.c_size_t(buf.numel()),
                                   _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_head_bwd_f32")
    return d_lo, d_gl


def ce_backward(logits, targets, class_w, loss2, grad_scale=1.0):
    """dlogits = grad_scale * d(ce)/d(logits) for the weighted-mean CE returned by head_forward (loss2 = [ce, sum w])."""
    B, C, Pp = logits.shape
    d = torch.empty_like(logits)
    dev = logits.device
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_ce_bwd_f32(_lib.ptr(logits), _lib.ptr(targets), _lib.ptr(class_w), _lib.ptr(loss2),
                                          ctypes.c_float(grad_scale), B, C, Pp, _lib.ptr(d), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_ce_bwd_f32")
    return d


def reg_loss_backward_stack(feat_T, G, reg, coef, n_total):
    """-> d_stack [n_total, 64, 64]: zeros in the first n_total - n matrices, coef * d(reg)/d(feat_T) in the last n (feat_T, G [n, 64, 64]): the
    gradient of the whole stack of feature transforms in one launch (include/ampnet_hip.h: ampnet_reg_loss_bwd_stack_f32)."""
    n = feat_T.shape[0]
    dev = feat_T.device
    if not (feat_T.is_contiguous() and G.is_contiguous()) or n_total < n:
        raise _lib.AmpnetError("reg_loss_backward_stack: contiguous [n, 64, 64] tensors and n_total >= n")
    out = torch.empty((n_total, 64, 64), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_reg_loss_bwd_stack_f32(_lib.ptr(feat_T), _lib.ptr(G), _lib.ptr(reg), ctypes.c_float(coef), n, n_total, _lib.ptr(out),
                                                      _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_reg_loss_bwd_stack_f32")
    return out


def reg_loss_backward(feat_T, G, reg, coef, d_feat_T):
    """d_feat_T += coef * d(reg)/d(feat_T); feat_T, G, d_feat_T [n, 64, 64] contiguous, reg [1]."""
    n = feat_T.shape[0]
    dev = feat_T.device
    if not (feat_T.is_contiguous() and G.is_contiguous() and d_feat_T.is_contiguous()):
        raise _lib.AmpnetError("reg_loss_backward: tensors must be contiguous")
    with torch.cuda.device(dev):
        rc = _lib.lib().ampnet_reg_loss_bwd_f32(_lib.ptr(feat_T), _lib.ptr(G), _lib.ptr(reg), ctypes.c_float(coef), n,
                                                _lib.ptr(d_feat_T), _lib.stream_ptr(dev))
    _lib.check(rc, "ampnet_reg_loss_bwd_f32")
