"""Shared machinery of the two baseline PointNet drop-ins (pointnet.py, light_pointnet_256.py): parameter holders with
the reference's state_dict keys, the eval forward (ampnet_pointnet_seg_fwd_f32, csrc/baseline.hip) and the train-mode
forward / backward (ampnet_pointnet_seg_train_fwd_f32 / ampnet_pointnet_seg_bwd_f32, csrc/baseline_train.hip) behind a
torch.autograd.Function, so that the reference's own loop -- logits, feat_T = net(pc); loss.backward(); optimizer.step()
(pointNet/baseline/train_segmentation.py:274-328) -- runs on the HIP path.  SURVEY row a12 / BASELINE.json config 1: the
reference's plumbing case, built for parity, not tuned."""
import ctypes

import torch
import torch.nn as nn

from ... import _lib, ops
from .pointnetAtt import _BN, _Conv, _Linear

N_LAYERS = 21          # AMPNET_POINTNET_LAYERS


class TnetHolder(nn.Module):
    def __init__(self, input_dim, output_dim, G, F1, F2, bias, device):
        super().__init__()
        self.output_dim = output_dim
        self.conv_1 = _Conv(input_dim, 64, bias, device)
        self.conv_2 = _Conv(64, 128, bias, device)
        self.conv_3 = _Conv(128, G, bias, device)
        self.bn_1, self.bn_2, self.bn_3 = _BN(64, device), _BN(128, device), _BN(G, device)
        self.bn_4, self.bn_5 = _BN(F1, device), _BN(F2, device)
        self.fc_1 = _Linear(G, F1, bias, device)
        self.fc_2 = _Linear(F1, F2, bias, device)
        self.fc_3 = _Linear(F2, output_dim * output_dim, True, device)

    def forward(self, x):
        raise _lib.AmpnetError("TransformationNet runs inside SegmentationPointNet's HIP launch sequence")

    def layers(self):
        return [(self.conv_1, self.bn_1), (self.conv_2, self.bn_2), (self.conv_3, self.bn_3),
                (self.fc_1, self.bn_4), (self.fc_2, self.bn_5), (self.fc_3, None)]


class BaseHolder(nn.Module):
    def __init__(self, point_dimension, return_local_features, G, F1, F2, bias, device):
        super().__init__()
        self.return_local_features = return_local_features
        self.input_transform = TnetHolder(point_dimension, point_dimension, G, F1, F2, bias, device)
        self.feature_transform = TnetHolder(64, 64, G, F1, F2, bias, device)
        self.conv_1 = _Conv(9, 64, bias, device)
        self.conv_2 = _Conv(64, 64, bias, device)
        self.conv_3 = _Conv(64, 64, bias, device)
        self.conv_4 = _Conv(64, 128, bias, device)
        self.conv_5 = _Conv(128, G, bias, device)
        self.bn_1, self.bn_2, self.bn_3 = _BN(64, device), _BN(64, device), _BN(64, device)
        self.bn_4, self.bn_5 = _BN(128, device), _BN(G, device)

    def forward(self, x):
        raise _lib.AmpnetError("the baseline BasePointNet runs inside SegmentationPointNet's HIP launch sequence")

    def layers(self):
        return (self.input_transform.layers() + self.feature_transform.layers()
                + [(self.conv_1, self.bn_1), (self.conv_2, self.bn_2), (self.conv_3, self.bn_3),
                   (self.conv_4, self.bn_4), (self.conv_5, self.bn_5)])


class SegHolder(nn.Module):
    """Subclasses set VARIANT / T_DIM / widths and build self.base_pointnet + conv_1..4, bn_1..3."""
    VARIANT = None
    T_DIM = None

    def _init_head(self, num_classes, G, H1, H2, H3, device):
        self.num_classes = num_classes
        self.conv_1 = _Conv(G + 64, H1, True, device)
        self.conv_2 = _Conv(H1, H2, True, device)
        self.conv_3 = _Conv(H2, H3, True, device)
        self.conv_4 = _Conv(H3, num_classes, True, device)
        self.bn_1, self.bn_2, self.bn_3 = _BN(H1, device), _BN(H2, device), _BN(H3, device)
        self._ws = ops.Workspace()
        self._key, self._arr, self._keep = None, None, None

    def _layer_table(self):
        layers = self.base_pointnet.layers() + [(self.conv_1, self.bn_1), (self.conv_2, self.bn_2),
                                                (self.conv_3, self.bn_3), (self.conv_4, None)]
        assert len(layers) == N_LAYERS
        tensors = []
        for lin, bn in layers:
            tensors += [lin.weight, getattr(lin, "bias", None)]
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [None] * 4
        key = tuple(0 if t is None else t.data_ptr() for t in tensors)
        if key != self._key:
            arr = (ctypes.c_void_p * (6 * N_LAYERS))()
            for i, t in enumerate(tensors):
                if t is not None:
                    _lib.require_gpu(t, "SegmentationPointNet parameter")
                    if t.dtype != torch.float32 or not t.is_contiguous():
                        raise _lib.AmpnetError("SegmentationPointNet parameters must be contiguous float32")
                arr[i] = None if t is None else t.data_ptr()
            self._key, self._arr, self._keep = key, arr, tensors
        return self._arr

    def _train_tables(self):
        """(parameters in autograd order, layer pointer table, a function that builds the gradient pointer table)."""
        layers = self.base_pointnet.layers() + [(self.conv_1, self.bn_1), (self.conv_2, self.bn_2),
                                                (self.conv_3, self.bn_3), (self.conv_4, None)]
        params, slots = [], []                      # slots[i] = indices into params of (weight, bias, bn.weight, bn.bias) or -1
        for lin, bn in layers:
            idx = []
            for t in (lin.weight, getattr(lin, "bias", None), None if bn is None else bn.weight, None if bn is None else bn.bias):
                if t is None:
                    idx.append(-1)
                else:
                    idx.append(len(params))
                    params.append(t)
            slots.append(idx)
        return params, slots

    def forward(self, x):
        """x [B, N, 9] -> (logits [B, num_classes, N], feature_transform [B, 64, 64])."""
        if self.training:
            _lib.require_gpu(x, "x")
            if x.dim() != 3 or x.shape[2] != 9 or x.dtype != torch.float32:
                raise _lib.AmpnetError(f"SegmentationPointNet: x must be [B, N, 9] float32, got {tuple(x.shape)} {x.dtype}")
            params, slots = self._train_tables()
            return _BaselineFn.apply(self, slots, x.contiguous(), *params)
        _lib.require_gpu(x, "x")
        if x.dim() != 3 or x.shape[2] != 9 or x.dtype != torch.float32:
            raise _lib.AmpnetError(f"SegmentationPointNet: x must be [B, N, 9] float32, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B, N, _ = x.shape
        dev = x.device
        L = _lib.lib()
        L.ampnet_pointnet_seg_workspace_bytes.restype = ctypes.c_size_t
        need = L.ampnet_pointnet_seg_workspace_bytes(self.VARIANT, B, N, self.num_classes)
        buf = self._ws.get(need, dev)
        logits = torch.empty((B, self.num_classes, N), dtype=torch.float32, device=dev)
        feat_T = torch.empty((B, 64, 64), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.ampnet_pointnet_seg_fwd_f32(self._layer_table(), self.VARIANT, _lib.ptr(x), B, N, self.num_classes,
                                               _lib.ptr(logits), _lib.ptr(feat_T), _lib.ptr(buf),
                                               ctypes.c_size_t(buf.numel()), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_pointnet_seg_fwd_f32")
        return logits, feat_T


class _BaselineFn(torch.autograd.Function):
    """Train-mode forward (batch statistics, running-statistics update) and backward of SegmentationPointNet through the C ABI.
    Every forward owns its tape (workspace) until its backward has run."""

    @staticmethod
    def forward(ctx, module, slots, x, *params):
        B, N, _ = x.shape
        dev = x.device
        L = _lib.lib()
        L.ampnet_pointnet_seg_train_workspace_bytes.restype = ctypes.c_size_t
        need = L.ampnet_pointnet_seg_train_workspace_bytes(module.VARIANT, B, N, module.num_classes)
        ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        logits = torch.empty((B, module.num_classes, N), dtype=torch.float32, device=dev)
        feat_T = torch.empty((B, 64, 64), dtype=torch.float32, device=dev)
        table = module._layer_table()
        with torch.cuda.device(dev):
            rc = L.ampnet_pointnet_seg_train_fwd_f32(table, module.VARIANT, _lib.ptr(x), B, N, module.num_classes, _lib.ptr(logits),
                                                     _lib.ptr(feat_T), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_pointnet_seg_train_fwd_f32")
        for m in module.modules():
            if isinstance(m, _BN):
                m.num_batches_tracked += 1
        ctx.save_for_backward(x)
        ctx.meta = (module, slots, ws, [tuple(p.shape) for p in params])
        ctx.mark_non_differentiable()
        return logits, feat_T

    @staticmethod
    def backward(ctx, dlogits, d_feat_T):
        module, slots, ws, shapes = ctx.meta
        x, = ctx.saved_tensors
        B, N, _ = x.shape
        dev = x.device
        grads = [torch.zeros(s, dtype=torch.float32, device=dev) for s in shapes]
        garr = (ctypes.c_void_p * (4 * N_LAYERS))()
        for i, idx in enumerate(slots):
            for j, k in enumerate(idx):
                garr[4 * i + j] = None if k < 0 else grads[k].data_ptr()
        dl = dlogits.contiguous().float()
        dft = None if d_feat_T is None else d_feat_T.contiguous().float()
        with torch.cuda.device(dev):
            rc = _lib.lib().ampnet_pointnet_seg_bwd_f32(module._layer_table(), garr, module.VARIANT, _lib.ptr(x), B, N, module.num_classes,
                                                        _lib.ptr(dl), _lib.ptr(dft), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_pointnet_seg_bwd_f32")
        ctx.meta = None
        return (None, None, None) + tuple(grads)


N_CLS_LAYERS = 20      # AMPNET_POINTNET_CLS_LAYERS


class ClsHolder(nn.Module):
    """ClassificationPointNet of both baseline files (pointnet.py:100-125, light_pointnet_256.py:100-125): parameter holder with the
    reference's state_dict keys (base_pointnet.*, fc_1..3, bn_1..2; dropout_1 has none); forward -> (log-probabilities [B, num_classes],
    feature_transform [B, 64, 64]) through ampnet_pointnet_cls_fwd_f32, train mode differentiable through ampnet_pointnet_cls_bwd_f32.
    Dropout draws come from the package's counter hash of (seed, step) -- not torch's generator (as in the AMP-Net head)."""
    VARIANT = None

    def _init_head(self, num_classes, dropout, G, C1, C2, bias, device):
        self.num_classes = num_classes
        self.p_drop = float(dropout)
        self.fc_1 = _Linear(G, C1, bias, device)
        self.fc_2 = _Linear(C1, C2, bias, device)
        self.fc_3 = _Linear(C2, num_classes, True, device)
        self.bn_1, self.bn_2 = _BN(C1, device), _BN(C2, device)
        self.dropout_1 = nn.Dropout(dropout)              # kept for module-tree parity; the kernel applies the mask
        self.seed, self._step = 0x243F6A88, 0
        self._ws = ops.Workspace()
        self._key, self._arr, self._keep = None, None, None

    def _layers(self):
        layers = self.base_pointnet.layers() + [(self.fc_1, self.bn_1), (self.fc_2, self.bn_2), (self.fc_3, None)]
        assert len(layers) == N_CLS_LAYERS
        return layers

    def _layer_table(self):
        tensors = []
        for lin, bn in self._layers():
            tensors += [lin.weight, getattr(lin, "bias", None)]
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [None] * 4
        key = tuple(0 if t is None else t.data_ptr() for t in tensors)
        if key != self._key:
            arr = (ctypes.c_void_p * (6 * N_CLS_LAYERS))()
            for i, t in enumerate(tensors):
                if t is not None:
                    _lib.require_gpu(t, "ClassificationPointNet parameter")
                    if t.dtype != torch.float32 or not t.is_contiguous():
                        raise _lib.AmpnetError("ClassificationPointNet parameters must be contiguous float32")
                arr[i] = None if t is None else t.data_ptr()
            self._key, self._arr, self._keep = key, arr, tensors
        return self._arr

    def _train_tables(self):
        params, slots = [], []
        for lin, bn in self._layers():
            idx = []
            for t in (lin.weight, getattr(lin, "bias", None), None if bn is None else bn.weight, None if bn is None else bn.bias):
                if t is None:
                    idx.append(-1)
                else:
                    idx.append(len(params))
                    params.append(t)
            slots.append(idx)
        return params, slots

    def forward(self, x):
        _lib.require_gpu(x, "x")
        if x.dim() != 3 or x.shape[2] != 9 or x.dtype != torch.float32:
            raise _lib.AmpnetError(f"ClassificationPointNet: x must be [B, N, 9] float32, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        if self.training:
            seed = (self.seed + 0x632BE5AB * self._step) & 0xFFFFFFFF
            self._step += 1
            params, slots = self._train_tables()
            return _BaselineClsFn.apply(self, slots, seed, x, *params)
        B, N, _ = x.shape
        dev = x.device
        L = _lib.lib()
        L.ampnet_pointnet_cls_workspace_bytes.restype = ctypes.c_size_t
        buf = self._ws.get(L.ampnet_pointnet_cls_workspace_bytes(self.VARIANT, B, N, self.num_classes), dev)
        out = torch.empty((B, self.num_classes), dtype=torch.float32, device=dev)
        feat_T = torch.empty((B, 64, 64), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.ampnet_pointnet_cls_fwd_f32(self._layer_table(), self.VARIANT, _lib.ptr(x), B, N, self.num_classes, 0, ctypes.c_float(0.0),
                                               ctypes.c_uint32(0), _lib.ptr(out), _lib.ptr(feat_T), _lib.ptr(buf), ctypes.c_size_t(buf.numel()),
                                               _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_pointnet_cls_fwd_f32")
        return out, feat_T


class _BaselineClsFn(torch.autograd.Function):
    """Train-mode forward / backward of ClassificationPointNet through the C ABI; every forward owns its tape until its backward has run."""

    @staticmethod
    def forward(ctx, module, slots, seed, x, *params):
        B, N, _ = x.shape
        dev = x.device
        L = _lib.lib()
        L.ampnet_pointnet_cls_workspace_bytes.restype = ctypes.c_size_t
        ws = torch.empty(int(L.ampnet_pointnet_cls_workspace_bytes(module.VARIANT, B, N, module.num_classes)), dtype=torch.uint8, device=dev)
        out = torch.empty((B, module.num_classes), dtype=torch.float32, device=dev)
        feat_T = torch.empty((B, 64, 64), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.ampnet_pointnet_cls_fwd_f32(module._layer_table(), module.VARIANT, _lib.ptr(x), B, N, module.num_classes, 1,
                                               ctypes.c_float(module.p_drop), ctypes.c_uint32(seed), _lib.ptr(out), _lib.ptr(feat_T), _lib.ptr(ws),
                                               ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_pointnet_cls_fwd_f32")
        for m in module.modules():
            if isinstance(m, _BN):
                m.num_batches_tracked += 1
        ctx.save_for_backward(x)
        ctx.meta = (module, slots, seed, ws, [tuple(p.shape) for p in params])
        return out, feat_T

    @staticmethod
    def backward(ctx, d_out, d_feat_T):
        module, slots, seed, ws, shapes = ctx.meta
        x, = ctx.saved_tensors
        B, N, _ = x.shape
        dev = x.device
        grads = [torch.zeros(s, dtype=torch.float32, device=dev) for s in shapes]
        garr = (ctypes.c_void_p * (4 * N_CLS_LAYERS))()
        for i, idx in enumerate(slots):
            for j, k in enumerate(idx):
                garr[4 * i + j] = None if k < 0 else grads[k].data_ptr()
        do = d_out.contiguous().float()
        dft = None if d_feat_T is None else d_feat_T.contiguous().float()
        with torch.cuda.device(dev):
            rc = _lib.lib().ampnet_pointnet_cls_bwd_f32(module._layer_table(), garr, module.VARIANT, _lib.ptr(x), B, N, module.num_classes,
                                                        ctypes.c_float(module.p_drop), ctypes.c_uint32(seed), _lib.ptr(do), _lib.ptr(dft), _lib.ptr(ws),
                                                        ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_pointnet_cls_bwd_f32")
        ctx.meta = None
        return (None, None, None, None) + tuple(grads)
