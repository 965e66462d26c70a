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


def window_offsets(np_cluster, device):
    """[Q+1] int32 device prefix offsets from a list of window sizes."""
    off = [0]
    for n in np_cluster:
        off.append(off[-1] + int(n))
    return torch.tensor(off, dtype=torch.int32, device=device), off[-1], max(int(n) for n in np_cluster)


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
