"""Host-side helpers of the hot path, same names and argument meaning as the reference's utils/utils.py.

`fps` runs on the MI355X through the C ABI (ampnet_fps_f32); the augmentation helpers stay numpy because
the reference applies them to host arrays before upload (train_pointnet-attention.py:390-405) and their
random draws must come from numpy's global stream in the reference's order for a run to be reproducible
against it.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib


# ---- a1: utils/utils.py:889-933 ------------------------------------------------------------------
def fps_indices(xyz, n_samples):
    """Batched FPS on the GPU.  xyz: torch tensor [B, N, D>=3] (or [N, D]) float32 on the GPU.
    Returns int32 indices [B, n_samples] (or [n_samples]) in selection order; index 0 is always first."""
    _lib.require_gpu(xyz, "xyz")
    single = xyz.dim() == 2
    x = xyz.unsqueeze(0) if single else xyz
    if x.dim() != 3 or x.shape[2] < 3:
        raise _lib.AmpnetError(f"fps: expected [B, N, D>=3], got {tuple(xyz.shape)}")
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    B, N, D = x.shape
    n_samples = int(n_samples)
    if not (1 <= n_samples <= N):
        raise IndexError(f"fps: n_samples={n_samples} out of range for {N} points")   # reference: IndexError at :928
    idx = torch.empty((B, n_samples), dtype=torch.int32, device=x.device)
    L = _lib.lib()
    L.ampnet_fps_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_fps_workspace_bytes(B, N)              # 0 for register-resident clouds (N <= 16384)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    with torch.cuda.device(x.device):
        rc = L.ampnet_fps_f32(_lib.ptr(x), B, N, D, n_samples, _lib.ptr(idx), _lib.ptr(ws), ctypes.c_size_t(need),
                              _lib.stream_ptr(x.device))
    _lib.check(rc, "ampnet_fps_f32")
    return idx[0] if single else idx


def knn_indices(xyz, centres, k):
    """Exact k-NN grouping on the GPU (BUILD-DEFINED: the reference has no k-NN, include/ampnet_hip.h: ampnet_knn_f32).
    xyz [B, N, D>=3] float32 GPU, centres [B, S] int32 point indices (e.g. fps_indices) -> int32 [B, S, k]: per centre the k
    points with the smallest (float32 squared distance, index), ascending; the centre itself comes first."""
    _lib.require_gpu(xyz, "xyz")
    _lib.require_gpu(centres, "centres")
    if xyz.dim() != 3 or xyz.shape[2] < 3 or centres.dim() != 2 or centres.shape[0] != xyz.shape[0]:
        raise _lib.AmpnetError(f"knn: expected xyz [B, N, D>=3] and centres [B, S], got {tuple(xyz.shape)} {tuple(centres.shape)}")
    if centres.dtype != torch.int32:
        raise _lib.AmpnetError("knn: centres must be int32")
    x = xyz.float().contiguous()
    c = centres.contiguous()
    B, N, D = x.shape
    S = c.shape[1]
    k = int(k)
    if not (1 <= k <= N):
        raise IndexError(f"knn: k={k} out of range for {N} points")
    if S and (int(c.min()) < 0 or int(c.max()) >= N):
        raise IndexError("knn: centre index out of range")
    out = torch.empty((B, S, k), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ampnet_knn_f32(_lib.ptr(x), B, N, D, _lib.ptr(c), S, k, _lib.ptr(out), _lib.stream_ptr(x.device))
    _lib.check(rc, "ampnet_knn_f32")
    return out


def gather_rows(pc, idx):
    """pc [B, N, D] f32 GPU, idx [B, S] int32 -> [B, S, D] (the `pc[sample_inds]` of utils.py:933)."""
    _lib.require_gpu(pc, "pc")
    B, N, D = pc.shape
    S = idx.shape[1]
    out = torch.empty((B, S, D), dtype=torch.float32, device=pc.device)
    with torch.cuda.device(pc.device):
        rc = _lib.lib().ampnet_gather_rows_f32(_lib.ptr(pc.contiguous()), _lib.ptr(idx.contiguous()), B, N, D, S,
                                               _lib.ptr(out), _lib.stream_ptr(pc.device))
    _lib.check(rc, "ampnet_gather_rows_f32")
    return out


def fps(pc, n_samples, device="cuda"):
    """Drop-in for the reference's fps(pc, n_samples): pc [N, D] -> the sampled rows (all D columns) in
    selection order.  numpy in -> numpy out (like the reference); a GPU tensor in -> GPU tensor out."""
    if isinstance(pc, torch.Tensor):
        t = pc if pc.is_cuda else pc.to(device)
        rows = gather_rows(t.float().unsqueeze(0), fps_indices(t, n_samples).unsqueeze(0))[0]
        return rows if pc.is_cuda else rows.cpu()
    arr = np.asarray(pc)
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(device)
    idx = fps_indices(t, n_samples).cpu().numpy().astype(np.int64)
    return arr[idx]


# ---- host helpers on the path (utils/utils.py:14-19, 582-632) --------------------------------------
def rm_padding(preds, targets):
    keep = targets != -1
    return preds[keep], targets[keep], keep


def rotate_point_cloud_z(batch_data, rotation_angle=None):
    """[B, N, 3] -> rotated about z by one angle for the whole batch; float64 product stored as float32."""
    if not rotation_angle:
        rotation_angle = np.random.uniform() * 2 * np.pi
    c, s = np.cos(rotation_angle), np.sin(rotation_angle)
    rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
    flat = np.asarray(batch_data).reshape(-1, 3)
    return np.dot(flat, rot).astype(np.float32).reshape(np.asarray(batch_data).shape)


def shuffle_data(data, labels):
    """One point permutation shared by the batch: data [B, N, ...], labels [B, N]."""
    idx = np.arange(labels.shape[1])
    np.random.shuffle(idx)
    return data[:, idx, :], labels[:, idx], idx


def shuffle_clusters(data, labels):
    """One cluster permutation shared by the batch: data [B, N, D, W], labels [B, N, W]."""
    idx = np.arange(labels.shape[2])
    np.random.shuffle(idx)
    return data[:, :, :, idx], labels[:, :, idx]


# ---- checkpoints (utils/utils.py:422-438): same dict keys, same file naming -------------------------
def get_cluster_centroid(pc):
    """pc [n, >=2] tensor -> tensor [2] = (mean x, mean y)   (utils/utils.py:538-543)."""
    return torch.stack([pc[:, 0].mean(0), pc[:, 1].mean(0)], dim=0)


def get_labels(cluster_lists):
    """list of clusters [n_i, >=10] (column 9 = ASPRS class code) -> list of LongTensor [n_i] with the segmentation
    labels 0 background, 1 tower, 2 lines, 3 low/medium vegetation, 4 high vegetation (utils/utils.py:546-579)."""
    from ..pointNet.datasets import segmentation_labels
    return [segmentation_labels(torch.as_tensor(c).squeeze(0)[:, 9]) for c in cluster_lists]


def save_checkpoint_segmen_model(name, task, epoch, epochs_since_improvement, base_pointnet, segmen_model,
                                 opt_pointnet, opt_segmen, accuracy, batch_size, learning_rate, number_of_points,
                                 weighing_method=None):
    state = {
        "base_pointnet": base_pointnet.state_dict(),
        "segmen_net": segmen_model.state_dict(),
        "opt_pointnet": opt_pointnet.state_dict(),
        "opt_segmen": opt_segmen.state_dict(),
        "task": task,
        "batch_size": batch_size,
        "lr": learning_rate,
        "number_of_points": number_of_points,
        "epoch": epoch,
        "epochs_since_improvement": epochs_since_improvement,
        "accuracy": accuracy,
    }
    os.makedirs("pointNet/checkpoints", exist_ok=True)
    torch.save(state, "pointNet/checkpoints/model_" + name + ".pth")
