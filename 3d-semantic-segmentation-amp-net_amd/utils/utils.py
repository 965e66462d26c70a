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


def fps_indices_ragged(rows, sizes, n_samples):
    """FPS of MANY clouds of unequal size in one launch per size class (include/ampnet_hip.h: ampnet_fps_ragged_f32): `rows` [total, D>=3]
    float32 GPU tensor = the clouds back to back, `sizes` their point counts, `n_samples` one int (clamped per cloud to its size) or a
    list.  Returns a list of int32 GPU tensors (indices relative to each cloud, selection order).  Clouds are bucketed by the kernel
    variant their size selects, so a small cloud never runs the 16384-point variant because a large one shares the call."""
    _lib.require_gpu(rows, "rows")
    if rows.dim() != 2 or rows.shape[1] < 3:
        raise _lib.AmpnetError(f"fps_indices_ragged: expected rows [total, D>=3], got {tuple(rows.shape)}")
    sizes = [int(n) for n in sizes]
    if not sizes or min(sizes) < 1 or sum(sizes) != rows.shape[0]:
        raise _lib.AmpnetError(f"fps_indices_ragged: sizes (sum {sum(sizes)}) do not tile the {rows.shape[0]} rows")
    want = [int(n_samples)] * len(sizes) if np.isscalar(n_samples) else [int(v) for v in n_samples]
    if len(want) != len(sizes) or min(want) < 1:
        raise _lib.AmpnetError("fps_indices_ragged: n_samples must be >= 1, one per cloud")
    want = [min(w, n) for w, n in zip(want, sizes)]
    x = rows if rows.dtype == torch.float32 else rows.float()
    x = x.contiguous()
    dev, D = x.device, x.shape[1]
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    edges = (256, 1024, 2048, 4096, 8192, 16384)                         # the size classes of csrc/fps.hip: fps_dispatch
    klass = [int(np.searchsorted(edges, n)) for n in sizes]
    out = [None] * len(sizes)
    L = _lib.lib()
    L.ampnet_fps_ragged_workspace_bytes.restype = ctypes.c_size_t
    for k in sorted(set(klass)):
        members = [i for i, c in enumerate(klass) if c == k]
        # the bucket's clouds as (offset, size) pairs of the SAME rows array: no copy; offsets need not be contiguous
        coff = np.empty(len(members) + 1, dtype=np.int32)
        ooff = np.zeros(len(members) + 1, dtype=np.int32)
        contiguous = all(starts[members[j]] + sizes[members[j]] == starts[members[j + 1]] for j in range(len(members) - 1))
        if contiguous:
            base = int(starts[members[0]])
            coff[:] = [int(starts[i]) - base for i in members] + [int(starts[members[-1]] + sizes[members[-1]]) - base]
            xs = x[base:base + int(coff[-1])]
        else:                                                           # interleaved classes: pack the bucket's rows once
            xs = torch.cat([x[int(starts[i]):int(starts[i]) + sizes[i]] for i in members])
            coff[:] = np.concatenate([[0], np.cumsum([sizes[i] for i in members])])
        ooff[1:] = np.cumsum([want[i] for i in members])
        max_n, total = max(sizes[i] for i in members), int(coff[-1])
        idx = torch.empty(int(ooff[-1]), dtype=torch.int32, device=dev)
        offs = torch.from_numpy(np.stack([coff, ooff])).to(dev)
        need = L.ampnet_fps_ragged_workspace_bytes(total, max_n)
        ws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
        with torch.cuda.device(dev):
            rc = L.ampnet_fps_ragged_f32(_lib.ptr(xs), D, _lib.ptr(offs[0]), _lib.ptr(offs[1]), len(members), total, max_n, _lib.ptr(idx),
                                         _lib.ptr(ws), ctypes.c_size_t(need), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_fps_ragged_f32")
        for j, i in enumerate(members):
            out[i] = idx[int(ooff[j]):int(ooff[j + 1])]
    return out


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


def kmeans_balanced(feat, k, size_min, size_max, n_init=5, max_iter=10, tol=1e-2, seed=0):
    """Size-constrained k-means on the GPU (include/ampnet_hip.h: ampnet_kmeans_balanced_f32; BUILD-DEFINED spec, the reference uses
    the third-party KMeansConstrained here).  feat [n, 3] float32 GPU tensor -> (labels int32 [n], centres [k, 3], inertia float)."""
    _lib.require_gpu(feat, "feat")
    f = feat.float().contiguous()
    if f.dim() != 2 or f.shape[1] != 3:
        raise _lib.AmpnetError(f"kmeans_balanced: feat must be [n, 3], got {tuple(feat.shape)}")
    n = f.shape[0]
    L = _lib.lib()
    L.ampnet_kmeans_workspace_bytes.restype = ctypes.c_size_t
    need = L.ampnet_kmeans_workspace_bytes(n, int(k))
    if need == 0:
        raise _lib.AmpnetError(f"kmeans_balanced: n={n}, k={k} out of range (k <= 32, n <= 65536)")
    ws = torch.empty(need, dtype=torch.uint8, device=f.device)
    labels = torch.empty(n, dtype=torch.int32, device=f.device)
    centres = torch.empty((int(k), 3), dtype=torch.float32, device=f.device)
    inertia = torch.zeros(1, dtype=torch.float64, device=f.device)
    with torch.cuda.device(f.device):
        rc = L.ampnet_kmeans_balanced_f32(_lib.ptr(f), n, int(k), int(size_min), int(size_max), int(n_init), int(max_iter), ctypes.c_float(tol),
                                          ctypes.c_uint32(seed & 0xFFFFFFFF), _lib.ptr(labels), _lib.ptr(centres), _lib.ptr(inertia), _lib.ptr(ws),
                                          ctypes.c_size_t(need), _lib.stream_ptr(f.device))
    _lib.check(rc, "ampnet_kmeans_balanced_f32")
    return labels, centres, float(inertia.item())


def get_cluster_centroid(pc):
    """mean x, y of a cluster [n, >= 2] -> tensor [2] (utils/utils.py:538-543)."""
    return torch.stack([pc[:, 0].mean(0), pc[:, 1].mean(0)], dim=0)


def kmeans_clustering(in_pc, n_points=2048, get_centroids=True, max_clusters=18, out_path='', file_name='', device="cuda", seed=0):
    """Drop-in for utils/utils.py:473-535: clusters of >= n_points points of one point cloud [1, n, dim] / [n, dim] by constrained k-means on
    (x, y, NDVI) = columns (0, 1, 8).  Returns (cluster_lists: list of [n_i, dim] tensors, centroids [k, 2]); optionally saves both like the
    reference (torch.save instead of pickle.dump: amp_test loads them without unpickling).  The clustering itself runs on the MI355X
    (kmeans_balanced above) instead of the third-party KMeansConstrained: same contract (k = floor(n / n_points) <= max_clusters
    clusters, every cluster >= n_points), a different -- build-defined, deterministic -- assignment."""
    in_pc = torch.as_tensor(in_pc)
    if in_pc.dim() == 3:
        in_pc = in_pc.squeeze(0)
    cluster_lists, centroids = [], []
    if in_pc.shape[0] >= 2 * n_points:
        k = min(int(in_pc.shape[0] // n_points), max_clusters)
        feat = in_pc[:, [0, 1, 8]].float().to(device)
        labels, _, _ = kmeans_balanced(feat, k, n_points, in_pc.shape[0], n_init=5, max_iter=10, tol=0.01, seed=seed)
        labels = labels.cpu()
        for c in range(k):                       # clusters in ascending label order (the reference groups by sorted label)
            pts = in_pc[labels == c]
            cluster_lists.append(pts)
            if get_centroids:
                centroids.append(get_cluster_centroid(pts))
    else:
        cluster_lists.append(in_pc)
        if get_centroids:
            centroids.append(get_cluster_centroid(in_pc))
    centroids = torch.stack(centroids, dim=0) if centroids else torch.FloatTensor()
    if out_path:
        os.makedirs(out_path, exist_ok=True)
        torch.save(cluster_lists, os.path.join(out_path, file_name + '_clusters_list') + '.pkl')
        torch.save(centroids, os.path.join(out_path, file_name + '_centroids') + '.pkl')
    return cluster_lists, centroids


def split_kmeans_windows(pc, n_points=2048, max_clusters=9, device="cuda", seed=0):
    """The array part of data_proc/3_kmeans.py:27-116 split_kmeans: a point cloud [n, D] (numpy) -> windows tensor [n_points, D, k] of
    exactly n_points points each (k = ceil(n / n_points) <= max_clusters; surplus points sampled away with random.sample, missing ones
    duplicated with numpy's RNG, like the reference), grouped by size-constrained k-means (size_min = size_max = n_points) on
    (x, y, NDVI) = columns (0, 1, 9).  Clouds below 2 * n_points: one window (random.sample down to n_points when larger)."""
    import random
    pc = np.asarray(pc, dtype=np.float32)
    if pc.shape[0] >= 2 * n_points:
        in_pc = pc
        k = int(np.ceil(in_pc.shape[0] / n_points))
        if k > max_clusters:
            k = max_clusters
            ix = random.sample(range(in_pc.shape[0]), n_points * max_clusters)
            in_pc = in_pc[ix, :]
        elif in_pc.shape[0] < n_points * k:
            extra = np.random.randint(0, in_pc.shape[0], n_points * k - in_pc.shape[0])
            in_pc = np.concatenate([in_pc, in_pc[extra, :]], axis=0)
        if in_pc.shape[0] % n_points != 0:
            in_pc = in_pc[:n_points * (in_pc.shape[0] // n_points), :]
        feat = torch.from_numpy(np.ascontiguousarray(in_pc[:, [0, 1, 9]])).to(device)
        labels, _, _ = kmeans_balanced(feat, k, n_points, n_points, n_init=5, max_iter=10, tol=1e-2, seed=seed)
        labels = labels.cpu().numpy()
        t = torch.from_numpy(in_pc)
        return torch.stack([t[labels == c] for c in range(k)], dim=2)        # [n_points, D, k]
    if pc.shape[0] > n_points:
        ix = random.sample(range(pc.shape[0]), n_points)
        pc = pc[ix, :]
    return torch.from_numpy(pc).unsqueeze(2)


def fps(pc, n_samples, device="cuda"):
    """Drop-in for the reference's fps(pc, n_samples): pc [N, D] -> the sampled rows (all D columns) in
    selection order.  numpy in -> numpy out (like the reference); a GPU tensor in -> GPU tensor out.

    Deviation: the reference (utils/utils.py:889-933) computes the distances in the dtype of `pc` (float64 for
    arrays read from LAS files); here they are float32 on the device, whatever the input dtype.  The training
    pipeline feeds float32 windows, for which the index lists are bit-identical (tests/golden/fps_*.npz); a float64
    input whose two farthest candidates differ by less than float32 rounding may pick the other one."""
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
def get_labels(cluster_lists):
    """list of clusters [n_i, >=10] (column 9 = ASPRS class code) -> list of LongTensor [n_i] with the segmentation
    labels 0 background, 1 tower, 2 lines, 3 low/medium vegetation, 4 high vegetation (utils/utils.py:546-579)."""
    from ..pointNet.datasets import segmentation_labels
    return [segmentation_labels(torch.as_tensor(c).squeeze(0)[:, 9]) for c in cluster_lists]


def save_checkpoint(name, epoch, epochs_since_improvement, model, optimizer, accuracy, batch_size, learning_rate, n_points,
                    weighing_method=None):
    """The baseline's checkpoint (utils/utils.py:441-456): same dict keys, same file name pointNet/checkpoints/checkpoint_<name>.pth."""
    state = {'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'batch_size': batch_size, 'lr': learning_rate,
             'number_of_points': n_points, 'epoch': epoch, 'epochs_since_improvement': epochs_since_improvement, 'accuracy': accuracy,
             'weighing_method': weighing_method}
    os.makedirs("pointNet/checkpoints", exist_ok=True)
    torch.save(state, 'pointNet/checkpoints/checkpoint_' + name + '.pth')
    return 'pointNet/checkpoints/checkpoint_' + name + '.pth'


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


def host_cpu_budget():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (cpu.max / cfs_quota_us) -- a container that
    SEES 256 hardware threads but is scheduled on 16 of them.  torch sizes its intra-op pool from the former: 256 threads spinning on a
    16-CPU quota next to DataLoader workers is what made an epoch with 12 workers 30 x slower than with 4 (bench.py train_att_epoch)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # cgroup v1
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, int(os.environ.get("AMPNET_CPU_THREADS", n))))


def limit_host_threads(reserve=0):
    """Sizes torch's intra-op pool of THIS process to the CPU budget minus `reserve` (the DataLoader workers); returns the budget."""
    import torch
    budget = host_cpu_budget()
    torch.set_num_threads(max(1, min(torch.get_num_threads(), budget - reserve)))
    return budget
