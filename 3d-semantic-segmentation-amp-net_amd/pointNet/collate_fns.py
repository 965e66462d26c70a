"""collate_seq_padd / collate_cls_padd with the reference's name, output layout and random draws (pointNet/collate_fns.py:4-55):
every sample is brought to exactly 2048 points per cluster (torch.randint with replacement when it has fewer,
random.sample when it has more -- the same calls in the same order, so a seeded run collates identically),
clusters are padded to 9 (data and centroids by replicating the last cluster, targets with -1), and the centroids
come out through the reference's `.view(-1, 9, 2)` reinterpretation of a [B, 2, 1, 9] tensor.
Runs in DataLoader workers: CPU only, never touches HIP."""
import os
import random

import torch

N_POINTS = 2048
MAX_WINDOWS = 9


def _pad_last(t, width, mode):
    """t [..., w] -> [..., width]; 'replicate' repeats the last slice, otherwise fills with the constant `mode`."""
    extra = width - t.shape[-1]
    if extra <= 0:
        return t
    if mode == "replicate":
        tail = t[..., -1:].expand(*t.shape[:-1], extra)
    else:
        tail = torch.full((*t.shape[:-1], extra), mode, dtype=t.dtype)
    return torch.cat([t, tail], dim=-1)


def _fill_padded(out, i, t, width, mode):
    """out[i, ..., :w] = t, out[i, ..., w:] = the last slice of t ('replicate') or the constant `mode` -- _pad_last written into place."""
    w = t.shape[-1]
    out[i, ..., :w] = t
    if w < width:
        if mode == "replicate":
            out[i, ..., w:] = t[..., -1:]
        else:
            out[i, ..., w:] = mode


def collate_seq_padd(batch):
    """batch: list of (pc [n, 9, w] float, labels [n, w] int, filename, centroids [2, w] float)
    -> (data [B, 2048, 9, 9] f32, targets [B, 2048, 9] i64, filenames, centroids [B, 9, 2] f32).
    The output tensors are allocated once and every sample is gathered / padded straight into its slice (the values and the random draws
    are those of the reference's pad + stack; half the memory traffic of building padded per-sample copies first: this runs in the
    DataLoader workers and was 93 ms per batch of 64, bench.py train_att_epoch)."""
    B = len(batch)
    feats = int(torch.as_tensor(batch[0][0]).shape[1]) if B else 9
    data = torch.empty((B, N_POINTS, feats, MAX_WINDOWS), dtype=torch.float32)
    targets = torch.empty((B, N_POINTS, MAX_WINDOWS), dtype=torch.int64)
    cents = torch.empty((B, 2, 1, MAX_WINDOWS), dtype=torch.float32)
    names = []
    for i, (pc, labels, name, cent) in enumerate(batch):
        pc = torch.as_tensor(pc).float()
        labels = torch.as_tensor(labels).long()
        cent = torch.as_tensor(cent).float().unsqueeze(1)            # [2, 1, w]
        n = pc.shape[0]
        if pc.shape[2] > MAX_WINDOWS:
            raise ValueError(f"{name}: {pc.shape[2]} clusters, at most {MAX_WINDOWS} fit")     # (torch.stack of the reference fails here as well)
        idx = None
        if n < N_POINTS:
            idx = torch.randint(0, n, (N_POINTS,))
        elif n > N_POINTS:
            idx = torch.as_tensor(random.sample(range(n), N_POINTS))
        if idx is not None:
            pc, labels = pc.index_select(0, idx), labels.index_select(0, idx)
        _fill_padded(data, i, pc, MAX_WINDOWS, "replicate")
        _fill_padded(targets, i, labels, MAX_WINDOWS, -1)
        _fill_padded(cents, i, cent, MAX_WINDOWS, "replicate")
        names.append(name)
    return data, targets, names, cents.view(-1, MAX_WINDOWS, 2)


def collate_cls_padd(batch):
    """The classification collate of the reference (pointNet/collate_fns.py:58-113) for the window-token classifier:
    batch: list of (pc [n, d, w] float, target, filename, centroids [2, w] float, labels_segmen [n, w] int)
    -> (data [B, 2048, d, 9] f32, targets [B, ...] i64, filenames, centroids [B, 9, 2] f32, labels_segmen [B, 2048, 9] i64).
    Points are resampled to 2048 per cluster and clusters padded to 9 exactly as in collate_seq_padd (same torch / python RNG calls
    in the same order; data and centroids replicate the last cluster, the segmentation labels are padded with -1; the centroids
    leave through the same `.view(-1, 9, 2)` reinterpretation).  `target` is stacked as given; hand it over as a sequence ([1], an
    array): the reference builds it with torch.LongTensor(t[1]) (:74), which for a bare int k ALLOCATES k uninitialised elements
    instead of holding k -- a bare int is taken as the label here."""
    data, targets, names, cents, seg = [], [], [], [], []
    for pc, target, name, cent, labels in batch:
        pc = torch.as_tensor(pc).float()
        labels = torch.as_tensor(labels).long()
        cent = torch.as_tensor(cent).float().unsqueeze(1)            # [2, 1, w]
        n = pc.shape[0]
        if n < N_POINTS:
            idx = torch.randint(0, n, (N_POINTS,))
            pc, labels = pc[idx], labels[idx]
        elif n > N_POINTS:
            idx = random.sample(range(n), N_POINTS)
            pc, labels = pc[idx], labels[idx]
        data.append(_pad_last(pc, MAX_WINDOWS, "replicate"))
        cents.append(_pad_last(cent, MAX_WINDOWS, "replicate"))
        seg.append(_pad_last(labels, MAX_WINDOWS, -1))
        targets.append(torch.as_tensor(target).long())
        names.append(name)
    return (torch.stack(data, 0), torch.stack(targets, 0), names, torch.stack(cents, 0).view(-1, MAX_WINDOWS, 2),
            torch.stack(seg, 0))


class RaggedBatch:
    """What collate_seq_ragged hands to the training step instead of the padded [B, 2048, 9, 9] / [B, 2048, 9] pair: the samples as the
    dataset returned them, concatenated, + the resampling map.  The padded batch is never built on the host; include/ampnet_hip.h:
    ampnet_collate_augment_f32 gathers it on the device.  Tensors only (pin_memory / to work attribute by attribute)."""
    __slots__ = ("pts", "lab", "idx", "meta", "n_points", "n_windows")

    def __init__(self, pts, lab, idx, meta, n_points=N_POINTS, n_windows=MAX_WINDOWS):
        self.pts, self.lab, self.idx, self.meta, self.n_points, self.n_windows = pts, lab, idx, meta, n_points, n_windows

    def __len__(self):
        return int(self.idx.shape[0])

    def pin_memory(self):                    # DataLoader(pin_memory=True) calls this on custom batch types
        return RaggedBatch(self.pts.pin_memory(), self.lab.pin_memory(), self.idx.pin_memory(), self.meta.pin_memory(), self.n_points, self.n_windows)

    def to(self, device, non_blocking=False):
        return RaggedBatch(*(t.to(device, non_blocking=non_blocking) for t in (self.pts, self.lab, self.idx, self.meta)), self.n_points, self.n_windows)

    def record_stream(self, stream):
        for t in (self.pts, self.lab, self.idx, self.meta):
            if t.is_cuda:
                t.record_stream(stream)

    @property
    def is_cuda(self):
        return self.pts.is_cuda

    def to_padded(self):
        """The (data [B, N, 9, W] f32, targets [B, N, W] i64) pair collate_seq_padd builds -- host reference of what the device kernel
        gathers (tests; AMPNET_HOST_AUG=1)."""
        B, N, W = len(self), self.n_points, self.n_windows
        data = torch.empty((B, N, 9, W), dtype=torch.float32)
        targets = torch.empty((B, N, W), dtype=torch.int64)
        meta = self.meta.cpu().tolist()
        pts, lab, idx = self.pts.cpu(), self.lab.cpu(), self.idx.cpu().long()
        for b, (n, w, po, lo) in enumerate(meta):
            pc = pts[po:po + n * 9 * w].view(n, 9, w).index_select(0, idx[b])
            lb = lab[lo:lo + n * w].view(n, w).index_select(0, idx[b]).long()
            _fill_padded(data, b, pc, W, "replicate")
            _fill_padded(targets, b, lb, W, -1)
        return data, targets


def _batch_buffer(numel, dtype):
    """Uninitialised [numel] tensor; inside a DataLoader worker its storage is shared memory already (torch's default_collate idiom), so
    handing the batch to the main process does not copy it again."""
    t = torch.empty(0, dtype=dtype)
    if torch.utils.data.get_worker_info() is not None:
        storage = t._typed_storage()._new_shared(numel, device=t.device)
        return t.new(storage).resize_(numel)
    return torch.empty(numel, dtype=dtype)


def collate_seq_ragged(batch):
    """collate_seq_padd without its 42 MB of host gathers per batch: same arguments, same random draws in the same order (so a seeded
    run resamples identically), but the result carries the samples RAGGED --
        (RaggedBatch, None, filenames, centroids [B, 9, 2] f32)
    and the resampling to 2048 points / padding to 9 clusters happens on the GPU inside the augmentation kernel
    (amp_step.train_loop -> ampnet_collate_augment_f32).  RaggedBatch.to_padded() rebuilds collate_seq_padd's tensors exactly.
    Samples may arrive unread (datasets.LazyKmeansSample, LidarKmeansDataset(lazy=True)): libampnet_host.so then reads, filters and
    relabels each file straight into its slice of the batch (include/ampnet_host.h) -- same bytes as the eager samples."""
    from .. import _hostlib
    from .datasets import LazyKmeansSample
    B = len(batch)
    names = []
    idx = torch.empty((B, N_POINTS), dtype=torch.int32)
    meta = torch.empty((B, 4), dtype=torch.int32)
    cents = torch.empty((B, 2, 1, MAX_WINDOWS), dtype=torch.float32)
    # upper bounds of the two buffers (a lazy sample may lose rows to the noise filter)
    tot_p = tot_l = 0
    eager = []
    for pc, labels, name, cent in batch:
        if isinstance(pc, LazyKmeansSample):
            n, feats, w = pc.n, 9, pc.w
            eager.append(None)
        else:
            pc = torch.as_tensor(pc).float()
            n, feats, w = pc.shape
            eager.append(pc)
        if feats != 9 or w > MAX_WINDOWS:
            raise ValueError(f"{name}: expected [n, 9, w <= {MAX_WINDOWS}], got [{n}, {feats}, {w}]")
        tot_p += n * 9 * w
        tot_l += n * w
    pts = _batch_buffer(tot_p, torch.float32)
    lab = _batch_buffer(tot_l, torch.int8)
    po = lo = 0
    ramp = None
    cbuf = torch.empty((2, 64), dtype=torch.float32)
    for i, (pc, labels, name, cent) in enumerate(batch):
        if eager[i] is None:
            w = pc.w
            cb = cbuf[:, :w].contiguous() if pc.want_centroids else None
            n = _hostlib.lib().ampnet_host_kmeans_file_ragged_f32(os.fsencode(pc.path), pc.offset, pc.n, pc.feats, w, pts.data_ptr() + 4 * po,
                                                                  lab.data_ptr() + lo, cb.data_ptr() if cb is not None else None)
            if n < 0:
                raise RuntimeError(f"{name}: ampnet_host_kmeans_file_ragged_f32 failed ({n})")
            cent = cb
        else:
            pc = eager[i]
            n, _, w = pc.shape
            pts[po:po + n * 9 * w] = pc.reshape(-1)
            lab[lo:lo + n * w] = torch.as_tensor(labels).reshape(-1).to(torch.int8)
        if n < N_POINTS:
            idx[i] = torch.randint(0, n, (N_POINTS,)).to(torch.int32)
        elif n > N_POINTS:
            idx[i] = torch.as_tensor(random.sample(range(n), N_POINTS), dtype=torch.int32)
        else:
            if ramp is None:
                ramp = torch.arange(N_POINTS, dtype=torch.int32)
            idx[i] = ramp
        meta[i, 0], meta[i, 1], meta[i, 2], meta[i, 3] = n, w, po, lo
        po += n * 9 * w
        lo += n * w
        _fill_padded(cents, i, torch.as_tensor(cent).float().unsqueeze(1), MAX_WINDOWS, "replicate")
        names.append(name)
    return RaggedBatch(pts[:po], lab[:lo], idx, meta), None, names, cents.view(-1, MAX_WINDOWS, 2)
