"""collate_seq_padd / collate_cls_padd with the reference's name, output layout and random draws (pointNet/collate_fns.py:4-55):
every sample is brought to exactly 2048 points per cluster (torch.randint with replacement when it has fewer,
random.sample when it has more -- the same calls in the same order, so a seeded run collates identically),
clusters are padded to 9 (data and centroids by replicating the last cluster, targets with -1), and the centroids
come out through the reference's `.view(-1, 9, 2)` reinterpretation of a [B, 2, 1, 9] tensor.
Runs in DataLoader workers: CPU only, never touches HIP."""
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
