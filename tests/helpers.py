"""Shared test helpers (host side only)."""
import numpy as np
import torch


def torch_params(d):
    return {k: torch.from_numpy(np.array(v, copy=True)) for k, v in d.items()}


def replay_augment(seed, pc, tg, train):
    """Replays the numpy draws of the reference's train_loop in order
    (train_pointnet-attention.py:390-405 with utils/utils.py:582-632): cluster permutation, angle,
    then per window: z-rotation of xyz (float64 product rounded to float32) and one point permutation.
    pc [B,N,9,W] f32, tg [B,N,W] i64 -> augmented copies."""
    np.random.seed(seed)
    W = pc.shape[3]
    idx = np.arange(W)
    np.random.shuffle(idx)
    pc = pc[:, :, :, idx].copy()
    tg = tg[:, :, idx].copy()
    angle = np.random.uniform() * 2 * np.pi
    if train:
        c, s = np.cos(angle), np.sin(angle)
        rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
        for w in range(W):
            xyz = pc[:, :, :3, w]
            pc[:, :, :3, w] = np.dot(xyz.reshape(-1, 3), rot).reshape(xyz.shape).astype(np.float32)
            pidx = np.arange(pc.shape[1])
            np.random.shuffle(pidx)
            pc[:, :, :, w] = pc[:, :, :, w][:, pidx, :]
            tg[:, :, w] = tg[:, :, w][:, pidx]
    return pc, tg


def baseline_state(synth, table, base):
    """Seeded state_dict {key: ndarray} for a baseline PointNet from {key: shape} (make_golden.py:sec_baseline uses the
    same function, so the fixture's outputs belong to exactly these weights)."""
    sd = {}
    for i, (k, shp) in enumerate(table.items()):
        if k.endswith("running_mean"):
            sd[k] = synth.uniform(base + i, shp, -0.3, 0.3)
        elif k.endswith("running_var"):
            sd[k] = synth.uniform(base + i, shp, 0.5, 1.5)
        elif ".bn" in k or k.startswith("bn"):
            lo, hi = (0.5, 1.5) if k.endswith("weight") else (-0.2, 0.2)
            sd[k] = synth.uniform(base + i, shp, lo, hi)
        else:
            fan = int(np.prod(shp[1:])) if len(shp) > 1 else int(shp[0])
            b = 1.0 / np.sqrt(fan)
            sd[k] = synth.uniform(base + i, shp, -b, b)
    return sd


def cls_sample_array(synth, seed, n):
    """A classification sample file's array [n, 11]: x, y, HAG, class code, I, R, G, B, NIR, NDVI, sampling flag (seeded)."""
    a = synth.uniform(seed, (n, 11), 0.0, 1.0).astype(np.float32)
    codes = np.array([15, 14, 3, 4, 5, 6, 9, 1], dtype=np.float32)
    a[:, 3] = codes[synth.randint(seed + 1, (n,), 0, len(codes))]
    a[:, 10] = (synth.randint(seed + 2, (n,), 0, 3) > 0).astype(np.float32)
    return a
