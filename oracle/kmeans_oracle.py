"""ORACLE -- test infrastructure, not product code.

CPU restatement of the size-constrained k-means this build defines (csrc/kmeans.hip, include/ampnet_hip.h:
ampnet_kmeans_balanced_f32).  PARITY UNPINNED: the reference delegates this step to the third-party package
k_means_constrained.KMeansConstrained (data_proc/3_kmeans.py:78-82, utils/utils.py:500-505), which is not part of the reference
repository (no version pinned, not importable here), and no reference test or fixture covers it.  What is pinned is the call
sites' contract: k clusters, every cluster >= size_min (== size_max == n_points for the training windows), features (x, y, NDVI).

Spec (float32 distances ((d0*d0 + d1*d1) + d2*d2), one rounding per operation):
  seeding     farthest-point seeding from a start point (init 0: point 0; init t: lowbias32(seed + 0x9E3779B9 * t) % n)
  assignment  (point, cluster) pairs in ascending (distance, point, cluster) order; sweep 1 fills every cluster to size_min,
              sweep 2 places the remaining points under capacity size_max
  update      cluster means (float64 sums in the kernel's order: 1024 strided partial sums, halving tree), float32 centres
  stop        summed squared centre shift <= tol * mean feature variance, or max_iter
  result      final assignment against the final centres; the init with the lowest inertia wins
Only tests/ may import this file."""
import numpy as np

T = 1024
MAXK = 32


def _hash32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _dist(F, c):
    d = F - c[None, :]
    sq = d * d
    return (sq[:, 0] + sq[:, 1]) + sq[:, 2]                # float32 throughout


def _strided_tree_sum(v):
    """sum of float64 v in the kernel's order: thread t adds v[t], v[t + 1024], ... sequentially, then a halving tree."""
    part = np.zeros(T, dtype=np.float64)
    for t in range(min(T, len(v))):
        s = 0.0
        for x in v[t::T]:
            s += float(x)
        part[t] = s
    w = T // 2
    while w > 0:
        part[:w] += part[w:2 * w]
        w //= 2
    return part[0]


def seed_centres(F, k, start):
    n = len(F)
    dmin = np.full(n, np.inf, dtype=np.float32)
    C = np.zeros((k, 3), dtype=np.float32)
    last = start
    for c in range(k):
        C[c] = F[last]
        dmin = np.minimum(_dist(F, F[last]), dmin)
        last = int(np.argmax(dmin))                          # first maximum
    return C


def assign(F, C, size_min, size_max):
    n, k = len(F), len(C)
    D = np.stack([_dist(F, C[c]) for c in range(k)], axis=1)          # [n, k] float32
    bits = D.view(np.uint32).astype(np.uint64).reshape(-1)
    ids = (np.arange(n, dtype=np.uint64)[:, None] * MAXK + np.arange(k, dtype=np.uint64)[None, :]).reshape(-1)
    order = np.argsort((bits << np.uint64(32)) | ids, kind="stable")
    pi = (ids[order] // MAXK).astype(np.int64)
    pc = (ids[order] % MAXK).astype(np.int64)
    labels = np.full(n, -1, dtype=np.int32)
    cnt = np.zeros(k, dtype=np.int64)
    assigned = 0
    for cap, target in ((size_min, min(size_min * k, n)), (size_max, n)):
        if assigned >= target:
            continue
        for i, c in zip(pi, pc):
            if labels[i] >= 0 or cnt[c] >= cap:
                continue
            labels[i] = c
            cnt[c] += 1
            assigned += 1
            if assigned >= target:
                break
    return labels, D


def kmeans_balanced(F, k, size_min, size_max, n_init=5, max_iter=10, tol=1e-2, seed=0):
    """-> (labels int32 [n], centres float32 [k, 3], inertia float)."""
    F = np.ascontiguousarray(F, dtype=np.float32)
    n = len(F)
    tol_abs = np.float32(float(np.float32(tol)) * float(F.astype(np.float64).var(axis=0).sum()) / 3.0)
    best = None
    for init in range(n_init):
        start = 0 if init == 0 else _hash32(seed + 0x9E3779B9 * init) % n
        C = seed_centres(F, k, start)
        for it in range(max_iter):
            labels, _ = assign(F, C, size_min, size_max)
            newC = np.zeros_like(C)
            shift = 0.0
            for c in range(k):
                sel = labels == c
                m = max(float(_strided_tree_sum(np.where(sel, 1.0, 0.0))), 1.0)
                part = 0.0
                for f in range(3):
                    nc = np.float32(_strided_tree_sum(np.where(sel, F[:, f].astype(np.float64), 0.0)) / m)
                    d = float(nc) - float(C[c, f])
                    part += d * d
                    newC[c, f] = nc
                shift += part
            C = newC
            if shift <= float(tol_abs):
                break
        labels, D = assign(F, C, size_min, size_max)
        inertia = _strided_tree_sum(D[np.arange(n), labels].astype(np.float64))
        if best is None or inertia < best[2]:
            best = (labels, C.copy(), inertia)
    return best


def lloyd_inertia(F, k, iters=25, start=0):
    """Unconstrained Lloyd from the same farthest-point seeding: the quality yardstick of the balanced assignment."""
    F = np.ascontiguousarray(F, dtype=np.float32)
    C = seed_centres(F, k, start)
    for _ in range(iters):
        D = np.stack([_dist(F, C[c]) for c in range(k)], axis=1)
        lab = D.argmin(1)
        for c in range(k):
            if (lab == c).any():
                C[c] = F[lab == c].mean(0)
    D = np.stack([_dist(F, C[c]) for c in range(k)], axis=1)
    return float(D.min(1).astype(np.float64).sum())
