"""The FPS cascade drop-in (package data_proc/sample_fps.py for the reference's data_proc/sample_fps.py:12-34) and the ragged launch
under it (ampnet_fps_ragged_f32): files of unequal size through both stages in one launch per size class, outputs equal to the C oracle
(oracle/fps_oracle.c, pinned to the reference's fps by tests/test_oracle_golden.py) run file by file -- bit for bit, integer indices."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402
from oracle import fps_oracle as F                # noqa: E402

pytestmark = pytest.mark.gpu


def _tile(synth, seed, n, D=11):
    """[n, D] float32 tile: x, y, z from the synthetic generator, column 3 = class codes incl. noise 30 / 7, the rest features."""
    xyz = synth.clouds(seed, 1, n)[0][:, :3]
    pc = synth.uniform(seed + 1, (n, D), 0.0, 1.0).astype(np.float32)
    pc[:, :3] = xyz
    cls = (synth.uniform(seed + 2, (n,), 0.0, 1.0) * 40).astype(np.int64)
    pc[:, 3] = np.where(cls == 0, 30, np.where(cls == 1, 7, np.array([2, 3, 5, 15, 14, 4])[cls % 6])).astype(np.float32)
    return pc


def _oracle_cascade(pc):
    out8 = None
    if pc.shape[0] > 8192:
        pc = pc[F.fps_indices_c(np.ascontiguousarray(pc[:, :3]), 8192)]
        out8 = pc
    if pc.shape[0] > 4096:
        pc = pc[F.fps_indices_c(np.ascontiguousarray(pc[:, :3]), 4096)]
    return out8, pc


@pytest.mark.parametrize("sizes,samples", [([300, 5000, 1100, 4097, 200, 9000], 256), ([8193, 12000, 16384, 9001], 8192),
                                           ([17000, 20000, 16385], 8192), ([5000, 40000, 700, 9000, 16390], [4096, 8192, 700, 4096, 100])])
def test_ragged_launch_matches_oracle_per_cloud(synth, sizes, samples):
    """ampnet_fps_ragged_f32: clouds of unequal size, several size classes (incl. the streaming one) in one call."""
    U = sub("utils.utils")
    clouds = [synth.clouds(900 + i, 1, n)[0][:, :3].copy() for i, n in enumerate(sizes)]
    rows = torch.from_numpy(np.concatenate(clouds)).cuda()
    got = U.fps_indices_ragged(rows, sizes, samples)
    want_s = [samples] * len(sizes) if np.isscalar(samples) else samples
    for c, g, s in zip(clouds, got, want_s):
        s = min(s, c.shape[0])
        assert g.shape == (s,)
        assert np.array_equal(g.cpu().numpy(), F.fps_indices_c(c, s))


def test_cascade_directory_matches_reference_semantics(synth, tmp_path):
    """Files of 2000 .. 21000 points with noise classes: class-30/7 filter, stage 8192 (only files above it), stage 4096, outputs of
    both directories equal to the oracle cascade file by file; a file <= 4096 points is written unchanged (sample_fps.py:32-34)."""
    S = sub("data_proc.sample_fps")
    in_dir, out_dir = tmp_path / "tiles", tmp_path / "out"
    in_dir.mkdir()
    sizes = [2000, 4500, 8600, 9100, 12000, 17500, 21000, 4200, 8400]
    raw = {}
    for i, n in enumerate(sizes):
        pc = _tile(synth, 7000 + 10 * i, n)
        raw[f"tile_{i}"] = pc
        with open(in_dir / f"tile_{i}.pkl", "wb") as f:
            pickle.dump(pc.astype(np.float64), f)                 # the reference's tiles are float64 on disk; .astype(float32) on load (:15)
    files = sorted(str(p) for p in in_dir.glob("*.pkl"))
    written = S.sample_files(files, str(out_dir), files_per_launch=5)      # two groups: batching must not change a file's result
    n8 = 0
    for name, pc in raw.items():
        clean = S.remove_noise(pc.astype(np.float64).astype(np.float32))
        assert clean.shape[0] < pc.shape[0] and not np.isin(clean[:, 3], [30, 7]).any()
        want8, want4 = _oracle_cascade(clean)
        p8 = out_dir / "towers_100x100_fps_8192" / f"{name}.pkl"
        assert p8.exists() == (want8 is not None), name
        if want8 is not None:
            n8 += 1
            got8 = pickle.load(open(p8, "rb"))
            assert got8.dtype == np.float32 and np.array_equal(got8, want8), name
        got4 = pickle.load(open(out_dir / "towers_100x100_fps_4096" / f"{name}.pkl", "rb"))
        assert np.array_equal(got4, want4), name
        assert got4.shape[0] == min(clean.shape[0], 4096)
    assert written == [n8, len(sizes)]
