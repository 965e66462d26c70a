"""GPU parity of ampnet_fps_f32 (through the C ABI) with the oracle and the reference's golden indices.
Bit-exact: every index must match."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                          # noqa: E402
from oracle import fps_oracle as F                # noqa: E402

pytestmark = pytest.mark.gpu


def _cloud(synth, name, seed, n):
    xyz = synth.clouds(seed, 1, n)[0]
    if name == "dup":
        xyz[n // 2:] = xyz[: n // 2]
    return xyz


@pytest.mark.parametrize("name", ["small", "dup", "c5", "tiny"])
def test_fps_matches_reference_golden(golden, synth, name):
    U = sub("utils.utils")
    g = golden("fps")
    seed, n, s = [int(v) for v in g[name + "_meta"]]
    xyz = _cloud(synth, name, seed, n)
    got = U.fps_indices(torch.from_numpy(xyz).cuda(), s).cpu().numpy()
    assert np.array_equal(got.astype(np.int64), g[name + "_idx"])


@pytest.mark.parametrize("n,s,ld", [(1, 1, 3), (64, 64, 3), (100, 37, 5), (257, 200, 3), (1500, 1500, 13),
                                    (2048, 512, 9), (4096, 1024, 3), (8192, 4096, 13), (16384, 8192, 3)])
def test_fps_matches_oracle_shapes(synth, n, s, ld):
    U = sub("utils.utils")
    pc = synth.uniform(1000 + n, (3, n, ld), -1.0, 1.0)
    pc[1, :, :3] = np.round(pc[1, :, :3] * 4) / 4        # coarse grid: many exact ties and duplicates
    got = U.fps_indices(torch.from_numpy(pc).cuda(), s).cpu().numpy()
    for c in range(3):
        want = F.fps_indices_c(pc[c], s)
        assert np.array_equal(got[c], want), f"cloud {c}: first mismatch at {np.argmax(got[c] != want)}"


def test_fps_drop_in_returns_rows(synth):
    U = sub("utils.utils")
    pc = np.concatenate([synth.clouds(5, 1, 3000)[0], synth.uniform(6, (3000, 10), 0, 1)], axis=1)
    out = U.fps(pc, 256)
    assert isinstance(out, np.ndarray) and out.shape == (256, 13)
    assert np.array_equal(out, F.fps(pc, 256))
    with pytest.raises(IndexError):
        U.fps(pc, 3001)


def test_fps_batch_c5_full_size_properties(synth):
    """BASELINE config 5: 16 clouds x 8192 points -> 4096 samples.  Size-independent properties:
    indices unique, start at 0, and the running minimum distance of the picks never increases."""
    U = sub("utils.utils")
    pc = synth.clouds(77, 16, 8192)
    idx = U.fps_indices(torch.from_numpy(pc).cuda(), 4096).cpu().numpy()
    assert idx.shape == (16, 4096) and (idx[:, 0] == 0).all()
    for c in (0, 7, 15):
        assert len(np.unique(idx[c])) == 4096
        assert np.array_equal(idx[c], F.fps_indices_c(pc[c], 4096))
