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


@pytest.mark.parametrize("n,s,ld", [(16385, 700, 3), (20000, 8192, 13), (40000, 1000, 3), (60000, 8192, 4)])
def test_fps_large_clouds_match_oracle(synth, n, s, ld):
    """Raw tiles before the first FPS stage have any size (data_proc/sample_fps.py:23-26: `if pc.shape[0] > 8192: fps(pc, 8192)`):
    clouds of more than 16384 points stream their coordinates (fps_stream_kernel) -- same indices as the C oracle, bit for bit."""
    U = sub("utils.utils")
    pc = synth.uniform(2000 + n, (2, n, ld), -1.0, 1.0)
    pc[1, :, :3] = np.round(pc[1, :, :3] * 16) / 16      # 33^3 grid cells for >= 16385 points: duplicates and exact ties
    got = U.fps_indices(torch.from_numpy(pc).cuda(), s).cpu().numpy()
    for c in range(2):
        want = F.fps_indices_c(pc[c], s)
        assert np.array_equal(got[c], want), f"cloud {c}: first mismatch at {np.argmax(got[c] != want)}"


@pytest.mark.parametrize("n,s", [(600, 600), (5000, 5000), (8192, 8192), (12000, 12000)])
def test_fps_all_duplicates_tail(synth, n, s):
    """Every point on a 5 x 5 x 5 grid: after <= 125 picks only duplicates of picked points are left (maximum distance 0) and the
    reference keeps picking the lowest remaining index (utils.py:927-931) -- the kernel's bitmap tail."""
    U = sub("utils.utils")
    pc = np.round(synth.uniform(3000 + n, (1, n, 3), -1.0, 1.0) * 2) / 2
    got = U.fps_indices(torch.from_numpy(pc).cuda(), s).cpu().numpy()[0]
    assert np.array_equal(got, F.fps_indices_c(pc[0], s))
    assert len(np.unique(got)) == s


def test_fps_round_stamps_diagnostic(synth):
    """The diagnostic build returns the same indices and four increasing time stamps per round."""
    import ctypes
    L = sub("_lib")
    pc = synth.clouds(77, 1, 8192)
    xyz = torch.from_numpy(pc).cuda()
    S = 512
    idx = torch.empty(S, dtype=torch.int32, device="cuda")
    stamps = torch.zeros(4 * S, dtype=torch.int64, device="cuda")
    rc = L.lib().ampnet_fps_round_stamps(L.ptr(xyz), 8192, 3, S, L.ptr(idx), L.ptr(stamps), L.stream_ptr(xyz.device))
    L.check(rc, "ampnet_fps_round_stamps")
    assert np.array_equal(idx.cpu().numpy(), F.fps_indices_c(pc[0], S))
    st = stamps.cpu().numpy().reshape(S, 4)[1:]
    assert (np.diff(st.reshape(-1)) > 0).all()
    d = np.median(np.diff(np.concatenate([st[:-1, 3:4], st[1:]], axis=1), axis=1), axis=0)      # update, barrier, fold, coordinates
    print(f"fps round anatomy (cycles, median): update+wave-reduce {d[0]:.0f}, slot+barrier {d[1]:.0f}, fold {d[2]:.0f}, coordinates {d[3]:.0f}; round {d.sum():.0f}")


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
