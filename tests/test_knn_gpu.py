"""k-NN grouping kernel (BUILD-DEFINED spec, include/ampnet_hip.h: ampnet_knn_f32) against the build's own CPU restatement
(oracle/fps_oracle.py: knn_indices).  The reference has no k-NN: parity against it is unpinned; what is pinned
is the stated spec -- exact float32 distances, (distance, index) order -- bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402
from oracle import fps_oracle                      # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,s,k", [(100, 10, 1), (1000, 64, 32), (2048, 256, 32), (777, 33, 64), (8192, 128, 32)])
def test_knn_matches_oracle(synth, n, s, k):
    U = sub("utils.utils")
    pc = synth.clouds(3, 2, n)
    x = torch.from_numpy(pc).cuda()
    cent = U.fps_indices(x, s)
    got = U.knn_indices(x, cent, k).cpu().numpy()
    for c in range(2):
        want = fps_oracle.knn_indices(pc[c], cent[c].cpu().numpy(), k)
        assert np.array_equal(got[c], want), (n, s, k, c)
        assert np.array_equal(got[c][:, 0], cent[c].cpu().numpy())       # the centre is its own nearest neighbour


def test_knn_ties_and_duplicates(synth):
    """Duplicate points and a regular grid (many equal distances): ties go to the lower index; a lane's stripe gets
    drained (all neighbours of a duplicated point sit in few stripes)."""
    U = sub("utils.utils")
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pc = np.concatenate([g, g[:64], np.zeros((128, 3), np.float32)], 0)[None]         # grid + duplicates + 128 copies of the origin
    x = torch.from_numpy(pc).cuda()
    cent = torch.tensor([[0, 5, 100, 1023, 1024, pc.shape[1] - 1]], dtype=torch.int32).cuda()
    for k in (1, 7, 40, 200):
        got = U.knn_indices(x, cent, k).cpu().numpy()[0]
        want = fps_oracle.knn_indices(pc[0], cent[0].cpu().numpy(), k)
        assert np.array_equal(got, want), k


def test_knn_argument_errors(synth):
    U = sub("utils.utils")
    x = torch.from_numpy(synth.clouds(4, 1, 64)).cuda()
    c = torch.zeros((1, 4), dtype=torch.int32).cuda()
    with pytest.raises(IndexError):
        U.knn_indices(x, c, 65)
    with pytest.raises(Exception):
        U.knn_indices(x.cpu(), c, 4)               # no CPU fallback
    with pytest.raises(IndexError):
        U.knn_indices(x, c + 64, 4)
    big = torch.zeros((1, 12289, 3), device="cuda")
    with pytest.raises(Exception):
        U.knn_indices(big, c, 4)                   # coordinates must fit LDS
