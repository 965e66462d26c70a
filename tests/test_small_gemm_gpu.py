"""The token-level GEMM of the backward (csrc/bwd_misc.hip: sgemm_mfma_kernel behind ampnet_small_gemm_f32): every operand layout the
backward uses, ragged sizes, leading dimensions wider than the matrix, unaligned bases, accumulate, the bias-gradient row sums -- against
float64 matmul of the same fp32 inputs.  Bar: 1e-5 * K^0.5 * max|A| max|B| (fp32 accumulation of K products in a fixed order); run twice: bitwise
equal (no atomics)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402

# (trans_a, trans_b, M, N, K): the shapes of one AMP-Net step (Q = 576 windows) + ragged ones
SHAPES = [
    (True, False, 256, 256, 576),     # dW of out_proj = G^T X
    (False, False, 576, 256, 256),    # dX of out_proj = G W
    (True, False, 768, 256, 576),     # dW of in_proj
    (False, False, 576, 256, 768),    # dX of in_proj
    (True, False, 16, 2, 576),        # dW of the positional fc_1 (K rows, two inputs)
    (False, False, 576, 16, 256),     # dX of the positional fc_2
    (False, False, 576, 128, 512),    # one K split of the feature T-Net's fc_3
    (True, False, 4096, 128, 576),    # dW of that fc_3
    (False, True, 64, 50, 37),        # forward-style X W^T, everything ragged
    (True, True, 33, 31, 70),
    (False, False, 1, 1, 1),
    (True, False, 40, 9, 5),
    (False, True, 100, 65, 1030),     # more than one pass of 16 x 8 chunks
]


def _operands(synth, seed, ta, tb, M, N, K, pad_a, pad_b, offset):
    a_shape = (K, M) if ta else (M, K)
    b_shape = (N, K) if tb else (K, N)
    A = torch.from_numpy(synth.uniform(seed, (a_shape[0], a_shape[1] + pad_a + offset), -1.0, 1.0).astype(np.float32)).cuda()
    Bm = torch.from_numpy(synth.uniform(seed + 1, (b_shape[0], b_shape[1] + pad_b + offset), -1.0, 1.0).astype(np.float32)).cuda()
    return A[:, offset:offset + a_shape[1]], Bm[:, offset:offset + b_shape[1]]


@pytest.mark.gpu
@pytest.mark.parametrize("ta,tb,M,N,K", SHAPES)
@pytest.mark.parametrize("layout", ["dense", "padded", "unaligned"])
def test_small_gemm_matches_float64(synth, ta, tb, M, N, K, layout):
    ops = sub("ops")
    pad_a, pad_b, offset = {"dense": (0, 0, 0), "padded": (4, 8, 0), "unaligned": (3, 5, 1)}[layout]
    A, Bm = _operands(synth, 4200 + M + N + K, ta, tb, M, N, K, pad_a, pad_b, offset)
    opA = (A.t() if ta else A).double().cpu()
    opB = (Bm.t() if tb else Bm).double().cpu()
    want = opA @ opB
    out, rs = ops.small_gemm(A, Bm, ta, tb, want_row_sums=True)
    again, rs2 = ops.small_gemm(A, Bm, ta, tb, want_row_sums=True)
    assert torch.equal(out, again) and torch.equal(rs, rs2)
    bar = 1e-5 * np.sqrt(K) * float(opA.abs().max() * opB.abs().max())
    assert (out.double().cpu() - want).abs().max().item() <= bar, ((out.double().cpu() - want).abs().max().item(), bar)
    assert (rs.double().cpu() - opA.sum(1)).abs().max().item() <= 2e-6 * K
    # accumulate into a wider output
    wide = torch.ones((M, N + 3), dtype=torch.float32, device="cuda")
    ops.small_gemm(A, Bm, ta, tb, out=wide[:, :N], accumulate=True)
    assert (wide[:, :N].double().cpu() - (want + 1.0)).abs().max().item() <= bar + 1e-6
    assert torch.all(wide[:, N:] == 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("ta,tb,M,N,K", [(True, False, 128, 128, 256), (False, False, 9, 128, 256), (True, True, 33, 31, 70)])
def test_small_gemm_k_scale(synth, ta, tb, M, N, K):
    """C = op(A) diag(k_scale) op(B): the form the pooled layers' per-slot matrices W^T diag(P2) W use."""
    ops = sub("ops")
    A, Bm = _operands(synth, 5200 + M + N + K, ta, tb, M, N, K, 0, 0, 0)
    ks = torch.from_numpy(synth.uniform(5300 + K, (K,), -2.0, 2.0)).cuda()
    opA = (A.t() if ta else A).double().cpu()
    opB = (Bm.t() if tb else Bm).double().cpu()
    want = (opA * ks.double().cpu()[None, :]) @ opB
    out, rs = ops.small_gemm(A, Bm, ta, tb, want_row_sums=True, k_scale=ks)
    assert (out.double().cpu() - want).abs().max().item() <= 2e-5 * np.sqrt(K) * float(opA.abs().max() * opB.abs().max())
    assert (rs.double().cpu() - (opA * ks.double().cpu()[None, :]).sum(1)).abs().max().item() <= 4e-6 * K


@pytest.mark.gpu
def test_small_gemm_rejects_bad_operands(synth):
    ops, L = sub("ops"), sub("_lib")
    A = torch.zeros((8, 4), device="cuda")
    with pytest.raises(L.AmpnetError):
        ops.small_gemm(A, torch.zeros((5, 3), device="cuda"))                 # inner dimensions differ
    with pytest.raises(L.AmpnetError):
        ops.small_gemm(A.t(), torch.zeros((8, 3), device="cuda"))             # rows not contiguous
    with pytest.raises(L.AmpnetError):
        ops.small_gemm(A.double(), torch.zeros((4, 3), device="cuda"))
