"""GPU parity of the backward kernels (through the C ABI) with float64 autograd of the oracle.

Tolerance: relative error of each gradient tensor in the L2 norm.  The T-Net FC BatchNorms normalise over only B
rows, which makes fp32 gradients of the reference itself noisy (tests/test_oracle_golden.py: 2e-3 at B = 16), so
the bar is 2e-2 for tensors behind those layers and 2e-3 elsewhere, plus a floor of 1e-5 of the global gradient norm."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402
from oracle import ampnet_oracle as O              # noqa: E402
from helpers import torch_params                   # noqa: E402

pytestmark = pytest.mark.gpu


def _check_grads(got, want, loose, tight, what):
    gtot = float(np.sqrt(sum(float(v.double().pow(2).sum()) for v in want.values())))
    bad = []
    for k, w in want.items():
        g = got[k].detach().cpu().double().reshape(w.shape)
        err = float((g - w).norm())
        ref = float(w.norm())
        tol = (loose if ("transform" in k) else tight) * ref + 1e-5 * gtot
        if not err <= tol:
            bad.append((k, err, ref))
    assert not bad, f"{what}: " + "; ".join(f"{k}: err {e:.3e} vs |g| {r:.3e}" for k, e, r in bad[:8])


@pytest.mark.parametrize("B,W,N", [(8, 2, 96), (16, 3, 160)])
def test_encoder_backward_matches_oracle_autograd(synth, params, B, W, N):
    ops = sub("ops")
    p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_params(5, params.ENC_PARAMS).items()}
    b = {k: torch.from_numpy(v).cuda() for k, v in synth.make_buffers(5, params.ENC_BUFFERS).items()}
    grads = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    pt = ops.PointerTable(params.ENC_PARAMS, p, "p")
    bt = ops.PointerTable(params.ENC_BUFFERS, b, "b")
    gt = ops.PointerTable(params.ENC_PARAMS, grads, "g")
    Q = B * W
    x = synth.windows(300 + B, Q, N)
    xd = torch.from_numpy(x.reshape(-1, 9)).cuda()
    off, total, mx = ops.window_offsets([N] * Q, xd.device)
    fws, bws = ops.Workspace(), ops.Workspace()
    local, glob, ft, _ = ops.encoder_forward(pt, bt, xd, off, Q, total, mx, W, True, fws)
    r1 = synth.uniform(401, (Q * N, 64), -1, 1)
    r2 = synth.uniform(402, (Q, 256), -1, 1)
    r3 = synth.uniform(403, (Q, 64, 64), -1, 1)          # slot-major rows, like feat_T
    ops.encoder_backward(pt, gt, xd, off, Q, total, mx, W, local, ft, torch.from_numpy(r1).cuda(), torch.from_numpy(r2).cuda(),
                         torch.from_numpy(r3).cuda(), fws, bws)
    # oracle: W encoder calls in float64, the same linear loss
    op = {k: v.double().requires_grad_(True) for k, v in torch_params(synth.make_params(5, params.ENC_PARAMS)).items()}
    ob = {k: v.double() for k, v in torch_params(synth.make_buffers(5, params.ENC_BUFFERS)).items()}
    xw = torch.from_numpy(x).double().reshape(B, W, N, 9)
    R1 = torch.from_numpy(r1).double().reshape(B, W, N, 64)
    R2 = torch.from_numpy(r2).double().reshape(B, W, 256)
    R3 = torch.from_numpy(r3).double().reshape(W, B, 64, 64)
    loss = 0.0
    for w in range(W):
        l, g, t = O.encoder(op, ob, xw[:, w], train=True)
        loss = loss + (l * R1[:, w]).sum() + (g * R2[:, w]).sum() + (t * R3[w]).sum()
    loss.backward()
    want = {k: v.grad for k, v in op.items()}
    assert all(torch.isfinite(g).all() for g in grads.values()), "a gradient was not written"
    _check_grads(grads, want, 2e-2, 2e-3, "encoder")
