"""Edge shapes through the public modules: one cluster, the maximum of 32 clusters, 2 / 8 classes, windows of 4, 33, 257
points (not multiples of the 32-row MFMA tile, the 256-row work item or the 512-row chunk).  Eval logits against the oracle
(north_star bar: 1e-3), train step: finite loss and gradients."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402
from helpers import torch_params                   # noqa: E402
from oracle import ampnet_oracle as O              # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,W,N,C", [(1, 1, 50, 5), (2, 32, 64, 5), (3, 2, 33, 2), (2, 5, 1000, 8), (5, 3, 257, 5), (2, 2, 4, 5)])
def test_edge_shapes(synth, params, B, W, N, C):
    M, S, T = sub("pointNet.model.pointnetAtt"), sub("pointNet.amp_step"), sub("trainer")
    hp_table = dict(params.HEAD_PARAMS)
    hp_table["conv_4.weight"] = (C, 64, 1)
    hp_table["conv_4.bias"] = (C,)
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=C, local_dim=64, device="cuda")
    ep, eb = synth.make_params(3, params.ENC_PARAMS), synth.make_buffers(3, params.ENC_BUFFERS)
    hp, hb = synth.make_params(4, hp_table), synth.make_buffers(4, params.HEAD_BUFFERS)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**ep, **eb}.items()}, strict=False)
    att.load_state_dict({k: torch.from_numpy(v) for k, v in {**hp, **hb}.items()}, strict=False)
    pc, tg, cent, _ = synth.sample_batch(900 + B, B, N, max_w=W)
    tg = np.where(tg >= 0, tg % C, tg)
    x = np.ascontiguousarray(pc.transpose(0, 3, 1, 2))
    t = np.ascontiguousarray(tg.transpose(0, 2, 1))
    enc.eval(); att.eval()
    with torch.no_grad():
        out = S.forward_batch(enc, att, x, t, cent, None, want_loss=False, want_preds=True)
        logits, _, _, _ = O.forward_windows(torch_params(ep), torch_params(eb), torch_params(hp), torch_params(hb),
                                            torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent), False, False)
    assert (out["logits"].cpu() - logits).abs().max().item() <= 1e-3
    if B >= 2:                                     # BatchNorm needs more than one window per slot in train mode
        enc.train(); att.train()
        o2 = T.forward_backward(enc, att, x, t, cent, torch.ones(C, device="cuda"))
        assert np.isfinite(float(o2["ce"][0])) and np.isfinite(float(o2["reg"]))
        assert all(torch.isfinite(p.grad).all().item() for m in (enc, att) for p in m.parameters())
