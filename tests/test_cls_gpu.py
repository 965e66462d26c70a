"""ClassificationWithAttention on the HIP path (pointNet/model/pointnetAtt.py:115-151) against the reference module's own
outputs and autograd gradients (tests/golden/cls.npz, made by tests/golden/make_golden.py:sec_cls).  Bars: outputs / attention weights
1e-4; gradients 5e-3 of each tensor's norm against the reference's fp32 autograd (BatchNorm over B = 16 rows)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402

pytestmark = pytest.mark.gpu


def _net(synth, params, g, dropout=0.0):
    Wn, B, C = [int(v) for v in g["meta"]]
    M = sub("pointNet.model.pointnetAtt")
    net = M.ClassificationWithAttention(256, 8, num_classes=C, dropout=dropout, num_w=Wn, device="cuda")
    table = params.cls_head_params(C, Wn)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(9, table).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(9, params.CLS_HEAD_BUFFERS).items()})
    r = net.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith("num_batches_tracked") for k in r.missing_keys)
    assert list(dict(net.named_parameters()).keys()) == list(table.keys())
    gl = torch.from_numpy(synth.uniform(91, (Wn, B, 256), 0.0, 2.0)).cuda()
    mask = torch.zeros(B, Wn, dtype=torch.bool)
    mask[1, 3:] = True
    mask[7, 4] = True
    return net, gl, mask.cuda(), (Wn, B, C)


def test_cls_head_eval_matches_reference(golden, synth, params):
    g = golden("cls")
    net, gl, mask, _ = _net(synth, params, g)
    net.eval()
    with torch.no_grad():
        out, aw = net(gl, None, mask)
    np.testing.assert_allclose(out.cpu().numpy(), g["eval_out"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(aw.cpu().numpy(), g["eval_weights"], rtol=1e-4, atol=1e-6)
    with pytest.raises(Exception):
        net(gl.cpu(), None, None)


def test_cls_head_train_step_matches_reference(golden, synth, params):
    g = golden("cls")
    net, gl, mask, (Wn, B, C) = _net(synth, params, g)
    net.train()
    glg = gl.clone().requires_grad_(True)
    out, aw = net(glg, None, mask)
    tgt = torch.from_numpy(synth.randint(92, (B,), 0, C)).cuda()
    loss = torch.nn.functional.cross_entropy(out, tgt)
    loss.backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["train_out"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(aw.cpu().numpy(), g["train_weights"], rtol=1e-4, atol=1e-6)
    assert abs(loss.item() - g["loss"].item()) <= 1e-4 * abs(g["loss"].item())
    ref = g["d_gl"].astype(np.float64)                       # [W, B, 256]
    got = glg.grad.double().cpu().numpy()
    assert np.linalg.norm(got - ref) <= 5e-3 * np.linalg.norm(ref)
    worst = 0.0
    gtot = np.sqrt(sum(float((g[f"grad/{k}"].astype(np.float64) ** 2).sum()) for k, _ in net.named_parameters()))
    for k, p in net.named_parameters():                      # fc_2.bias (in front of bn_2) has an analytically zero gradient
        ref = g[f"grad/{k}"].astype(np.float64)
        err = np.linalg.norm(p.grad.double().cpu().numpy().reshape(ref.shape) - ref)
        worst = max(worst, err / (np.linalg.norm(ref) + 1e-3 * gtot))
        assert err <= 5e-3 * np.linalg.norm(ref) + 1e-5 * gtot, (k, err, np.linalg.norm(ref), gtot)
    print(f"classification head: worst relative gradient error vs the reference's autograd {worst:.2e}")
    sd = net.state_dict()
    for k in sd:
        if "running" in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"buf/{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
    assert int(net.bn_2.num_batches_tracked) == 1


def test_cls_head_dropout_matches_oracle_masks(golden, synth, params):
    """Attention dropout 0.3: the HIP forward / backward equal the oracle evaluated with the same keep-mask."""
    from oracle import ampnet_oracle as O
    from helpers import torch_params
    g = golden("cls")
    net, gl, mask, (Wn, B, C) = _net(synth, params, g, dropout=0.3)
    net.train()
    seed = net.seed
    glg = gl.clone().requires_grad_(True)
    out, aw = net(glg, None, mask)
    tgt = torch.from_numpy(synth.randint(92, (B,), 0, C)).cuda()
    torch.nn.functional.cross_entropy(out, tgt).backward()
    keep = torch.from_numpy(O.keep_mask(seed, 0, B * 8 * Wn * Wn, 0.3)).double().reshape(B * 8, Wn, Wn)
    p = {k: v.double().requires_grad_(True) for k, v in torch_params(synth.make_params(9, params.cls_head_params(C, Wn))).items()}
    b = {k: v.double() for k, v in torch_params(synth.make_buffers(9, params.CLS_HEAD_BUFFERS)).items()}
    gl64 = gl.double().cpu().requires_grad_(True)
    o64, a64 = O.cls_head(p, b, gl64, mask.cpu(), True, drop_p=0.3, drop_mask=keep)
    torch.nn.functional.cross_entropy(o64, tgt.cpu()).backward()
    assert (out.detach().double().cpu() - o64.detach()).abs().max().item() <= 1e-4
    assert (aw.double().cpu() - a64.detach()).abs().max().item() <= 1e-5
    gtot64 = float(np.sqrt(sum(float(v.grad.pow(2).sum()) for v in p.values())))
    for k, v in net.named_parameters():
        ref = p[k].grad
        assert float((v.grad.double().cpu().reshape(ref.shape) - ref).norm()) <= 5e-3 * float(ref.norm()) + 1e-5 * gtot64, k
    assert float((glg.grad.double().cpu() - gl64.grad).norm()) <= 5e-3 * float(gl64.grad.norm())
