"""GPU parity of the encoder / head forward (through the C ABI) against the reference's golden outputs and the
oracle.  Tolerance: north_star asks logits within 1e-3 (fp32) of the reference CPU forward; the intermediate
tensors here are held to 2e-4 of their scale."""
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


def _close(got, want, tol, what):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max()
    scale = max(1.0, np.abs(want).max())
    assert err <= tol * scale, f"{what}: max|diff| {err:.3e} > {tol:.1e} * {scale:.3g}"


def _enc_tables(synth, params, seed):
    ops = sub("ops")
    p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_params(seed, params.ENC_PARAMS).items()}
    b = {k: torch.from_numpy(v).cuda() for k, v in synth.make_buffers(seed, params.ENC_BUFFERS).items()}
    return ops, p, b, ops.PointerTable(params.ENC_PARAMS, p, "enc params"), ops.PointerTable(params.ENC_BUFFERS, b, "enc buffers")


def test_encoder_eval_matches_reference_golden(golden, synth, params):
    g = golden("encoder")
    ops, p, b, pt, bt = _enc_tables(synth, params, 1)
    x = synth.windows(21, 2, 256)
    xd = torch.from_numpy(x.reshape(-1, 9)).cuda()
    off, total, mx = ops.window_offsets([256, 256], xd.device)
    local, glob, ft, it = ops.encoder_forward(pt, bt, xd, off, 2, total, mx, 1, False, ops.Workspace(), want_in_T=True)
    _close(it, g["eval_in_T"], 2e-5, "input transform")
    _close(ft, g["eval_feat_T"], 2e-5, "feature transform")
    _close(local.reshape(2, 256, 64), g["eval_local"], 2e-5, "local")
    _close(glob, g["eval_global"], 2e-5, "global")


def test_encoder_train_matches_reference_golden(golden, synth, params):
    g = golden("encoder")
    ops, p, b, pt, bt = _enc_tables(synth, params, 1)
    x = synth.windows(22, 4, 128)
    xd = torch.from_numpy(x.reshape(-1, 9)).cuda()
    off, total, mx = ops.window_offsets([128] * 4, xd.device)
    local, glob, ft, _ = ops.encoder_forward(pt, bt, xd, off, 4, total, mx, 1, True, ops.Workspace())
    # B = 4 rows in the T-Net FC BatchNorms: the reference itself sits 4e-4 from fp64 here (test_oracle_golden.py)
    _close(local.reshape(4, 128, 64), g["train_local"], 2e-4, "local")
    _close(glob, g["train_global"], 2e-4, "global")
    _close(ft, g["train_feat_T"], 2e-4, "feature transform")
    for k in ["bn_1.running_mean", "bn_1.running_var", "bn_6.running_var", "input_transform.bn_4.running_mean",
              "input_transform.bn_4.running_var", "feature_transform.bn_3.running_var", "feature_transform.bn_5.running_mean"]:
        _close(b[k], g["train_" + k], 1e-4, k)


@pytest.mark.parametrize("B,W,N", [(3, 2, 96), (8, 3, 160), (4, 9, 512), (16, 2, 544), (8, 2, 300), (4, 2, 1100)])
def test_encoder_train_slots_match_oracle(synth, params, B, W, N):
    """All B*W windows in one launch sequence == W oracle encoder calls on the B windows of each slot."""
    ops, p, b, pt, bt = _enc_tables(synth, params, 5)
    x = synth.windows(100 + B, B * W, N)                       # window q = b * W + w
    xd = torch.from_numpy(x.reshape(-1, 9)).cuda()
    off, total, mx = ops.window_offsets([N] * (B * W), xd.device)
    local, glob, ft, it = ops.encoder_forward(pt, bt, xd, off, B * W, total, mx, W, True, ops.Workspace(), want_in_T=True)
    op = torch_params(synth.make_params(5, params.ENC_PARAMS))
    ob = torch_params(synth.make_buffers(5, params.ENC_BUFFERS))
    op = {k: v.double() for k, v in op.items()}
    ob = {k: v.double() for k, v in ob.items()}
    xw = torch.from_numpy(x).double().reshape(B, W, N, 9)
    local = local.reshape(B, W, N, 64)
    glob = glob.reshape(B, W, 256)
    ft = ft.reshape(W, B, 64, 64)                               # slot-major
    for w in range(W):
        l, gg, t = O.encoder(op, ob, xw[:, w], train=True)
        _close(local[:, w], l.float(), 3e-4, f"local slot {w}")
        _close(glob[:, w], gg.float(), 3e-4, f"global slot {w}")
        _close(ft[w], t.float(), 3e-4, f"feat_T slot {w}")
    for k in ob:
        _close(b[k], ob[k].float(), 2e-4, k)


def test_encoder_eval_ragged_windows_match_oracle(synth, params):
    """Inference feeds clusters of different sizes (test_pointnet_att_segmen.py:160-164)."""
    ops, p, b, pt, bt = _enc_tables(synth, params, 6)
    sizes = [37, 512, 1, 700, 2048, 33, 129]
    xs = [synth.windows(200 + i, 1, n)[0] for i, n in enumerate(sizes)]
    xd = torch.from_numpy(np.concatenate(xs, 0)).cuda()
    off, total, mx = ops.window_offsets(sizes, xd.device)
    local, glob, ft, _ = ops.encoder_forward(pt, bt, xd, off, len(sizes), total, mx, 1, False, ops.Workspace())
    op = torch_params(synth.make_params(6, params.ENC_PARAMS))
    ob = torch_params(synth.make_buffers(6, params.ENC_BUFFERS))
    r0 = 0
    for i, n in enumerate(sizes):
        l, gg, t = O.encoder(op, ob, torch.from_numpy(xs[i])[None], train=False)
        _close(local[r0:r0 + n], l[0], 5e-5, f"local window {i}")
        _close(glob[i], gg[0], 5e-5, f"global window {i}")
        _close(ft[i], t[0], 5e-5, f"feat_T window {i}")
        r0 += n


def _head_tables(synth, params, seed):
    ops = sub("ops")
    p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_params(seed, params.HEAD_PARAMS).items()}
    b = {k: torch.from_numpy(v).cuda() for k, v in synth.make_buffers(seed, params.HEAD_BUFFERS).items()}
    return ops, p, b, ops.PointerTable(params.HEAD_PARAMS, p, "head params"), ops.PointerTable(params.HEAD_BUFFERS, b, "head buffers")


def _head_inputs(synth):
    gl = synth.uniform(31, (3, 2, 256), 0.0, 2.0)            # [W, B, E] like the reference
    lo = synth.uniform(32, (2, 768, 64), -1.0, 1.0)
    cent = synth.uniform(33, (2, 3, 2), -1.0, 1.0)
    return gl, lo, cent


@pytest.mark.parametrize("case", ["uniform_masked", "ragged_nomask"])
def test_head_eval_matches_reference_golden(golden, synth, params, case):
    g = golden("head")
    ops, p, b, pt, bt = _head_tables(synth, params, 2)
    gl, lo, cent = _head_inputs(synth)
    npc = [256, 256, 256] if case == "uniform_masked" else [100, 300, 368]
    mask = torch.tensor([[False, False, True], [False, False, False]]) if case == "uniform_masked" else None
    B, W = 2, 3
    gld = torch.from_numpy(np.ascontiguousarray(gl.transpose(1, 0, 2)).reshape(B * W, 256)).cuda()    # row b*W+w
    lod = torch.from_numpy(lo.reshape(-1, 64)).cuda()
    off, total, mx = ops.window_offsets(npc * B, lod.device)
    logits, _, _ = ops.head_forward(pt, bt, gld, lod, torch.from_numpy(cent).cuda(), off, mask, B, W, total, mx, 5,
                                    False, 0.3, 0, ops.Workspace())
    _close(logits, g[case], 1e-4, "logits")     # north_star bar: 1e-3


def test_head_train_dropout_matches_oracle(synth, params):
    """Train mode: batch statistics + the counter-based dropout, whose keep-masks the oracle restates bit for bit."""
    ops, p, b, pt, bt = _head_tables(synth, params, 7)
    B, W, npc = 4, 3, [160, 96, 224]
    Pp = sum(npc)
    gl = synth.uniform(41, (W, B, 256), 0.0, 2.0)
    lo = synth.uniform(42, (B, Pp, 64), -1.0, 1.0)
    cent = synth.uniform(43, (B, W, 2), -1.0, 1.0)
    tg = synth.randint(44, (B, Pp), -1, 5)
    mask = torch.tensor([[False, True, False]] + [[False] * 3] * 3)
    drop_p, seed = 0.3, 1234
    gld = torch.from_numpy(np.ascontiguousarray(gl.transpose(1, 0, 2)).reshape(B * W, 256)).cuda()
    lod = torch.from_numpy(lo.reshape(-1, 64)).cuda()
    off, total, mx = ops.window_offsets(npc * B, lod.device)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0])
    logits, preds, loss = ops.head_forward(pt, bt, gld, lod, torch.from_numpy(cent).cuda(), off, mask, B, W, total, mx, 5,
                                           True, drop_p, seed, ops.Workspace(), targets=torch.from_numpy(tg), class_w=cw,
                                           want_preds=True)
    op = {k: v.double() for k, v in torch_params(synth.make_params(7, params.HEAD_PARAMS)).items()}
    ob = {k: v.double() for k, v in torch_params(synth.make_buffers(7, params.HEAD_BUFFERS)).items()}

    def km(stream, n):
        return torch.from_numpy(O.keep_mask(seed, stream, n, drop_p).astype(np.float64))
    masks = {"att": km(0, B * 8 * W * W).reshape(B * 8, W, W),
             "d2": km(1, B * Pp * 128).reshape(B, Pp, 128).transpose(1, 2),
             "d3": km(2, B * Pp * 64).reshape(B, Pp, 64).transpose(1, 2)}
    want = O.head(op, ob, torch.from_numpy(gl).double(), torch.from_numpy(lo).double(), torch.from_numpy(cent).double(),
                  npc, mask, True, drop_p=drop_p, drop_masks=masks)
    _close(logits, want.float(), 2e-4, "train logits")
    ce, _ = O.loss_terms(want, torch.from_numpy(tg), torch.eye(64, dtype=torch.float64)[None])
    assert abs(loss[0].item() - ce.item()) <= 2e-5 * max(1.0, abs(ce.item()))
    pw = O.predictions(want).numpy()
    assert (preds.cpu().numpy() != pw).mean() < 1e-3
    for k in ob:
        _close(b[k], ob[k].float(), 1e-4, k)
