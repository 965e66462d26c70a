"""The GRU variant on the HIP path: SegmentationWithGRU (pointNet/model/pointnetAtt.py:212-258) and the reference's GRU
train_loop (pointNet/rnn/train_pointnetGRU.py:335-441) against what the reference itself returned (tests/golden/gru.npz, made by
tests/golden/make_golden.py:sec_gru) and against the float64 oracle.  Bars: eval logits 1e-3 (observed ~1e-5); loss terms 1e-4;
gradients of the first train step -- the GRU head's own tensors within 2e-3 of the reference's fp32 gradients stored in full, the encoder's
with the noise-scaled float64 bar of tests/test_step_gpu.py (they pass through the T-Net FC BatchNorms over B rows); with dropout on,
the HIP step equals the oracle run with the same keep-masks."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402
from helpers import torch_params                   # noqa: E402

pytestmark = pytest.mark.gpu


class _NoOpt:
    def zero_grad(self):
        pass


def _models(synth, params, enc_seed, head_seed, p_drop=None):
    M = sub("pointNet.model.pointnetAtt")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    gru = M.SegmentationWithGRU(num_classes=5, global_feat_size=256, hidden_size=64, device="cuda")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(enc_seed, params.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(enc_seed, params.ENC_BUFFERS).items()})
    r = enc.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith("num_batches_tracked") for k in r.missing_keys)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(head_seed, params.GRU_HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(head_seed, params.HEAD_BUFFERS).items()})
    r = gru.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith("num_batches_tracked") for k in r.missing_keys)
    if p_drop is not None:
        gru.p_drop = p_drop
    return enc, gru


def test_gru_state_dict_keys_match_reference(synth, params):
    _, gru = _models(synth, params, 5, 6)
    want = set(params.GRU_HEAD_PARAMS) | set(params.HEAD_BUFFERS) | {"bn_2.num_batches_tracked", "bn_3.num_batches_tracked"}
    assert set(gru.state_dict().keys()) == want
    assert list(dict(gru.named_parameters()).keys()) == list(params.GRU_HEAD_PARAMS.keys())       # nn.Module order of the reference
    M = sub("pointNet.model.pointnetAtt")
    with pytest.raises(AttributeError):
        M.ClassificationFromGRU(num_classes=5)(torch.zeros(2, 5, 256, device="cuda"))            # the reference's forward fails the same way


def test_gru_head_eval_matches_reference(golden, synth, params):
    g = golden("gru")
    _, gru = _models(synth, params, 5, 6)
    gru.eval()
    seq = torch.from_numpy(synth.uniform(71, (2, 3, 256), 0.0, 2.0)).cuda()
    lo = torch.from_numpy(synth.uniform(72, (2, 768, 64), -1.0, 1.0)).cuda()
    with torch.no_grad():
        a = gru(seq, lo, [256, 256, 256])
        b = gru(seq, lo, [100, 300, 368])
    for got, key in ((a, "uniform"), (b, "ragged")):
        err = np.abs(got.cpu().numpy() - g[key]).max()
        print(f"GRU head eval logits ({key}): max abs err {err:.2e}")
        assert err <= 1e-3
    with pytest.raises(Exception):
        gru(seq.cpu(), lo.cpu(), [256, 256, 256])                                               # no CPU fallback


def _batch(golden, synth):
    g = golden("gru")
    B, N, W = [int(v) for v in g["meta"]]
    pc, tg, cent, _ = synth.sample_batch(43, B, N, max_w=W, w_real=[int(v) for v in g["w_real"]])
    data = (torch.from_numpy(pc), torch.from_numpy(tg), [f"f{i}" for i in range(B)], torch.from_numpy(cent))
    return g, data, pc, tg


def _oracle_grads(synth, params, pc, tg, dt, drop_p=0.0, masks=None, relu_shift=0.0):
    """Gradients of the GRU step from the oracle.  relu_shift != 0 evaluates the same graph with every BatchNorm output moved by that
    amount before its ReLU: values change by 1e-5 (nothing), but every activation within 1e-5 of the ReLU kink switches side -- the
    difference to the unshifted gradients measures what a rounding-level change of a pre-activation can do to each tensor."""
    from oracle import ampnet_oracle as O
    orig = O.batchnorm_rows
    if relu_shift:
        O.batchnorm_rows = lambda *a, **k: orig(*a, **k) + relu_shift
    try:
        return _oracle_grads_inner(O, synth, params, pc, tg, dt, drop_p, masks)
    finally:
        O.batchnorm_rows = orig


def _kink_allowance(synth, params, pc, tg, w64, drop_p=0.0, masks=None):
    out = {k: 0.0 for k in w64}
    for sft in (1e-5, -1e-5):
        _, _, ws, _ = _oracle_grads(synth, params, pc, tg, torch.float64, drop_p, masks, relu_shift=sft)
        for k in w64:
            out[k] = max(out[k], float((ws[k] - w64[k]).norm()))
    return out


def _oracle_grads_inner(O, synth, params, pc, tg, dt, drop_p, masks):
    d = lambda dd, gr: {k: v.to(dt).requires_grad_(gr) for k, v in torch_params(dd).items()}      # noqa: E731
    ep, eb = d(synth.make_params(7, params.ENC_PARAMS), True), d(synth.make_buffers(7, params.ENC_BUFFERS), False)
    hp, hb = d(synth.make_params(8, params.GRU_HEAD_PARAMS), True), d(synth.make_buffers(8, params.HEAD_BUFFERS), False)
    lg, tpc, tf = O.forward_windows_gru(ep, eb, hp, hb, torch.from_numpy(pc).to(dt), torch.from_numpy(tg), True, True, drop_p=drop_p,
                                        drop_masks=None if masks is None else {k: v.to(dt) for k, v in masks.items()})
    c, r = O.loss_terms(lg, tpc, tf, class_w=(1.0,) * 5)
    (c + 0.001 * r).backward()
    out = {("enc", k): v.grad.double() for k, v in ep.items()}
    out.update({("gru", k): v.grad.double() for k, v in hp.items()})
    return c.item(), r.item(), out, lg.detach()


def test_gru_train_loop_eval_matches_reference(golden, synth, params):
    S = sub("pointNet.gru_step")
    enc, gru = _models(synth, params, 7, 8)
    g, data, _, _ = _batch(golden, synth)
    ce = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=-1)
    m, tpc, preds, _ = S.train_loop(data, _NoOpt(), _NoOpt(), ce, enc, gru, None, "segmentation", False, torch.Tensor(), 0, 0)
    for k, key in (("ce", "ce_loss"), ("reg", "reg_loss"), ("loss", "loss")):
        assert abs(m[key].item() - g[f"eval_{k}"].item()) <= 5e-5 * abs(g[f"eval_{k}"].item()), k
    assert np.array_equal(tpc.numpy(), g["eval_targets"])
    assert (preds.numpy() != g["eval_preds"]).mean() < 1e-3


def test_gru_train_loop_first_step_matches_reference(golden, synth, params):
    """One train step of the package's train_loop (fused path, FusedAdam) against the reference's own step (dropout p = 0)."""
    S = sub("pointNet.gru_step")
    T = sub("trainer")
    enc, gru = _models(synth, params, 7, 8, p_drop=0.0)
    g, data, pc, tg = _batch(golden, synth)
    ce = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=-1)
    opt_p, opt_g = T.FusedAdam(enc.parameters(), lr=1e-3), T.FusedAdam(gru.parameters(), lr=1e-3)
    m, tpc, preds, _ = S.train_loop(data, opt_p, opt_g, ce, enc, gru, None, "segmentation", True, torch.Tensor(), 0, 0)
    for k, key in (("ce", "ce_loss"), ("reg", "reg_loss"), ("loss", "loss")):
        assert abs(m[key].item() - g[f"s1_{k}"].item()) <= 1e-4 * abs(g[f"s1_{k}"].item()), (k, m[key].item(), g[f"s1_{k}"].item())
    assert (preds.numpy() != g["s1_preds"]).mean() < 2e-3
    _, _, w64, _ = _oracle_grads(synth, params, pc, tg, torch.float64)
    _, _, w32, _ = _oracle_grads(synth, params, pc, tg, torch.float32)
    kink = _kink_allowance(synth, params, pc, tg, w64)
    gtot = float(np.sqrt(sum(float(w.pow(2).sum()) for w in w64.values())))
    bad, worst = [], 0.0
    for tag, mod in (("enc", enc), ("gru", gru)):
        for k, p in mod.named_parameters():
            got = p.grad.double().cpu()
            w = w64[(tag, k)]
            nrm, err = float(w.norm()), float((got.reshape(w.shape) - w).norm())
            noise = float((w32[(tag, k)] - w).norm())
            gn = g[f"s1_{tag}_gnorm/{k}"]
            if abs(float(got.norm()) - gn[0]) > 1.5e-2 * gn[0] + 1e-5 * gtot:
                bad.append(("gnorm", tag, k, float(got.norm()), float(gn[0])))
            key = f"s1_{tag}_grad/{k}"
            if key in g.files:                                        # the GRU head's tensors, stored in full by the reference run
                ref = torch.from_numpy(g[key].astype(np.float64)).reshape(w.shape)
                noise = max(noise, float((ref - w).norm()))
            # one activation 5e-6 from the ReLU kink of bn_3 (channel 15) moves conv_3.weight's gradient by 6e-4 when it switches side
            # (tests/diagnostics/diag_gru_conv3.py, diag_gru_bnsum.py): the allowance is what the float64 oracle itself says such switches cost
            if err > 3.0 * noise + 2e-4 * nrm + 1e-5 * gtot + 1.5 * kink[(tag, k)]:
                bad.append(("f64", tag, k, err / (nrm + 1e-30), noise / (nrm + 1e-30), kink[(tag, k)] / (nrm + 1e-30)))
            worst = max(worst, err / (nrm + 1e-5 * gtot))
            ps = g[f"s1_{tag}_psum/{k}"]
            have = p.detach().double().abs().sum().item()
            # a bias in front of a BatchNorm (conv_2.bias, conv_3.bias) has an analytically zero gradient: Adam divides rounding noise by
            # its own magnitude and moves every element by +-lr in a direction that is noise, in the reference as well
            atol = 1.1e-3 * p.numel() if gn[0] < 1e-6 * gtot else 2.1e-3 * max(1.0, 0.02 * p.numel())
            if abs(have - ps[1]) > 2e-4 * abs(ps[1]) + atol:
                bad.append(("psum", tag, k, have, float(ps[1])))
    print(f"GRU train step: worst relative gradient error vs float64 {worst:.2e}")
    assert not bad, bad
    sd = gru.state_dict()
    for k in sd:
        if "running" in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"s1_gru_buf/{k}"], rtol=1e-3, atol=1e-4, err_msg=k)
    assert int(gru.bn_2.num_batches_tracked) == 1 and int(enc.bn_1.num_batches_tracked) == int(g["meta"][2])


def test_gru_reference_style_loop_with_autograd_and_dropout(golden, synth, params):
    """The reference's own loop body on the drop-in modules -- W encoder calls, SegmentationWithGRU(global_seq, local_feats, np_cluster),
    torch's CrossEntropyLoss, loss.backward() -- with dropout 0.3: equals the oracle evaluated with the same keep-masks."""
    from oracle import ampnet_oracle as O
    enc, gru = _models(synth, params, 7, 8)
    g, data, pc, tg = _batch(golden, synth)
    B, N, W = [int(v) for v in g["meta"]]
    enc.train(); gru.train()
    x = torch.from_numpy(pc).cuda()
    t = torch.from_numpy(tg).cuda()
    seed = gru.seed                                                   # first train-mode call: step counter 0
    lo_feats, gl_feats, targets_pc, ft = [], [], [], None
    for w in range(W):
        out, ft = enc(x[:, :, :, w])
        lo_feats.append(out[:, :, -64:])
        gl_feats.append(out[:, 0, :-64].view(-1, 1, 256))
        targets_pc.append(t[:, :, w])
    logits = gru(torch.cat(gl_feats, 1), torch.cat(lo_feats, 1), [N] * W)
    tp = torch.cat(targets_pc, 1)
    ce = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=-1)(logits, tp)
    reg = torch.norm(torch.eye(64, device="cuda") - torch.bmm(ft, ft.transpose(2, 1)))
    (ce + 0.001 * reg).backward()
    P = W * N
    masks = {"d2": torch.from_numpy(O.keep_mask(seed, 1, B * P * 128, 0.3).reshape(B, P, 128)).permute(0, 2, 1).double(),
             "d3": torch.from_numpy(O.keep_mask(seed, 2, B * P * 64, 0.3).reshape(B, P, 64)).permute(0, 2, 1).double()}
    c64, r64, w64, lg64 = _oracle_grads(synth, params, pc, tg, torch.float64, drop_p=0.3, masks=masks)
    c32, r32, w32, _ = _oracle_grads(synth, params, pc, tg, torch.float32, drop_p=0.3, masks=masks)
    assert abs(ce.item() - c64) <= 1e-4 * abs(c64) and abs(reg.item() - r64) <= 1e-4 * abs(r64)
    assert (logits.detach().double().cpu() - lg64).abs().max().item() <= 1e-3
    kink = _kink_allowance(synth, params, pc, tg, w64, 0.3, masks)
    gtot = float(np.sqrt(sum(float(w.pow(2).sum()) for w in w64.values())))
    bad = []
    for tag, mod in (("enc", enc), ("gru", gru)):
        for k, p in mod.named_parameters():
            w = w64[(tag, k)]
            err = float((p.grad.double().cpu().reshape(w.shape) - w).norm())
            noise = float((w32[(tag, k)] - w).norm())
            if err > 3.0 * noise + 2e-4 * float(w.norm()) + 1e-5 * gtot + 1.5 * kink[(tag, k)]:
                bad.append((tag, k, err / (float(w.norm()) + 1e-30), noise / (float(w.norm()) + 1e-30)))
    assert not bad, bad


@pytest.mark.parametrize("B,sizes", [(1, [7]), (1, [33, 1, 100]), (3, [5, 64, 31, 257]), (2, [2048] * 2 + [100] * 16)])
def test_gru_head_edge_shapes_match_oracle(synth, params, B, sizes):
    """Ragged / tiny / many windows (W = 1 .. 18, window sizes that are no multiple of the 32-row tiles) through the reference-signature
    forward in eval mode and one train-mode forward + backward, against the oracle."""
    from oracle import ampnet_oracle as O
    _, gru = _models(synth, params, 5, 6, p_drop=0.0)
    W, P = len(sizes), sum(sizes)
    seq = torch.from_numpy(synth.uniform(81, (B, W, 256), 0.0, 2.0))
    lo = torch.from_numpy(synth.uniform(82, (B, P, 64), -1.0, 1.0))
    hp = torch_params(synth.make_params(6, params.GRU_HEAD_PARAMS))
    hb = torch_params(synth.make_buffers(6, params.HEAD_BUFFERS))
    gru.eval()
    with torch.no_grad():
        got = gru(seq.cuda(), lo.cuda(), sizes)
    want = O.gru_head(hp, hb, seq, lo, sizes, train=False)
    assert got.shape == (B, 5, P)
    assert (got.cpu() - want).abs().max().item() <= 1e-3
    if B * P >= 8:                                             # a BatchNorm over a handful of rows is not a meaningful gradient test
        gru.train()
        s2, l2 = seq.cuda().requires_grad_(True), lo.cuda().requires_grad_(True)
        out = gru(s2, l2, sizes)
        tgt = torch.from_numpy(synth.randint(83, (B, P), 0, 5)).cuda()
        torch.nn.functional.cross_entropy(out, tgt).backward()
        p64 = {k: v.double().requires_grad_(True) for k, v in hp.items()}
        b64 = {k: v.double() for k, v in hb.items()}
        s64, l64 = seq.double().requires_grad_(True), lo.double().requires_grad_(True)
        o64 = O.gru_head(p64, b64, s64, l64, sizes, train=True)
        torch.nn.functional.cross_entropy(o64, tgt.cpu()).backward()
        assert (out.detach().double().cpu() - o64.detach()).abs().max().item() <= 2e-3
        for got_g, want_g, name in ((s2.grad, s64.grad, "global_seq"), (l2.grad, l64.grad, "local_feats")):
            err = float((got_g.double().cpu() - want_g).norm()) / (float(want_g.norm()) + 1e-12)
            assert err <= 2e-2, (name, err)
