"""End-to-end step parity on the GPU: the package's train_loop against what the reference's own train_loop
returned on the same seeded batch (tests/golden/step.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402

pytestmark = pytest.mark.gpu


class _NoOpt:
    def zero_grad(self):
        pass


def _models(synth, params, dropout=0.3):
    M = sub("pointNet.model.pointnetAtt")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, dropout=dropout, device="cuda")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(3, params.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(3, params.ENC_BUFFERS).items()})
    r = enc.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith("num_batches_tracked") for k in r.missing_keys)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(4, params.HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(4, params.HEAD_BUFFERS).items()})
    r = att.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith("num_batches_tracked") for k in r.missing_keys)
    return enc, att


TIGHT = {("att", "conv_3.weight"), ("att", "conv_3.bias"), ("att", "conv_4.weight"), ("att", "conv_4.bias"), ("att", "bn_3.weight"), ("att", "bn_3.bias")}


def _grad_rtol(tag, key):
    """Relative bar of one gradient tensor against the REFERENCE's own fp32 gradient (tests/golden/step.npz).
    Measured on the MI355X (tests/diagnostics/diag_step_grads.py, B = 16, N = 64, W = 3): the reference's fp32 gradients sit 1e-3 .. 5e-3
    (bn_5.bias: 1.3e-2) from a float64 evaluation of the same graph on EVERY tensor whose value depends on a T-Net -- the encoder
    and, through the global features, the attention and conv_2 of the head -- and so does any other fp32 evaluation (torch's own:
    same figures).  Only the last two head layers are free of that noise: there the HIP gradients agree with the reference to
    1e-4 .. 1e-6 where the golden file stores them in full (torch's fp32 evaluation of conv_3.weight is itself 8e-4 from float64).
    Hence 2e-3 for those and 1.5e-2 (observed worst 8.7e-3) for the rest; the float64-arbitrated check of _check_against_f64
    below is the sharper statement (1e-4 on the well-conditioned tensors, noise-scaled on the others)."""
    return 2e-3 if (tag, key) in TIGHT else 1.5e-2


def _check_against_f64(synth, params, g, pc, tg, cent, seed, got):
    """Arbiter = the oracle in float64 on the batch train_loop saw (the augmentation replayed from the same numpy seed).
    A HIP gradient may be no further from float64 than 3x the distance of the better-known fp32 evaluations of the same graph
    (the reference's own gradient from the golden file where it is stored in full, torch-CPU fp32 of the oracle otherwise)
    + 2e-4 of its norm (observed ratio <= 1.85); the well-conditioned tensors (TIGHT) within 1e-4 outright (observed 1e-5)."""
    from oracle import ampnet_oracle as O
    from helpers import replay_augment, torch_params
    apc, atg = replay_augment(seed, pc, tg, True)
    want = {}
    for dt in (torch.float64, torch.float32):
        d = lambda dd, gr: {k: v.to(dt).requires_grad_(gr) for k, v in torch_params(dd).items()}      # noqa: E731
        ep, eb = d(synth.make_params(3, params.ENC_PARAMS), True), d(synth.make_buffers(3, params.ENC_BUFFERS), False)
        hp, hb = d(synth.make_params(4, params.HEAD_PARAMS), True), d(synth.make_buffers(4, params.HEAD_BUFFERS), False)
        lg, tpc, tf, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(apc).to(dt), torch.from_numpy(atg),
                                           torch.from_numpy(cent).to(dt), True, True)
        c, r = O.loss_terms(lg, tpc, tf)
        (c + 0.001 * r).backward()
        want[dt] = {("enc", k): v.grad.double() for k, v in ep.items()}
        want[dt].update({("att", k): v.grad.double() for k, v in hp.items()})
    gtot = float(np.sqrt(sum(float(w.pow(2).sum()) for w in want[torch.float64].values())))
    bad = []
    for key, w in want[torch.float64].items():
        nrm = float(w.norm())
        err = float((got[key].reshape(w.shape) - w).norm())
        noise = float((want[torch.float32][key] - w).norm())
        gk = f"s1_{key[0]}_grad/{key[1]}"
        if gk in g.files:
            noise = max(noise, float((torch.from_numpy(g[gk].astype(np.float64)).reshape(w.shape) - w).norm()))
        tol = (1e-4 * nrm if key in TIGHT else 3.0 * noise + 2e-4 * nrm) + 1e-5 * gtot
        if not err <= tol:
            bad.append((key, err / (nrm + 1e-30), noise / (nrm + 1e-30)))
    assert not bad, "; ".join(f"{k}: rel err {e:.2e} (fp32 noise {n:.2e})" for k, e, n in bad[:8])


def _batch(golden, synth):
    g = golden("step")
    B, N, W = [int(v) for v in g["meta"]]
    pc, tg, cent, _ = synth.sample_batch(41, B, N, max_w=W, w_real=[int(v) for v in g["w_real"]])
    data = (torch.from_numpy(pc), torch.from_numpy(tg), [f"f{i}" for i in range(B)], torch.from_numpy(cent))
    return g, data


def test_state_dict_keys_match_reference(synth, params):
    enc, att = _models(synth, params)
    want = set(params.ENC_PARAMS) | set(params.ENC_BUFFERS) | {k.replace("running_mean", "num_batches_tracked") for k in params.ENC_BUFFERS if k.endswith("running_mean")}
    assert set(enc.state_dict().keys()) == want
    want = set(params.HEAD_PARAMS) | set(params.HEAD_BUFFERS) | {"bn_2.num_batches_tracked", "bn_3.num_batches_tracked"}
    assert set(att.state_dict().keys()) == want
    assert sum(p.numel() for p in enc.parameters()) == 883401 and sum(p.numel() for p in att.parameters()) == 317621


def test_train_loop_eval_matches_reference(golden, synth, params):
    S = sub("pointNet.amp_step")
    enc, att = _models(synth, params)
    g, data = _batch(golden, synth)
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    np.random.seed(777)
    m, tpc, preds, _ = S.train_loop(data, _NoOpt(), _NoOpt(), ce, enc, att, None, "segmentation", False, 0, 0)
    assert abs(m["ce_loss"].item() - g["eval_ce"].item()) <= 5e-5 * abs(g["eval_ce"].item())
    assert abs(m["reg_loss"].item() - g["eval_reg"].item()) <= 5e-5 * abs(g["eval_reg"].item())
    assert abs(m["loss"].item() - g["eval_loss"].item()) <= 5e-5 * abs(g["eval_loss"].item())
    assert np.array_equal(tpc.numpy(), g["eval_targets"])
    assert (preds.numpy() != g["eval_preds"]).mean() < 1e-3


def test_reference_signature_forward_eval(golden, synth, params):
    """BasePointNet.forward / SegmentationWithAttention.forward with the reference's own shapes and returns."""
    enc, att = _models(synth, params)
    enc.eval(); att.eval()
    x = torch.from_numpy(synth.windows(21, 2, 256)).cuda()
    with torch.no_grad():
        out, ft = enc(x)
    assert out.shape == (2, 256, 320) and ft.shape == (2, 64, 64)
    assert torch.equal(out[:, 0, :256], out[:, 5, :256])
    gl = torch.stack([out[:, 0, :-64]] * 3, 0)                       # [W, B, 256]
    lo = torch.cat([out[:, :, -64:]] * 3, 1)                         # [B, 3 * 256, 64]
    with torch.no_grad():
        logits, zero = att(gl, lo, torch.zeros(2, 3, 2).cuda(), [256, 256, 256])
    assert logits.shape == (2, 5, 768) and zero == 0
    with pytest.raises(Exception):
        enc(x.cpu())                                                 # no CPU fallback


def test_train_loop_two_steps_match_reference(golden, synth, params):
    """The reference's own train_loop ran two train steps on this seeded batch (dropout p = 0, make_golden.py:sec_step).
    Step 1 pins loss terms and every gradient norm; both steps pin the Adam update (parameter |sum| after the step)."""
    S = sub("pointNet.amp_step")
    T = sub("trainer")
    enc, att = _models(synth, params, dropout=0.0)
    g, data = _batch(golden, synth)
    W = int(g["meta"][2])
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    opt_p = T.FusedAdam(enc.parameters(), lr=1e-3)
    opt_a = T.FusedAdam(att.parameters(), lr=1e-3)
    for step in (1, 2):
        np.random.seed(1000 + step)
        d = (data[0].clone(), data[1].clone(), data[2], data[3])
        m, tpc, preds, _ = S.train_loop(d, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
        rt = 1e-4 if step == 1 else 2e-3          # step 2 inherits Adam's sign-flip noise (test_oracle_golden.py)
        for k in ("ce", "reg", "loss"):
            want = g[f"s{step}_{k}"].item()
            assert abs(m[k + "_loss" if k != "loss" else "loss"].item() - want) <= rt * abs(want), (step, k)
        assert np.array_equal(tpc.numpy(), g[f"s{step}_targets"])
        if step == 1:
            assert (preds.numpy() != g["s1_preds"]).mean() < 2e-3
            gtot = np.sqrt(sum(float(g[k][0]) ** 2 for k in g.files if k.startswith("s1_") and "_gnorm/" in k))
            for tag, mod in (("enc", enc), ("att", att)):
                for k, p in mod.named_parameters():
                    gn = g[f"s1_{tag}_gnorm/{k}"]
                    got = p.grad.double()
                    rt_g = _grad_rtol(tag, k)
                    assert abs(got.norm().item() - gn[0]) <= rt_g * gn[0] + 1e-5 * gtot, (k, got.norm().item(), gn[0])
                    key = f"s1_{tag}_grad/{k}"
                    if key in g.files:
                        err = np.linalg.norm(got.cpu().numpy() - g[key].astype(np.float64))
                        assert err <= rt_g * gn[0] + 1e-5 * gtot, (k, err, gn[0])
            hip = {(tag, k): p.grad.double().cpu() for tag, mod in (("enc", enc), ("att", att)) for k, p in mod.named_parameters()}
            _check_against_f64(synth, params, g, data[0].numpy(), data[1].numpy(), data[3].numpy(), 1000 + step, hip)
        for tag, mod in (("enc", enc), ("att", att)):
            for k, p in mod.named_parameters():
                ps = g[f"s{step}_{tag}_psum/{k}"]
                np.testing.assert_allclose(p.detach().double().abs().sum().item(), ps[1], rtol=2e-4,
                                           atol=2.1e-3 * step * max(1.0, 0.02 * p.numel()), err_msg=k)
    assert int(enc.bn_1.num_batches_tracked) == 2 * W and int(att.bn_2.num_batches_tracked) == 2
    for tag, mod in (("enc", enc), ("att", att)):
        sd = mod.state_dict()
        for k in sd:
            if "running" in k:
                np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"final_{tag}_buf/{k}"], rtol=1e-2, atol=5e-3, err_msg=k)   # after two noisy Adam steps


def test_reference_style_loop_with_autograd(golden, synth, params):
    """The reference's own train_loop body, driven through the drop-in modules: W encoder calls, torch.cat, the module
    forward with the reference signature, torch's CrossEntropyLoss + torch.norm reg, loss.backward(), torch.optim.Adam.
    Must reproduce the reference's step-1 loss and gradients like the fused path does."""
    import torch.nn.functional as F
    enc, att = _models(synth, params, dropout=0.0)
    g, data = _batch(golden, synth)
    S = sub("pointNet.amp_step")
    ce_loss = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]).cuda(), reduction="mean", ignore_index=-1)
    opt_p = torch.optim.Adam(enc.parameters(), lr=1e-3)
    opt_a = torch.optim.Adam(att.parameters(), lr=1e-3)
    np.random.seed(1001)
    x, t = S.augment_batch(data[0].clone(), data[1].clone(), True)          # [B, W, N, 9], [B, W, N]
    B, W, N, _ = x.shape
    enc.train(); att.train()
    opt_p.zero_grad(); opt_a.zero_grad()
    lo, gl, tp = [], [], []
    for w in range(W):
        out, feat_transform = enc(torch.from_numpy(x[:, w]).cuda())
        lo.append(out[:, :, -64:])
        gl.append(out[:, 0, :-64].view(-1, 1, 256))
        tp.append(torch.from_numpy(t[:, w]).cuda())
    lo, gl, targets_pc = torch.cat(lo, 1), torch.cat(gl, 1), torch.cat(tp, 1)
    mask = (targets_pc.view(B, -1, W) == -1).all(1)
    logits, _ = att(gl.transpose(0, 1), lo, data[3].cuda(), [N] * W, mask)
    ce = ce_loss(logits, targets_pc)
    eye = torch.eye(64, device="cuda")
    reg = torch.norm(eye - torch.bmm(feat_transform, feat_transform.transpose(2, 1)))
    loss = ce + 0.001 * reg
    loss.backward()
    opt_p.step(); opt_a.step()
    assert abs(ce.item() - g["s1_ce"].item()) <= 1e-4 * abs(g["s1_ce"].item())
    assert abs(reg.item() - g["s1_reg"].item()) <= 1e-4 * abs(g["s1_reg"].item())
    gtot = np.sqrt(sum(float(g[k][0]) ** 2 for k in g.files if k.startswith("s1_") and "_gnorm/" in k))
    for tag, mod in (("enc", enc), ("att", att)):
        for k, p in mod.named_parameters():
            gn = g[f"s1_{tag}_gnorm/{k}"]
            assert abs(p.grad.double().norm().item() - gn[0]) <= _grad_rtol(tag, k) * gn[0] + 1e-5 * gtot, k
            ps = g[f"s1_{tag}_psum/{k}"]
            np.testing.assert_allclose(p.detach().double().abs().sum().item(), ps[1], rtol=2e-4,
                                       atol=2.1e-3 * max(1.0, 0.02 * p.numel()), err_msg=k)


def test_validation_split_miou_identical_to_oracle(synth, params):
    """north_star: identical mIoU on a fixed synthetic validation split.  8 samples (2 batches of 4), W = 9 cluster
    slots, N = 512: per-class IoU (utils/get_metrics.py definition), their mean and the accuracy from the HIP
    predictions equal the oracle's on the same weights."""
    sys.path.insert(0, os.path.join(ROOT))
    from oracle import ampnet_oracle as O
    from helpers import torch_params
    S = sub("pointNet.amp_step")
    M = sub("utils.get_metrics")
    enc, att = _models(synth, params)
    enc.eval(); att.eval()
    ep = torch_params(synth.make_params(3, params.ENC_PARAMS)); eb = torch_params(synth.make_buffers(3, params.ENC_BUFFERS))
    hp = torch_params(synth.make_params(4, params.HEAD_PARAMS)); hb = torch_params(synth.make_buffers(4, params.HEAD_BUFFERS))
    all_hip, all_ora, all_tg = [], [], []
    for b in range(2):
        pc, tg, cent, _ = synth.sample_batch(700 + b, 4, 512, max_w=9)
        x = np.ascontiguousarray(pc.transpose(0, 3, 1, 2))
        t = np.ascontiguousarray(tg.transpose(0, 2, 1))
        with torch.no_grad():
            out = S.forward_batch(enc, att, x, t, cent, None, want_loss=False, want_preds=True)
            logits, tpc, _, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent), False, False)
        all_hip.append(out["preds"].cpu().numpy().reshape(-1))
        all_ora.append(O.predictions(logits).numpy().reshape(-1))
        all_tg.append(tpc.numpy().reshape(-1))
    hip, ora, tgt = np.concatenate(all_hip), np.concatenate(all_ora), np.concatenate(all_tg)
    keep = tgt != -1
    assert (hip != ora).sum() <= 3                              # fp32 argmax ties only
    iou_h = [M.get_iou_obj(hip[keep], tgt[keep], c) for c in range(5)]
    iou_o = [O.iou_obj(ora[keep], tgt[keep], c) for c in range(5)]
    assert np.allclose(iou_h, iou_o, atol=2e-5, equal_nan=True)
    assert abs(np.nanmean(iou_h) - np.nanmean(iou_o)) < 1e-5


def test_training_reduces_loss_and_learns_the_labels(synth, params):
    """End-to-end sanity of forward + backward + FusedAdam: 60 steps on one fixed synthetic batch (the labels are a learnable
    function of the features, synthetic.labels_for) must cut the loss and lift the accuracy well above the class prior."""
    T = sub("trainer")
    M = sub("pointNet.model.pointnetAtt")
    torch.manual_seed(0)
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cuda")
    enc.train(); att.train()
    tr = T.Trainer(enc, att, lr=1e-3, class_w=torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda"))
    pc, tg, cent, _ = synth.sample_batch(321, 16, 256, max_w=3)
    x = torch.from_numpy(np.ascontiguousarray(pc.transpose(0, 3, 1, 2))).cuda()
    t = torch.from_numpy(np.ascontiguousarray(tg.transpose(0, 2, 1))).cuda()
    c = torch.from_numpy(cent).cuda()
    hist = []
    for _ in range(60):
        out = tr.step(x, t, c)
        hist.append(float(out["ce"][0]))
    assert all(np.isfinite(hist))
    assert np.mean(hist[-5:]) < 0.6 * np.mean(hist[:3]), (hist[:3], hist[-5:])
    keep = t.reshape(16, -1) != -1
    acc = ((out["preds"] == t.reshape(16, -1)) & keep).sum().item() / keep.sum().item()
    prior = max((t[t != -1] == k).float().mean().item() for k in range(5))
    assert acc > prior + 0.1, (acc, prior)
