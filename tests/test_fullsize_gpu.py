"""Parity of the BENCHMARKED code path at BASELINE.json's full sizes (configs[1] verbatim, configs[2] shape).

Why a separate file: `pw_gemm` / `pw_bwd` are persistent kernels -- a workgroup walks several blocks of rows and re-stages
its prologue constants when the BatchNorm slot changes.  That loop only iterates more than once when Q * chunks exceeds the
resident grid (256..512 workgroups), which none of the small-shape tests reach.  The shapes here give 1152 (B = 32) and 2304
(B = 64) blocks of rows, i.e. 2..9 iterations per workgroup with slot changes in between -- the code `bench.py` times.

Reference lines: pointNet/model/pointnetAtt.py:80-112,176-209; pointNet/self-attention/train_pointnet-attention.py:337-475.
Arbiter: the oracle (oracle/ampnet_oracle.py, pinned to the reference by tests/test_oracle_golden.py).  Tolerances:
logits 1e-3 absolute (north_star's bar, fp32), loss terms 1e-4 relative, running statistics 2e-4 of their scale,
gradients: distance to the float64 oracle within 4x the distance of the oracle's own float32 evaluation + 1e-3 of the
tensor's norm (tests/test_backward_gpu.py explains the yardstick)."""
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

N_POINTS, N_WIN = 2048, 9


def _models(synth, params, dropout=0.3):
    M = sub("pointNet.model.pointnetAtt")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, dropout=dropout, device="cuda")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(3, params.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(3, params.ENC_BUFFERS).items()})
    enc.load_state_dict(sd, strict=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(4, params.HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(4, params.HEAD_BUFFERS).items()})
    att.load_state_dict(sd, strict=False)
    return enc, att


def _oracle_state(synth, params, dt, grad):
    d = lambda dd, g: {k: v.to(dt).requires_grad_(g) for k, v in torch_params(dd).items()}      # noqa: E731
    return (d(synth.make_params(3, params.ENC_PARAMS), grad), d(synth.make_buffers(3, params.ENC_BUFFERS), False),
            d(synth.make_params(4, params.HEAD_PARAMS), grad), d(synth.make_buffers(4, params.HEAD_BUFFERS), False))


def _device_batch(pc, tg, cent):
    x = torch.from_numpy(np.ascontiguousarray(pc.transpose(0, 3, 1, 2))).cuda()       # [B, W, N, 9]
    t = torch.from_numpy(np.ascontiguousarray(tg.transpose(0, 2, 1))).cuda()          # [B, W, N]
    return x, t, torch.from_numpy(cent).cuda()


def test_config2_eval_forward_matches_oracle(synth, params):
    """BASELINE.json configs[1] verbatim: forward only, B = 32, W = 9, N = 2048, fp32; logits within 1e-3 of the CPU forward."""
    S = sub("pointNet.amp_step")
    B = 32
    enc, att = _models(synth, params)
    enc.eval(); att.eval()
    pc, tg, cent, _ = synth.sample_batch(100, B, N_POINTS, max_w=N_WIN)
    x, t, centd = _device_batch(pc, tg, cent)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    with torch.no_grad():
        out = S.forward_batch(enc, att, x, t, centd, cw, want_loss=True, want_preds=True)
    torch.cuda.synchronize()
    ep, eb, hp, hb = _oracle_state(synth, params, torch.float32, False)
    with torch.no_grad():
        logits, tpc, t_feat, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(pc), torch.from_numpy(tg),
                                                   torch.from_numpy(cent), False, False)
        ce, _ = O.loss_terms(logits, tpc, t_feat)
        want_preds = O.predictions(logits).numpy()
    got = out["logits"].cpu()
    assert got.shape == logits.shape == (B, 5, N_WIN * N_POINTS)
    err = (got - logits).abs().max().item()
    assert err <= 1e-3, f"config 2 logits: max |diff| {err:.3e} > 1e-3"
    assert abs(out["ce"][0].item() - ce.item()) <= 1e-4 * abs(ce.item())
    assert (out["preds"].cpu().numpy() != want_preds).mean() < 1e-3
    assert torch.equal(out["targets_pc"].cpu(), tpc)
    ft = out["feat_last"].cpu()
    assert (ft - t_feat).abs().max().item() <= 2e-4 * max(1.0, t_feat.abs().max().item())


def test_config3_size_eval_equals_sub_batches(synth, params):
    """576 windows in one launch sequence (2304 blocks of rows: every persistent workgroup iterates ~9 times) equal the
    same samples run 8 at a time (72 windows: one block per workgroup).  Eval mode uses running statistics and the attention is
    per sample, so the two must agree to fp32 summation noise -- no CPU oracle time needed at B = 64."""
    S = sub("pointNet.amp_step")
    B, SB = 64, 8
    enc, att = _models(synth, params)
    enc.eval(); att.eval()
    pc, tg, cent, _ = synth.sample_batch(100, B, N_POINTS, max_w=N_WIN)
    x, t, centd = _device_batch(pc, tg, cent)
    with torch.no_grad():
        full = S.forward_batch(enc, att, x, t, centd, None, want_loss=False, want_preds=True)
        lg_full, pr_full = full["logits"].clone(), full["preds"].clone()
        for b0 in range(0, B, SB):
            part = S.forward_batch(enc, att, x[b0:b0 + SB].contiguous(), t[b0:b0 + SB].contiguous(), centd[b0:b0 + SB].contiguous(),
                                   None, want_loss=False, want_preds=True)
            d = (part["logits"] - lg_full[b0:b0 + SB]).abs().max().item()
            assert d <= 2e-5, f"samples {b0}..{b0 + SB}: sub-batch logits differ by {d:.3e}"
            assert (part["preds"] != pr_full[b0:b0 + SB]).float().mean().item() < 1e-4


def _check_grads(got, want64, want32, what):
    gtot = float(np.sqrt(sum(float(v.double().pow(2).sum()) for v in want64.values())))
    rel_floor = 0.0
    for k, w in want64.items():
        ref = float(w.norm())
        if ref >= 1e-3 * gtot:
            rel_floor = max(rel_floor, float((want32[k].double() - w).norm()) / ref)
    bad, worst = [], 0.0
    for k, w in want64.items():
        g = got[k].detach().cpu().double().reshape(w.shape)
        err, ref = float((g - w).norm()), float(w.norm())
        noise = float((want32[k].double() - w).norm())
        tol = max(4.0 * noise, 1.5 * rel_floor * ref) + 1e-3 * ref + 1e-5 * gtot
        worst = max(worst, err / (ref + 1e-5 * gtot))
        if not err <= tol:
            bad.append((k, err, noise, ref))
    assert not bad, f"{what} (worst torch-fp32 relative noise {rel_floor:.2e}): " + "; ".join(
        f"{k}: err {e:.3e} fp32-noise {n:.3e} |g| {r:.3e}" for k, e, n, r in bad[:8])
    return worst, rel_floor


def test_config3_shape_train_step_matches_oracle(synth, params):
    """A configs[2]-shaped train step (B = 32 samples x 9 windows x 2048 points, dropout 0.3) through the fused path
    (trainer.forward_backward, what bench.py times): loss terms, logits, per-slot running statistics and every parameter
    gradient against the oracle driven with the SAME dropout keep-masks."""
    T = sub("trainer")
    B, drop_p = 32, 0.3
    enc, att = _models(synth, params, dropout=drop_p)
    enc.train(); att.train()
    pc, tg, cent, _ = synth.sample_batch(101, B, N_POINTS, max_w=N_WIN)
    x, t, centd = _device_batch(pc, tg, cent)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    seed = att.seed & 0xFFFFFFFF                              # the first train step of a fresh module (att._step == 0)
    out = T.forward_backward(enc, att, x, t, centd, cw)
    torch.cuda.synchronize()
    got = {"enc/" + k: p.grad.clone() for k, p in enc.named_parameters()}
    got.update({"att/" + k: p.grad.clone() for k, p in att.named_parameters()})
    assert all(torch.isfinite(g).all() for g in got.values())
    Pp = N_WIN * N_POINTS
    keep = {s: O.keep_mask(seed, s, n, drop_p) for s, n in ((0, B * 8 * N_WIN * N_WIN), (1, B * Pp * 128), (2, B * Pp * 64))}
    want, aux = {}, {}
    for dt in (torch.float64, torch.float32):
        ep, eb, hp, hb = _oracle_state(synth, params, dt, True)
        masks = {"att": torch.from_numpy(keep[0]).to(dt).reshape(B * 8, N_WIN, N_WIN),
                 "d2": torch.from_numpy(keep[1]).to(dt).reshape(B, Pp, 128).transpose(1, 2),
                 "d3": torch.from_numpy(keep[2]).to(dt).reshape(B, Pp, 64).transpose(1, 2)}
        logits, tpc, t_feat, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(pc).to(dt), torch.from_numpy(tg),
                                                   torch.from_numpy(cent).to(dt), True, True, drop_p=drop_p, drop_masks=masks)
        ce, reg = O.loss_terms(logits, tpc, t_feat)
        (ce + 0.001 * reg).backward()
        want[dt] = {"enc/" + k: v.grad for k, v in ep.items()}
        want[dt].update({"att/" + k: v.grad for k, v in hp.items()})
        if dt == torch.float64:
            aux = dict(ce=ce.item(), reg=reg.item(), logits=logits.detach().float(), eb=eb, hb=hb)
        del logits, masks, ce, reg
    assert abs(out["ce"][0].item() - aux["ce"]) <= 1e-4 * abs(aux["ce"]), (out["ce"][0].item(), aux["ce"])
    assert abs(out["reg"].item() - aux["reg"]) <= 1e-4 * abs(aux["reg"]), (out["reg"].item(), aux["reg"])
    err = (out["logits"].cpu() - aux["logits"]).abs().max().item()
    assert err <= 1e-3, f"train logits: max |diff| {err:.3e}"
    for mod, bufs in ((enc, aux["eb"]), (att, aux["hb"])):
        sd = mod.state_dict()
        for k, v in bufs.items():                            # per-slot running statistics, W sequential updates
            d = (sd[k].cpu().double() - v).abs().max().item()
            assert d <= 2e-4 * max(1.0, v.abs().max().item()), (k, d)
    worst, floor = _check_grads(got, want[torch.float64], want[torch.float32], "config-3-shaped step")
    print(f"full-size train step: worst relative gradient error {worst:.2e} (torch fp32 noise floor {floor:.2e})")


def _bench_pin_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_bench_pin", os.path.join(ROOT, "tests", "golden", "make_bench_pin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_shape_train_forward_matches_oracle(synth, params):
    """The step bench.py TIMES, verbatim (BASELINE.json configs[2]: B = 64 samples x 9 windows x 2048 points, dropout 0.3, rank-0
    batch seed 100, fresh modules): 2304 blocks of rows, so every persistent pw_gemm workgroup runs its train-mode statistics /
    max-pool epilogue ~9 times with slot changes in between.  trainer.forward_backward's loss terms, logits and the per-slot running
    statistics against the oracle's float32 train-mode forward with the same dropout keep-masks (no autograd: affordable at B = 64),
    and the oracle's numbers against tests/golden/bench_pin.json, which bench.py checks its first step against.
    Bars: ce / reg 1e-4 relative, logits 1e-3 absolute (north_star), running statistics 2e-4 of their scale, pin file 2e-5 relative
    (the oracle is torch-CPU float32: its summation order depends on the host's thread count)."""
    import json
    T = sub("trainer")
    pin_mod = _bench_pin_module()
    B = 64
    enc, att = _models(synth, params, dropout=pin_mod.DROP_P)
    assert att.seed == pin_mod.ATT_SEED and att._step == 0
    enc.train(); att.train()
    pc, tg, cent, _ = synth.sample_batch(100, B, N_POINTS, max_w=N_WIN)
    x, t, centd = _device_batch(pc, tg, cent)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    out = T.forward_backward(enc, att, x, t, centd, cw)
    torch.cuda.synchronize()
    got_ce, got_reg, got_logits = out["ce"][0].item(), out["reg"].item(), out["logits"].cpu()
    assert all(torch.isfinite(p.grad).all() for m in (enc, att) for p in m.parameters())
    want = pin_mod.train_forward(synth, params, B, pin_mod.ATT_SEED)
    assert abs(got_ce - want["ce"]) <= 1e-4 * abs(want["ce"]), (got_ce, want["ce"])
    assert abs(got_reg - want["reg"]) <= 1e-4 * abs(want["reg"]), (got_reg, want["reg"])
    err = (got_logits - want["logits"]).abs().max().item()
    assert err <= 1e-3, f"B = 64 train logits: max |diff| {err:.3e}"
    for mod, bufs in ((enc, want["eb"]), (att, want["hb"])):
        sd = mod.state_dict()
        for k, v in bufs.items():
            d = (sd[k].cpu() - v).abs().max().item()
            assert d <= 2e-4 * max(1.0, v.abs().max().item()), (k, d)
    pin = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_pin.json")))["train_B64"]
    assert abs(want["ce"] - pin["ce"]) <= 2e-5 * abs(pin["ce"]), ("bench_pin.json is stale: rerun tests/golden/make_bench_pin.py", want["ce"], pin["ce"])
    assert abs(want["reg"] - pin["reg"]) <= 2e-5 * abs(pin["reg"]), (want["reg"], pin["reg"])
    print(f"bench shape: ce {got_ce:.7f} vs oracle {want['ce']:.7f}, reg {got_reg:.5f} vs {want['reg']:.5f}, logits max diff {err:.2e}")


def test_train_step_is_bitwise_reproducible(synth, params):
    """No float atomics, every reduction in a fixed order ("Determinism"): two fresh model pairs stepped on the
    same batch give bit-identical losses, logits, gradients and parameters after Adam -- at a shape where the persistent kernels
    iterate and the last-arriver finalisations run (B = 16: 576 blocks of rows)."""
    T = sub("trainer")
    B = 16
    pc, tg, cent, _ = synth.sample_batch(103, B, N_POINTS, max_w=N_WIN)
    x, t, centd = _device_batch(pc, tg, cent)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    runs = []
    for _ in range(2):
        enc, att = _models(synth, params, dropout=0.3)
        tr = T.Trainer(enc, att, lr=1e-3, class_w=cw)
        rec = []
        for _s in range(2):
            out = tr.step(x, t, centd)
            rec.append((out["ce"].clone(), out["reg"].clone(), out["logits"].clone(), [p.grad.clone() for m in (enc, att) for p in m.parameters()]))
        torch.cuda.synchronize()
        rec.append([p.detach().clone() for m in (enc, att) for p in m.parameters()] + [b.clone() for m in (enc, att) for b in m.buffers()])
        runs.append(rec)
    a, b = runs
    for s in range(2):
        assert torch.equal(a[s][0], b[s][0]) and torch.equal(a[s][1], b[s][1]), f"step {s}: loss terms differ between two runs"
        assert torch.equal(a[s][2], b[s][2]), f"step {s}: logits differ"
        for i, (ga, gb) in enumerate(zip(a[s][3], b[s][3])):
            assert torch.equal(ga, gb), f"step {s}: gradient tensor {i} differs between two runs"
    for i, (pa, pb) in enumerate(zip(a[2], b[2])):
        assert torch.equal(pa, pb), f"parameter / buffer {i} differs after two steps"
