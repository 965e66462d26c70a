"""RCCL de-risk on a one-GPU box (round-2 review item 8): the `nccl` backend of torch.distributed IS RCCL on ROCm, and until a multi-GPU
node runs bench.py the data-parallel exchanges had only ever executed on gloo.  Here ONE rank initialises `nccl` on the box's GPU and, under
AMPNET_FORCE_COLLECTIVES=1, drives the complete exchange path of a train step through the real collective calls (tests/rccl_worker.py):
async head all-reduce inside forward_backward, work.wait() + encoder all-reduce in reduce_gradients, FusedAdam's 1 / world scale, and -- in
the second case -- the 36 global-batch BatchNorm exchanges of the C callback with its launch-stream check, plus the loss and epoch-metric
all-reduces.  With one rank every collective is the identity, so the result must EQUAL the plain single-process step (no process group):
loss terms, every gradient, the parameters after Adam and the running statistics.  Bars: plain all-reduce path bit-equal; global-batch
BatchNorm path within 1e-5 / 1e-4 (its statistics take the Chan-merge route through the gathered partials instead of the direct one).
No scaling claim: a one-rank ring moves no bytes.  The measured host cost of the exchanges per step is printed (and bench.py reports it)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sub                           # noqa: E402

pytestmark = pytest.mark.gpu
B, N, W = 32, 64, 3          # the shape of tests/test_syncbn_gpu.py (its 1e-2 gradient bar was measured there: 4.9e-3)


def _run_rank(out, sync):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AMPNET_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out, str(int(sync)), str(B), str(N), str(W)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out, weights_only=True)


def _plain_step():
    import syncbn_worker as Wk
    synth, params = sub("synthetic"), sub("params")
    M, T = sub("pointNet.model.pointnetAtt"), sub("trainer")
    enc, att = Wk.build(synth, params, M)
    x, t, c = Wk.batch(synth, B, N, W)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    opt_p, opt_a = T.FusedAdam(enc.parameters(), lr=1e-3), T.FusedAdam(att.parameters(), lr=1e-3)
    out = T.fused_train_step(enc, att, opt_p, opt_a, x.cuda(), t.cuda(), c.cuda(), cw)
    assert not out["pending"]
    torch.cuda.synchronize()
    res = {"ce": out["ce"][0:1].detach().cpu(), "reg": out["reg"].detach().reshape(1).cpu()}
    for tag, m in (("enc", enc), ("att", att)):
        for k, p in m.named_parameters():
            res[f"grad/{tag}/{k}"] = p.grad.detach().cpu().clone()
            res[f"param/{tag}/{k}"] = p.detach().cpu().clone()
        for k, b in m.named_buffers():
            if "running" in k:
                res[f"buf/{tag}/{k}"] = b.detach().cpu().clone()
    import time
    for _ in range(2):
        T.fused_train_step(enc, att, opt_p, opt_a, x.cuda(), t.cuda(), c.cuda(), cw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        T.fused_train_step(enc, att, opt_p, opt_a, x.cuda(), t.cuda(), c.cuda(), cw)
    torch.cuda.synchronize()
    res["ms_per_step"] = torch.tensor((time.perf_counter() - t0) / 5 * 1e3)
    return res


def test_gradient_allreduce_on_rccl_equals_plain_step(tmp_path):
    plain = _plain_step()
    got = _run_rank(str(tmp_path / "rccl.pt"), False)
    assert int(got["n_pending"]) == 1 and float(got["metrics_ok"]) == 1.0
    for k, v in plain.items():
        if k == "ms_per_step":
            continue
        assert torch.equal(got[k], v), f"{k}: the one-rank RCCL step differs from the plain step (max |diff| {(got[k] - v).abs().max().item():.3e})"
    print(f"one-rank RCCL step {got['ms_per_step'].item():.3f} ms vs plain {plain['ms_per_step'].item():.3f} ms "
          f"(B = {B}, N = {N}, W = {W}: the two all-reduces on the host path)")


def test_sync_batchnorm_on_rccl_equals_plain_step(tmp_path):
    plain = _plain_step()
    got = _run_rank(str(tmp_path / "rccl_sync.pt"), True)
    assert abs(got["ce"].item() - plain["ce"].item()) <= 1e-5 * abs(plain["ce"].item())
    assert abs(got["reg"].item() - plain["reg"].item()) <= 1e-5 * abs(plain["reg"].item())
    gtot = float(np.sqrt(sum(float(v.double().pow(2).sum()) for k, v in plain.items() if k.startswith("grad/"))))
    worst = 0.0
    for k, v in plain.items():
        if k.startswith("grad/"):
            e = float((got[k].double() - v.double()).norm()) / (float(v.double().norm()) + 1e-5 * gtot)
            worst = max(worst, e)
            assert e <= 1e-2, (k, e)          # same bar as the two-rank gloo test (T-Net FC BatchNorm rounding amplification)
        elif k.startswith("buf/"):
            np.testing.assert_allclose(got[k].numpy(), v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    print(f"one-rank RCCL step with global-batch BatchNorm: worst relative gradient difference {worst:.2e}; "
          f"{got['ms_per_step'].item():.3f} ms per step vs plain {plain['ms_per_step'].item():.3f} ms = the host cost of the 36 exchanges + loss all-reduce")
