"""Global-batch BatchNorm under data parallelism (SURVEY section 8(e) option A; include/ampnet_hip.h: ampnet_set_collective): a train step
of TWO ranks, each on half of a seeded batch, must equal the single-process step on the whole batch -- loss terms, every gradient after
the gradient all-reduce and the 1 / world average, running statistics.  The reference is single-device, so its BatchNorm always sees
the whole batch (pointNet/model/pointnetAtt.py:80-112); with per-rank statistics (the default) the same comparison is off by tens of
per cent, which the test also shows.  Two gloo ranks share the box's one GPU; dropout is 0 (the masks are indexed by local rows)."""
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
B, N, W = 32, 64, 3


def _run_ranks(out, sync):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "syncbn_worker.py"), out, str(int(sync)), str(B), str(N), str(W)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out, weights_only=True)


def test_two_rank_step_equals_single_process_step(tmp_path):
    import syncbn_worker as Wk
    synth, params = sub("synthetic"), sub("params")
    M, T = sub("pointNet.model.pointnetAtt"), sub("trainer")
    enc, att = Wk.build(synth, params, M)
    x, t, c = Wk.batch(synth, B, N, W)
    one = Wk.step(T, enc, att, x, t, c)                                   # the whole batch in this process
    two = _run_ranks(str(tmp_path / "sync.pt"), True)
    loc = _run_ranks(str(tmp_path / "local.pt"), False)
    assert abs(two["ce"].item() - one["ce"].item()) <= 1e-5 * abs(one["ce"].item())
    assert abs(two["reg"].item() - one["reg"].item()) <= 1e-5 * abs(one["reg"].item())
    gtot = float(np.sqrt(sum(float(v.double().pow(2).sum()) for k, v in one.items() if k.startswith("grad/"))))
    worst_sync, worst_local, bad = 0.0, 0.0, []
    for k, v in one.items():
        if k.startswith("grad/"):
            nrm = float(v.double().norm())
            e2 = float((two[k].double() - v.double()).norm()) / (nrm + 1e-5 * gtot)
            e1 = float((loc[k].double() - v.double()).norm()) / (nrm + 1e-5 * gtot)
            worst_sync, worst_local = max(worst_sync, e2), max(worst_local, e1)
            # the two evaluations differ in summation order only; the T-Net FC BatchNorms amplify that rounding (tests/test_step_gpu.py)
            if e2 > 1e-2:                                                  # observed worst 4.9e-3 (input T-Net tensors), 1.46 with per-rank statistics
                bad.append((k, e2))
        elif k.startswith("buf/"):
            np.testing.assert_allclose(two[k].numpy(), v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    print(f"2-rank step vs 1-rank step on the same batch: worst relative gradient error {worst_sync:.2e} with global-batch BatchNorm, "
          f"{worst_local:.2e} with per-rank BatchNorm")
    assert not bad, bad
    assert worst_local > 10 * worst_sync                                  # the comparison is sensitive to what it tests
