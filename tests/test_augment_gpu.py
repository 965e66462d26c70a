"""Device-side input pipeline of train_loop (include/ampnet_hip.h: ampnet_augment_f32) against the host augmentation that
replays the reference's numpy draws (amp_step.augment_batch, pinned against the reference in tests/test_host_cpu.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,N,W", [(4, 256, 9), (3, 100, 5), (2, 2048, 9)])
def test_device_augmentation_equals_host(synth, train, B, N, W):
    S = sub("pointNet.amp_step")
    pc, tg, _, _ = synth.sample_batch(55, B, N, max_w=W)          # [B, N, 9, W], [B, N, W]
    pct, tgt = torch.from_numpy(pc), torch.from_numpy(tg)
    np.random.seed(4242)
    xh, th = S.augment_batch(pct.clone(), tgt.clone(), train)
    draws_after_host = np.random.uniform()
    np.random.seed(4242)
    xd, td = S.augment_batch_device(pct.clone(), tgt.clone(), train, "cuda")
    assert np.random.uniform() == draws_after_host                  # the same number of draws was consumed
    xd, td = xd.cpu().numpy(), td.cpu().numpy()
    assert xd.shape == xh.shape and td.shape == th.shape
    assert np.array_equal(td, th)
    assert np.array_equal(xd[..., 3:], xh[..., 3:])                  # features: moved, never touched
    assert np.array_equal(xd[..., 2], xh[..., 2])                    # z is not rotated
    if train:
        # x, y: float64 rotation rounded to float32; a BLAS fused multiply-add may differ in the last float64 bit
        diff = xd[..., :2] != xh[..., :2]
        assert diff.mean() < 1e-5
        assert np.abs(xd[..., :2] - xh[..., :2]).max() <= 2.4e-7 * max(1.0, np.abs(xh[..., :2]).max())
    else:
        assert np.array_equal(xd, xh)


def test_train_loop_device_and_host_pipelines_agree(synth, params, monkeypatch):
    """The same seeded eval step through both pipelines: identical targets, predictions and loss."""
    from test_step_gpu import _models, _NoOpt
    S = sub("pointNet.amp_step")
    enc, att = _models(synth, params)
    pc, tg, cent, _ = synth.sample_batch(41, 4, 256, max_w=3)
    data = (torch.from_numpy(pc), torch.from_numpy(tg), ["f"] * 4, torch.from_numpy(cent))
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    res = []
    for host in ("1", "0"):
        monkeypatch.setenv("AMPNET_HOST_AUG", host)
        np.random.seed(9)
        m, tpc, preds, _ = S.train_loop(data, _NoOpt(), _NoOpt(), ce, enc, att, None, "segmentation", False, 0, 0)
        res.append((m["loss"].item(), tpc.numpy().copy(), preds.numpy().copy()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
