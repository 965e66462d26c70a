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


@pytest.mark.parametrize("train", [True, False])
def test_ragged_collate_plus_device_gather_equals_padded_path(synth, train):
    """include/ampnet_hip.h: ampnet_collate_augment_f32 -- the resampling to 2048 points and the padding to 9 clusters that collate_seq_padd
    does on the host (pointNet/collate_fns.py:33-45), done inside the augmentation kernel from the ragged batch of collate_seq_ragged: the
    device tensors must be BIT-identical to collate_seq_padd -> ampnet_augment_f32 under the same python / torch / numpy seeds (fewer
    points than 2048, more, exactly 2048; 1 .. 9 clusters)."""
    import random
    C, S = sub("pointNet.collate_fns"), sub("pointNet.amp_step")
    batch = []
    for seed, n, w in [(161, 2048, 1), (162, 2048, 3), (163, 1500, 5), (164, 3000, 9), (165, 2048, 9), (166, 700, 2)]:
        win = synth.windows(seed, w, n)
        pc = np.ascontiguousarray(win.transpose(1, 2, 0))
        lab = synth.labels_for(win, seed).transpose(1, 0).copy()
        cent = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], 0).astype(np.float32)
        batch.append((pc, lab, f"f{seed}", cent))
    random.seed(9); torch.manual_seed(9)
    data, tg, _, cents = C.collate_seq_padd(batch)
    random.seed(9); torch.manual_seed(9)
    rb, _, _, cents2 = C.collate_seq_ragged(batch)
    assert torch.equal(cents, cents2)
    np.random.seed(31)
    x1, t1 = S.augment_batch_device(data, tg, train, "cuda")
    np.random.seed(31)
    x2, t2 = S.augment_ragged_device(rb, train, "cuda")
    assert torch.equal(x1, x2), f"{(x1 != x2).sum().item()} of {x1.numel()} values differ"
    assert torch.equal(t1, t2)
    assert (t2 == -1).any() and x2.shape == (6, 9, 2048, 9)
