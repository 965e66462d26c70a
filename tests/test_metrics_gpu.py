"""Accuracy / IoU from confusion counts taken on the device (ampnet_confusion_i64) equal the reference-style host functions
(utils/get_metrics.py:6-31 after rm_padding, utils/utils.py:14-19) to the last bit: they are float32 quotients of the same integers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,absent", [(1, None), (4097, None), (9 * 2048 * 8, 4), (9 * 2048 * 64, None)])
def test_confusion_metrics_equal_host_metrics(synth, n, absent):
    G, U = sub("utils.get_metrics"), sub("utils.utils")
    preds = synth.randint(501, (n,), 0, 5)
    tg = synth.randint(502, (n,), 0, 5)
    if absent is not None:                                   # a label that occurs nowhere: IoU = 0 / 0 = nan in both paths
        preds[preds == absent] = 0
        tg[tg == absent] = 0
    tg[synth.uniform01(503, (n,)) < 0.2] = -1
    p, t = torch.from_numpy(preds), torch.from_numpy(tg)
    counts = G.confusion_device(p.cuda(), t.cuda(), 5).cpu().numpy()
    assert counts[-1] == int((tg == -1).sum()) and counts.sum() == n
    acc, ious = G.metrics_from_confusion(counts, 5)
    p2, t2, _ = U.rm_padding(p, t)
    if len(p2):
        want_acc = G.get_accuracy(p2, t2, {}, "segmentation")["accuracy"]
        assert acc == want_acc or (np.isnan(acc) and np.isnan(want_acc))
    for c in range(5):
        want = G.get_iou_obj(p2, t2, c)
        assert ious[c] == want or (np.isnan(ious[c]) and np.isnan(want)), (c, ious[c], want)


def test_train_loop_device_outputs(synth, params):
    """device_outputs=True returns what the default path downloads."""
    import test_step_gpu as TS
    S = sub("pointNet.amp_step")
    enc, att = TS._models(synth, params)
    pc, tg, cent, _ = synth.sample_batch(41, 4, 64, max_w=3)
    data = (torch.from_numpy(pc), torch.from_numpy(tg), ["f"] * 4, torch.from_numpy(cent))
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    np.random.seed(5)
    m1, t1, p1, _ = S.train_loop(data, TS._NoOpt(), TS._NoOpt(), ce, enc, att, None, "segmentation", False, 0, 0)
    np.random.seed(5)
    m2, t2, p2, _ = S.train_loop(data, TS._NoOpt(), TS._NoOpt(), ce, enc, att, None, "segmentation", False, 0, 0, device_outputs=True)
    assert t2.is_cuda and p2.is_cuda
    assert torch.equal(t1, t2.cpu()) and torch.equal(p1, p2.cpu()) and m1["loss"].item() == m2["loss"].item()
