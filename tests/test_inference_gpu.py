"""The inference path the reference actually runs (pointNet/self-attention/test_pointnet_att_segmen.py:127-181: batch 1, one file of up
to 18-25 ragged clusters per step) and its several-files-per-launch form (amp_test.segment_files -> ampnet_head_fwd_files_f32):
per-file predictions must be IDENTICAL to the batch-1 path -- eval-mode BatchNorm uses running statistics and the attention is per file,
so nothing a file computes depends on what shares its launch; files with different cluster counts share padded window slots (zero-row
windows, masked keys)."""
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


def _models(synth, params):
    M = sub("pointNet.model.pointnetAtt")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cuda")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(3, params.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(3, params.ENC_BUFFERS).items()})
    enc.load_state_dict(sd, strict=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(4, params.HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(4, params.HEAD_BUFFERS).items()})
    att.load_state_dict(sd, strict=False)
    enc.eval(); att.eval()
    return enc, att


def test_files_per_launch_equals_batch_one(synth, params):
    A = sub("pointNet.amp_test")
    enc, att = _models(synth, params)
    files = [synth.test_file_clusters(4000 + 100 * i, w, 2048 if i % 2 == 0 else 300, 400) for i, w in enumerate([18, 3, 25, 1, 9, 18])]
    dev = torch.device("cuda")
    one = [A.segment_file(enc, att, cl, ce, dev) for cl, ce in files]
    for group in (2, 4, 6):
        got = []
        for g0 in range(0, len(files), group):
            got += A.segment_files(enc, att, files[g0:g0 + group], dev)
        assert len(got) == len(one)
        for i, ((p1, t1), (p2, t2)) in enumerate(zip(one, got)):
            assert torch.equal(t1, t2), f"file {i}: targets differ"
            assert torch.equal(p1, p2), f"file {i}, {group} files per launch: {(p1 != p2).sum().item()} of {p1.numel()} predictions differ from the batch-1 path"


def test_batch_one_file_matches_oracle(synth, params):
    """One file of 5 ragged clusters through segment_file against the oracle's eval forward of the same clusters (encoder per cluster,
    attention over the file's tokens): predictions equal up to near-ties, logits within 1e-3."""
    A = sub("pointNet.amp_test")
    enc, att = _models(synth, params)
    clusters, cent = synth.test_file_clusters(4500, 5, 700, 300)
    preds, targets = A.segment_file(enc, att, clusters, cent, torch.device("cuda"))
    d = lambda dd: {k: v for k, v in torch_params(dd).items()}      # noqa: E731
    ep, eb = d(synth.make_params(3, params.ENC_PARAMS)), d(synth.make_buffers(3, params.ENC_BUFFERS))
    hp, hb = d(synth.make_params(4, params.HEAD_PARAMS)), d(synth.make_buffers(4, params.HEAD_BUFFERS))
    with torch.no_grad():
        los, gls = [], []
        for c in clusters:
            lo, gl, _ = O.encoder(ep, eb, c[None, :, :9].float(), False)
            los.append(lo[0]); gls.append(gl)
        gl = torch.stack(gls, 0)                                    # [W, 1, 256]
        logits = O.head(hp, hb, gl, torch.cat(los, 0)[None], cent[None].float(), [int(c.shape[0]) for c in clusters], None, False)
    want = O.predictions(logits).reshape(-1)
    assert (preds != want).float().mean().item() < 1e-3
    assert targets.numel() == preds.numel() == sum(int(c.shape[0]) for c in clusters)
