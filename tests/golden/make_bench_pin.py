#!/usr/bin/env python3
"""Pins what bench.py's FIRST step must compute (tests/golden/bench_pin.json).

Runs the oracle (oracle/ampnet_oracle.py, pinned to the reference by tests/test_oracle_golden.py) in float32 on the host on exactly
the inputs bench.py builds for rank 0 -- synthetic.make_params(3 / 4), synthetic.sample_batch(100, B, 2048, max_w=9) -- and records

  train_B64 : train-mode forward of the fresh modules (BASELINE.json configs[2], B = 64, dropout 0.3 with the keep-masks of the first
              step of a fresh SegmentationWithAttention: seed = att.seed, step 0): ce, reg, sum of class weights
  fwd_B32   : eval forward (configs[1], B = 32): ce

bench.py compares the loss terms of its first step with these (1e-4 relative) and exits non-zero on a mismatch or on a non-finite
loss in the timed region; tests/test_fullsize_gpu.py::test_bench_shape_train_forward_matches_oracle re-derives the B = 64 numbers on
the GPU box and checks the file against them.  No reference code is involved: inputs, weights and the oracle are this repository's.

    python tests/golden/make_bench_pin.py            (about 3 minutes on 8 cores, 20 GB)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sub                             # noqa: E402
from oracle import ampnet_oracle as O                # noqa: E402

N_POINTS, N_WIN, DROP_P = 2048, 9, 0.3
ATT_SEED = 0x5EED                                    # SegmentationWithAttention.seed of this package (pointNet/model/pointnetAtt.py); the test checks it


def _state(synth, P):
    t = lambda d: {k: torch.from_numpy(np.array(v)) for k, v in d.items()}      # noqa: E731
    return (t(synth.make_params(3, P.ENC_PARAMS)), t(synth.make_buffers(3, P.ENC_BUFFERS)),
            t(synth.make_params(4, P.HEAD_PARAMS)), t(synth.make_buffers(4, P.HEAD_BUFFERS)))


def train_forward(synth, P, B, seed, batch_seed=100):
    """Train-mode oracle forward at the bench shape -> dict(ce, reg, logits, buffers)."""
    pc, tg, cent, _ = synth.sample_batch(batch_seed, B, N_POINTS, max_w=N_WIN)
    ep, eb, hp, hb = _state(synth, P)
    Pp = N_WIN * N_POINTS
    keep = {s: O.keep_mask(seed, s, n, DROP_P) for s, n in ((0, B * 8 * N_WIN * N_WIN), (1, B * Pp * 128), (2, B * Pp * 64))}
    masks = {"att": torch.from_numpy(keep[0]).float().reshape(B * 8, N_WIN, N_WIN),
             "d2": torch.from_numpy(keep[1]).float().reshape(B, Pp, 128).transpose(1, 2),
             "d3": torch.from_numpy(keep[2]).float().reshape(B, Pp, 64).transpose(1, 2)}
    del keep
    with torch.no_grad():
        logits, tpc, t_feat, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent),
                                                   True, True, drop_p=DROP_P, drop_masks=masks)
        ce, reg = O.loss_terms(logits, tpc, t_feat)
    return dict(ce=float(ce), reg=float(reg), logits=logits, eb=eb, hb=hb)


def eval_forward(synth, P, B, batch_seed=100):
    pc, tg, cent, _ = synth.sample_batch(batch_seed, B, N_POINTS, max_w=N_WIN)
    ep, eb, hp, hb = _state(synth, P)
    with torch.no_grad():
        logits, tpc, t_feat, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent), False, False)
        ce, _ = O.loss_terms(logits, tpc, t_feat)
    return dict(ce=float(ce))


def main():
    synth, P = sub("synthetic"), sub("params")
    t0 = time.time()
    tr = train_forward(synth, P, 64, ATT_SEED)
    print(f"train B=64: ce {tr['ce']:.8f} reg {tr['reg']:.6f} ({time.time() - t0:.0f} s)", flush=True)
    t1 = time.time()
    ev = eval_forward(synth, P, 32)
    print(f"eval B=32: ce {ev['ce']:.8f} ({time.time() - t1:.0f} s)", flush=True)
    out = {"note": "oracle float32 on bench.py's rank-0 inputs; made by tests/golden/make_bench_pin.py",
           "train_B64": {"ce": tr["ce"], "reg": tr["reg"], "dropout_seed": ATT_SEED, "drop_p": DROP_P, "batch_seed": 100},
           "fwd_B32": {"ce": ev["ce"], "batch_seed": 100}, "rel_tol": 1e-4}
    with open(os.path.join(ROOT, "tests", "golden", "bench_pin.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote tests/golden/bench_pin.json")


if __name__ == "__main__":
    main()
