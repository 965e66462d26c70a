import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_backward_gpu as T
from conftest import sub
synth, params = sub("synthetic"), sub("params")
def chk(got, want64, want32, what):
    worst = []
    for k, w in want64.items():
        g = got[k].detach().cpu().double().reshape(w.shape)
        err = float((g - w).norm()); ref = float(w.norm()); noise = float((want32[k].double() - w).norm())
        worst.append((err / max(ref, 1e-30), noise / max(ref, 1e-30), k))
    worst.sort(reverse=True)
    print(what, " | ".join(f"{k}: {e:.2e} (torch fp32 {n:.2e})" for e, n, k in worst[:4]))
T._check_grads = chk
for (B, W, N) in [(16, 2, 288), (16, 2, 300), (16, 2, 320), (16, 2, 256), (16, 2, 260), (16, 2, 512), (16, 2, 544)]:
    print(B, W, N, end=" ")
    T.test_encoder_backward_matches_oracle_autograd.__wrapped__(synth, params, B, W, N) if hasattr(T.test_encoder_backward_matches_oracle_autograd, "__wrapped__") else T.test_encoder_backward_matches_oracle_autograd(synth, params, B, W, N)
