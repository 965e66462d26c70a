"""Edge-shape sweep through the public modules (eval forward vs the oracle; train step finite)."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import sub
from helpers import torch_params
from oracle import ampnet_oracle as O
synth, P = sub("synthetic"), sub("params")
M = sub("pointNet.model.pointnetAtt"); S = sub("pointNet.amp_step"); T = sub("trainer")
bad = 0
for (B, W, N, C) in [(1, 1, 50, 5), (2, 32, 64, 5), (3, 2, 33, 2), (2, 5, 1000, 8), (1, 9, 2048, 5), (5, 3, 257, 5), (2, 2, 4, 5)]:
    hp_table = dict(P.HEAD_PARAMS); hp_table["conv_4.weight"] = (C, 64, 1); hp_table["conv_4.bias"] = (C,)
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=C, local_dim=64, device="cuda")
    ep, eb = synth.make_params(3, P.ENC_PARAMS), synth.make_buffers(3, P.ENC_BUFFERS)
    hp, hb = synth.make_params(4, hp_table), synth.make_buffers(4, P.HEAD_BUFFERS)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**ep, **eb}.items()}, strict=False)
    att.load_state_dict({k: torch.from_numpy(v) for k, v in {**hp, **hb}.items()}, strict=False)
    pc, tg, cent, _ = synth.sample_batch(900 + B, B, N, max_w=W)
    tg = tg % C
    x = np.ascontiguousarray(pc.transpose(0, 3, 1, 2)); t = np.ascontiguousarray(tg.transpose(0, 2, 1))
    enc.eval(); att.eval()
    with torch.no_grad():
        out = S.forward_batch(enc, att, x, t, cent, None, want_loss=False, want_preds=True)
        logits, tpc, _, _ = O.forward_windows(torch_params(ep), torch_params(eb), torch_params(hp), torch_params(hb), torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent), False, False)
    err = (out["logits"].cpu() - logits).abs().max().item()
    msg = f"B={B} W={W} N={N} C={C}: eval logits max err {err:.2e}"
    if not err <= 1e-3:
        bad += 1; msg += "  <-- FAIL"
    if B >= 2:
        enc.train(); att.train()
        cw = torch.ones(C, device="cuda")
        o2 = T.forward_backward(enc, att, x, t, cent, cw)
        fin = all(torch.isfinite(p.grad).all().item() for m in (enc, att) for p in m.parameters())
        msg += f" | train ce {float(o2['ce'][0]):.4f} grads finite {fin}"
        if not fin: bad += 1
    print(msg)
print("FAILURES", bad)
