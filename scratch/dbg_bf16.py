import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import sub
synth, params = sub("synthetic"), sub("params")
from test_bf16_gpu import _models
T = sub("trainer"); L = sub("_lib")
for (B, N, W) in ((8, 256, 3), (64, 512, 3)):
    pc, tg, cent, _ = synth.sample_batch(812, B, N, max_w=W)
    x = np.ascontiguousarray(pc.transpose(0, 3, 1, 2)); t = np.ascontiguousarray(tg.transpose(0, 2, 1))
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    res = {}
    for mode in ("bf16", "fp32"):
        L.set_matrix_precision(mode)
        enc, att = _models(synth, params, 0.0)
        enc.train(); att.train()
        out = T.forward_backward(enc, att, x, t, cent, cw)
        torch.cuda.synchronize()
        res[mode] = (float(out["ce"][0]), float(out["reg"]), {("e." if m is enc else "a.") + k: p.grad.double().clone() for m in (enc, att) for k, p in m.named_parameters()})
    L.set_matrix_precision("fp32")
    gb, gf = res["bf16"][2], res["fp32"][2]
    print("B", B, "ce", res["bf16"][0], res["fp32"][0], "reg", res["bf16"][1], res["fp32"][1])
    tot = np.sqrt(sum((gf[k] ** 2).sum().item() for k in gf))
    for k in gf:
        nf, nb = gf[k].norm().item(), gb[k].norm().item()
        cos = (gf[k] * gb[k]).sum().item() / max(nf * nb, 1e-30)
        if nf / tot > 0.02 or cos < 0.98:
            print(f"  {k:45s} |g| {nf:.3e} ({nf/tot:.3f} of total) bf16 |g| {nb:.3e} cos {cos:.4f}")
