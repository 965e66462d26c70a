"""Where does the 6e-4 error of the GRU head's conv_3.weight gradient sit?  (noise spread over the matrix, or a block of it?)"""
import os, sys, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gru_gpu as TG
from conftest import sub
synth, params = sub("synthetic"), sub("params")
g = np.load(os.path.join(ROOT, "tests/golden/gru.npz"))
S, T = sub("pointNet.gru_step"), sub("trainer")
for p_drop in (0.0,):
    enc, gru = TG._models(synth, params, 7, 8, p_drop=p_drop)
    B, N, W = [int(v) for v in g["meta"]]
    pc, tg, cent, _ = synth.sample_batch(43, B, N, max_w=W, w_real=[int(v) for v in g["w_real"]])
    x, t = S.relayout_batch(torch.from_numpy(pc), torch.from_numpy(tg), "cuda")
    enc.train(); gru.train()
    out = T.forward_backward(enc, gru, x, t, None, None)
    torch.cuda.synchronize()
    _, _, w64, _ = TG._oracle_grads(synth, params, pc, tg, torch.float64)
    _, _, w32, _ = TG._oracle_grads(synth, params, pc, tg, torch.float32)
    for k in ("conv_3.weight", "conv_3.bias", "conv_2.weight", "conv_4.weight", "bn_2.weight", "bn_3.weight", "gru_global.weight_hh_l0", "gru_global.weight_ih_l0"):
        got = dict(gru.named_parameters())[k].grad.double().cpu()
        w = w64[("gru", k)].reshape(got.shape)
        e = got - w
        print(f"{k:28s} rel err {float(e.norm() / w.norm()):.2e}  torch-f32 {float((w32[('gru', k)].reshape(got.shape) - w).norm() / w.norm()):.2e}  |g| {float(w.norm()):.3e}  max|e| {float(e.abs().max()):.2e}")
    got = dict(gru.named_parameters())["conv_3.weight"].grad.double().cpu().reshape(64, 128)
    w = w64[("gru", "conv_3.weight")].reshape(64, 128)
    e = got - w
    print("row err norms (64 output channels):", np.array2string((e.norm(dim=1) / w.norm(dim=1)).numpy(), precision=1, max_line_width=200))
    print("col err norms (128 inputs):", np.array2string((e.norm(dim=0) / w.norm(dim=0)).numpy(), precision=1, max_line_width=200))
    ref = torch.from_numpy(g["s1_gru_grad/conv_3.weight"].astype(np.float64)).reshape(64, 128)
    print("reference fp32 vs f64:", float((ref - w).norm() / w.norm()), " HIP vs reference fp32:", float((got - ref).norm() / w.norm()))
