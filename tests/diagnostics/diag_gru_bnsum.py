"""CPU: conditioning of the BatchNorm-backward sums of the GRU head's bn_3 per channel (float64 oracle)."""
import os, sys, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ampnet_oracle as O
from helpers import torch_params
PK = "3d-semantic-segmentation-amp-net_amd"
synth, params = importlib.import_module(PK + ".synthetic"), importlib.import_module(PK + ".params")
g = np.load(os.path.join(ROOT, "tests/golden/gru.npz"))
B, N, W = [int(v) for v in g["meta"]]
pc, tg, _, _ = synth.sample_batch(43, B, N, max_w=W, w_real=[int(v) for v in g["w_real"]])
dt = torch.float64
d = lambda dd, gr: {k: v.to(dt).requires_grad_(gr) for k, v in torch_params(dd).items()}
ep, eb = d(synth.make_params(7, params.ENC_PARAMS), True), d(synth.make_buffers(7, params.ENC_BUFFERS), False)
hp, hb = d(synth.make_params(8, params.GRU_HEAD_PARAMS), True), d(synth.make_buffers(8, params.HEAD_BUFFERS), False)
kept = {}
orig = O.batchnorm_rows
def hook(z, gamma, beta, bufs, key, train):
    y = orig(z, gamma, beta, bufs, key, train)
    if key in ("bn_3.", "bn_2.") and z.shape[0] > 1000:
        y.retain_grad(); kept[key] = (z, y)
    return y
O.batchnorm_rows = hook
lg, tpc, tf = O.forward_windows_gru(ep, eb, hp, hb, torch.from_numpy(pc).to(dt), torch.from_numpy(tg), True, True)
c, r = O.loss_terms(lg, tpc, tf, class_w=(1.0,) * 5)
(c + 0.001 * r).backward()
for key in ("bn_3.", "bn_2."):
    z, y = kept[key]
    dy = y.grad
    zh = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-5)
    k1 = dy.abs().sum(0) / dy.sum(0).abs()
    k2 = (dy * zh).abs().sum(0) / (dy * zh).sum(0).abs()
    print(key, "cancellation of sum(dy): max", float(k1.max()), "at", int(k1.argmax()), " median", float(k1.median()))
    print(key, "cancellation of sum(dy*zhat): max", float(k2.max()), "at", int(k2.argmax()), " median", float(k2.median()))
    if key == "bn_3.":
        print("   channel 15:", float(k1[15]), float(k2[15]), " active fraction", float((y[:, 15] > 0).double().mean()))
        # fp32 sequential accumulation in blocks of 256 rows vs exact
        d32 = dy[:, 15].float()
        acc = sum(float(np.float32(0) + sum((np.float32(v) for v in d32[i:i + 256].numpy()), np.float32(0))) for i in range(0, d32.shape[0], 256))
        print("   sum(dy)[15] exact", float(dy[:, 15].sum()), " fp32 blocks-of-256", acc)
z, y = kept["bn_3."]
print("gamma/beta[15]", float(hp["bn_3.weight"][15]), float(hp["bn_3.bias"][15]), " var[15]", float(z[:, 15].var(unbiased=False)), " median var", float(z.var(0, unbiased=False).median()))
print("min |y| ch15:", float(y[:, 15].abs().min()), " count |y|<1e-5:", int((y[:, 15].abs() < 1e-5).sum()), " all channels count |y|<1e-6:", int((y.abs() < 1e-6).sum()))
yy = y.detach()
near = (yy.abs() < 2e-6).nonzero()
print("near-zero entries (row, ch, y, dy):", [(int(r), int(c), float(yy[r, c]), float(y.grad[r, c])) for r, c in near[:10]])
dy = y.grad
print("|dy| ch15: max", float(dy[:, 15].abs().max()), " norm", float(dy[:, 15].norm()))
