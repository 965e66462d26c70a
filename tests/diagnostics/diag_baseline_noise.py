"""Where the baseline classifier's gradients leave the float64 autograd: per parameter, HIP path vs an fp32 CPU evaluation of the same graph
(python tests/diagnostics/diag_baseline_noise.py [light] [drop_p])."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest                                             # noqa: E402
from test_baseline_cls_gpu import _net, _f64_forward, N_CLS, VARIANTS   # noqa: E402
synth = conftest.sub("synthetic")
from oracle import ampnet_oracle as O                      # noqa: E402
light = len(sys.argv) > 1 and sys.argv[1] == "light"
drop_p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
tag, modname, pdim, base = VARIANTS[1 if light else 0]
net, _ = _net(synth, modname, pdim, base, drop_p, "cuda")
net.train()
Bn, N = 16, 256
x = torch.from_numpy(synth.windows(87, Bn, N)).cuda()
y = torch.from_numpy((synth.uniform(88, (Bn,), 0.0, 1.0) * N_CLS).astype(np.int64).clip(0, N_CLS - 1)).cuda()
sd64 = {k: v.detach().double().cpu().clone().requires_grad_("running" not in k and "num_batches" not in k) for k, v in net.state_dict().items() if "num_batches" not in k}
sd32 = {k: v.detach().float().clone().requires_grad_(v.requires_grad) for k, v in sd64.items()}
seed = net.seed & 0xFFFFFFFF
out, ft = net(x)
loss = torch.nn.functional.nll_loss(out, y) + 0.001 * torch.norm(torch.eye(64, device="cuda") - torch.bmm(ft, ft.transpose(2, 1)))
loss.backward()
c2 = net.fc_3.weight.shape[1]
keep = torch.from_numpy(O.keep_mask(seed, 0, Bn * c2, drop_p)).double().reshape(Bn, c2) if drop_p > 0 else None
for sd, dt in ((sd64, torch.float64), (sd32, torch.float32)):
    o, T = _f64_forward(sd, x.to(dt).cpu(), pdim, None if keep is None else keep.to(dt), drop_p)
    l = torch.nn.functional.nll_loss(o, y.cpu()) + 0.001 * torch.norm(torch.eye(64, dtype=dt) - torch.bmm(T, T.transpose(2, 1)))
    l.backward()
    if dt == torch.float64:
        o64k, T64k = o.detach(), T.detach()
        print("forward: max |log-prob diff| vs f64", (out.detach().double().cpu() - o.detach()).abs().max().item(), " feat_T", (ft.detach().double().cpu() - T.detach()).abs().max().item())
print("cpu fp32 forward: max |log-prob diff| vs f64", (o.detach().double() - o64k).abs().max().item(), " feat_T", (T.detach().double() - T64k).abs().max().item())
print(f"{'parameter':58s} {'|g64|':>10s} {'hip rel':>10s} {'cpu32 rel':>10s}")
for k, p in net.named_parameters():
    w = sd64[k].grad
    n = w.norm().item()
    print(f"{k:58s} {n:10.3e} {(p.grad.double().cpu().reshape(w.shape) - w).norm().item() / n:10.2e} {(sd32[k].grad.double() - w).norm().item() / n:10.2e}")
