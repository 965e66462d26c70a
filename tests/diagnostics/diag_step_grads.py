"""Where does each gradient tensor of the step golden sit: HIP vs the reference's fp32 gradient (golden), vs the float64 oracle,
and the reference / torch-fp32 oracle vs float64 (the noise yardsticks).  python tests/diagnostics/diag_step_grads.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sub
from oracle import ampnet_oracle as O
from helpers import torch_params, replay_augment
synth, params = sub("synthetic"), sub("params")
S, T, M = sub("pointNet.amp_step"), sub("trainer"), sub("pointNet.model.pointnetAtt")
g = np.load(os.path.join(ROOT, "tests", "golden", "step.npz"))
B, N, W = [int(v) for v in g["meta"]]
pc, tg, cent, _ = synth.sample_batch(41, B, N, max_w=W, w_real=[int(v) for v in g["w_real"]])
enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, dropout=0.0, device="cuda")
enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**synth.make_params(3, params.ENC_PARAMS), **synth.make_buffers(3, params.ENC_BUFFERS)}.items()}, strict=False)
att.load_state_dict({k: torch.from_numpy(v) for k, v in {**synth.make_params(4, params.HEAD_PARAMS), **synth.make_buffers(4, params.HEAD_BUFFERS)}.items()}, strict=False)
ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
class NoOpt:
    def zero_grad(self): pass
    def step(self): pass
np.random.seed(1001)
data = (torch.from_numpy(pc.copy()), torch.from_numpy(tg.copy()), ["f"] * B, torch.from_numpy(cent))
S.train_loop(data, NoOpt(), NoOpt(), ce, enc, att, None, "segmentation", True, 0, 0)
got = {"enc/" + k: p.grad.double().cpu() for k, p in enc.named_parameters()}
got.update({"att/" + k: p.grad.double().cpu() for k, p in att.named_parameters()})
apc, atg = replay_augment(1001, pc, tg, True)
want = {}
for dt in (torch.float64, torch.float32):
    d = lambda dd, gr: {k: v.to(dt).requires_grad_(gr) for k, v in torch_params(dd).items()}
    ep, eb = d(synth.make_params(3, params.ENC_PARAMS), True), d(synth.make_buffers(3, params.ENC_BUFFERS), False)
    hp, hb = d(synth.make_params(4, params.HEAD_PARAMS), True), d(synth.make_buffers(4, params.HEAD_BUFFERS), False)
    lg, tpc, tf, _ = O.forward_windows(ep, eb, hp, hb, torch.from_numpy(apc).to(dt), torch.from_numpy(atg), torch.from_numpy(cent).to(dt), True, True)
    c, r = O.loss_terms(lg, tpc, tf)
    (c + 0.001 * r).backward()
    want[dt] = {"enc/" + k: v.grad.double() for k, v in ep.items()}
    want[dt].update({"att/" + k: v.grad.double() for k, v in hp.items()})
print(f"{'tensor':44s} {'|g|':>10s} {'hip-ref':>9s} {'hip-f64':>9s} {'ref-f64':>9s} {'o32-f64':>9s}")
for k, w in want[torch.float64].items():
    tag, name = k.split("/", 1)
    nrm = float(w.norm()) + 1e-30
    key = f"s1_{tag}_grad/{name}"
    ref = torch.from_numpy(g[key].astype(np.float64)) if key in g.files else None
    e = lambda a, b: float((a.reshape(b.shape) - b).norm()) / nrm
    print(f"{k:44s} {nrm:10.3e} {e(got[k], ref) if ref is not None else float('nan'):9.2e} {e(got[k], w):9.2e} "
          f"{e(ref, w) if ref is not None else float('nan'):9.2e} {e(want[torch.float32][k], w):9.2e}")
