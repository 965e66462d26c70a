import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import sub
synth = sub("synthetic"); S = sub("pointNet.amp_step"); M = sub("pointNet.model.pointnetAtt"); T = sub("trainer")
B, W, N = 64, 9, 2048
enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cuda")
op, oa = T.FusedAdam(enc.parameters(), lr=1e-3), T.FusedAdam(att.parameters(), lr=1e-3)
ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
pc, tg, cent, _ = synth.sample_batch(5, B, N, max_w=W)
for pin in (False, True):
    data = (torch.from_numpy(pc), torch.from_numpy(tg), ["f"] * B, torch.from_numpy(cent))
    if pin:
        data = (data[0].pin_memory(), data[1].pin_memory(), data[2], data[3])
    for host in ("1", "0"):
        os.environ["AMPNET_HOST_AUG"] = host
        for _ in range(2):
            S.train_loop(data, op, oa, ce, enc, att, None, "segmentation", True, 0, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            m, tpc, preds, _ = S.train_loop(data, op, oa, ce, enc, att, None, "segmentation", True, 0, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"pinned={pin} host_aug={host}: {dt*1e3:.1f} ms per train_loop call ({B*W*N/dt/1e6:.1f} M points/s end to end)")
