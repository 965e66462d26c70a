"""Run N fused train steps only (for rocprofv3 --kernel-trace --stats): python3 tools/prof_step.py [steps] [precision]"""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
PKG = "3d-semantic-segmentation-amp-net_amd"
synth = importlib.import_module(PKG + ".synthetic")
M = importlib.import_module(PKG + ".pointNet.model.pointnetAtt")
T = importlib.import_module(PKG + ".trainer")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if len(sys.argv) > 2:
    importlib.import_module(PKG + "._lib").set_matrix_precision(sys.argv[2])     # fp32 | bf16 | bf16_train | bf16_store
B, W, N = 64, 9, 2048
enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cuda")
enc.train(); att.train()
tr = T.Trainer(enc, att)
pc, tg, cent, _ = synth.sample_batch(5, B, N, max_w=W)
x = torch.from_numpy(pc.transpose(0, 3, 1, 2).copy()).cuda()
t = torch.from_numpy(tg.transpose(0, 2, 1).copy()).cuda()
c = torch.from_numpy(cent).cuda()
for _ in range(steps):
    tr.step(x, t, c)
torch.cuda.synchronize()
print("done", steps)
