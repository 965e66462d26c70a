"""Run N eval forwards only (for rocprofv3 --kernel-trace --stats): python3 tools/prof_fwd.py [steps] [B]"""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
PKG = "3d-semantic-segmentation-amp-net_amd"
synth = importlib.import_module(PKG + ".synthetic")
M = importlib.import_module(PKG + ".pointNet.model.pointnetAtt")
S = importlib.import_module(PKG + ".pointNet.amp_step")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, N = 9, 2048
enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cuda")
enc.eval(); att.eval()
pc, tg, cent, _ = synth.sample_batch(5, B, N, max_w=W)
x = torch.from_numpy(pc.transpose(0, 3, 1, 2).copy()).cuda()
t = torch.from_numpy(tg.transpose(0, 2, 1).copy()).cuda()
c = torch.from_numpy(cent).cuda()
with torch.no_grad():
    for _ in range(steps):
        S.forward_batch(enc, att, x, t, c, None, want_loss=False, want_preds=True)
torch.cuda.synchronize()
print("done", steps)
