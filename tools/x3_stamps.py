"""Phase timeline of wave 0 / workgroup 0 of the split pooled GEMM (tools/build_stamps.sh builds the probe library):
AMPNET_LIB_PATH=tools/lib_stamps.so python3 tools/x3_stamps.py"""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
PKG = "3d-semantic-segmentation-amp-net_amd"
L = importlib.import_module(PKG + "._lib")
synth = importlib.import_module(PKG + ".synthetic")
M = importlib.import_module(PKG + ".pointNet.model.pointnetAtt")
T = importlib.import_module(PKG + ".trainer")
L.set_matrix_precision("f32x3")
B, W, N = 64, 9, 2048
enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cuda")
enc.train(); att.train()
tr = T.Trainer(enc, att)
pc, tg, cent, _ = synth.sample_batch(5, B, N, max_w=W)
x = torch.from_numpy(pc.transpose(0, 3, 1, 2).copy()).cuda()
t = torch.from_numpy(tg.transpose(0, 2, 1).copy()).cuda()
c = torch.from_numpy(cent).cuda()
for _ in range(3):
    tr.step(x, t, c)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
rc = L.lib().ampnet_debug_pw_stamps(buf, 4096)
a = np.frombuffer(buf, dtype=np.uint64)
n = int(a[0])
ids = (a[1:1 + n] >> np.uint64(48)).astype(int)
ts = (a[1:1 + n] & np.uint64(0xffffffffffff)).astype(np.int64)
print("rc", rc, "stamps", n)
names = {1: "block begin", 2: "tile begin", 3: "k loop issued", 4: "epilogue done", 5: "reduction done", 6: "partials written", 7: "waves arrived"}
dt = np.diff(ts)
# s_memtime ticks at 100 MHz on this part?  report raw ticks and the shares
tot = {}
for i in range(n - 1):
    key = f"{names.get(ids[i], ids[i])} -> {names.get(ids[i + 1], ids[i + 1])}"
    tot.setdefault(key, []).append(int(dt[i]))
span = int(ts[-1] - ts[0])
print("span ticks", span)
for k, v in tot.items():
    print(f"{k:40s} n={len(v):4d} mean={np.mean(v):10.1f} sum={np.sum(v):10d} share={np.sum(v) / span:.3f}")
print("first 40 deltas:", [(int(ids[i]), int(dt[i])) for i in range(min(40, n - 1))])

# ---- the split backward (Gram form): wave 0 = W role, wave 4 = D role of workgroup 0
buf2 = (ctypes.c_ulonglong * 4096)()
if hasattr(L.lib(), "ampnet_debug_bx_stamps") and L.lib().ampnet_debug_bx_stamps(buf2, 4096) == 0:
    b = np.frombuffer(buf2, dtype=np.uint64).reshape(2, 2048)
    nm = {1: "block begin", 2: "products issued", 3: "next block staged", 4: "barrier passed"}
    for role, name in ((0, "W wave 0"), (1, "D wave 4")):
        n = int(b[role][0])
        ids = (b[role][1:1 + n] >> np.uint64(48)).astype(int)
        ts = (b[role][1:1 + n] & np.uint64(0xffffffffffff)).astype(np.int64)
        d = np.diff(ts)
        agg = {}
        for i in range(n - 1):
            if 0 < d[i] < 10**7:
                agg.setdefault(f"{nm.get(ids[i], ids[i])} -> {nm.get(ids[i + 1], ids[i + 1])}", []).append(int(d[i]))
        print(name, "stamps", n)
        for k, v in agg.items():
            print(f"   {k:44s} n={len(v):4d} mean={np.mean(v):9.1f} median={np.median(v):9.1f}")
