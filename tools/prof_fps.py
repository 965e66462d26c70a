"""FPS + k-NN only (BASELINE configs[4]: 16 clouds x 8192 -> 4096, k = 32), for rocprofv3 passes: python3 tools/prof_fps.py [reps]"""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
PKG = "3d-semantic-segmentation-amp-net_amd"
synth = importlib.import_module(PKG + ".synthetic")
U = importlib.import_module(PKG + ".utils.utils")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
xyz = torch.from_numpy(synth.clouds(200, 16, 8192)).cuda()
for _ in range(reps):
    idx = U.fps_indices(xyz, 4096)
    grp = U.knn_indices(xyz, idx, 32)
torch.cuda.synchronize()
print("done", reps)
