"""Drop-in for the reference's data_proc/sample_fps.py (:12-34): the farthest-point-sampling cascade over a directory of tiles.

Per file the reference does, one file at a time on the host:
    pc = pickle.load(f).astype(float32)                      # [n, D], column 3 = ASPRS class
    pc = pc[pc[:, 3] != 30];  pc = pc[pc[:, 3] != 7]         # drop noise
    if n > 8192: pc = fps(pc, 8192)  -> <out>/towers_100x100_fps_8192/<name>.pkl
    if n > 4096: pc = fps(pc, 4096)  -> <out>/towers_100x100_fps_4096/<name>.pkl      (else: pc unchanged goes there)
Here many files go through each stage in ONE launch per size class (utils.fps_indices_ragged -> ampnet_fps_ragged_f32): FPS is a chain
of dependent rounds on one compute unit per cloud, so throughput comes from running hundreds of clouds side by side (256 CUs), not from a
faster single cloud.  Same outputs, file by file and bit for bit (tests/test_fps_cascade_gpu.py checks against the C oracle); stage 2
samples the stage-1 OUTPUT (rows in selection order), exactly as the reference's second `fps(pc, 4096)` does.
"""
import glob
import os
import pickle

import numpy as np
import torch

from .. import _lib
from .._safe_load import load_numpy_pickle
from ..utils.utils import fps_indices_ragged

STAGES = (8192, 4096)                        # data_proc/sample_fps.py:23, :28
NOISE_CLASSES = (30, 7)                      # :20-21


def remove_noise(pc):
    """Rows whose class (column 3) is 30 or 7 dropped, in the reference's order (two successive filters)."""
    for c in NOISE_CLASSES:
        pc = pc[np.where(pc[:, 3] != c)]
    return pc


def cascade(clouds, device="cuda", stages=STAGES):
    """clouds: list of [n_i, D >= 4] arrays (noise already removed).  Returns one dict per stage {cloud index: sampled rows [S, D]} for
    the clouds that stage applied to, and the list of final arrays (every cloud, sampled or untouched)."""
    cur = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
    per_stage = []
    dev = torch.device(device)
    for S in stages:
        todo = [i for i, c in enumerate(cur) if c.shape[0] > S]
        done = {}
        if todo:
            # only x, y, z travel: 12 B per point up, 4 B per sample down; the D-column rows are gathered on the host
            xyz = torch.from_numpy(np.concatenate([cur[i][:, :3] for i in todo])).to(dev)
            idx = fps_indices_ragged(xyz, [cur[i].shape[0] for i in todo], S)
            flat = torch.cat(idx).cpu().numpy().astype(np.int64)
            for j, i in enumerate(todo):
                cur[i] = cur[i][flat[j * S:(j + 1) * S]]
                done[i] = cur[i]
        per_stage.append(done)
    return per_stage, cur


def sample_files(files, out_path, files_per_launch=256, device="cuda", allow_pickle=None, dirs=("towers_100x100_fps_8192", "towers_100x100_fps_4096")):
    """The reference's loop over `files` (:12-34) in groups of files_per_launch.  Returns the number of files written per directory."""
    if not torch.cuda.is_available():
        raise _lib.AmpnetError("sample_fps: farthest-point sampling runs on the GPU (no CPU fallback)")
    for d in dirs:
        os.makedirs(os.path.join(out_path, d), exist_ok=True)
    written = [0, 0]
    for g0 in range(0, len(files), files_per_launch):
        group = files[g0:g0 + files_per_launch]
        names = [os.path.basename(p).split('.')[0] for p in group]
        clouds = [remove_noise(np.asarray(load_numpy_pickle(p, allow_pickle)).astype(np.float32)) for p in group]
        per_stage, final = cascade(clouds, device)
        for i, pc in per_stage[0].items():
            with open(os.path.join(out_path, dirs[0], names[i]) + '.pkl', 'wb') as f:
                pickle.dump(pc, f)
            written[0] += 1
        for i, pc in enumerate(final):                           # sampled to 4096 or left as it was (:28-34)
            with open(os.path.join(out_path, dirs[1], names[i]) + '.pkl', 'wb') as f:
                pickle.dump(pc, f)
            written[1] += 1
    return written


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--in_path", default='/dades/LIDAR/towers_detection/datasets/towers_100x100/*pkl')      # the reference's constants (:8-9)
    ap.add_argument("--out_path", default='/dades/LIDAR/towers_detection/datasets')
    ap.add_argument("--files_per_launch", type=int, default=256)
    a = ap.parse_args()
    n = sample_files(sorted(glob.glob(a.in_path)), a.out_path, a.files_per_launch)
    print(f"wrote {n[0]} files to towers_100x100_fps_8192 and {n[1]} to towers_100x100_fps_4096")
