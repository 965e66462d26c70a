"""AMP-Net inference driver on the HIP path: same function name, arguments and outputs as the reference's test()
(pointNet/self-attention/test_pointnet_att_segmen.py:31-284): one file per step, its pre-computed constrained
k-means clusters read from `<cluster_dir>/<file>_clusters_list.pkl` / `_centroids.pkl` (default `k_means_25/`),
per-file accuracy and per-class IoU, a CSV row appended to <out_path>/../IoU-results-v2.csv.

The reference pushes every cluster (they have different sizes) through the encoder in a Python loop and repeats /
concatenates the attention tokens per point; here all clusters of a file are ONE ragged launch sequence."""
import os
import time

import numpy as np
import torch

from .._safe_load import load_tensor_list
from ..utils.get_metrics import get_accuracy, get_iou_obj
from ..utils.utils import get_labels
from .datasets import LidarDataset4Test
from .model.pointnetAtt import BasePointNet, SegmentationWithAttention

CLASS_KEYS = ['bckg', 'tower', 'cables', 'low_veg', 'high_veg']


def _load_list(path, allow_pickle=None):
    """The cluster pickles are lists of tensors: torch.save files load with torch's restricted unpickler; files written by the
    reference's kmeans_clustering (plain pickle.dump, utils/utils.py:526-533) need full unpickling, which executes what the
    file says and is therefore an explicit opt-in (allow_pickle=True or AMPNET_ALLOW_PICKLE=1) -- _safe_load.py."""
    return load_tensor_list(path, allow_pickle)


def segment_file(base_pointnet, segmen_net, clusters_list, centroids, device):
    """clusters_list: list of [n_i, >=10] tensors (cols 0..8 features, col 9 class code); centroids [W, 2]
    -> (preds [sum n_i] int64 cpu, targets [sum n_i] int64 cpu)."""
    targets = torch.cat(get_labels([c.clone() for c in clusters_list]), dim=0)
    sizes = [int(c.shape[0]) for c in clusters_list]
    rows = torch.cat([torch.as_tensor(c)[:, :9].float() for c in clusters_list], dim=0).to(device)
    cent = torch.as_tensor(centroids).float().reshape(1, len(sizes), 2).to(device)
    with torch.no_grad():
        local, glob, _ = base_pointnet.forward_windows(rows, np_cluster=sizes)
        logits, preds, _ = segmen_net.forward_rows(glob, local, cent, sizes, None, want_preds=True)
    return preds.reshape(-1).cpu(), targets.reshape(-1)


def segment_files(base_pointnet, segmen_net, files, device):
    """Several files per launch sequence: files = list of (clusters_list, centroids) as segment_file takes them.  All clusters of all
    files go through the encoder as ONE ragged launch sequence and through the head as one (ampnet_head_fwd_files_f32: a file's clusters
    occupy the first slots of its row of W_max window slots, the rest are zero-row windows masked out of its attention).  Returns a list of
    (preds, targets) per file, identical to segment_file's file by file: eval-mode BatchNorm uses running statistics and the attention
    is per file, so nothing a file computes depends on its neighbours (tests/test_inference_gpu.py)."""
    from .. import ops
    if not files:
        return []
    n_files = len(files)
    sizes = [[int(c.shape[0]) for c in cl] for cl, _ in files]
    W = max(len(s) for s in sizes)
    targets = [torch.cat(get_labels([c.clone() for c in cl]), dim=0).reshape(-1) for cl, _ in files]
    flat = [n for s in sizes for n in s]
    rows = torch.cat([torch.as_tensor(c)[:, :9].float() for cl, _ in files for c in cl], dim=0).to(device)
    cent = torch.zeros(n_files, W, 2)
    mask = torch.ones(n_files, W, dtype=torch.uint8)
    slot_sizes, real_slots = [], []
    for f, (s, (_, ce)) in enumerate(zip(sizes, files)):
        cent[f, :len(s)] = torch.as_tensor(ce).float().reshape(len(s), 2)
        mask[f, :len(s)] = 0
        slot_sizes += s + [0] * (W - len(s))
        real_slots += [f * W + w for w in range(len(s))]
    with torch.no_grad():
        local, glob, _ = base_pointnet.forward_windows(rows, np_cluster=flat)
        gl = torch.zeros(n_files * W, glob.shape[1], dtype=torch.float32, device=device)
        gl[torch.tensor(real_slots, device=device)] = glob                      # real clusters into their slots (device copy)
        off, total, mx = ops.window_offsets(slot_sizes, rows.device)
        pt, bt = segmen_net._tables()
        _, preds = ops.head_forward_files(pt, bt, gl, local, cent.to(device), off, mask.to(device), n_files, W, total, mx, segmen_net.num_classes,
                                          segmen_net._ws)
    preds = preds.cpu()
    out, r0 = [], 0
    for s, t in zip(sizes, targets):
        n = sum(s)
        out.append((preds[r0:r0 + n], t))
        r0 += n
    return out


def test(dataset_path, out_path, n_points, number_of_workers, model_checkpoint, path_list_files, cluster_dir='k_means_25',
         device='cuda', allow_pickle=None, files_per_launch=1):
    """files_per_launch (not in the reference, which runs batch 1): how many files share one launch sequence (segment_files); the
    per-file metrics, their order and the CSV row are the same for any value."""
    start = time.time()
    device = torch.device(device)
    checkpoint = torch.load(model_checkpoint, map_location=device, weights_only=True)
    with open(os.path.join(path_list_files, 'test_seg_files.txt')) as f:
        test_files = f.read().splitlines()
    ds = LidarDataset4Test(dataset_path, task='segmentation', number_of_points=n_points, files=test_files, fixed_num_points=False,
                           allow_pickle=allow_pickle)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=number_of_workers, drop_last=False)
    base_pointnet = BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device=device)
    segmen_net = SegmentationWithAttention(256, 8, local_dim=64, num_classes=5, device=device)
    base_pointnet.load_state_dict(checkpoint['base_pointnet'])
    segmen_net.load_state_dict(checkpoint['segmen_net'])
    base_pointnet.eval()
    segmen_net.eval()
    total_params = sum(p.numel() for p in base_pointnet.parameters()) + sum(p.numel() for p in segmen_net.parameters())
    print(f"Total Trainable Params: {total_params}")
    iou = {k: [] for k in CLASS_KEYS}
    accuracy = []
    def results():
        group = []
        for pc, file_name in loader:
            name = file_name[0].split('/')[-1].split('.')[0]
            clusters = _load_list(os.path.join(cluster_dir, name + '_clusters_list.pkl'), allow_pickle)
            centroids = _load_list(os.path.join(cluster_dir, name + '_centroids.pkl'), allow_pickle)
            if files_per_launch <= 1:
                yield segment_file(base_pointnet, segmen_net, clusters, torch.as_tensor(centroids), device)
                continue
            group.append((clusters, torch.as_tensor(centroids)))
            if len(group) == files_per_launch:
                yield from segment_files(base_pointnet, segmen_net, group, device)
                group = []
        if group:
            yield from segment_files(base_pointnet, segmen_net, group, device)

    for preds, targets in results():
        accuracy.append(get_accuracy(preds.numpy(), targets.numpy(), {}, 'segmentation')['accuracy'])
        present = set(targets.numpy().reshape(-1).tolist())
        per = [get_iou_obj(preds, targets, c) if c in present else None for c in range(5)]
        for c, k in enumerate(CLASS_KEYS):
            if per[c] is not None:
                iou[k].append(per[c])
        miou = np.nanmean(np.array([per[1], per[2], per[3], per[4], per[0]], dtype=np.float64))
        print([per[1], per[2], miou])
    iou_arr = [np.mean(iou['tower']), np.mean(iou['low_veg']), np.mean(iou['high_veg']), np.mean(iou['bckg']), np.mean(iou['cables'])]
    mean_iou = float(np.mean(iou_arr))
    print('mean_iou: ', mean_iou, ' accuracy: ', float(np.mean(accuracy)))
    minutes = round((time.time() - start) / 60, 3)
    print("--- TOTAL TIME: %s min ---" % minutes)
    model_name = model_checkpoint.split('/')[-1].split('.')[0]
    out_path = os.path.join(out_path, 'preds_Att')
    os.makedirs(out_path + '/figures', exist_ok=True)
    with open(os.path.join(os.path.dirname(out_path), 'IoU-results-v2.csv'), 'a') as fid:
        fid.write('%s,%s,%s,%s,%s,%s,%s,%s,%s,%s,%s\n' % (
            model_name, n_points, round(float(np.mean(iou['tower'])), 3), round(float(np.mean(iou['low_veg'])), 3),
            round(float(np.mean(iou['high_veg'])), 3), round(float(np.mean(iou['cables'])), 3), round(float(np.mean(iou['bckg'])), 3),
            round(mean_iou, 4), round(float(np.mean(accuracy)), 3), total_params, minutes))
    return dict(mean_iou=mean_iou, accuracy=float(np.mean(accuracy)), iou={k: float(np.mean(v)) if v else float('nan') for k, v in iou.items()})
