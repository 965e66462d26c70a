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
from .datasets import LidarDataset4Test
from .model.pointnetAtt import BasePointNet, SegmentationWithAttention

CLASS_KEYS = ['bckg', 'tower', 'cables', 'low_veg', 'high_veg']


def _load_list(path, allow_pickle=None):
    """The cluster pickles are lists of tensors: torch.save files load with torch's restricted unpickler; files written by the
    reference's kmeans_clustering (plain pickle.dump, utils/utils.py:526-533) need full unpickling, which executes what the
    file says and is therefore an explicit opt-in (allow_pickle=True or AMPNET_ALLOW_PICKLE=1) -- _safe_load.py."""
    return load_tensor_list(path, allow_pickle)


class _Staging:
    """Host side of a launch: ONE page-locked buffer the clusters of a file (or group of files) are concatenated into, one upload, and
    one page-locked buffer the predictions + labels come back through as bytes.  Two of each, used alternately, so that a call that
    leaves its results on the device (device_outputs=True) does not have to finish before the next call fills the other buffer; a buffer
    is reused only after the upload that read it has completed (event)."""

    def __init__(self):
        self.inb, self.outb, self.ev, self.k = [None, None], [None, None], [None, None], 0

    def take(self, rows, cols):
        self.k ^= 1
        k = self.k
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        if self.inb[k] is None or self.inb[k].shape[0] < rows or self.inb[k].shape[1] != cols:
            self.inb[k] = torch.empty((max(rows, 1 << 16), cols), dtype=torch.float32).pin_memory()
        return k, self.inb[k][:rows]

    def uploaded(self, k, device):
        self.ev[k] = torch.cuda.Event()
        self.ev[k].record(torch.cuda.current_stream(device))

    def out(self, k, n):
        if self.outb[k] is None or self.outb[k].numel() < 2 * n:
            self.outb[k] = torch.empty(2 * max(n, 1 << 16), dtype=torch.uint8).pin_memory()
        return self.outb[k][:2 * n].view(2, n)            # contiguous: a strided page-locked view would take the slow element-wise copy path


_STAGE = {}
_LUT = {}


def _label_lut(device):
    """ASPRS class code -> segmentation label as a device table (utils/utils.py:546-579: 15 -> 1, 14 -> 2, 3 / 4 -> 3, 5 -> 4, else 0)."""
    key = str(device)
    if key not in _LUT:
        t = torch.zeros(256, dtype=torch.int64)
        t[15], t[14], t[3], t[4], t[5] = 1, 2, 3, 3, 4
        _LUT[key] = t.to(device)
    return _LUT[key]


def _upload_clusters(cluster_tensors, device):
    """list of [n_i, >= 10] float tensors -> (rows [sum n_i, 9] f32 device, targets [sum n_i] int64 device, slot): one concatenation into
    page-locked memory, one asynchronous upload; the labels come from the class-code column ON THE DEVICE (one table look-up) instead of
    a clone + five masked assignments per cluster on the host."""
    st = _STAGE.setdefault(str(device), _Staging())
    total = sum(int(c.shape[0]) for c in cluster_tensors)
    k, buf = st.take(total, 10)
    torch.cat([torch.as_tensor(c)[:, :10].float() for c in cluster_tensors], dim=0, out=buf)
    raw = buf.to(device, non_blocking=True)
    st.uploaded(k, device)
    rows = raw[:, :9].contiguous()
    targets = _label_lut(device)[raw[:, 9].long().clamp_(0, 255)]
    return rows, targets, (st, k)


def _download(preds, targets, slot, device):
    """(preds, targets) int64 device vectors -> int64 CPU vectors through one byte-sized page-locked transfer (labels are 0 .. 4)."""
    st, k = slot
    n = preds.numel()
    out = st.out(k, n)
    out.copy_(torch.stack([preds.reshape(-1), targets.reshape(-1)]).to(torch.uint8), non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    host = torch.from_numpy(out.numpy().astype(np.int64))          # a fresh pageable array (Tensor.to on page-locked memory page-locks its result: ms per call)
    return host[0], host[1]


def segment_file(base_pointnet, segmen_net, clusters_list, centroids, device, device_outputs=False):
    """clusters_list: list of [n_i, >=10] tensors (cols 0..8 features, col 9 class code); centroids [W, 2]
    -> (preds [sum n_i] int64, targets [sum n_i] int64), on the CPU like the reference's loop hands them to the metrics
    (test_pointnet_att_segmen.py:127-181), or left on the device without any synchronisation (device_outputs=True: test() counts the
    confusion matrix there and downloads once per run)."""
    device = torch.device(device)
    sizes = [int(c.shape[0]) for c in clusters_list]
    rows, targets, slot = _upload_clusters(clusters_list, device)
    cent = torch.as_tensor(centroids).float().reshape(1, len(sizes), 2).to(device, non_blocking=True)
    with torch.no_grad():
        local, glob, _ = base_pointnet.forward_windows(rows, np_cluster=sizes)
        logits, preds, _ = segmen_net.forward_rows(glob, local, cent, sizes, None, want_preds=True)
    if device_outputs:
        return preds.reshape(-1), targets
    return _download(preds, targets, slot, device)


def segment_files(base_pointnet, segmen_net, files, device, device_outputs=False):
    """Several files per launch sequence: files = list of (clusters_list, centroids) as segment_file takes them.  All clusters of all
    files go through the encoder as ONE ragged launch sequence and through the head as one (ampnet_head_fwd_files_f32: a file's clusters
    occupy the first slots of its row of W_max window slots, the rest are zero-row windows masked out of its attention).  Returns a list of
    (preds, targets) per file, identical to segment_file's file by file: eval-mode BatchNorm uses running statistics and the attention
    is per file, so nothing a file computes depends on its neighbours (tests/test_inference_gpu.py).  One upload and one download per group."""
    from .. import ops
    if not files:
        return []
    device = torch.device(device)
    n_files = len(files)
    sizes = [[int(c.shape[0]) for c in cl] for cl, _ in files]
    W = max(len(s) for s in sizes)
    flat = [n for s in sizes for n in s]
    rows, targets, slot = _upload_clusters([c for cl, _ in files for c in cl], device)
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
        _, preds = ops.head_forward_files(pt, bt, gl, local, cent.to(device, non_blocking=True), off, mask.to(device, non_blocking=True), n_files, W,
                                          total, mx, segmen_net.num_classes, segmen_net._ws)
    if device_outputs:
        p_all, t_all = preds.reshape(-1), targets
    else:
        p_all, t_all = _download(preds, targets, slot, device)
    out, r0 = [], 0
    for s in sizes:
        n = sum(s)
        out.append((p_all[r0:r0 + n], t_all[r0:r0 + n]))
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
                yield segment_file(base_pointnet, segmen_net, clusters, torch.as_tensor(centroids), device, device_outputs=True)
                continue
            group.append((clusters, torch.as_tensor(centroids)))
            if len(group) == files_per_launch:
                yield from segment_files(base_pointnet, segmen_net, group, device, device_outputs=True)
                group = []
        if group:
            yield from segment_files(base_pointnet, segmen_net, group, device, device_outputs=True)

    # Predictions and labels stay on the device: one confusion-count kernel per file is queued behind its forward, the counts of the
    # whole run come down ONCE.  Accuracy and IoU are the reference's float32 quotients of those integers (utils/get_metrics.py:
    # metrics_from_confusion, bit-identical to get_accuracy / get_iou_obj: tests/test_metrics_gpu.py); a class enters a file's IoU only
    # if the file's TARGETS hold it (test_pointnet_att_segmen.py:192-219).
    from ..utils.get_metrics import confusion_device, metrics_from_confusion
    counts = [confusion_device(preds, targets, 5) for preds, targets in results()]
    counts_h = torch.stack(counts).cpu().numpy() if counts else np.zeros((0, 26), dtype=np.int64)
    for c in counts_h:
        acc, per_all = metrics_from_confusion(c, 5)
        accuracy.append(acc)
        in_targets = c[:25].reshape(5, 5).sum(axis=1) > 0
        per = [per_all[k] if in_targets[k] else None for k in range(5)]
        for k, key in enumerate(CLASS_KEYS):
            if per[k] is not None:
                iou[key].append(per[k])
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
