"""Training / inference drivers of the GRU variant on the HIP path: same function names, arguments, files read and
written as the reference's train_gru (pointNet/rnn/train_pointnetGRU.py:32-332) and test (pointNet/rnn/test_pointnet_gru_segmen.py:28-251).

  train_gru  reads <path_list_files>/train_seg_files.txt, val_seg_files.txt and <dataset_folder>/kmeans_<name>.pt, trains
             BasePointNet(3, True, 256) + SegmentationWithGRU(5, 256, 64) with an unweighted CrossEntropyLoss(ignore_index=-1) and
             2 x Adam(lr), writes pointNet/checkpoints/model_<name>.pth whenever the mean validation loss improves
  test       one file per step; the reference clusters every file in situ with kmeans_clustering(pc, n_points, True, MAX_CLUSTERS = 18)
             -- here that is the on-device constrained k-means (utils.kmeans_clustering) -- and all clusters of a file are one ragged
             launch sequence through the encoder and the GRU head."""
import datetime
import os
import time

import numpy as np
import torch

from ..trainer import FusedAdam
from ..utils.get_metrics import get_accuracy, get_iou_obj
from ..utils.utils import get_labels, kmeans_clustering, save_checkpoint_segmen_model
from .amp_train import IOU_NAMES, reduce_epoch_metrics
from .collate_fns import collate_seq_padd
from .datasets import LidarDataset4Test, LidarKmeansDataset
from .gru_step import GLOBAL_FEAT_SIZE, HIDDEN_SIZE, train_loop
from .model.pointnetAtt import BasePointNet, SegmentationWithGRU

NUM_CLASSES = 5
MAX_CLUSTERS = 18
CLASS_KEYS = ['bckg', 'tower', 'cables', 'low_veg', 'high_veg']


def _epoch(loader, train, pointnet, pred_net, opt_p, opt_g, ce_loss, epoch):
    """As amp_train._epoch: prefetched uploads, predictions stay on the device, confusion counts per step, one download per epoch."""
    from ..utils.get_metrics import confusion_device, metrics_from_confusion
    from .prefetch import DevicePrefetcher
    sums = dict(loss=[], ce=[], reg=[], acc=[])
    ious = {k: [] for k in IOU_NAMES}
    dev = next(pointnet.parameters()).device
    counts, scalars = [], []
    for data in DevicePrefetcher(loader, dev):
        metrics, targets, preds, _ = train_loop(data, opt_p, opt_g, ce_loss, pointnet, pred_net, None, 'segmentation', train, None, epoch, 0,
                                                device_outputs=True)
        counts.append(confusion_device(preds, targets, len(IOU_NAMES)))
        scalars.append(torch.stack([metrics['loss'].reshape(()), metrics['ce_loss'].reshape(()), metrics['reg_loss'].reshape(())]))
    if counts:
        counts_h, scalars_h = torch.stack(counts).cpu().numpy(), torch.stack(scalars).cpu().numpy()
        for c, sc in zip(counts_h, scalars_h):
            acc, per = metrics_from_confusion(c, len(IOU_NAMES))
            sums['acc'].append(acc)
            for name, v in zip(IOU_NAMES, per):
                ious[name].append(v)
            sums['loss'].append(float(sc[0]))
            sums['ce'].append(float(sc[1]))
            sums['reg'].append(float(sc[2]))
    return reduce_epoch_metrics(sums, ious, device=dev)


def train_gru(task, dataset_folder, path_list_files, output_folder, n_points, n_windows, batch_size, epochs, learning_rate,
              weighing_method='EFS', beta=0.999, number_of_workers=4, model_checkpoint=None, c_sample=False, use_kmeans=True, device='cuda'):
    if task != 'segmentation':
        raise NotImplementedError("the reference's GRU train_loop only reaches a model call for task='segmentation'")
    start = time.time()
    device = torch.device(device)
    with open(os.path.join(path_list_files, 'train_seg_files.txt')) as f:
        train_files = f.read().splitlines()
    with open(os.path.join(path_list_files, 'val_seg_files.txt')) as f:
        val_files = f.read().splitlines()
    name = 'GRU' + str(GLOBAL_FEAT_SIZE) + 'h' + str(HIDDEN_SIZE)
    train_ds = LidarKmeansDataset(dataset_folder, task=task, number_of_points=n_points, files=train_files, fixed_num_points=c_sample)
    val_ds = LidarKmeansDataset(dataset_folder, task=task, number_of_points=n_points, files=val_files, fixed_num_points=c_sample)
    # persistent workers, forked before the first pinned batch exists (amp_train.start_workers: a later fork stalls the GPU queues)
    mk = lambda ds: torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=number_of_workers,   # noqa: E731
                                                drop_last=True, collate_fn=collate_seq_padd, pin_memory=True,
                                                persistent_workers=number_of_workers > 0)
    train_loader, val_loader = mk(train_ds), mk(val_ds)
    from .amp_train import start_workers
    start_workers(val_loader, train_loader)
    print(f'Dataset folder: {dataset_folder}\nSamples for training: {len(train_ds)}\nSamples for validation: {len(val_ds)}')
    pointnet = BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=GLOBAL_FEAT_SIZE, device=device)
    pred_net = SegmentationWithGRU(num_classes=NUM_CLASSES, global_feat_size=GLOBAL_FEAT_SIZE, hidden_size=HIDDEN_SIZE, device=device)
    ce_loss = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=-1)
    opt_p = FusedAdam(pointnet.parameters(), lr=learning_rate)
    opt_g = FusedAdam(pred_net.parameters(), lr=learning_rate)
    print(f"Total Trainable Params: {sum(p.numel() for p in pointnet.parameters()) + sum(p.numel() for p in pred_net.parameters())}")
    best_vloss, since = 1_000_000., 0
    history = []
    for epoch in range(epochs):
        t0 = time.time()
        tr = _epoch(train_loader, True, pointnet, pred_net, opt_p, opt_g, ce_loss, epoch)
        with torch.no_grad():
            va = _epoch(val_loader, False, pointnet, pred_net, opt_p, opt_g, ce_loss, epoch)
        history.append((tr, va))
        print(f"epoch {epoch}: train loss {tr['loss']:.4f} acc {tr['acc']:.3f} | val loss {va['loss']:.4f} acc {va['acc']:.3f} "
              f"iou tower {va['iou_tower']:.3f} | {time.time() - t0:.1f} s", flush=True)
        if va['loss'] < best_vloss:
            best_vloss, since = va['loss'], 0
            stamp = datetime.datetime.now().strftime("%m-%d-%H:%M")
            save_checkpoint_segmen_model(stamp + name, task, epoch, since, pointnet, pred_net, opt_p, opt_g, va['acc'], batch_size,
                                         learning_rate, n_points, weighing_method)
        else:
            since += 1
        if since > 100:                                    # train_pointnetGRU.py:327-328
            break
    print("--- TOTAL TIME: %s h ---" % (round((time.time() - start) / 3600, 3)))
    return history


def segment_file(base_pointnet, segmen_net, clusters_list, device):
    """clusters_list: list of [n_i, >=10] tensors (cols 0..8 features, col 9 class code) -> (preds [sum n_i] cpu, targets [sum n_i] cpu)."""
    targets = torch.cat(get_labels([c.clone() for c in clusters_list]), dim=0)
    sizes = [int(c.shape[0]) for c in clusters_list]
    rows = torch.cat([torch.as_tensor(c)[:, :9].float() for c in clusters_list], dim=0).to(device)
    with torch.no_grad():
        local, glob, _ = base_pointnet.forward_windows(rows, np_cluster=sizes)
        logits, preds, _ = segmen_net.forward_rows(glob, local, sizes, 1, want_preds=True)
    return preds.reshape(-1).cpu(), targets.reshape(-1)


def test(dataset_folder, output_folder, n_points, number_of_workers, model_checkpoint, path_list_files, device='cuda', allow_pickle=None):
    start = time.time()
    device = torch.device(device)
    checkpoint = torch.load(model_checkpoint, map_location=device, weights_only=True)
    with open(os.path.join(path_list_files, 'test_seg_files.txt')) as f:
        test_files = f.read().splitlines()
    ds = LidarDataset4Test(dataset_folder, task='segmentation', number_of_points=n_points, files=test_files, fixed_num_points=False,
                           allow_pickle=allow_pickle)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=number_of_workers, drop_last=False)
    base_pointnet = BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=GLOBAL_FEAT_SIZE, device=device)
    segmen_net = SegmentationWithGRU(num_classes=5, global_feat_size=GLOBAL_FEAT_SIZE, hidden_size=HIDDEN_SIZE, device=device)
    base_pointnet.load_state_dict(checkpoint['base_pointnet'])
    segmen_net.load_state_dict(checkpoint['segmen_net'])
    base_pointnet.eval()
    segmen_net.eval()
    total_params = sum(p.numel() for p in base_pointnet.parameters()) + sum(p.numel() for p in segmen_net.parameters())
    print(f"Total Trainable Params: {total_params}")
    iou = {k: [] for k in CLASS_KEYS}
    accuracy = []
    for pc, file_name in loader:
        clusters_list, _ = kmeans_clustering(pc, n_points=n_points, get_centroids=True, max_clusters=MAX_CLUSTERS)
        preds, targets = segment_file(base_pointnet, segmen_net, clusters_list, device)
        accuracy.append(get_accuracy(preds.numpy(), targets.numpy(), {}, 'segmentation')['accuracy'])
        present = set(targets.numpy().reshape(-1).tolist())
        for c, k in enumerate(CLASS_KEYS):
            if c in present:
                iou[k].append(get_iou_obj(preds, targets, c))
    iou_arr = [np.mean(iou['tower']), np.mean(iou['low_veg']), np.mean(iou['high_veg']), np.mean(iou['bckg']), np.mean(iou['cables'])]
    mean_iou = float(np.mean(iou_arr))
    print('mean_iou: ', mean_iou, ' accuracy: ', float(np.mean(accuracy)))
    minutes = round((time.time() - start) / 60, 3)
    print("--- TOTAL TIME: %s min ---" % minutes)
    model_name = model_checkpoint.split('/')[-1].split('.')[0]
    os.makedirs(output_folder, exist_ok=True)
    with open(os.path.join(os.path.dirname(output_folder.rstrip('/')) or '.', 'IoU-results-v2.csv'), 'a') as fid:
        fid.write('%s,%s,%s,%s,%s,%s,%s,%s,%s,%s,%s\n' % (
            model_name, n_points, round(float(np.mean(iou['tower'])), 3), round(float(np.mean(iou['low_veg'])), 3),
            round(float(np.mean(iou['high_veg'])), 3), round(float(np.mean(iou['cables'])), 3), round(float(np.mean(iou['bckg'])), 3),
            round(mean_iou, 4), round(float(np.mean(accuracy)), 3), total_params, minutes))
    return dict(mean_iou=mean_iou, accuracy=float(np.mean(accuracy)), iou={k: float(np.mean(v)) if v else float('nan') for k, v in iou.items()})
