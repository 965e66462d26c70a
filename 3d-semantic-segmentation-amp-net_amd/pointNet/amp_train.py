"""AMP-Net training driver on the HIP path: same function name, arguments, files read and files written as the
reference's train_att (pointNet/self-attention/train_pointnet-attention.py:29-334):

  reads   <path_list_files>/train_seg_files.txt, val_seg_files.txt; <dataset_folder>/kmeans_<name>.pt
  trains  BasePointNet(3, True, 256) + SegmentationWithAttention(256, 8, 5, local_dim=64), CE weights [1,2,2,1,1],
          2 x Adam(lr), MultiStepLR milestones [150, 250, 350] gamma 0.5 stepped per epoch
  writes  pointNet/checkpoints/model_<name>.pth (same dict keys) whenever the mean validation loss improves.

Data parallel: launched under torch.distributed.run (one process per GPU) every rank trains on a rank-strided
shard of the file list and gradients are all-reduced per step (RCCL); epoch metrics (incl. the validation loss that
drives the checkpoint decision) are reduced over all ranks; rank 0 logs and writes checkpoints."""
import datetime
import os
import time

import numpy as np
import torch

from ..trainer import FusedAdam, shard_indices
from ..utils.get_metrics import get_accuracy, get_iou_obj
from ..utils.utils import limit_host_threads, rm_padding, save_checkpoint_segmen_model
from .amp_step import train_loop
from .collate_fns import collate_seq_padd, collate_seq_ragged
from .datasets import LidarKmeansDataset
from .model.pointnetAtt import BasePointNet, SegmentationWithAttention

GLOBAL_FEAT_SIZE = 256
ATT_HEADS = 8
IOU_NAMES = ['bckg', 'tower', 'cables', 'low_veg', 'high_veg']


def _dist_setup():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        return dist.get_rank(), world, local
    return 0, 1, 0


def start_workers(*loaders):
    """Fork the worker processes of persistent-worker DataLoaders right away (DataLoader does it at the first iter()); the batches they
    prefetch meanwhile are dropped when the epoch loop asks for its own iterator.  No-op for loaders without workers."""
    for ld in loaders:
        if getattr(ld, "persistent_workers", False) and ld.num_workers > 0 and len(ld) > 0:
            iter(ld)


def _epoch(loader, train, pointnet, att_net, opt_p, opt_a, ce_loss, epoch):
    """One pass over the loader.  Default: predictions and targets never leave the GPU -- per batch one confusion-count kernel and four
    scalars stay queued on the device and are downloaded ONCE at the end of the epoch, so no step synchronises with the host (the
    reference downloads both tensors every step and takes accuracy / IoU with numpy, train_pointnet-attention.py:217-240).
    AMPNET_HOST_METRICS=1 or AMPNET_HOST_AUG=1: the reference's per-step host path."""
    sums = dict(loss=[], ce=[], reg=[], acc=[])
    ious = {k: [] for k in IOU_NAMES}
    dev = next(pointnet.parameters()).device
    on_device = os.environ.get("AMPNET_HOST_AUG") != "1" and os.environ.get("AMPNET_HOST_METRICS") != "1"
    if on_device:
        from ..utils.get_metrics import confusion_device, metrics_from_confusion
        from .prefetch import DevicePrefetcher
        counts, scalars = [], []
        for data in DevicePrefetcher(loader, dev):                # batch i + 1 uploads while step i computes
            metrics, targets, preds, _ = train_loop(data, opt_p, opt_a, ce_loss, pointnet, att_net, None, 'segmentation', train, epoch, 0,
                                                    device_outputs=True)
            counts.append(confusion_device(preds, targets, len(IOU_NAMES)))
            scalars.append(torch.stack([metrics['loss'].reshape(()), metrics['ce_loss'].reshape(()), metrics['reg_loss'].reshape(())]))
        if counts:
            counts_h = torch.stack(counts).cpu().numpy()
            scalars_h = torch.stack(scalars).cpu().numpy()
            for c, sc in zip(counts_h, scalars_h):
                acc, per = metrics_from_confusion(c, len(IOU_NAMES))
                sums['acc'].append(acc)
                for name, v in zip(IOU_NAMES, per):
                    ious[name].append(v)
                sums['loss'].append(float(sc[0]))
                sums['ce'].append(float(sc[1]))
                sums['reg'].append(float(sc[2]))
        return reduce_epoch_metrics(sums, ious, device=dev)
    for data in loader:
        metrics, targets, preds, _ = train_loop(data, opt_p, opt_a, ce_loss, pointnet, att_net, None, 'segmentation', train, epoch, 0)
        preds, targets, _ = rm_padding(preds.reshape(-1), targets.reshape(-1))
        sums['acc'].append(get_accuracy(preds, targets, {}, 'segmentation')['accuracy'])
        for c, name in enumerate(IOU_NAMES):
            ious[name].append(get_iou_obj(preds, targets, c))
        sums['loss'].append(metrics['loss'].item())
        sums['ce'].append(metrics['ce_loss'].item())
        sums['reg'].append(metrics['reg_loss'].item())
    return reduce_epoch_metrics(sums, ious, device=dev)


def reduce_epoch_metrics(sums, ious, device=None):
    """Per-batch metric lists of this rank -> epoch means over ALL ranks' batches (the reference's mean over batches,
    train_pointnet-attention.py:280-312, with the per-class IoU as nanmean).  Under torch.distributed every rank contributes
    (sum, count) pairs to one SUM all-reduce, so every rank sees the same validation loss and takes the same
    checkpoint / early-stop decision; single process: plain means."""
    keys = list(sums.keys()) + ['iou_' + k for k in ious.keys()]
    vals = [np.asarray(v, dtype=np.float64) for v in sums.values()] + [np.asarray(v, dtype=np.float64) for v in ious.values()]
    acc = np.zeros((len(keys), 2), dtype=np.float64)
    for i, v in enumerate(vals):
        ok = ~np.isnan(v)
        acc[i] = (v[ok].sum(), ok.sum())
    from ..trainer import _collectives_on
    dist, _, on = _collectives_on()
    if on:
        on_gpu = dist.get_backend() == "nccl"
        t = torch.from_numpy(acc).to(device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        acc = t.cpu().numpy()
    return {k: (float(a[0] / a[1]) if a[1] > 0 else float('nan')) for k, a in zip(keys, acc)}


def train_att(task, dataset_folder, path_list_files, output_folder, n_points, batch_size, epochs, learning_rate,
              weighing_method='EFS', beta=0.999, number_of_workers=4, model_checkpoint=None, device='cuda', sync_bn=None):
    """sync_bn (default: AMPNET_SYNC_BN=1 in the environment): under data parallelism, BatchNorm statistics and the loss normalisation
    over the GLOBAL batch, i.e. the single-device semantics of the reference at batch_size x world (trainer.enable_sync_batchnorm);
    off, every rank normalises over its own batch_size samples."""
    if task != 'segmentation':
        raise NotImplementedError("only the segmentation task is on the AMP-Net hot path")
    start = time.time()
    rank, world, local = _dist_setup()
    limit_host_threads(reserve=number_of_workers)        # the main process's torch pool next to the loader's workers (utils.host_cpu_budget)
    if world > 1 and (sync_bn if sync_bn is not None else os.environ.get("AMPNET_SYNC_BN") == "1"):
        from ..trainer import enable_sync_batchnorm
        enable_sync_batchnorm()
    device = torch.device('cuda', local)
    with open(os.path.join(path_list_files, 'train_seg_files.txt')) as f:
        train_files = f.read().splitlines()
    with open(os.path.join(path_list_files, 'val_seg_files.txt')) as f:
        val_files = f.read().splitlines()
    if world > 1:
        train_files = [train_files[i] for i in shard_indices(len(train_files), rank, world)]
        val_files = [val_files[i] for i in shard_indices(len(val_files), rank, world)]
    name = 'ATT' + 'g' + str(GLOBAL_FEAT_SIZE) + 'w100' + 'xyz'
    # lazy samples (with the ragged collate): the workers hand collate_seq_ragged file positions and libampnet_host.so reads, filters and
    # relabels every sample straight into the batch (include/ampnet_host.h) -- a third of the worker time of the numpy statement
    padded = os.environ.get("AMPNET_PADDED_COLLATE") == "1" or os.environ.get("AMPNET_HOST_AUG") == "1"
    train_ds = LidarKmeansDataset(dataset_folder, task=task, number_of_points=n_points, files=train_files, lazy=not padded)
    val_ds = LidarKmeansDataset(dataset_folder, task=task, number_of_points=n_points, files=val_files, lazy=not padded)
    # pin_memory: the collated batch (42 MB of points + 9 MB of labels at B = 64) lands in page-locked memory in the loader's pinning
    # thread, so train_loop's single upload runs at PCIe rate instead of through a pageable staging copy (bench.py: train_loop_inclusive)
    # collate: the reference's collate_seq_padd builds the padded [B, 2048, 9, 9] batch in the workers (42 MB of gathers per batch of 64:
    # the epoch then runs at the loader's pace, bench.py train_att_epoch); collate_seq_ragged makes the same draws and leaves resampling
    # and padding to the augmentation kernel.  AMPNET_PADDED_COLLATE=1 (or the numpy augmentation path) selects the reference's.
    # persistent workers, forked NOW: a fork of this process once it holds page-locked memory (the loaders' pinned batches) stalls the GPU
    # queues for seconds, more each epoch (the kernel write-protects the parent's pages for copy-on-write and the driver revalidates what
    # it had pinned: 3.6 / 5.3 / 5.9 s at the start of epochs 2 / 3 / 4 with the default per-epoch workers, tools/prof_loader.py) --
    # so both loaders fork once, before the first pinned allocation, and keep their workers
    mk = lambda ds: torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=number_of_workers,   # noqa: E731
                                                drop_last=True, collate_fn=collate_seq_padd if padded else collate_seq_ragged, pin_memory=True,
                                                persistent_workers=number_of_workers > 0)
    train_loader, val_loader = mk(train_ds), mk(val_ds)
    start_workers(val_loader, train_loader)
    if rank == 0:
        print(f'Dataset folder: {dataset_folder}\nSamples for training: {len(train_ds)} (per rank), validation: {len(val_ds)}')

    pointnet = BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=GLOBAL_FEAT_SIZE, device=device)
    att_net = SegmentationWithAttention(GLOBAL_FEAT_SIZE, ATT_HEADS, num_classes=5, local_dim=64, device=device)
    att_net.seed = (att_net.seed ^ (rank * 0x9E3779B9)) & 0xFFFFFFFF      # data parallel: every rank draws its own dropout masks
    c_weights = torch.FloatTensor([1, 2, 2, 1, 1]).to(device)
    ce_loss = torch.nn.CrossEntropyLoss(weight=c_weights, reduction='mean', ignore_index=-1)
    opt_p = FusedAdam(pointnet.parameters(), lr=learning_rate)
    opt_a = FusedAdam(att_net.parameters(), lr=learning_rate)
    sched = [torch.optim.lr_scheduler.MultiStepLR(o, milestones=[150, 250, 350], gamma=0.5) for o in (opt_p, opt_a)]
    best_vloss, epoch_ini, since = 1_000_000., 0, 0
    if model_checkpoint:
        ck = torch.load(model_checkpoint, map_location=device, weights_only=True)
        pointnet.load_state_dict(ck['base_pointnet'])
        att_net.load_state_dict(ck['segmen_net'])
        opt_p.load_state_dict(ck['opt_pointnet'])
        opt_a.load_state_dict(ck['opt_segmen'])
        batch_size, learning_rate, epoch_ini = ck['batch_size'], ck['lr'], ck['epoch']
    if world > 1:                                      # same start on every rank
        import torch.distributed as dist
        for m in (pointnet, att_net):
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src=0)
    history = []
    for epoch in range(epoch_ini, epochs):
        t0 = time.time()
        tr = _epoch(train_loader, True, pointnet, att_net, opt_p, opt_a, ce_loss, epoch)
        with torch.no_grad():
            va = _epoch(val_loader, False, pointnet, att_net, opt_p, opt_a, ce_loss, epoch)
        for s in sched:
            s.step()
        history.append((tr, va))
        if rank == 0:
            print(f"epoch {epoch}: train loss {tr['loss']:.4f} acc {tr['acc']:.3f} | val loss {va['loss']:.4f} acc {va['acc']:.3f} "
                  f"iou tower {va['iou_tower']:.3f} | {time.time() - t0:.1f} s", flush=True)
        if va['loss'] < best_vloss:
            best_vloss, since = va['loss'], 0
            if rank == 0:
                stamp = datetime.datetime.now().strftime("%m-%d-%H:%M")
                save_checkpoint_segmen_model(stamp + name, task, epoch, since, pointnet, att_net, opt_p, opt_a, va['acc'],
                                             batch_size, learning_rate, n_points, weighing_method)
        else:
            since += 1
    if rank == 0:
        print("--- TOTAL TIME: %s h ---" % (round((time.time() - start) / 3600, 3)))
    return history
