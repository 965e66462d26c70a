"""Baseline PointNet segmentation on the HIP path: the per-batch step, the training driver and the test driver with the names,
arguments, files and return values of the reference's pointNet/baseline/train_segmentation.py (train :33-271, train_loop :274-328)
and pointNet/baseline/test_segmentation.py.  BASELINE.json config 1 (batch 4, N = 512, 9 features, 5 classes).

The step is the reference's own: z-rotation of the cloud (numpy RNG, also in validation, :284), logits, feat_T = pointnet(pc),
CrossEntropyLoss(weight [1,2,2,1,1], ignore_index -1), reg = || I - F F^T ||, loss.backward(), optimizer.step() -- the module's
forward and backward run through the C ABI (pointNet/model/_baseline.py), the two scalar losses are torch ops on the GPU."""
import datetime
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from ..utils.get_metrics import get_accuracy, get_iou_obj
from ..utils.utils import rotate_point_cloud_z, save_checkpoint
from .datasets import LidarDatasetExpanded

GLOBAL_FEAT_SIZE = 256
NUM_CLASES = 5


def train_loop(data, optimizer, ce_loss, pointnet, w_tensorboard=None, train=True, epoch=0, last_epoch=0, first_batch_val=False):
    """Drop-in for train_segmentation.py:274-328 -> (metrics {'ce_loss', 'reg_loss', 'loss'}, targets [B, N] cpu, preds [B, N] cpu, last_epoch)."""
    metrics = {'accuracy': []}
    pc, targets, filenames = data
    dev = next(pointnet.parameters()).device
    pc = np.array(torch.as_tensor(pc).numpy(), dtype=np.float32, copy=True)
    pc[:, :, :3] = rotate_point_cloud_z(pc[:, :, :3])
    pc = torch.from_numpy(pc).to(dev)
    targets = torch.as_tensor(targets).to(dev)
    optimizer.zero_grad()
    pointnet = pointnet.train() if train else pointnet.eval()
    with torch.set_grad_enabled(train):
        logits, feat_transform = pointnet(pc)
        metrics['ce_loss'] = ce_loss(logits, targets).view(-1, 1)
        preds = F.log_softmax(logits.detach(), dim=1).max(1)[1].cpu()
        identity = torch.eye(feat_transform.shape[-1], device=dev)
        metrics['reg_loss'] = torch.norm(identity - torch.bmm(feat_transform, feat_transform.transpose(2, 1)))
        if train:
            metrics['loss'] = metrics['ce_loss'] + 0.001 * metrics['reg_loss']
            metrics['loss'].backward()
            optimizer.step()
        else:
            metrics['loss'] = metrics['ce_loss']
    return metrics, targets.detach().cpu(), preds, last_epoch


def _model(model, device):
    if model == 'light':
        from .model.light_pointnet_256 import SegmentationPointNet
        return SegmentationPointNet(num_classes=NUM_CLASES, point_dimension=2, device=device)
    from .model.pointnet import SegmentationPointNet
    return SegmentationPointNet(num_classes=NUM_CLASES, point_dimension=3, device=device)


def train(dataset_folder, path_list_files, output_folder, n_points, batch_size, epochs, learning_rate, number_of_workers=0,
          model_checkpoint=None, c_sample=False, model='pointnet', device='cuda'):
    """Drop-in for train_segmentation.py:33-271: Adam, MultiStepLR [50, 100, 300] gamma 0.5, checkpoint (utils.save_checkpoint keys)
    whenever the mean validation loss improves.  `model`: 'pointnet' (pointNet/model/pointnet.py, the one that runs in the reference,
    SURVEY F4) or 'light' (light_pointnet_256.py with point_dimension=2).  Returns the per-epoch history."""
    start = time.time()
    with open(os.path.join(path_list_files, 'train_seg_files.txt')) as f:
        train_files = f.read().splitlines()
    with open(os.path.join(path_list_files, 'val_seg_files.txt')) as f:
        val_files = f.read().splitlines()
    mk = lambda files: torch.utils.data.DataLoader(                                            # noqa: E731
        LidarDatasetExpanded(dataset_folder=dataset_folder, task='segmentation', number_of_points=n_points, files=files, fixed_num_points=True),
        batch_size=batch_size, shuffle=True, num_workers=number_of_workers, drop_last=True)
    train_loader, val_loader = mk(train_files), mk(val_files)
    pointnet = _model(model, torch.device(device))
    optimizer = torch.optim.Adam(pointnet.parameters(), lr=learning_rate)
    ce_loss = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]).to(device), reduction='mean', ignore_index=-1)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[50, 100, 300], gamma=0.5)
    if model_checkpoint:
        ck = torch.load(model_checkpoint, map_location=device, weights_only=True)
        pointnet.load_state_dict(ck['model'])
        optimizer.load_state_dict(ck['optimizer'])
    best, since, history = 1_000_000., 0, []
    name = datetime.datetime.now().strftime("%m-%d-%H:%M") + 'segBase100_' + str(GLOBAL_FEAT_SIZE)
    for epoch in range(epochs):
        ep = dict(train_loss=[], val_loss=[], train_acc=[], val_acc=[], iou_tower_val=[])
        for data in train_loader:
            m, targets, preds, _ = train_loop(data, optimizer, ce_loss, pointnet, None, True, epoch, -1)
            ep['train_loss'].append(m['loss'].item())
            ep['train_acc'].append(get_accuracy(preds.view(-1), targets.view(-1), {}, 'segmentation')['accuracy'])
        scheduler.step()
        with torch.no_grad():
            for data in val_loader:
                m, targets, preds, _ = train_loop(data, optimizer, ce_loss, pointnet, None, False, epoch, -1)
                ep['val_loss'].append(m['loss'].item())
                ep['val_acc'].append(get_accuracy(preds.view(-1), targets.view(-1), {}, 'segmentation')['accuracy'])
                ep['iou_tower_val'].append(get_iou_obj(targets.view(-1), preds.view(-1), 1))
        row = {k: float(np.nanmean(v)) if v else float('nan') for k, v in ep.items()}
        history.append(row)
        print(f"epoch {epoch}: train loss {row['train_loss']:.4f} acc {row['train_acc']:.3f} | val loss {row['val_loss']:.4f} acc {row['val_acc']:.3f}", flush=True)
        if row['val_loss'] < best:
            save_checkpoint(name, epoch, since, pointnet, optimizer, row['val_acc'], batch_size, learning_rate, n_points)
            best, since = row['val_loss'], 0
        else:
            since += 1
            if since > 100:
                break
    print("--- TOTAL TIME: %s h ---" % (round((time.time() - start) / 3600, 3)))
    return history


def test(dataset_folder, n_points, output_folder, number_of_workers, model_checkpoint, path_list_files='pointNet/data/train_test_files/RGBN',
         model='pointnet', device='cuda'):
    """Drop-in for pointNet/baseline/test_segmentation.py: one file per step (batch 1, all its points), per-class IoU, mean IoU and
    accuracy over the test list.  Returns the summary dict (the reference prints it and appends a CSV row)."""
    ck = torch.load(model_checkpoint, map_location=device, weights_only=True)
    with open(os.path.join(path_list_files, 'test_seg_files.txt')) as f:
        files = f.read().splitlines()
    ds = LidarDatasetExpanded(dataset_folder=dataset_folder, task='segmentation', number_of_points=n_points, files=files, fixed_num_points=False)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=number_of_workers, drop_last=False)
    net = _model(model, torch.device(device))
    net.load_state_dict(ck['model'])
    net.eval()
    names = ['bckg', 'tower', 'cables', 'low_veg', 'high_veg']
    iou = {k: [] for k in names}
    acc = []
    with torch.no_grad():
        for pc, labels, _ in loader:
            logits, _ = net(pc.float().to(device))
            preds = F.log_softmax(logits, dim=1).max(1)[1].cpu().view(-1)
            t = labels.view(-1)
            acc.append(get_accuracy(preds, t, {}, 'segmentation')['accuracy'])
            for c, k in enumerate(names):
                if (t == c).any():
                    iou[k].append(get_iou_obj(t, preds, c))
    out = {'iou': {k: float(np.mean(v)) if v else float('nan') for k, v in iou.items()}, 'accuracy': float(np.mean(acc))}
    out['mean_iou'] = float(np.nanmean(list(out['iou'].values())))
    print('mean_iou: ', out['mean_iou'], ' accuracy: ', out['accuracy'])
    return out
