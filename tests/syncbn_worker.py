"""Rank program of tests/test_syncbn_gpu.py: one data-parallel train step (forward + backward + gradient all-reduce) on this rank's
share of a seeded batch, with or without global-batch BatchNorm; rank 0 saves loss terms, averaged gradients and running statistics.
Launched by torch.distributed.run (gloo, 127.0.0.1); every rank uses cuda:0 (one-GPU box)."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "3d-semantic-segmentation-amp-net_amd"


def build(synth, params, M):
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cuda")
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, dropout=0.0, device="cuda")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(3, params.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(3, params.ENC_BUFFERS).items()})
    enc.load_state_dict(sd, strict=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(4, params.HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(4, params.HEAD_BUFFERS).items()})
    att.load_state_dict(sd, strict=False)
    enc.train(); att.train()
    return enc, att


def batch(synth, B, N, W):
    pc, tg, cent, _ = synth.sample_batch(77, B, N, max_w=W)
    x = torch.from_numpy(pc.transpose(0, 3, 1, 2).copy())       # [B, W, N, 9]
    t = torch.from_numpy(tg.transpose(0, 2, 1).copy())
    return x, t, torch.from_numpy(cent)


def step(T, enc, att, x, t, c):
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device="cuda")
    out = T.forward_backward(enc, att, x.cuda(), t.cuda(), c.cuda(), cw)
    world = T.reduce_gradients(out["grad_bufs"], ())
    res = {"ce": out["ce"][0:1].detach().cpu(), "reg": out["reg"].detach().reshape(1).cpu()}
    for tag, m in (("enc", enc), ("att", att)):
        for k, p in m.named_parameters():
            res[f"grad/{tag}/{k}"] = (p.grad / world).detach().cpu()
        for k, b in m.named_buffers():
            if "running" in k:
                res[f"buf/{tag}/{k}"] = b.detach().cpu()
    return res


if __name__ == "__main__":
    out_path, sync, B, N, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    synth, params = importlib.import_module(PKG + ".synthetic"), importlib.import_module(PKG + ".params")
    M, T = importlib.import_module(PKG + ".pointNet.model.pointnetAtt"), importlib.import_module(PKG + ".trainer")
    enc, att = build(synth, params, M)
    x, t, c = batch(synth, B, N, W)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    if sync:
        assert T.enable_sync_batchnorm()
    res = step(T, enc, att, x[sl], t[sl], c[sl])
    if rank == 0:
        torch.save(res, out_path)
    dist.barrier()
    if sync:
        T.disable_sync_batchnorm()
    dist.destroy_process_group()
