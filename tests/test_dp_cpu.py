"""Data-parallel host logic on CPU with gloo, world_size 2: the flat-bucket gradient all-reduce + averaging of
trainer.reduce_gradients gives every rank the parameters a single process gets from the averaged gradients, and
shard_indices partitions the samples.  (The kernels themselves need the GPU; the N > 1 GPU path runs the same
Python through RCCL.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sub                           # noqa: E402


def _fill_grads(store, rank):
    g = torch.Generator().manual_seed(100 + rank)
    store.flat.copy_(torch.randn(store.flat.shape, generator=g))
    store.attach()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, M, P = sub("trainer"), sub("pointNet.model.pointnetAtt"), sub("params")
        torch.manual_seed(0)                                   # same initial weights on every rank
        att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cpu")
        store = T.GradStore(att, att._param_table())
        _fill_grads(store, rank)
        opt = torch.optim.Adam(att.parameters(), lr=1e-2)
        w = T.reduce_gradients([store.flat], [opt])
        assert w == world
        opt.step()
        torch.save({k: v.detach().clone() for k, v in att.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
        torch.save(store.flat.clone(), os.path.join(out_dir, f"grad{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    T, M = sub("trainer"), sub("pointNet.model.pointnetAtt")
    torch.manual_seed(0)
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cpu")
    store = T.GradStore(att, att._param_table())
    acc = torch.zeros_like(store.flat)
    for r in range(world):
        _fill_grads(store, r)
        acc += store.flat
    store.flat.copy_(acc / world)
    store.attach()
    opt = torch.optim.Adam(att.parameters(), lr=1e-2)
    opt.step()
    want = att.state_dict()
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for k in want:
        assert torch.equal(r0[k], r1[k]), k                                   # ranks stay in lock-step
        torch.testing.assert_close(r0[k], want[k], rtol=1e-6, atol=1e-7)      # == single process on the mean gradient
    g0 = torch.load(tmp_path / "grad0.pt", weights_only=True)        # averaged gradients (alignment gaps excluded)
    expected = (acc / world).clone()
    for n, v in store.views.items():
        off = (v.data_ptr() - store.flat.data_ptr()) // 4
        torch.testing.assert_close(g0[off:off + v.numel()], expected[off:off + v.numel()], rtol=1e-6, atol=1e-7)


def test_shard_indices_partition():
    T = sub("trainer")
    for n, world in ((10, 2), (4096, 8), (7, 3)):
        shards = [T.shard_indices(n, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == n // world
        flat = sorted(i for s in shards for i in s)
        assert len(set(flat)) == len(flat) and set(flat) <= set(range(n))
    assert T.shard_indices(7, 0, 3, drop_last=False) == [0, 3, 6]


def test_grad_store_views_alias_flat_buffer():
    T, M, P = sub("trainer"), sub("pointNet.model.pointnetAtt"), sub("params")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, device="cpu")
    st = T.GradStore(enc, P.ENC_PARAMS)
    st.attach()
    assert st.flat.numel() >= 883401
    for n, p in enc.named_parameters():
        assert p.grad.shape == p.shape and p.grad.data_ptr() % 256 == st.flat.data_ptr() % 256
    st.flat.fill_(2.0)
    assert all(float(p.grad.sum()) == 2.0 * p.numel() for p in enc.parameters())


def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                  # a cold shell: bench.py must start its own ranks
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` from a plain shell starts two ranks (torch.distributed.run, 127.0.0.1), relays rank 0's JSON line
    and exits 0; --dry-run keeps the GPU out of it (gloo), the rendezvous and launcher are the ones the GPU run uses."""
    import json
    r = _run_bench({}, "--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "ranks": 2, "backend": "gloo", "rank_sum": 3.0}


def test_bench_launcher_reports_a_failed_rank():
    r = _run_bench({"AMPNET_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--dry-run")
    assert r.returncode != 0


def _metrics_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = sub("pointNet.amp_train")
        sums = dict(loss=[1.0 + rank, 2.0 + rank], acc=[0.5 * (rank + 1)] * 2)
        ious = dict(tower=[float("nan"), 0.25 * (rank + 1)], bckg=[0.5, 0.75] if rank == 0 else [float("nan")] * 2)
        torch.save(A.reduce_epoch_metrics(sums, ious), os.path.join(out_dir, f"m{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_epoch_metrics_are_reduced_over_ranks(tmp_path):
    """Every rank must see the same validation loss (checkpoint decision, amp_train.py): mean over all ranks' batches."""
    world, port = 2, 31000 + os.getpid() % 2000
    mp.spawn(_metrics_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    m0 = torch.load(tmp_path / "m0.pt", weights_only=True)
    m1 = torch.load(tmp_path / "m1.pt", weights_only=True)
    assert m0 == m1
    assert m0["loss"] == pytest.approx((1 + 2 + 2 + 3) / 4) and m0["acc"] == pytest.approx(0.75)
    assert m0["iou_tower"] == pytest.approx((0.25 + 0.5) / 2) and m0["iou_bckg"] == pytest.approx(0.625)
