"""Fused AMP-Net training step on the HIP path and the optimiser that goes with it.

fused_train_step() is what the package's train_loop runs in train mode: encoder forward (all B*W windows, per-slot
BatchNorm statistics) -> head forward with fused weighted CE -> CE / reg-loss gradients -> head backward -> encoder
backward -> gradient all-reduce when torch.distributed is initialised (RCCL, one flat bucket per network) ->
optimizer.step() for the two optimisers the caller owns (FusedAdam below, or any torch optimiser: gradients are
left in p.grad).  The reference does the same work with W serial encoder calls and autograd
(train_pointnet-attention.py:396-470).
"""
import ctypes

import torch

from . import _lib, ops
from . import params as P

READY = True


class GradStore:
    """Flat float32 gradient buffer of a module with one 256-byte aligned view per parameter (p.grad = view), so the
    backward kernels write straight into what the optimiser and the all-reduce read."""

    def __init__(self, module, table):
        named = dict(module.named_parameters())
        self.names = list(table.keys())
        dev = next(module.parameters()).device
        offs, total = P.offsets({n: tuple(named[n].shape) for n in self.names})
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = {n: self.flat[o:o + k].view(named[n].shape) for n, (o, k) in offs.items()}
        self._table = None
        self.params = named

    @property
    def table(self):
        """Pointer table for the backward kernels (GPU only, built on first use)."""
        if self._table is None:
            self._table = ops.PointerTable({n: tuple(self.params[n].shape) for n in self.names}, self.views, "gradients")
        return self._table

    def attach(self):
        for n in self.names:
            self.params[n].grad = self.views[n]


def _store(module, table):
    st = getattr(module, "_grad_store", None)
    if st is None or st.flat.device != next(module.parameters()).device:
        st = GradStore(module, table)
        module._grad_store = st
    return st


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam's interface and state_dict layout (step / exp_avg / exp_avg_sq per parameter) on the
    multi-tensor HIP kernel ampnet_adam_step_f32.  No weight decay, no amsgrad (the reference uses neither)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = 1.0

    def _advance(self, group):
        """Bump the step counters of a group's parameters (state created on first use); returns (parameters with a gradient, step)."""
        ps = [p for p in group["params"] if p.grad is not None]
        steps = set()
        for p in ps:
            _lib.require_gpu(p, "parameter")
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not torch.is_tensor(st["step"]):          # checkpoints written by torch < 1.12 keep Adam's step as a Python int
                st["step"] = torch.tensor(float(st["step"]))
            st["step"] += 1
            steps.add(int(st["step"].item()))
        if len(steps) > 1:
            raise _lib.AmpnetError("FusedAdam: parameters of one group must share their step count")
        return ps, (steps.pop() if steps else 0)

    @staticmethod
    def _launch(jobs, lr, b1, b2, eps, step, grad_scale):
        """ONE multi-tensor launch over jobs = [(optimizer, parameters)]: they share every scalar of the update."""
        ps = [p for _, plist in jobs for p in plist]
        st = [o.state[p] for o, plist in jobs for p in plist]
        n = len(ps)
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])   # noqa: E731
        numel = (ctypes.c_long * n)(*[p.numel() for p in ps])
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
        dev = ps[0].device
        with torch.cuda.device(dev):
            rc = _lib.lib().ampnet_adam_step_f32(arr(ps), arr(grads), arr([s["exp_avg"] for s in st]), arr([s["exp_avg_sq"] for s in st]), numel, n,
                                                 ctypes.c_float(lr), ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(eps), step,
                                                 ctypes.c_float(grad_scale), _lib.stream_ptr(dev))
        _lib.check(rc, "ampnet_adam_step_f32")

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise _lib.AmpnetError("FusedAdam: closures are not supported")
        for group in self.param_groups:
            ps, step = self._advance(group)
            if ps:
                self._launch([(self, ps)], group["lr"], *group["betas"], group["eps"], step, self.grad_scale)
        return None

    @staticmethod
    @torch.no_grad()
    def step_together(optimizers):
        """optimizer.step() of several FusedAdam instances (the reference keeps one Adam per network, train_pointnet-attention.py:140-141) in
        as few launches as their hyper-parameters allow: groups with the same (lr, betas, eps, step, grad_scale, device) share one
        multi-tensor launch -- for the reference's recipe, one launch for both networks.  Same arithmetic, element by element."""
        buckets = {}
        for opt in optimizers:
            for group in opt.param_groups:
                ps, step = opt._advance(group)
                if ps:
                    key = (float(group["lr"]), tuple(float(b) for b in group["betas"]), float(group["eps"]), step, float(opt.grad_scale), ps[0].device)
                    buckets.setdefault(key, []).append((opt, ps))
        for (lr, (b1, b2), eps, step, gs, _), jobs in buckets.items():
            FusedAdam._launch(jobs, lr, b1, b2, eps, step, gs)


def _dist_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_world_size()
    return None, 1


def _collectives_on():
    """(dist, world, on): `on` = the data-parallel exchanges are to be issued -- world > 1, or a one-rank group under
    AMPNET_FORCE_COLLECTIVES=1 (tests/test_rccl_gpu.py: the whole exchange path, async head all-reduce, work.wait(), the sync-BatchNorm
    callback and its stream check, then runs on RCCL with one rank, where every collective is the identity and the step must equal the
    plain single-process step)."""
    import os
    dist, world = _dist_world()
    return dist, world, dist is not None and (world > 1 or os.environ.get("AMPNET_FORCE_COLLECTIVES") == "1")


_SYNC = {"on": False}


def enable_sync_batchnorm():
    """Global-batch BatchNorm under data parallelism (include/ampnet_hip.h: ampnet_set_collective; SURVEY section 8(e) option A): every
    train-mode BatchNorm of the HIP launch sequences merges its statistics over all ranks, and forward_backward normalises the loss
    over the global batch, so that a step of N ranks equals the single-process step on the concatenated batch.  The exchanges are
    torch.distributed collectives (RCCL under the nccl backend) on views of a scratch tensor this function keeps alive.
    Call after init_process_group, with the rank's device current.  Returns True when it took effect (world size > 1)."""
    dist, world, on = _collectives_on()
    if not on:
        return False
    L = _lib.lib()
    L.ampnet_collective_scratch_bytes.restype = ctypes.c_size_t
    nbytes = L.ampnet_collective_scratch_bytes(world)
    dev = torch.device("cuda", torch.cuda.current_device())
    scratch = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    base = scratch.data_ptr()
    proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    state = {"error": None}

    def collective(ctx, op, send, recv, n, stream):
        try:
            if (stream or 0) != torch.cuda.current_stream(dev).cuda_stream:
                raise _lib.AmpnetError("sync BatchNorm: the launch stream is not torch's current stream")
            s0, r0 = (send - base) // 4, (recv - base) // 4
            if op == 1:
                dist.all_reduce(scratch[s0:s0 + n], op=dist.ReduceOp.SUM)
            else:
                out, inp = scratch[r0:r0 + world * n], scratch[s0:s0 + n]
                try:
                    dist.all_gather_into_tensor(out, inp)
                except (RuntimeError, NotImplementedError):
                    dist.all_gather(list(out.chunk(world)), inp)
            return 0
        except Exception as e:                       # an exception must not unwind through the C frames
            state["error"] = e
            return 1

    cb = proto(collective)
    with torch.cuda.device(dev):
        rc = L.ampnet_set_collective(cb, None, dist.get_rank(), world, ctypes.c_void_p(base), ctypes.c_size_t(nbytes))
    _lib.check(rc, "ampnet_set_collective")
    _SYNC.update(on=True, cb=cb, scratch=scratch, state=state, world=world)
    return True


def disable_sync_batchnorm():
    if _SYNC.get("on"):
        _lib.check(_lib.lib().ampnet_set_collective(None, None, 0, 1, None, ctypes.c_size_t(0)), "ampnet_set_collective")
    _SYNC.clear()
    _SYNC["on"] = False


def _global_loss(loss2, reg):
    """Loss terms of the GLOBAL batch from the rank's (weighted-mean CE, sum of weights) and reg = || I - F F^T || over its own samples:
    ce_g = sum_r ce_r sw_r / sum_r sw_r, reg_g = sqrt(sum_r reg_r^2).  Returns (loss2 for ampnet_ce_bwd_f32, reg for ampnet_reg_loss_bwd_f32,
    world): the backward seeds are world x d(global loss)/d(local tensors), because the gradient all-reduce is followed by 1 / world."""
    dist, world = _dist_world()
    pack = torch.stack([loss2[0] * loss2[1], loss2[1], reg.reshape(-1)[0] ** 2])
    dist.all_reduce(pack, op=dist.ReduceOp.SUM)
    return torch.stack([pack[0] / pack[1], pack[1] / world]), pack[2].sqrt().reshape(1), world


def forward_backward(pointnet, att_net, x, t, centroids, class_w, reg_weight=0.001, overlap_allreduce=False):
    """Forward + loss + backward of one batch; gradients land in p.grad (views of the modules' flat buffers).
    x [B, W, N, 9] f32, t [B, W, N] i64 (host or device), centroids [B, W, 2].
    overlap_allreduce (data parallel, opt-in: fused_train_step sets it): the head's flat gradient buffer is complete before the encoder
    backward starts, so its SUM all-reduce is issued there asynchronously and travels under the encoder backward; the work handle is
    returned in out["pending"] = {data_ptr of the buffer: handle} and MUST be handed to reduce_gradients (which waits for it instead of
    reducing that buffer again).  Callers that do their own gradient handling leave it off and get purely local gradients."""
    dev = next(pointnet.parameters()).device
    if dev.type != "cuda":
        raise _lib.AmpnetError("the AMP-Net HIP path needs the model on the GPU")
    if not (pointnet.training and att_net.training):
        raise _lib.AmpnetError("forward_backward needs both modules in train mode")
    xd = torch.as_tensor(x).to(dev, non_blocking=True).float()
    B, W, N, _ = xd.shape
    targets_pc = torch.as_tensor(t).reshape(B, W * N)
    tgd = targets_pc.to(dev, non_blocking=True).long()          # any integer dtype in (numpy int32 labels): the C ABI takes int64
    cent = torch.as_tensor(centroids).to(dev).float().contiguous() if centroids is not None else None
    Q, rows = B * W, B * W * N
    eg = _store(pointnet, P.ENC_PARAMS)
    hg = _store(att_net, att_net._param_table())
    ept, ebt = pointnet._tables()
    hpt, hbt = att_net._tables()
    xr = xd.reshape(rows, 9)
    off, total, mx = ops.window_offsets([N] * Q, dev)
    # ---- forward ----
    local, glob, feat_T, _ = ops.encoder_forward(ept, ebt, xr, off, Q, total, mx, W, True, pointnet._ws)
    gru = getattr(att_net, "head_kind", "attention") == "gru"      # SegmentationWithGRU: no centroids, no key-padding mask
    seed = (att_net.seed + 0x632BE5AB * att_net._step) & 0xFFFFFFFF
    att_net._step += 1
    if gru:
        if class_w is None:
            class_w = torch.ones(att_net.num_classes, dtype=torch.float32, device=dev)
        logits, preds, loss2 = ops.gru_head_forward(hpt, hbt, glob, local, off, B, W, total, mx, att_net.num_classes, True, att_net.p_drop,
                                                    seed, att_net._ws, targets=tgd, class_w=class_w, want_preds=True)
    else:
        mask = ops.pad_mask(tgd, W)                                 # the reference's literal (targets.view(B, -1, W) == -1).all(dim=1), one launch
        logits, preds, loss2 = ops.head_forward(hpt, hbt, glob, local, cent, off, mask, B, W, total, mx, att_net.num_classes, True,
                                                att_net.p_drop, seed, att_net._ws, targets=tgd, class_w=class_w, want_preds=True)
    # num_batches_tracked: + W for the encoder's 16 BatchNorms (W encoder calls in the reference), + 1 for the head's two; one launch
    counters = pointnet._bn_counters() + [att_net.bn_2.num_batches_tracked, att_net.bn_3.num_batches_tracked]
    torch._foreach_add_(counters, [W] * (len(counters) - 2) + [1, 1])
    feat_last = feat_T[-B:]
    reg, G = ops.reg_loss(feat_last, keep_G=True)
    # ---- backward ----
    if not hasattr(att_net, "_bws"):
        att_net._bws, pointnet._bws = ops.Workspace(), ops.Workspace()
    if _SYNC["on"]:
        if _SYNC["state"]["error"] is not None:
            raise _SYNC["state"]["error"]
        loss2, reg_b, world = _global_loss(loss2, reg)      # reported ce / reg are the global batch's; seeds scaled by world (see there)
        reg, reg_weight = reg_b, reg_weight * world
    dlog = ops.ce_backward(logits, tgd, class_w, loss2)
    if gru:
        d_lo, d_gl = ops.gru_head_backward(hpt, hg.table, glob, local, off, B, W, total, mx, att_net.num_classes, att_net.p_drop, seed,
                                           dlog, att_net._ws, att_net._bws)
    else:
        d_lo, d_gl = ops.head_backward(hpt, hg.table, local, cent, off, B, W, total, mx, att_net.num_classes, att_net.p_drop, seed,
                                       dlog, att_net._ws, att_net._bws)
    # data parallel: the head's gradient buffer is complete here -- its all-reduce travels while the encoder backward runs
    pending = {}
    if overlap_allreduce:
        dist, world, on = _collectives_on()
        if on:
            pending[hg.flat.data_ptr()] = dist.all_reduce(hg.flat, op=dist.ReduceOp.SUM, async_op=True)
    d_ft = ops.reg_loss_backward_stack(feat_last, G, reg, reg_weight, feat_T.shape[0])      # zeros + the regulariser's gradient, one launch
    ops.encoder_backward(ept, eg.table, xr, off, Q, total, mx, W, local, feat_T, d_lo, d_gl, d_ft, pointnet._ws, pointnet._bws)
    eg.attach()
    hg.attach()
    return dict(logits=logits, preds=preds, ce=loss2, reg=reg, targets_pc=targets_pc, B=B, grad_bufs=(eg.flat, hg.flat), pending=pending)


def reduce_gradients(grad_bufs, optimizers, pending=None):
    """Data-parallel gradient exchange: ONE all-reduce (SUM) per network over its flat gradient buffer (4.8 MB in all,
    latency-bound on xGMI), the 1 / world_size average folded into FusedAdam's kernel (or applied to p.grad for other
    optimisers).  `pending` = forward_backward's out["pending"]: an all-reduce already started for a buffer (the head's, overlapped with
    the encoder backward) is waited for here instead of being issued again.  No-op when torch.distributed is not initialised.
    Returns the world size."""
    dist, world, on = _collectives_on()
    if on:
        pending = dict(pending or {})
        for flat in grad_bufs:
            work = pending.pop(flat.data_ptr(), None)
            if work is not None:
                work.wait()
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        for work in pending.values():                 # handles of buffers not listed: never leave a collective un-waited
            work.wait()
        for opt in optimizers:
            if isinstance(opt, FusedAdam):
                opt.grad_scale = 1.0 / world
            else:
                for g in opt.param_groups:
                    for p in g["params"]:
                        if p.grad is not None:
                            p.grad.mul_(1.0 / world)
    return world


def shard_indices(n_samples, rank, world, drop_last=True):
    """Rank-strided shard of range(n_samples): rank r gets r, r + world, ...; with drop_last every rank gets the same
    count (the collectives need equal step counts)."""
    idx = list(range(rank, n_samples, world))
    if drop_last:
        idx = idx[: n_samples // world]
    return idx


def fused_train_step(pointnet, att_net, optimizer_pointnet, optimizer_att, x, t, centroids, class_w):
    out = forward_backward(pointnet, att_net, x, t, centroids, class_w, overlap_allreduce=True)
    reduce_gradients(out["grad_bufs"], (optimizer_pointnet, optimizer_att), out["pending"])
    if isinstance(optimizer_pointnet, FusedAdam) and isinstance(optimizer_att, FusedAdam):
        FusedAdam.step_together((optimizer_pointnet, optimizer_att))          # one multi-tensor launch for both networks
    else:
        optimizer_pointnet.step()
        optimizer_att.step()
    return out


class Trainer:
    """Owns the two FusedAdam optimisers of the reference recipe and runs fused steps (bench.py, smoke())."""

    def __init__(self, pointnet, att_net, lr=1e-3, class_w=None, world_size=1):
        self.pointnet, self.att_net = pointnet, att_net
        self.opt_p = FusedAdam(pointnet.parameters(), lr=lr)
        self.opt_a = FusedAdam(att_net.parameters(), lr=lr)
        dev = next(pointnet.parameters()).device
        self.class_w = (class_w if class_w is not None else torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0])).to(dev)
        pointnet.train()
        att_net.train()

    def step(self, x, t, centroids):
        return fused_train_step(self.pointnet, self.att_net, self.opt_p, self.opt_a, x, t, centroids, self.class_w)
