"""torch.autograd bridges for the drop-in modules.

When a caller drives the modules the reference's way -- pointnet(x) per window, att_net(...), a torch loss on the
logits, loss.backward(), optimizer.step() (train_pointnet-attention.py:396-470) -- autograd needs a backward for
the HIP forward.  These Functions call the C-ABI backward entry points; every forward in grad mode keeps a private
workspace alive until its backward has run (the reference makes W encoder calls before one backward).
The package's own train_loop does not go through autograd (trainer.fused_train_step).
"""
import torch

from . import _lib, ops
from . import params as P


def _ordered_params(module, table):
    named = dict(module.named_parameters())
    return [named[n] for n in table.keys()]


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, x, *params):
        module, pt, bt, off, Q, total, mx, n_slots = meta
        ws = ops.Workspace()
        local, glob, feat_T, _ = ops.encoder_forward(pt, bt, x, off, Q, total, mx, n_slots, True, ws)
        # tensors go through save_for_backward: keeping an OUTPUT on ctx directly makes an output -> grad_fn -> ctx -> output cycle
        # the garbage collector cannot break, i.e. the whole activation workspace leaks when the backward never runs
        ctx.save_for_backward(x, local, feat_T)
        ctx.meta = (module, pt, off, Q, total, mx, n_slots, ws)
        return local, glob, feat_T

    @staticmethod
    def backward(ctx, d_local, d_glob, d_ft):
        module, pt, off, Q, total, mx, n_slots, ws = ctx.meta
        x, local, feat_T = ctx.saved_tensors
        named = dict(module.named_parameters())
        grads = {n: torch.empty_like(named[n]) for n in P.ENC_PARAMS}
        gt = ops.PointerTable(P.ENC_PARAMS, grads, "encoder gradients")
        ops.encoder_backward(pt, gt, x, off, Q, total, mx, n_slots, local, feat_T,
                             d_local.contiguous().float(), d_glob.contiguous().float(), d_ft.contiguous().float(),
                             ws, ops.Workspace())
        ctx.meta = None
        return (None, None) + tuple(grads[n] for n in P.ENC_PARAMS)


def encoder_apply(module, pt, bt, rows, off, Q, total, mx, n_slots):
    params = _ordered_params(module, P.ENC_PARAMS)
    out = _EncoderFn.apply((module, pt, bt, off, Q, total, mx, n_slots), rows.contiguous(), *params)
    module._bump_batches(n_slots)
    return out


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, gl, lo, *params):
        module, pt, bt, cent, off, mask, B, W, total, mx, n_classes, p_drop, seed = meta
        ws = ops.Workspace()
        logits, _, _ = ops.head_forward(pt, bt, gl, lo, cent, off, mask, B, W, total, mx, n_classes, True, p_drop, seed, ws)
        ctx.save_for_backward(lo)
        ctx.meta = (module, pt, cent, off, B, W, total, mx, n_classes, p_drop, seed, ws)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module, pt, cent, off, B, W, total, mx, n_classes, p_drop, seed, ws = ctx.meta
        lo, = ctx.saved_tensors
        table = module._param_table()
        named = dict(module.named_parameters())
        grads = {n: torch.empty_like(named[n]) for n in table}
        gt = ops.PointerTable(table, grads, "head gradients")
        d_lo, d_gl = ops.head_backward(pt, gt, lo, cent, off, B, W, total, mx, n_classes, p_drop, seed,
                                       dlogits.contiguous().float(), ws, ops.Workspace())
        ctx.meta = None
        return (None, d_gl, d_lo) + tuple(grads[n] for n in table)


def head_apply(module, pt, bt, gl_rows, lo_rows, centroids, off, mask, B, W, total, mx, n_classes, p_drop, seed,
               targets=None, class_w=None, want_preds=False):
    """Grad-mode head forward: returns (logits, preds or None, None) -- the loss is the caller's (a torch loss on the
    logits back-propagates through _HeadFn)."""
    params = _ordered_params(module, module._param_table())
    cent = centroids.to(gl_rows.device).float().contiguous()
    logits = _HeadFn.apply((module, pt, bt, cent, off, mask, B, W, total, mx, n_classes, p_drop, seed),
                           gl_rows.contiguous().float(), lo_rows.contiguous().float(), *params)
    torch._foreach_add_([module.bn_2.num_batches_tracked, module.bn_3.num_batches_tracked], 1)
    preds = logits.detach().argmax(dim=1) if want_preds else None
    return logits, preds, None


class _GruHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, gl, lo, *params):
        module, pt, bt, off, B, W, total, mx, n_classes, p_drop, seed = meta
        ws = ops.Workspace()
        logits, _, _ = ops.gru_head_forward(pt, bt, gl, lo, off, B, W, total, mx, n_classes, True, p_drop, seed, ws)
        ctx.save_for_backward(gl, lo)
        ctx.meta = (module, pt, off, B, W, total, mx, n_classes, p_drop, seed, ws)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module, pt, off, B, W, total, mx, n_classes, p_drop, seed, ws = ctx.meta
        gl, lo = ctx.saved_tensors
        table = module._param_table()
        named = dict(module.named_parameters())
        grads = {n: torch.empty_like(named[n]) for n in table}
        gt = ops.PointerTable(table, grads, "GRU head gradients")
        d_lo, d_gl = ops.gru_head_backward(pt, gt, gl, lo, off, B, W, total, mx, n_classes, p_drop, seed,
                                           dlogits.contiguous().float(), ws, ops.Workspace())
        ctx.meta = None
        return (None, d_gl, d_lo) + tuple(grads[n] for n in table)


def gru_head_apply(module, pt, bt, gl_rows, lo_rows, off, B, W, total, mx, n_classes, p_drop, seed, want_preds=False):
    """Grad-mode forward of SegmentationWithGRU: (logits, preds or None, None); a torch loss on the logits back-propagates
    through _GruHeadFn (the reference's loop, pointNet/rnn/train_pointnetGRU.py:403-433)."""
    params = _ordered_params(module, module._param_table())
    logits = _GruHeadFn.apply((module, pt, bt, off, B, W, total, mx, n_classes, p_drop, seed),
                              gl_rows.contiguous().float(), lo_rows.contiguous().float(), *params)
    torch._foreach_add_([module.bn_2.num_batches_tracked, module.bn_3.num_batches_tracked], 1)
    preds = logits.detach().argmax(dim=1) if want_preds else None
    return logits, preds, None


class _ClsHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, gl, *params):
        module, pt, bt, mask, B, W, n_classes, p_drop, seed = meta
        ws = ops.Workspace()
        out, aw = ops.cls_head_forward(pt, bt, gl, mask, B, W, n_classes, True, p_drop, seed, ws)
        ctx.save_for_backward(gl)
        ctx.meta = (module, pt, B, W, n_classes, p_drop, seed, ws)
        ctx.mark_non_differentiable(aw)
        return out, aw

    @staticmethod
    def backward(ctx, d_out, _d_aw):
        module, pt, B, W, n_classes, p_drop, seed, ws = ctx.meta
        gl, = ctx.saved_tensors
        table = module._param_table()
        named = dict(module.named_parameters())
        grads = {n: torch.empty_like(named[n]) for n in table}
        gt = ops.PointerTable(table, grads, "classification head gradients")
        d_gl = ops.cls_head_backward(pt, gt, gl, B, W, n_classes, p_drop, seed, d_out.contiguous().float(), ws, ops.Workspace())
        ctx.meta = None
        return (None, d_gl) + tuple(grads[n] for n in table)


def cls_head_apply(module, pt, bt, gl_rows, mask, B, W, n_classes, p_drop, seed):
    params = _ordered_params(module, module._param_table())
    out, aw = _ClsHeadFn.apply((module, pt, bt, mask, B, W, n_classes, p_drop, seed), gl_rows.contiguous().float(), *params)
    module.bn_2.num_batches_tracked += 1
    return out, aw
