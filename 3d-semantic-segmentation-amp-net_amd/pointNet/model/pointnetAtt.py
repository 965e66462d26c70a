"""Drop-in AMP-Net modules on the MI355X HIP path.

Same class names, constructor arguments, forward signatures, return values and `state_dict` keys as the
reference's pointNet/model/pointnetAtt.py (TransformationNet :7-47, BasePointNet :50-112,
SegmentationWithAttention :154-209, SegmentationWithGRU :212-258), so reference checkpoints load and the reference's train/test scripts
run unchanged on top of them.  The modules own ordinary nn.Parameters / buffers (as holders: their own
`forward` is never used); every forward goes through the C ABI (ops.encoder_forward / ops.head_forward).
There is no CPU or torch fallback: on a CPU tensor the call raises.

Beyond the reference signatures, BasePointNet.forward_windows() and SegmentationWithAttention.forward_rows()
take ALL windows of a step at once (what the reference does with W serial encoder calls and a Python
repeat/cat loop); the package's train_loop uses those.

Only the AMP-Net configuration is implemented in HIP: point_dimension=3, global_feat_dim=256, local_dim=64,
embed_dim=256, num_heads=8, num_classes<=8 (train_pointnet-attention.py:110-118); other values raise.
"""
import torch
import torch.nn as nn

from ... import _lib, ops
from ... import params as P


class _Conv(nn.Module):
    """Holder with nn.Conv1d(k=1)'s parameter names and shapes."""

    def __init__(self, cin, cout, bias, device):
        super().__init__()
        k = 1.0 / cin ** 0.5
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, device=device).uniform_(-k, k))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout, device=device).uniform_(-k, k))


class _Linear(nn.Module):
    def __init__(self, cin, cout, bias, device):
        super().__init__()
        k = 1.0 / cin ** 0.5
        self.weight = nn.Parameter(torch.empty(cout, cin, device=device).uniform_(-k, k))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout, device=device).uniform_(-k, k))


class _BN(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c, device=device))
        self.bias = nn.Parameter(torch.zeros(c, device=device))
        self.register_buffer("running_mean", torch.zeros(c, device=device))
        self.register_buffer("running_var", torch.ones(c, device=device))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long, device=device))


class _MHA(nn.Module):
    """Holder with nn.MultiheadAttention's parameter names (packed in-projection)."""

    def __init__(self, e, device):
        super().__init__()
        k = (6.0 / (4 * e)) ** 0.5                       # xavier_uniform_ on [3e, e]
        self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e, device=device).uniform_(-k, k))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * e, device=device))
        self.out_proj = _Linear(e, e, True, device)
        with torch.no_grad():
            self.out_proj.bias.zero_()


def _named_tensors(module):
    d = {k: v for k, v in module.named_parameters()}
    d.update({k: v for k, v in module.named_buffers() if not k.endswith("num_batches_tracked")})
    return d


class _TableCache:
    """PointerTables rebuilt only when a tensor moved (load_state_dict copies in place, .to() may not)."""

    def __init__(self):
        self.key = None
        self.tables = None

    def get(self, module, ptable, btable, what):
        t = _named_tensors(module)
        key = tuple(v.data_ptr() for v in t.values())
        if key != self.key:
            self.tables = (ops.PointerTable(ptable, t, what + " parameters"), ops.PointerTable(btable, t, what + " buffers"))
            self.key = key
        return self.tables


class TransformationNet(nn.Module):
    """Parameter holder of one T-Net (pointnetAtt.py:9-26).  Its computation is part of BasePointNet's
    launch sequence (csrc/encoder.hip: run_tnet); calling it on its own is not supported."""

    def __init__(self, input_dim, output_dim, device):
        super().__init__()
        self.device = device
        self.output_dim = output_dim
        self.conv_1 = _Conv(input_dim, 64, False, device)
        self.conv_2 = _Conv(64, 128, False, device)
        self.conv_3 = _Conv(128, 256, False, device)
        self.bn_1, self.bn_2, self.bn_3 = _BN(64, device), _BN(128, device), _BN(256, device)
        self.bn_4, self.bn_5 = _BN(256, device), _BN(128, device)
        self.fc_1 = _Linear(256, 256, False, device)
        self.fc_2 = _Linear(256, 128, False, device)
        self.fc_3 = _Linear(128, output_dim * output_dim, True, device)

    def forward(self, x):
        raise _lib.AmpnetError("TransformationNet runs inside BasePointNet's HIP launch sequence; call BasePointNet")


class BasePointNet(nn.Module):

    def __init__(self, point_dimension=2, return_local_features=False, global_feat_dim=256, device='cuda'):
        super().__init__()
        if point_dimension != P.POINT_DIM or global_feat_dim != P.GLOBAL_DIM:
            raise NotImplementedError("the HIP encoder is built for point_dimension=3, global_feat_dim=256 "
                                      "(train_pointnet-attention.py:110-113)")
        self.global_feat_dim = global_feat_dim
        self.point_dimension = point_dimension
        self.return_local_features = return_local_features
        self.input_transform = TransformationNet(point_dimension, point_dimension, device)
        self.feature_transform = TransformationNet(64, 64, device)
        self.conv_1 = _Conv(9 + point_dimension, 64, False, device)
        self.conv_2 = _Conv(64, 64, False, device)
        self.conv_3 = _Conv(64, 64, False, device)
        self.conv_4 = _Conv(64, 128, False, device)
        self.conv_5 = _Conv(128, 128, False, device)
        self.conv_6 = _Conv(128, global_feat_dim, False, device)
        self.bn_1, self.bn_2, self.bn_3 = _BN(64, device), _BN(64, device), _BN(64, device)
        self.bn_4, self.bn_5, self.bn_6 = _BN(128, device), _BN(128, device), _BN(global_feat_dim, device)
        self._cache = _TableCache()
        self._ws = ops.Workspace()

    def _tables(self):
        return self._cache.get(self, P.ENC_PARAMS, P.ENC_BUFFERS, "BasePointNet")

    def _bn_counters(self):
        return [m.num_batches_tracked for m in self.modules() if isinstance(m, _BN)]

    def _bump_batches(self, n):
        # one multi-tensor launch for the 16 counters (16 single-element kernels per step otherwise)
        torch._foreach_add_(self._bn_counters(), n)

    def forward_windows(self, x, np_cluster=None, n_slots=1):
        """All windows of a step in one launch sequence.

        x: [Q, N, 9] (uniform windows) or [rows, 9] with np_cluster = list of Q window sizes.
        n_slots: train mode only -- windows q with equal q % n_slots share BatchNorm batch statistics
        (Q = B * n_slots, q = b * n_slots + w), i.e. n_slots = W reproduces the reference's W encoder calls.
        Returns (local [rows, 64], global [Q, 256], feature_transform [Q, 64, 64]); in train mode the
        feature transforms are slot-major (the last Q / n_slots rows belong to the last slot)."""
        if x.dim() == 3:
            sizes = [x.shape[1]] * x.shape[0]
            rows = x.reshape(-1, x.shape[2])
        else:
            if np_cluster is None:
                raise _lib.AmpnetError("forward_windows: [rows, 9] input needs np_cluster")
            sizes, rows = [int(n) for n in np_cluster], x
        _lib.require_gpu(rows, "x")
        off, total, mx = ops.window_offsets(sizes, rows.device)
        pt, bt = self._tables()
        train = self.training
        if train and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ...autograd import encoder_apply
            return encoder_apply(self, pt, bt, rows.float(), off, len(sizes), total, mx, n_slots)
        local, glob, feat_T, _ = ops.encoder_forward(pt, bt, rows.float(), off, len(sizes), total, mx,
                                                     n_slots if train else 1, train, self._ws)
        if train:
            self._bump_batches(n_slots)
        return local, glob, feat_T

    def forward(self, x):
        """Reference signature (pointnetAtt.py:80-112): x [B, N, 9] ->
        (cat([global.repeat(N), local], 2) [B, N, 320], feature_transform [B, 64, 64]) when
        return_local_features else (global [B, 256], feature_transform)."""
        B, N = x.shape[0], x.shape[1]
        local, glob, feat_T = self.forward_windows(x, n_slots=1)
        if self.return_local_features:
            out = torch.cat([glob.unsqueeze(1).expand(B, N, self.global_feat_dim), local.view(B, N, 64)], dim=2)
            return out, feat_T
        return glob, feat_T


class SegmentationWithAttention(nn.Module):

    def __init__(self, embed_dim, num_heads, num_classes=2, local_dim=128, dropout=0.3, device='cuda'):
        super().__init__()
        if embed_dim != P.GLOBAL_DIM or num_heads != P.HEADS or local_dim != P.LOCAL_DIM or not (1 <= num_classes <= 8):
            raise NotImplementedError("the HIP head is built for embed_dim=256, num_heads=8, local_dim=64, "
                                      "num_classes<=8 (train_pointnet-attention.py:118)")
        self.embed_dim = embed_dim
        self.device = device
        self.num_classes = num_classes
        self.p_drop = float(dropout)
        self.fc1 = _Linear(2, 16, True, device)
        self.fc2 = _Linear(16, embed_dim, True, device)
        self.attention = _MHA(embed_dim, device)
        self.conv_2 = _Conv(local_dim + embed_dim, embed_dim // 2, True, device)
        self.conv_3 = _Conv(embed_dim // 2, 64, True, device)
        self.conv_4 = _Conv(64, num_classes, True, device)
        self.bn_2 = _BN(embed_dim // 2, device)
        self.bn_3 = _BN(64, device)
        self._cache = _TableCache()
        self._ws = ops.Workspace()
        self._step = 0
        self.seed = 0x5EED

    def _param_table(self):
        from collections import OrderedDict
        table = OrderedDict(P.HEAD_PARAMS)
        table["conv_4.weight"] = (self.num_classes, 64, 1)
        table["conv_4.bias"] = (self.num_classes,)
        return table

    def _tables(self):
        return self._cache.get(self, self._param_table(), P.HEAD_BUFFERS, "SegmentationWithAttention")

    def forward_rows(self, gl_rows, lo_rows, centroids, np_cluster, attn_mask=None, targets=None, class_w=None,
                     want_preds=False):
        """gl_rows [B*W, 256] (row b*W+w), lo_rows [B*P, 64] -> (logits [B, C, P], preds or None, loss or None)."""
        B, W = centroids.shape[0], centroids.shape[1]
        sizes = [int(n) for n in np_cluster] * B
        off, total, mx = ops.window_offsets(sizes, lo_rows.device)
        pt, bt = self._tables()
        train = self.training
        seed = (self.seed + 0x632BE5AB * self._step) & 0xFFFFFFFF
        if train:
            self._step += 1
        if train and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ...autograd import head_apply
            return head_apply(self, pt, bt, gl_rows, lo_rows, centroids, off, attn_mask, B, W, total, mx,
                              self.num_classes, self.p_drop, seed, targets, class_w, want_preds)
        out = ops.head_forward(pt, bt, gl_rows, lo_rows, centroids, off, attn_mask, B, W, total, mx, self.num_classes,
                               train, self.p_drop, seed, self._ws, targets=targets, class_w=class_w, want_preds=want_preds)
        if train:
            torch._foreach_add_([self.bn_2.num_batches_tracked, self.bn_3.num_batches_tracked], 1)
        return out

    def forward(self, gl_feats, lo_feats, centroids, np_cluster, attn_mask=None):
        """Reference signature (pointnetAtt.py:176-209): gl_feats [W, B, 256], lo_feats [B, P, 64],
        centroids [B, W, 2], np_cluster list of W sizes, attn_mask [B, W] bool -> (logits [B, C, P], 0)."""
        W, B = gl_feats.shape[0], gl_feats.shape[1]
        gl_rows = gl_feats.transpose(0, 1).reshape(B * W, self.embed_dim)
        lo_rows = lo_feats.reshape(-1, lo_feats.shape[2])
        logits, _, _ = self.forward_rows(gl_rows, lo_rows, centroids, np_cluster, attn_mask)
        return logits, 0


class _GRU(nn.Module):
    """Holder with nn.GRU(input, hidden, num_layers=1)'s parameter names, shapes and U(-1/sqrt(H), 1/sqrt(H)) initialisation."""

    def __init__(self, cin, hidden, device):
        super().__init__()
        k = 1.0 / hidden ** 0.5
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * hidden, cin, device=device).uniform_(-k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * hidden, hidden, device=device).uniform_(-k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * hidden, device=device).uniform_(-k, k))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * hidden, device=device).uniform_(-k, k))


class SegmentationWithGRU(nn.Module):
    """The GRU variant of the sequence model (pointnetAtt.py:212-258): nn.GRU(256 -> 64, batch_first, h0 = 0) over the
    window tokens, hidden state of step w broadcast over the points of window w, then conv_2 / bn_2 / conv_3 / bn_3 / conv_4 with
    Dropout(0.3) twice.  Runs through ampnet_gru_head_fwd_f32 / _bwd_f32 (csrc/gru_head.hip)."""
    head_kind = "gru"

    def __init__(self, num_classes, global_feat_size, hidden_size, device):
        super().__init__()
        if global_feat_size != P.GLOBAL_DIM or hidden_size != P.GRU_HIDDEN or not (1 <= num_classes <= 8):
            raise NotImplementedError("the HIP GRU head is built for global_feat_size=256, hidden_size=64, num_classes<=8 "
                                      "(pointNet/rnn/train_pointnetGRU.py:27-28,128)")
        self.hidden_size = hidden_size
        self.device = device
        self.num_classes = num_classes
        self.p_drop = 0.3                                   # nn.Dropout(0.3), pointnetAtt.py:225
        self.gru_global = _GRU(global_feat_size, hidden_size, device)
        self.conv_2 = _Conv(64 + 64, 128, True, device)
        self.conv_3 = _Conv(128, 64, True, device)
        self.conv_4 = _Conv(64, num_classes, True, device)
        self.bn_2 = _BN(128, device)
        self.bn_3 = _BN(64, device)
        self._cache = _TableCache()
        self._ws = ops.Workspace()
        self._step = 0
        self.seed = 0x6A09E667

    def _param_table(self):
        from collections import OrderedDict
        table = OrderedDict(P.GRU_HEAD_PARAMS)
        table["conv_4.weight"] = (self.num_classes, 64, 1)
        table["conv_4.bias"] = (self.num_classes,)
        return table

    def _tables(self):
        return self._cache.get(self, self._param_table(), P.HEAD_BUFFERS, "SegmentationWithGRU")

    def next_seed(self):
        seed = (self.seed + 0x632BE5AB * self._step) & 0xFFFFFFFF
        if self.training:
            self._step += 1
        return seed

    def forward_rows(self, gl_rows, lo_rows, np_cluster, B, targets=None, class_w=None, want_preds=False):
        """gl_rows [B*W, 256] (row b*W+w), lo_rows [B*P, 64], np_cluster: the W window sizes of a sample
        -> (logits [B, C, P], preds or None, loss or None)."""
        W = len(np_cluster)
        sizes = [int(n) for n in np_cluster] * B
        off, total, mx = ops.window_offsets(sizes, lo_rows.device)
        pt, bt = self._tables()
        train = self.training
        seed = self.next_seed()
        if train and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ...autograd import gru_head_apply
            return gru_head_apply(self, pt, bt, gl_rows, lo_rows, off, B, W, total, mx, self.num_classes, self.p_drop, seed, want_preds)
        out = ops.gru_head_forward(pt, bt, gl_rows.contiguous().float(), lo_rows.contiguous().float(), off, B, W, total, mx, self.num_classes,
                                   train, self.p_drop, seed, self._ws, targets=targets, class_w=class_w, want_preds=want_preds)
        if train:
            torch._foreach_add_([self.bn_2.num_batches_tracked, self.bn_3.num_batches_tracked], 1)
        return out

    def forward(self, global_seq, local_feats, np_cluster):
        """Reference signature (pointnetAtt.py:232-250): global_seq [B, W, 256], local_feats [B, P, 64], np_cluster list of W sizes
        -> logits [B, C, P]."""
        B, W = global_seq.shape[0], global_seq.shape[1]
        logits, _, _ = self.forward_rows(global_seq.reshape(B * W, -1), local_feats.reshape(-1, local_feats.shape[2]), np_cluster, B)
        return logits

    def initHidden(self, x):
        return torch.zeros(1, x.shape[0], self.hidden_size, device=x.device)


class ClassificationFromGRU(nn.Module):
    """Parameter-compatible holder of pointnetAtt.py:261-279.  The reference's forward reads self.embed_dim, which its __init__ never
    sets, and no reference script reaches it (train_pointnetGRU.py:405-407 leaves the classification branch without logits): calling it
    raises AttributeError there; here it raises with the reason."""

    def __init__(self, num_classes=2, dropout=0.3, num_w=5, embed_dim=256, device='cuda'):
        super().__init__()
        self.conv_1 = _Conv(num_w, 1, True, device)
        self.fc_2 = _Linear(embed_dim, 128, True, device)
        self.fc_3 = _Linear(128, num_classes, True, device)
        self.bn_2 = _BN(128, device)

    def forward(self, x):
        raise AttributeError("'ClassificationFromGRU' object has no attribute 'embed_dim' (the reference's forward fails the same way, "
                             "pointnetAtt.py:274; the classification task is not runnable in the reference)")


class ClassificationWithAttention(nn.Module):
    """pointnetAtt.py:115-151 on the HIP path (ampnet_cls_head_fwd_f32 / _bwd_f32, csrc/cls_head.hip): MultiheadAttention over the window
    tokens, conv_1 = Conv1d(num_w -> 1, 1) over the attention output re-viewed as [B, W, E] (a view of the sequence-first tensor, as the
    reference writes it), fc_2 -> bn_2 -> ReLU -> fc_3.  forward(gl_feats [W, B, E], centroids, attn_mask) -> (out [B, C], weights
    [B, W, W]); centroids are unused by the reference (the positional encoding is commented out there)."""

    def __init__(self, embed_dim, num_heads, num_classes=2, dropout=0.3, num_w=9, device='cuda'):
        super().__init__()
        if embed_dim != P.GLOBAL_DIM or num_heads != P.HEADS or not (1 <= num_classes <= 16) or not (1 <= num_w <= 32):
            raise NotImplementedError("the HIP classification head is built for embed_dim=256, num_heads=8, num_classes<=16, num_w<=32")
        self.embed_dim = embed_dim
        self.num_classes, self.num_w = num_classes, num_w
        self.p_drop = float(dropout)
        self.attention = _MHA(embed_dim, device)
        self.conv_1 = _Conv(num_w, 1, True, device)
        self.fc_2 = _Linear(embed_dim, 128, True, device)
        self.fc_3 = _Linear(128, num_classes, True, device)
        self.bn_2 = _BN(128, device)
        self._cache = _TableCache()
        self._ws = ops.Workspace()
        self._step = 0
        self.seed = 0x3C6EF372

    def _param_table(self):
        return P.cls_head_params(self.num_classes, self.num_w)

    def _tables(self):
        return self._cache.get(self, self._param_table(), P.CLS_HEAD_BUFFERS, "ClassificationWithAttention")

    def forward(self, gl_feats, centroids=None, attn_mask=None):
        W, B = gl_feats.shape[0], gl_feats.shape[1]
        if W != self.num_w:
            raise _lib.AmpnetError(f"ClassificationWithAttention: {W} tokens per sample, conv_1 was built for num_w={self.num_w}")
        gl_rows = gl_feats.transpose(0, 1).reshape(B * W, self.embed_dim)
        pt, bt = self._tables()
        train = self.training
        seed = (self.seed + 0x632BE5AB * self._step) & 0xFFFFFFFF
        if train:
            self._step += 1
        if train and torch.is_grad_enabled() and (gl_feats.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ...autograd import cls_head_apply
            return cls_head_apply(self, pt, bt, gl_rows, attn_mask, B, W, self.num_classes, self.p_drop, seed)
        out = ops.cls_head_forward(pt, bt, gl_rows.contiguous().float(), attn_mask, B, W, self.num_classes, train, self.p_drop, seed, self._ws)
        if train:
            self.bn_2.num_batches_tracked += 1
        return out
