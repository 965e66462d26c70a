"""ORACLE -- test infrastructure, not product code.

CPU fp32 restatement (torch tensor algebra, no nn.Module, no HIP) of the AMP-Net per-window hot path
of marionacaros/3D-semantic-segmentation-AMP-Net.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file; the product package never does.

Pinned: tests/test_oracle_golden.py checks every function here against outputs of the reference
itself (tests/golden/*.npz, produced by tests/golden/make_golden.py which imports /root/reference).

Reference lines restated (paths relative to /root/reference):
  tnet()            pointNet/model/pointnetAtt.py:28-47   TransformationNet.forward
  encoder()         pointNet/model/pointnetAtt.py:80-112  BasePointNet.forward
  mha()             torch.nn.MultiheadAttention as configured at pointnetAtt.py:163-165,187-190
  head()            pointNet/model/pointnetAtt.py:176-209 SegmentationWithAttention.forward
  forward_windows() pointNet/self-attention/train_pointnet-attention.py:396-435 (window loop, mask, head call)
  loss_terms()      train_pointnet-attention.py:127,138,445,463-467
  adam_step()       torch.optim.Adam as configured at train_pointnet-attention.py:140-141
  iou / accuracy    utils/get_metrics.py:6-31, utils/utils.py:14-19

Parameters are plain dicts {state_dict key: tensor}; BN running buffers are dicts too and are
updated in place in train mode exactly like nn.BatchNorm1d (momentum 0.1, unbiased running var).
"""
import math
import numpy as np
import torch

EPS = 1e-5
MOM = 0.1


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def batchnorm_rows(z, gamma, beta, bufs, key, train):
    """BatchNorm1d over the rows of z [R, C] (for Conv1d inputs R = B*N, for Linear inputs R = B)."""
    if train:
        mean = z.mean(0)
        var = z.var(0, unbiased=False)
        if bufs is not None:
            r = z.shape[0]
            with torch.no_grad():
                unb = var * (r / max(r - 1, 1))
                bufs[key + "running_mean"].mul_(1 - MOM).add_(MOM * mean)
                bufs[key + "running_var"].mul_(1 - MOM).add_(MOM * unb)
    else:
        mean = bufs[key + "running_mean"]
        var = bufs[key + "running_var"]
    return (z - mean) / torch.sqrt(var + EPS) * gamma + beta


def _w2(p, key):
    w = p[key]
    return w.reshape(w.shape[0], -1)          # Conv1d k=1 weight [Co, Ci, 1] -> [Co, Ci]


def tnet(p, bufs, pre, x, train):
    """x [B, n, k] -> [B, k, k]   (pointnetAtt.py:28-47)."""
    B, n, k = x.shape
    h = x.reshape(B * n, k)
    for i, _ in ((1, 64), (2, 128), (3, 256)):
        h = h @ _w2(p, f"{pre}conv_{i}.weight").t()
        h = torch.relu(batchnorm_rows(h, p[f"{pre}bn_{i}.weight"], p[f"{pre}bn_{i}.bias"], bufs, f"{pre}bn_{i}.", train))
    g = h.reshape(B, n, 256).max(1).values                       # MaxPool1d(num_points)
    g = torch.relu(batchnorm_rows(g @ p[pre + "fc_1.weight"].t(), p[pre + "bn_4.weight"], p[pre + "bn_4.bias"], bufs, pre + "bn_4.", train))
    g = torch.relu(batchnorm_rows(g @ p[pre + "fc_2.weight"].t(), p[pre + "bn_5.weight"], p[pre + "bn_5.bias"], bufs, pre + "bn_5.", train))
    g = g @ p[pre + "fc_3.weight"].t() + p[pre + "fc_3.bias"]
    return g.reshape(B, k, k) + torch.eye(k, dtype=x.dtype)


def encoder(p, bufs, x, train, point_dim=3):
    """x [B, N, 9] -> (local [B, N, 64], global [B, 256], feature_transform [B, 64, 64])

    BasePointNet.forward with return_local_features=True (pointnetAtt.py:80-112); the reference returns
    cat([global.repeat(N), local]) which train_loop immediately slices apart again
    (train_pointnet-attention.py:411-413) -- the two pieces are returned directly here.
    """
    B, N, _ = x.shape
    xyz = x[:, :, :point_dim]
    t_in = tnet(p, bufs, "input_transform.", xyz, train)
    h = torch.cat([torch.bmm(xyz, t_in), x], dim=2).reshape(B * N, -1)        # 12 channels
    for i in (1, 2):
        h = torch.relu(batchnorm_rows(h @ _w2(p, f"conv_{i}.weight").t(), p[f"bn_{i}.weight"], p[f"bn_{i}.bias"], bufs, f"bn_{i}.", train))
    h = h.reshape(B, N, 64)
    t_feat = tnet(p, bufs, "feature_transform.", h, train)
    local = torch.bmm(h, t_feat)
    h = local.reshape(B * N, 64)
    for i in (3, 4, 5, 6):
        h = torch.relu(batchnorm_rows(h @ _w2(p, f"conv_{i}.weight").t(), p[f"bn_{i}.weight"], p[f"bn_{i}.bias"], bufs, f"bn_{i}.", train))
    glob = h.reshape(B, N, -1).max(1).values
    return local, glob, t_feat


def mha(p, x, key_padding_mask, heads, drop_mask=None, drop_p=0.0):
    """Self-attention, sequence-first x [L, B, E]; key_padding_mask [B, L] bool (True = ignore).
    drop_mask: optional keep-mask [B*heads, L, L] applied to the softmax output, scaled by 1/(1-p)."""
    L, B, E = x.shape
    d = E // heads
    qkv = x.reshape(L * B, E) @ p["attention.in_proj_weight"].t() + p["attention.in_proj_bias"]
    q, k, v = qkv.split(E, dim=1)

    def heads_first(t):                                   # [L*B, E] -> [B*heads, L, d]
        return t.reshape(L, B * heads, d).transpose(0, 1)

    q, k, v = heads_first(q) * (1.0 / math.sqrt(d)), heads_first(k), heads_first(v)
    s = torch.bmm(q, k.transpose(1, 2))                   # [B*heads, L, L]
    if key_padding_mask is not None:
        m = key_padding_mask.reshape(B, 1, 1, L).expand(B, heads, L, L).reshape(B * heads, L, L)
        s = s.masked_fill(m, float("-inf"))
    a = torch.softmax(s, dim=-1)
    if drop_mask is not None:
        a = a * drop_mask * (1.0 / (1.0 - drop_p))
    o = torch.bmm(a, v).transpose(0, 1).reshape(L * B, E)
    o = o @ p["attention.out_proj.weight"].t() + p["attention.out_proj.bias"]
    return o.reshape(L, B, E)


def head(p, bufs, gl, lo, centroids, np_cluster, mask, train, heads=8,
         drop_p=0.0, drop_masks=None):
    """SegmentationWithAttention.forward (pointnetAtt.py:176-209).

    gl [W, B, 256], lo [B, sum(np_cluster), 64], centroids [B, W, 2], mask [B, W] bool or None
    -> logits [B, C, sum(np_cluster)].
    drop_masks: None, or dict with keep-masks 'att' [B*heads, W, W], 'd2' [B, 128, P], 'd3' [B, 64, P]."""
    W, B, E = gl.shape
    pos = torch.nn.functional.leaky_relu(centroids @ p["fc1.weight"].t() + p["fc1.bias"], 0.01)
    pos = pos @ p["fc2.weight"].t() + p["fc2.bias"]        # [B, W, E]
    tok = gl + pos.transpose(0, 1)
    tok = mha(p, tok, mask, heads, None if drop_masks is None else drop_masks.get("att"), drop_p)
    rep = torch.cat([tok[i].unsqueeze(1).expand(B, int(np_cluster[i]), E) for i in range(W)], dim=1)
    emb = torch.cat([lo, rep], dim=2)                      # [B, P, 320]
    P = emb.shape[1]
    h = emb.reshape(B * P, -1) @ _w2(p, "conv_2.weight").t() + p["conv_2.bias"]
    h = torch.relu(batchnorm_rows(h, p["bn_2.weight"], p["bn_2.bias"], bufs, "bn_2.", train))
    if drop_masks is not None:
        h = h * drop_masks["d2"].transpose(1, 2).reshape(B * P, -1) * (1.0 / (1.0 - drop_p))
    h = h @ _w2(p, "conv_3.weight").t() + p["conv_3.bias"]
    h = torch.relu(batchnorm_rows(h, p["bn_3.weight"], p["bn_3.bias"], bufs, "bn_3.", train))
    if drop_masks is not None:
        h = h * drop_masks["d3"].transpose(1, 2).reshape(B * P, -1) * (1.0 / (1.0 - drop_p))
    h = h @ _w2(p, "conv_4.weight").t() + p["conv_4.bias"]
    return h.reshape(B, P, -1).transpose(1, 2)             # [B, C, P]


def forward_windows(enc_p, enc_b, head_p, head_b, pc, targets, centroids, train_enc, train_head,
                    drop_p=0.0, drop_masks=None):
    """The deterministic core of train_loop (train_pointnet-attention.py:396-435): no shuffles, no rotation.

    pc [B, N, 9, W], targets [B, N, W] (-1 = padded), centroids [B, W, 2]
    -> logits [B, C, W*N], targets_pc [B, W*N], feature_transform of the LAST window [B, 64, 64]."""
    B, N, _, W = pc.shape
    lo, gl, tg, t_feat = [], [], [], None
    for w in range(W):                                      # the encoder sees one window slot at a time
        l, g, t_feat = encoder(enc_p, enc_b, pc[:, :, :, w], train_enc)
        lo.append(l)
        gl.append(g)
        tg.append(targets[:, :, w])
    lo = torch.cat(lo, dim=1)
    gl = torch.stack(gl, dim=0)                             # [W, B, 256]
    targets_pc = torch.cat(tg, dim=1)
    # NOTE the reference reshapes with view(B, -1, W) on the concatenated [B, W*N] targets
    # (train_pointnet-attention.py:428-431): column j of that view is NOT cluster j.  Restated literally.
    tm = targets_pc.reshape(B, -1, W)
    mask = (tm == -1).all(1)                                # [B, W]
    logits = head(head_p, head_b, gl, lo, centroids, [N] * W, mask, train_head, drop_p=drop_p, drop_masks=drop_masks)
    return logits, targets_pc, t_feat, mask


def gru_cell_sequence(p, x):
    """nn.GRU(256 -> H, num_layers=1, batch_first=True) with h0 = 0 (pointnetAtt.py:219,234-235; gate order r, z, n as torch
    documents it): x [B, L, 256] -> all hidden states [B, L, H]."""
    wih, whh = p["gru_global.weight_ih_l0"], p["gru_global.weight_hh_l0"]
    bih, bhh = p["gru_global.bias_ih_l0"], p["gru_global.bias_hh_l0"]
    B, L, _ = x.shape
    H = whh.shape[1]
    h = torch.zeros(B, H, dtype=x.dtype)
    out = []
    for t in range(L):
        gi = x[:, t] @ wih.t() + bih
        gh = h @ whh.t() + bhh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        out.append(h)
    return torch.stack(out, dim=1)


def gru_head(p, bufs, global_seq, lo, np_cluster, train, drop_p=0.0, drop_masks=None):
    """SegmentationWithGRU.forward (pointnetAtt.py:232-250): global_seq [B, W, 256], lo [B, sum(np_cluster), 64]
    -> logits [B, C, sum(np_cluster)].  drop_masks: None or dict with keep-masks 'd2' [B, 128, P], 'd3' [B, 64, P]."""
    B, W, _ = global_seq.shape
    hs = gru_cell_sequence(p, global_seq)                       # [B, W, H]
    rep = torch.cat([hs[:, i].unsqueeze(1).expand(B, int(np_cluster[i]), hs.shape[2]) for i in range(W)], dim=1)
    emb = torch.cat([lo, rep], dim=2)                           # [B, P, 128]
    P = emb.shape[1]
    h = emb.reshape(B * P, -1) @ _w2(p, "conv_2.weight").t() + p["conv_2.bias"]
    h = torch.relu(batchnorm_rows(h, p["bn_2.weight"], p["bn_2.bias"], bufs, "bn_2.", train))
    if drop_masks is not None:
        h = h * drop_masks["d2"].transpose(1, 2).reshape(B * P, -1) * (1.0 / (1.0 - drop_p))
    h = h @ _w2(p, "conv_3.weight").t() + p["conv_3.bias"]
    h = torch.relu(batchnorm_rows(h, p["bn_3.weight"], p["bn_3.bias"], bufs, "bn_3.", train))
    if drop_masks is not None:
        h = h * drop_masks["d3"].transpose(1, 2).reshape(B * P, -1) * (1.0 / (1.0 - drop_p))
    h = h @ _w2(p, "conv_4.weight").t() + p["conv_4.bias"]
    return h.reshape(B, P, -1).transpose(1, 2)


def forward_windows_gru(enc_p, enc_b, head_p, head_b, pc, targets, train_enc, train_head, drop_p=0.0, drop_masks=None):
    """The model part of the GRU train_loop (pointNet/rnn/train_pointnetGRU.py:385-403): pc [B, N, 9, W], targets [B, N, W]
    -> logits [B, C, W*N], targets_pc [B, W*N], feature_transform of the LAST window."""
    B, N, _, W = pc.shape
    lo, gl, tg, t_feat = [], [], [], None
    for w in range(W):
        l, g, t_feat = encoder(enc_p, enc_b, pc[:, :, :, w], train_enc)
        lo.append(l)
        gl.append(g)
        tg.append(targets[:, :, w])
    logits = gru_head(head_p, head_b, torch.stack(gl, dim=1), torch.cat(lo, dim=1), [N] * W, train_head, drop_p=drop_p, drop_masks=drop_masks)
    return logits, torch.cat(tg, dim=1), t_feat


def cls_head(p, bufs, gl, mask, train, heads=8, drop_p=0.0, drop_mask=None):
    """ClassificationWithAttention.forward (pointnetAtt.py:134-151): gl [W, B, 256], mask [B, W] bool or None
    -> (out [B, C], attention weights [B, W, W] = mean over the heads of the (dropped) probabilities)."""
    W, B, E = gl.shape
    d = E // heads
    qkv = gl.reshape(W * B, E) @ p["attention.in_proj_weight"].t() + p["attention.in_proj_bias"]
    q, k, v = qkv.split(E, dim=1)
    hf = lambda t: t.reshape(W, B * heads, d).transpose(0, 1)            # noqa: E731
    q, k, v = hf(q) * (1.0 / math.sqrt(d)), hf(k), hf(v)
    s = torch.bmm(q, k.transpose(1, 2))
    if mask is not None:
        s = s.masked_fill(mask.reshape(B, 1, 1, W).expand(B, heads, W, W).reshape(B * heads, W, W), float("-inf"))
    a = torch.softmax(s, dim=-1)
    if drop_mask is not None:
        a = a * drop_mask * (1.0 / (1.0 - drop_p))
    o = torch.bmm(a, v).transpose(0, 1).reshape(W * B, E) @ p["attention.out_proj.weight"].t() + p["attention.out_proj.bias"]
    o = o.reshape(W, B, E)
    x = o.reshape(-1, W, E)                                   # the reference's .view(-1, W, E): a re-interpretation, not a transpose
    x = torch.relu((x * p["conv_1.weight"].reshape(1, W, 1)).sum(1) + p["conv_1.bias"])      # Conv1d(W -> 1, 1): [B, E]
    h = x @ p["fc_2.weight"].t() + p["fc_2.bias"]
    h = torch.relu(batchnorm_rows(h, p["bn_2.weight"], p["bn_2.bias"], bufs, "bn_2.", train))
    return h @ p["fc_3.weight"].t() + p["fc_3.bias"], a.reshape(B, heads, W, W).mean(1)


def loss_terms(logits, targets_pc, t_feat, class_w=(1.0, 2.0, 2.0, 1.0, 1.0)):
    """(ce, reg): weighted CE with ignore_index -1, mean over non-ignored weights; Frobenius norm of
    I - F F^T over the whole [B, 64, 64] tensor (train_pointnet-attention.py:127,138,445,463-464)."""
    B, C, P = logits.shape
    lg = logits.transpose(1, 2).reshape(B * P, C)
    t = targets_pc.reshape(-1)
    keep = t != -1
    lsm = lg - torch.logsumexp(lg, dim=1, keepdim=True)
    w = torch.tensor(class_w, dtype=lg.dtype)
    tk = t.clamp(min=0)
    wi = w[tk] * keep
    nll = -lsm.gather(1, tk[:, None])[:, 0]
    ce = (wi * nll).sum() / wi.sum()
    eye = torch.eye(t_feat.shape[-1], dtype=t_feat.dtype)
    reg = torch.sqrt(((eye - torch.bmm(t_feat, t_feat.transpose(1, 2))) ** 2).sum())
    return ce, reg


def adam_step(param, grad, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update (no weight decay, no amsgrad); step counts from 1. In place."""
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------------------
# metrics (utils/get_metrics.py:6-31, utils/utils.py:14-19)
# ----------------------------------------------------------------------------------------------
def rm_padding(preds, targets):
    keep = targets != -1
    return preds[keep], targets[keep], keep


def iou_obj(preds, targets, label):
    """TP / (GT_pos + FP) for one label over flat int vectors; NaN when the denominator is 0.
    The reference's quotient is a float32 division (an int64 torch tensor over a numpy integer,
    get_metrics.py:12-14), restated as such."""
    preds = np.asarray(preds).reshape(-1)
    targets = np.asarray(targets).reshape(-1)
    det = preds == label
    tp = np.logical_and(det, preds == targets).sum()
    fp = det.sum() - tp
    gt = (targets == label).sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float32(tp) / np.float32(gt + fp))


def accuracy(preds, targets):
    preds = np.asarray(preds).reshape(-1)
    targets = np.asarray(targets).reshape(-1)
    return float(np.float32((preds == targets).sum()) / np.float32(len(preds)))     # float32 quotient, as above


def predictions(logits):
    """argmax over classes of log_softmax(logits) (train_pointnet-attention.py:449-450); first max wins."""
    return torch.log_softmax(logits, dim=1).max(1)[1]


# ----------------------------------------------------------------------------------------------
# dropout keep-masks: the counter-based generator the HIP kernels use, restated bit for bit
# ----------------------------------------------------------------------------------------------
def _mix32(x):
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    return (x ^ (x >> np.uint64(16))) & np.uint64(0xFFFFFFFF)


def keep_mask(seed, stream, n, p):
    """keep[i] = hash32(i ^ hash32(seed + stream * 0x9E3779B9)) >= p * 2^32  (lowbias32 mixer)."""
    base = _mix32(np.array([(seed + stream * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint64))[0]
    idx = np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    h = _mix32(idx ^ base)
    # the probability crosses the C ABI as a float (include/ampnet_hip.h: `float drop_p`), so the threshold is that of the
    # float32-rounded value: float(0.3f) * 2^32 and 0.3 * 2^32 differ by 51 -- one element in 8e7 (found at B = 32, N = 2048)
    thr = np.uint64(min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF))
    return (h >= thr)
