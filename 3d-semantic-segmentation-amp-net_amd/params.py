"""Parameter / buffer inventory of the AMP-Net hot path and its flat-buffer layout.

The names are the reference's `state_dict` keys verbatim (reference modules
pointNet/model/pointnetAtt.py:9-26 TransformationNet, :52-78 BasePointNet, :155-174
SegmentationWithAttention) so reference checkpoints load into the drop-in modules and back.

Order matters: index i in ENC_PARAMS / HEAD_PARAMS is the index the C ABI uses in its pointer
tables (include/ampnet_hip.h, enum ampnet_enc_param / ampnet_head_param).  tests/test_abi.py checks
this table against the names and sizes exported by the library itself.
"""
from collections import OrderedDict

POINT_DIM = 3          # T-Net on x,y,z        (train_pointnet-attention.py:110-113)
N_FEATS = 9            # features per point    (datasets.py:359)
GLOBAL_DIM = 256       # GLOBAL_FEAT_SIZE      (train_pointnet-attention.py:26)
LOCAL_DIM = 64
HEADS = 8              # ATT_HEADS             (train_pointnet-attention.py:25)
NUM_CLASSES = 5
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _tnet(prefix, k):
    p = [
        (prefix + "conv_1.weight", (64, k, 1)),
        (prefix + "conv_2.weight", (128, 64, 1)),
        (prefix + "conv_3.weight", (256, 128, 1)),
        (prefix + "bn_1.weight", (64,)), (prefix + "bn_1.bias", (64,)),
        (prefix + "bn_2.weight", (128,)), (prefix + "bn_2.bias", (128,)),
        (prefix + "bn_3.weight", (256,)), (prefix + "bn_3.bias", (256,)),
        (prefix + "bn_4.weight", (256,)), (prefix + "bn_4.bias", (256,)),
        (prefix + "bn_5.weight", (128,)), (prefix + "bn_5.bias", (128,)),
        (prefix + "fc_1.weight", (256, 256)),
        (prefix + "fc_2.weight", (128, 256)),
        (prefix + "fc_3.weight", (k * k, 128)),
        (prefix + "fc_3.bias", (k * k,)),
    ]
    return p


def _bn_bufs(prefix, c):
    return [(prefix + "running_mean", (c,)), (prefix + "running_var", (c,))]


ENC_PARAMS = OrderedDict(
    _tnet("input_transform.", POINT_DIM)
    + _tnet("feature_transform.", 64)
    + [
        ("conv_1.weight", (64, N_FEATS + POINT_DIM, 1)),
        ("conv_2.weight", (64, 64, 1)),
        ("conv_3.weight", (64, 64, 1)),
        ("conv_4.weight", (128, 64, 1)),
        ("conv_5.weight", (128, 128, 1)),
        ("conv_6.weight", (GLOBAL_DIM, 128, 1)),
        ("bn_1.weight", (64,)), ("bn_1.bias", (64,)),
        ("bn_2.weight", (64,)), ("bn_2.bias", (64,)),
        ("bn_3.weight", (64,)), ("bn_3.bias", (64,)),
        ("bn_4.weight", (128,)), ("bn_4.bias", (128,)),
        ("bn_5.weight", (128,)), ("bn_5.bias", (128,)),
        ("bn_6.weight", (GLOBAL_DIM,)), ("bn_6.bias", (GLOBAL_DIM,)),
    ]
)

_TNET_BN = [("bn_1.", 64), ("bn_2.", 128), ("bn_3.", 256), ("bn_4.", 256), ("bn_5.", 128)]
ENC_BUFFERS = OrderedDict(
    sum([_bn_bufs("input_transform." + n, c) for n, c in _TNET_BN], [])
    + sum([_bn_bufs("feature_transform." + n, c) for n, c in _TNET_BN], [])
    + sum([_bn_bufs(n, c) for n, c in
           [("bn_1.", 64), ("bn_2.", 64), ("bn_3.", 64), ("bn_4.", 128), ("bn_5.", 128), ("bn_6.", GLOBAL_DIM)]], [])
)

HEAD_PARAMS = OrderedDict([
    ("fc1.weight", (16, 2)), ("fc1.bias", (16,)),
    ("fc2.weight", (GLOBAL_DIM, 16)), ("fc2.bias", (GLOBAL_DIM,)),
    ("attention.in_proj_weight", (3 * GLOBAL_DIM, GLOBAL_DIM)),
    ("attention.in_proj_bias", (3 * GLOBAL_DIM,)),
    ("attention.out_proj.weight", (GLOBAL_DIM, GLOBAL_DIM)),
    ("attention.out_proj.bias", (GLOBAL_DIM,)),
    ("conv_2.weight", (128, LOCAL_DIM + GLOBAL_DIM, 1)), ("conv_2.bias", (128,)),
    ("conv_3.weight", (64, 128, 1)), ("conv_3.bias", (64,)),
    ("conv_4.weight", (NUM_CLASSES, 64, 1)), ("conv_4.bias", (NUM_CLASSES,)),
    ("bn_2.weight", (128,)), ("bn_2.bias", (128,)),
    ("bn_3.weight", (64,)), ("bn_3.bias", (64,)),
])

HEAD_BUFFERS = OrderedDict(_bn_bufs("bn_2.", 128) + _bn_bufs("bn_3.", 64))

# SegmentationWithGRU (pointnetAtt.py:214-230; HIDDEN_SIZE = 64, pointNet/rnn/train_pointnetGRU.py:27-28): state_dict order
GRU_HIDDEN = 64
GRU_HEAD_PARAMS = OrderedDict([
    ("gru_global.weight_ih_l0", (3 * GRU_HIDDEN, GLOBAL_DIM)),
    ("gru_global.weight_hh_l0", (3 * GRU_HIDDEN, GRU_HIDDEN)),
    ("gru_global.bias_ih_l0", (3 * GRU_HIDDEN,)),
    ("gru_global.bias_hh_l0", (3 * GRU_HIDDEN,)),
    ("conv_2.weight", (128, LOCAL_DIM + GRU_HIDDEN, 1)), ("conv_2.bias", (128,)),
    ("conv_3.weight", (64, 128, 1)), ("conv_3.bias", (64,)),
    ("conv_4.weight", (NUM_CLASSES, 64, 1)), ("conv_4.bias", (NUM_CLASSES,)),
    ("bn_2.weight", (128,)), ("bn_2.bias", (128,)),
    ("bn_3.weight", (64,)), ("bn_3.bias", (64,)),
])


def cls_head_params(num_classes=2, num_w=9):
    """ClassificationWithAttention (pointnetAtt.py:115-133), state_dict order."""
    return OrderedDict([
        ("attention.in_proj_weight", (3 * GLOBAL_DIM, GLOBAL_DIM)), ("attention.in_proj_bias", (3 * GLOBAL_DIM,)),
        ("attention.out_proj.weight", (GLOBAL_DIM, GLOBAL_DIM)), ("attention.out_proj.bias", (GLOBAL_DIM,)),
        ("conv_1.weight", (1, num_w, 1)), ("conv_1.bias", (1,)),
        ("fc_2.weight", (128, GLOBAL_DIM)), ("fc_2.bias", (128,)),
        ("fc_3.weight", (num_classes, 128)), ("fc_3.bias", (num_classes,)),
        ("bn_2.weight", (128,)), ("bn_2.bias", (128,)),
    ])


CLS_HEAD_BUFFERS = OrderedDict(_bn_bufs("bn_2.", 128))


def numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def offsets(table, align=64):
    """name -> (offset, numel) in a flat float32 buffer; every tensor starts `align` floats aligned
    (256 B) so kernels may use 16-byte vector loads on any parameter."""
    out, off = OrderedDict(), 0
    for name, shape in table.items():
        out[name] = (off, numel(shape))
        off += (numel(shape) + align - 1) // align * align
    return out, off


def n_params(table):
    return sum(numel(s) for s in table.values())


assert n_params(ENC_PARAMS) == 883401, n_params(ENC_PARAMS)
assert n_params(HEAD_PARAMS) == 317621, n_params(HEAD_PARAMS)
