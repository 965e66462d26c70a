"""Regenerates tests/golden/*.npz by RUNNING THE REFERENCE (read-only at /root/reference).

Run only in the build container:   python tests/golden/make_golden.py [section ...]
The GPU box has no /root/reference; tests there use the committed .npz files.

The fixtures hold OUTPUTS of the reference (and the few random draws it made); inputs and weights
are regenerated from (seed, shape) by the package's `synthetic` module, so nothing of the reference
(source, weights it shipped, data) is stored.  Absent third-party packages that the hot path never
calls are replaced by empty stub modules in sys.modules; no reference file
is modified or copied.
"""
import importlib
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import baseline_state, cls_sample_array   # noqa: E402

pkg = importlib.import_module("3d-semantic-segmentation-amp-net_amd")
synth = importlib.import_module("3d-semantic-segmentation-amp-net_amd.synthetic")
P = importlib.import_module("3d-semantic-segmentation-amp-net_amd.params")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return lambda *a, **k: None

    _stub("pointNet_2")
    _stub("pointNet_2.models")
    _stub("pointNet_2.models.pointnet2_utils", PointNetSetAbstraction=_Dummy, PointNetFeaturePropagation=_Dummy)
    _stub("k_means_constrained", KMeansConstrained=_Dummy)
    _stub("progressbar", progressbar=lambda x, *a, **k: x)
    _stub("laspy")
    _stub("prettytable", PrettyTable=_Dummy)
    _stub("torchsummary", summary=lambda *a, **k: None)      # imported by pointNet/rnn/train_pointnetGRU.py:7, never called on the path
    if "torch.utils.tensorboard" not in sys.modules:
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            _stub("torch.utils.tensorboard", SummaryWriter=_Dummy)


def load_script(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = [path]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def ref_models(enc_seed, head_seed, dropout=0.3):
    from pointNet.model.pointnetAtt import BasePointNet, SegmentationWithAttention
    enc = BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cpu")
    att = SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, dropout=dropout, device="cpu")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(enc_seed, P.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(enc_seed, P.ENC_BUFFERS).items()})
    missing = enc.load_state_dict(sd, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    assert not missing.unexpected_keys
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(head_seed, P.HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(head_seed, P.HEAD_BUFFERS).items()})
    missing = att.load_state_dict(sd, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    assert not missing.unexpected_keys
    return enc, att


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------
def sec_fps():
    """a1: utils/utils.py:889-933.  The reference returns rows, not indices: column 3 carries the row id."""
    from utils.utils import fps
    out = {}
    cases = [("small", 11, 1000, 100), ("dup", 12, 512, 300), ("c5", 13, 8192, 4096), ("tiny", 14, 64, 64)]
    for name, seed, n, s in cases:
        xyz = synth.clouds(seed, 1, n)[0]
        if name == "dup":
            xyz[n // 2:] = xyz[: n // 2]            # every point twice -> distance ties and zeros
        pc = np.concatenate([xyz, np.arange(n, dtype=np.float32)[:, None]], axis=1)
        got = fps(pc, s)
        out[f"{name}_meta"] = np.array([seed, n, s], dtype=np.int64)
        out[f"{name}_idx"] = got[:, 3].astype(np.int64)
    save("fps", **out)


def sec_encoder():
    """a2/a3: pointnetAtt.py:28-47, 80-112 -- eval forward and one train-mode forward."""
    enc, _ = ref_models(1, 2)
    x = torch.from_numpy(synth.windows(21, 2, 256))
    enc.eval()
    with torch.no_grad():
        out, ft = enc(x)
        t_in = enc.input_transform(x[:, :, :3])
    res = dict(eval_local=out[:, :, -64:], eval_global=out[:, 0, :-64], eval_feat_T=ft, eval_in_T=t_in)
    # train mode: batch statistics + running-stat update
    xt = torch.from_numpy(synth.windows(22, 4, 128))
    enc.train()
    with torch.no_grad():
        out, ft = enc(xt)
    res.update(train_local=out[:, :, -64:], train_global=out[:, 0, :-64], train_feat_T=ft)
    sd = enc.state_dict()
    for k in ["bn_1.running_mean", "bn_1.running_var", "bn_6.running_var", "input_transform.bn_4.running_mean",
              "input_transform.bn_4.running_var", "feature_transform.bn_3.running_var",
              "feature_transform.bn_5.running_mean", "bn_6.num_batches_tracked"]:
        res["train_" + k] = sd[k]
    save("encoder", **res)


def sec_head():
    """a4: pointnetAtt.py:176-209 eval forward: uniform clusters with one padded cluster, ragged clusters, no mask."""
    _, att = ref_models(1, 2)
    att.eval()
    gl = torch.from_numpy(synth.uniform(31, (3, 2, 256), 0.0, 2.0))
    lo = torch.from_numpy(synth.uniform(32, (2, 768, 64), -1.0, 1.0))
    cent = torch.from_numpy(synth.uniform(33, (2, 3, 2), -1.0, 1.0))
    mask = torch.tensor([[False, False, True], [False, False, False]])
    with torch.no_grad():
        a, _ = att(gl, lo, cent, [256, 256, 256], mask)
        b, _ = att(gl, lo, cent, [100, 300, 368], None)
    save("head", uniform_masked=a, ragged_nomask=b)


# B = 16: the T-Net FC BatchNorms normalise over the B rows of one window slot; with B = 3 the reference's own
# fp32 gradients sit 4 % from an fp64 evaluation of the same graph (ill-conditioned), useless as a pin.
STEP_B, STEP_N, STEP_W = 16, 64, 3
STEP_WREAL = [3, 2, 3, 1, 2, 3, 3, 2, 1, 3, 2, 3, 3, 1, 2, 3]


def sec_step():
    """a5/a6/a7: train_pointnet-attention.py:337-475 train_loop itself, train=False then train=True twice.

    Dropout is constructed with p=0 (a constructor argument of the reference head, pointnetAtt.py:155) so
    that the train-mode step is deterministic; the numpy RNG the augmentations draw from is seeded and the
    draws are recorded (cluster permutation, angle, point permutations)."""
    tr = load_script(os.path.join(REF, "pointNet/self-attention/train_pointnet-attention.py"), "ref_train_att")
    enc, att = ref_models(3, 4, dropout=0.0)
    B, N, W = STEP_B, STEP_N, STEP_W
    pc, tg, cent, w_real = synth.sample_batch(41, B, N, max_w=W, w_real=STEP_WREAL)
    names = [f"f{i}" for i in range(B)]
    data = (torch.from_numpy(pc), torch.from_numpy(tg), names, torch.from_numpy(cent))
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    opt_p = torch.optim.Adam(enc.parameters(), lr=1e-3)
    opt_a = torch.optim.Adam(att.parameters(), lr=1e-3)
    res = dict(meta=np.array([B, N, W], dtype=np.int64), w_real=w_real)

    np.random.seed(777)
    with torch.no_grad():
        m, tpc, preds, _ = tr.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", False, 0, 0)
    res.update(eval_ce=m["ce_loss"].detach().numpy(), eval_reg=m["reg_loss"].detach().numpy(),
               eval_loss=m["loss"].detach().numpy(), eval_targets=tpc.numpy(), eval_preds=preds.numpy())

    for step in (1, 2):
        np.random.seed(1000 + step)
        data = (torch.from_numpy(pc.copy()), torch.from_numpy(tg.copy()), names, torch.from_numpy(cent))
        m, tpc, preds, _ = tr.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
        res[f"s{step}_ce"] = m["ce_loss"].detach().numpy()
        res[f"s{step}_reg"] = m["reg_loss"].detach().numpy()
        res[f"s{step}_loss"] = m["loss"].detach().numpy()
        res[f"s{step}_preds"] = preds.numpy()
        res[f"s{step}_targets"] = tpc.numpy()
        for tag, mod in (("enc", enc), ("att", att)):
            for k, p in mod.named_parameters():
                g = p.grad.detach().double()
                res[f"s{step}_{tag}_gnorm/{k}"] = np.array([g.norm().item(), g.sum().item()])
                if p.numel() <= 4096:
                    res[f"s{step}_{tag}_grad/{k}"] = p.grad.detach().numpy()
                res[f"s{step}_{tag}_psum/{k}"] = np.array([p.detach().double().sum().item(),
                                                           p.detach().double().abs().sum().item()])
    for tag, mod in (("enc", enc), ("att", att)):
        for k, v in mod.state_dict().items():
            if "running" in k or "num_batches" in k:
                res[f"final_{tag}_buf/{k}"] = v.numpy()
    # small parameters in full after two Adam steps
    for k in ["input_transform.fc_3.bias", "bn_6.weight", "conv_1.weight"]:
        res[f"final_enc_param/{k}"] = enc.state_dict()[k].numpy()
    for k in ["conv_4.weight", "fc1.weight", "attention.out_proj.bias"]:
        res[f"final_att_param/{k}"] = att.state_dict()[k].numpy()
    save("step", **res)


def sec_metrics():
    """a11: utils/get_metrics.py:6-31, utils/utils.py:14-19."""
    from utils.get_metrics import get_iou_obj, get_accuracy
    from utils.utils import rm_padding
    preds = torch.from_numpy(synth.randint(51, (4000,), 0, 5))
    tgt = synth.randint(52, (4000,), 0, 5)
    tgt[synth.uniform01(53, (4000,)) < 0.2] = -1
    tgt = torch.from_numpy(tgt)
    p2, t2, keep = rm_padding(preds, tgt)
    ious = [get_iou_obj(p2, t2, c) for c in range(5)]
    acc = get_accuracy(p2, t2, {}, "segmentation", None)["accuracy"]
    # a class absent from both preds and targets -> nan
    p3 = torch.tensor([0, 0, 1, 1]); t3 = torch.tensor([0, 1, 1, 1])
    with np.errstate(all="ignore"):
        absent = get_iou_obj(p3, t3, 4)
    save("metrics", ious=np.array(ious), acc=np.array(acc), n_keep=np.array(int(keep.sum())), absent=np.array(absent))


def sec_collate():
    """a9: pointNet/collate_fns.py:4-55 with the python/torch RNGs seeded."""
    from pointNet.collate_fns import collate_seq_padd
    res = {}
    specs = [(61, 2048, 1), (62, 2048, 3), (63, 1500, 5), (64, 3000, 9), (65, 2048, 9)]
    batch = []
    for seed, n, w in specs:
        win = synth.windows(seed, w, n)                               # [w, n, 9]
        pc = np.ascontiguousarray(win.transpose(1, 2, 0))             # [n, 9, w]
        lab = synth.labels_for(win, seed).transpose(1, 0).copy()     # [n, w]
        cent = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], 0).astype(np.float32)   # [2, w]
        batch.append((pc, lab, f"f{seed}", cent))
    random.seed(5); torch.manual_seed(5)
    data, tg, names, cents = collate_seq_padd(batch)
    res["data_shape"] = np.array(data.shape); res["tg_shape"] = np.array(tg.shape)
    res["cents"] = cents.numpy()
    res["data_sum"] = data.double().sum(dim=(1, 2)).numpy()          # [B, 9] per-cluster sums
    res["data_probe"] = data[:, ::97, :, :].numpy()
    res["tg_probe"] = tg[:, ::97, :].numpy()
    res["tg_sum"] = tg.sum(dim=1).numpy()
    save("collate", **res)


def sec_dataset():
    """a10: pointNet/datasets.py:295-460 LidarKmeansDataset on a synthetic kmeans_<name>.pt."""
    import tempfile
    from pointNet.datasets import LidarKmeansDataset
    ds_mod = importlib.import_module("3d-semantic-segmentation-amp-net_amd.synthetic")
    with tempfile.TemporaryDirectory() as d:
        raw = ds_mod.kmeans_file_tensor(71, 96, 3)
        torch.save(torch.from_numpy(raw), os.path.join(d, "kmeans_tile71.pt"))
        ds = LidarKmeansDataset(d, task="segmentation", number_of_points=2048, files=["tile71.pt"],
                                fixed_num_points=True, c_sample=False, sort_kmeans=False, get_centroids=True)
        pc, lab, fn, cent = ds[0]
    save("dataset", pc=np.asarray(pc), labels=lab.numpy(), centroids=np.asarray(cent))


def sec_baseline():
    """a12 / config 1: pointNet/model/pointnet.py:128-154 SegmentationPointNet(5, point_dimension=3) and
    pointNet/model/light_pointnet_256.py:128-153 SegmentationPointNet(5, point_dimension=2, device='cpu'), eval, [4,512,9]."""
    from pointNet.model.pointnet import SegmentationPointNet
    from pointNet.model.light_pointnet_256 import SegmentationPointNet as LightSeg
    for tag, net, base in (("baseline", SegmentationPointNet(num_classes=5, point_dimension=3), 9000),
                           ("baseline_light", LightSeg(num_classes=5, point_dimension=2, device="cpu"), 9500)):
        table = {k: tuple(v.shape) for k, v in net.state_dict().items() if "num_batches" not in k}
        sd = {k: torch.from_numpy(v) for k, v in baseline_state(synth, table, base).items()}
        net.load_state_dict(sd, strict=False)
        net.eval()
        x = torch.from_numpy(synth.windows(81, 4, 512))
        with torch.no_grad():
            logits, ft = net(x)
        names = np.array(list(table.keys()))
        shapes = np.array([";".join(map(str, s)) for s in table.values()])
        save(tag, logits=logits.numpy(), feat_T=ft.numpy(), names=names, shapes=shapes, seed_base=np.array([base]))


def _baseline_train_section(tr, mk, base, Bn, N, tag):
    """Two steps of the reference's baseline train_loop on a seeded [Bn, N, 9] batch -> fixture `tag` (see sec_baseline_train).
    The same two steps are then repeated with torch's default dtype set to float64 (the reference's code unchanged: its
    `torch.Tensor(pc)`, `torch.eye` and the module's parameters all follow the default dtype), and the float64 loss terms, gradient norms,
    small gradients and running statistics are stored next to the float32 ones (suffix 64): their distance is the reference's OWN float32
    noise, the yardstick for what any float32 implementation can be held to at the second step."""
    res = {}
    _baseline_two_steps(tr, mk, base, Bn, N, res, "")
    torch.set_default_dtype(torch.float64)
    try:
        _baseline_two_steps(tr, mk, base, Bn, N, res, "64")
    finally:
        torch.set_default_dtype(torch.float32)
    save(tag, seed_base=np.array([base]), shape=np.array([Bn, N]), **res)


def _baseline_two_steps(tr, mk, base, Bn, N, res, sfx):
    net = mk()
    table = {k: tuple(v.shape) for k, v in net.state_dict().items() if "num_batches" not in k}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in baseline_state(synth, table, base).items()}, strict=False)
    x = synth.windows(83, Bn, N)
    t = synth.labels_for(x, 83)
    t[0, :40] = -1
    ce = torch.nn.CrossEntropyLoss(weight=torch.tensor([1., 2., 2., 1., 1.]), reduction="mean", ignore_index=-1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    full = sfx == ""
    for step in (1, 2):
        np.random.seed(2100 + step)
        data = (torch.from_numpy(x.copy()), torch.from_numpy(t.copy()), ["f"] * Bn)
        m, tpc, preds, _ = tr.train_loop(data, opt, ce, net, None, True, 0, 0)
        res[f"s{step}_ce{sfx}"] = m["ce_loss"].detach().numpy()
        res[f"s{step}_reg{sfx}"] = m["reg_loss"].detach().numpy()
        res[f"s{step}_loss{sfx}"] = m["loss"].detach().numpy()
        res[f"s{step}_preds{sfx}"] = preds.numpy().astype(np.int8)
        for k, p in net.named_parameters():
            g = p.grad.detach().double()
            res[f"s{step}_gnorm{sfx}/{k}"] = np.array([g.norm().item(), g.sum().item()])
            if p.numel() <= 2048:
                res[f"s{step}_grad{sfx}/{k}"] = p.grad.detach().numpy().astype(np.float32)
            else:                                   # every stride-th element (<= 1024 of them): the float32-to-float64 distance of large tensors
                res[f"s{step}_gsample{sfx}/{k}"] = p.grad.detach().reshape(-1)[::-(-p.numel() // 1024)].numpy().astype(np.float32)
            if full:
                res[f"s{step}_psum/{k}"] = np.array([p.detach().double().sum().item(), p.detach().double().abs().sum().item()])
        for k, v in net.state_dict().items():
            if "running" in k:
                res[f"s{step}_buf{sfx}/{k}"] = v.numpy().astype(np.float32)
    np.random.seed(2109)
    with torch.no_grad():
        data = (torch.from_numpy(x.copy()), torch.from_numpy(t.copy()), ["f"] * Bn)
        m, _, preds, _ = tr.train_loop(data, opt, ce, net, None, False, 0, 0)
    res[f"eval_ce{sfx}"] = m["ce_loss"].numpy()
    if full:
        res["eval_preds"] = preds.numpy()


def sec_baseline_train16():
    """a12 at B = 16 (round-2 review: with B = 4 rows in the T-Net FC BatchNorms the second optimisation step is fp32 noise in ANY
    implementation, so it pinned nothing): the reference's own train_loop (pointNet/baseline/train_segmentation.py:274-328) on a seeded
    [16, 512, 9] batch, two train steps + one eval pass; every gradient norm, the small gradients in full at BOTH steps, parameter sums,
    running statistics after each step."""
    tr = load_script(os.path.join(REF, "pointNet/baseline/train_segmentation.py"), "ref_train_seg")
    from pointNet.model.pointnet import SegmentationPointNet
    from pointNet.model.light_pointnet_256 import SegmentationPointNet as LightSeg
    _baseline_train_section(tr, lambda: SegmentationPointNet(num_classes=5, point_dimension=3), 9000, 16, 512, "baseline_train_b16")
    _baseline_train_section(tr, lambda: LightSeg(num_classes=5, point_dimension=2, device="cpu"), 9500, 16, 512, "baseline_light_train_b16")
    # BASELINE.json config 1's own shape [4, 512, 9] with the same float64 arbiter (round-3 review: the B = 4 bars were pinned to ONE
    # kernel's summation order; the float32-to-float64 distance of the reference itself is the yardstick instead)
    _baseline_train_section(tr, lambda: SegmentationPointNet(num_classes=5, point_dimension=3), 9000, 4, 512, "baseline_train_b4")
    _baseline_train_section(tr, lambda: LightSeg(num_classes=5, point_dimension=2, device="cpu"), 9500, 4, 512, "baseline_light_train_b4")


def sec_baseline_cls():
    """f4, the baseline classification modules: pointNet/model/pointnet.py:100-125 ClassificationPointNet(num_classes, dropout, point_dimension=3)
    and pointNet/model/light_pointnet_256.py:100-125 (point_dimension=2, the only value its BasePointNet accepts, :71).  Their drivers cannot
    run in the reference as committed, so the MODULES are pinned: eval output (log-probabilities, feature transform) on
    [4, 512, 9]; one train-mode step's forward and autograd on [16, 512, 9] with dropout 0 -- NLL of the log-probabilities against seeded labels
    + 0.001 * the feature-transform regulariser (the loss recipe of the baseline drivers, train_segmentation.py:300-306) -- with every
    gradient, and the running statistics after it."""
    from pointNet.model.pointnet import ClassificationPointNet
    from pointNet.model.light_pointnet_256 import ClassificationPointNet as LightCls
    n_cls = 4
    for tag, mk, base in (("baseline_cls", lambda dp: ClassificationPointNet(n_cls, dropout=dp, point_dimension=3), 9700),
                          ("baseline_light_cls", lambda dp: LightCls(n_cls, dropout=dp, point_dimension=2, device="cpu"), 9800)):
        net = mk(0.3)
        table = {k: tuple(v.shape) for k, v in net.state_dict().items() if "num_batches" not in k}
        state = {k: torch.from_numpy(v) for k, v in baseline_state(synth, table, base).items()}
        net.load_state_dict(state, strict=False)
        net.eval()
        res = {"names": np.array(list(table.keys())), "shapes": np.array([";".join(map(str, s)) for s in table.values()])}
        with torch.no_grad():
            out, ft = net(torch.from_numpy(synth.windows(84, 4, 512)))
        res["eval_out"], res["eval_feat_T"] = out.numpy(), ft.numpy()
        y = torch.from_numpy((synth.uniform(86, (16,), 0.0, 1.0) * n_cls).astype(np.int64).clip(0, n_cls - 1))
        res["labels"] = y.numpy()
        # the train step twice: as the reference runs it (float32) and with torch's default dtype float64 (suffix 64: the arbiter; the
        # distance between the two is the reference's own float32 noise on this graph)
        for sfx, dt in (("", torch.float32), ("64", torch.float64)):
            torch.set_default_dtype(dt)
            try:
                net = mk(0.0)
                net.load_state_dict(state, strict=False)
                net.train()
                x = torch.from_numpy(synth.windows(85, 16, 512)).to(dt)
                out, ft = net(x)
                nll = torch.nn.functional.nll_loss(out, y)
                reg = torch.norm(torch.eye(64) - torch.bmm(ft, ft.transpose(2, 1)))
                (nll + 0.001 * reg).backward()
            finally:
                torch.set_default_dtype(torch.float32)
            res.update({f"nll{sfx}": np.array([nll.item()]), f"reg{sfx}": np.array([reg.item()])})
            if sfx == "":
                res.update(train_out=out.detach().numpy(), train_feat_T=ft.detach().numpy())
            for k, p in net.named_parameters():
                g = p.grad.detach()
                res[f"gnorm{sfx}/{k}"] = np.array([g.double().norm().item(), g.double().sum().item()])
                if g.numel() <= 4096:
                    res[f"grad{sfx}/{k}"] = g.numpy().astype(np.float32)
                else:                               # large tensors: norm + sum above and every stride-th element (<= 2048 of them)
                    stride = -(-g.numel() // 2048)
                    res[f"gsample{sfx}/{k}"] = g.reshape(-1)[::stride].numpy().astype(np.float32)
            if sfx == "":
                for k, v in net.state_dict().items():
                    if "running" in k:
                        res[f"buf/{k}"] = v.numpy().copy()
        save(tag, seed_base=np.array([base]), **res)


def sec_gru():
    """f4: SegmentationWithGRU (pointnetAtt.py:212-258) eval forwards (uniform and ragged windows) and the reference's GRU train_loop
    (pointNet/rnn/train_pointnetGRU.py:335-441): eval, then two train steps.  The module hard-codes nn.Dropout(0.3); its p is set to 0
    on the instance for the train steps so that they are deterministic (dropout parity is checked against the oracle with shared masks)."""
    from pointNet.model.pointnetAtt import BasePointNet, SegmentationWithGRU
    tr = load_script(os.path.join(REF, "pointNet/rnn/train_pointnetGRU.py"), "ref_train_gru")

    def models(enc_seed, head_seed):
        enc = BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device="cpu")
        gru = SegmentationWithGRU(num_classes=5, global_feat_size=256, hidden_size=64, device="cpu")
        sd = {k: torch.from_numpy(v) for k, v in synth.make_params(enc_seed, P.ENC_PARAMS).items()}
        sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(enc_seed, P.ENC_BUFFERS).items()})
        assert not enc.load_state_dict(sd, strict=False).unexpected_keys
        sd = {k: torch.from_numpy(v) for k, v in synth.make_params(head_seed, P.GRU_HEAD_PARAMS).items()}
        sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(head_seed, P.HEAD_BUFFERS).items()})
        missing = gru.load_state_dict(sd, strict=False)
        assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
        assert not missing.unexpected_keys
        return enc, gru

    _, gru = models(5, 6)
    gru.eval()
    seq = torch.from_numpy(synth.uniform(71, (2, 3, 256), 0.0, 2.0))
    lo = torch.from_numpy(synth.uniform(72, (2, 768, 64), -1.0, 1.0))
    with torch.no_grad():
        a = gru(seq, lo, [256, 256, 256])
        b = gru(seq, lo, [100, 300, 368])
        hs, _ = gru.gru_global(seq, gru.initHidden(seq))
    res = dict(uniform=a.numpy(), ragged=b.numpy(), hidden=hs.numpy())

    enc, gru = models(7, 8)
    gru.dropout.p = 0.0
    B, N, W = STEP_B, STEP_N, STEP_W
    pc, tg, cent, w_real = synth.sample_batch(43, B, N, max_w=W, w_real=STEP_WREAL)
    names = [f"f{i}" for i in range(B)]
    ce = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=-1)          # train_pointnetGRU.py:148: unweighted
    opt_p = torch.optim.Adam(enc.parameters(), lr=1e-3)
    opt_g = torch.optim.Adam(gru.parameters(), lr=1e-3)
    res.update(meta=np.array([B, N, W], dtype=np.int64), w_real=w_real)
    data = (torch.from_numpy(pc), torch.from_numpy(tg), names, torch.from_numpy(cent))
    with torch.no_grad():
        m, tpc, preds, _ = tr.train_loop(data, opt_p, opt_g, ce, enc, gru, None, "segmentation", False, torch.Tensor(), 0, 0)
    res.update(eval_ce=m["ce_loss"].detach().numpy(), eval_reg=m["reg_loss"].detach().numpy(), eval_loss=m["loss"].detach().numpy(),
               eval_targets=tpc.numpy(), eval_preds=preds.numpy())
    for step in (1, 2):
        data = (torch.from_numpy(pc.copy()), torch.from_numpy(tg.copy()), names, torch.from_numpy(cent))
        m, tpc, preds, _ = tr.train_loop(data, opt_p, opt_g, ce, enc, gru, None, "segmentation", True, torch.Tensor(), 0, 0)
        res[f"s{step}_ce"] = m["ce_loss"].detach().numpy()
        res[f"s{step}_reg"] = m["reg_loss"].detach().numpy()
        res[f"s{step}_loss"] = m["loss"].detach().numpy()
        res[f"s{step}_preds"] = preds.numpy()
        for tag, mod in (("enc", enc), ("gru", gru)):
            for k, p in mod.named_parameters():
                g = p.grad.detach().double()
                res[f"s{step}_{tag}_gnorm/{k}"] = np.array([g.norm().item(), g.sum().item()])
                if tag == "gru" and p.numel() <= 49152 and step == 1:
                    res[f"s1_gru_grad/{k}"] = p.grad.detach().numpy()
                res[f"s{step}_{tag}_psum/{k}"] = np.array([p.detach().double().sum().item(), p.detach().double().abs().sum().item()])
        if step == 1:
            for k, v in gru.state_dict().items():
                if "running" in k:
                    res[f"s1_gru_buf/{k}"] = v.numpy().copy()
    save("gru", **res)


def sec_cls():
    """f4: ClassificationWithAttention (pointnetAtt.py:115-151): eval forward (with a key-padding mask) and a train-mode forward +
    backward of a cross-entropy on its output with dropout 0 (constructor argument): output, attention weights, every gradient."""
    from pointNet.model.pointnetAtt import ClassificationWithAttention
    Wn, B, C = 5, 16, 3
    table = P.cls_head_params(C, Wn)
    net = ClassificationWithAttention(256, 8, num_classes=C, dropout=0.0, num_w=Wn)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(9, table).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(9, P.CLS_HEAD_BUFFERS).items()})
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    assert list(dict(net.named_parameters()).keys()) == list(table.keys())
    gl = torch.from_numpy(synth.uniform(91, (Wn, B, 256), 0.0, 2.0))
    mask = torch.zeros(B, Wn, dtype=torch.bool)
    mask[1, 3:] = True
    mask[7, 4] = True
    net.eval()
    with torch.no_grad():
        out, aw = net(gl, None, mask)
    res = dict(meta=np.array([Wn, B, C]), eval_out=out.numpy(), eval_weights=aw.numpy())
    net.train()
    glg = gl.clone().requires_grad_(True)
    out, aw = net(glg, None, mask)
    tgt = torch.from_numpy(synth.randint(92, (B,), 0, C))
    loss = torch.nn.functional.cross_entropy(out, tgt)
    loss.backward()
    res.update(train_out=out.detach().numpy(), train_weights=aw.detach().numpy(), loss=loss.detach().numpy(), d_gl=glg.grad.numpy())
    for k, p_ in net.named_parameters():
        res[f"grad/{k}"] = p_.grad.numpy()
    for k, v in net.state_dict().items():
        if "running" in k:
            res[f"buf/{k}"] = v.numpy()
    save("cls", **res)



def sec_cls_data():
    """The classification side of the kept data API: LidarDataset (pointNet/datasets.py:9-142), LidarInferenceDataset (:518-565),
    collate_cls_padd (pointNet/collate_fns.py:58-113), run on seeded pickled arrays with the numpy / python / torch RNGs seeded."""
    import pickle
    import tempfile
    from pointNet.datasets import LidarDataset, LidarInferenceDataset
    from pointNet.collate_fns import collate_cls_padd
    res = {}
    files = ["pc_81.pkl", "tower_82.pkl", "tower_83.pkl"]
    sizes = [300, 150, 200]                                           # more than / fewer than / exactly number_of_points = 200
    with tempfile.TemporaryDirectory() as d:
        for f, n, seed in zip(files, sizes, (81, 82, 83)):
            with open(os.path.join(d, f), "wb") as fh:
                pickle.dump(cls_sample_array(synth, seed, n), fh)
        for task in ("classification", "segmentation"):
            for cs in (False, True):
                ds = LidarDataset(d, task=task, number_of_points=200, files=files, fixed_num_points=True, c_sample=cs)
                np.random.seed(17)
                for i in range(len(files)):
                    pc, lab, fn = ds[i]
                    tag = f"{task[:3]}_{int(cs)}_{i}"
                    res["ds_pc_" + tag] = np.asarray(pc)
                    res["ds_lab_" + tag] = np.asarray(lab)
                res[f"ds_counts_{task[:3]}_{int(cs)}"] = np.array([len(ds), ds.len_towers, ds.len_landscape])
        inf = LidarInferenceDataset(d, files=files, c_sample=True)
        for i in range(len(files)):
            pc, fn = inf[i]
            res[f"inf_pc_{i}"] = pc.numpy()
    # collate_cls_padd on window samples (same generator as sec_collate) with a class target and per-point labels
    specs = [(91, 2048, 1), (92, 1500, 4), (93, 3000, 9)]
    batch = []
    for seed, n, w in specs:
        win = synth.windows(seed, w, n)
        pc = np.ascontiguousarray(win.transpose(1, 2, 0))
        lab = synth.labels_for(win, seed).transpose(1, 0).copy()
        cent = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], 0).astype(np.float32)
        batch.append((pc, [seed % 2], f"f{seed}", cent, lab))
    random.seed(6); torch.manual_seed(6)
    data, tg, names, cents, seg = collate_cls_padd(batch)
    res["c_data_shape"] = np.array(data.shape); res["c_tg"] = tg.numpy(); res["c_cents"] = cents.numpy()
    res["c_data_sum"] = data.double().sum(dim=(1, 2)).numpy()
    res["c_data_probe"] = data[:, ::97, :, :].numpy()
    res["c_seg_probe"] = seg[:, ::97, :].numpy()
    res["c_seg_sum"] = seg.sum(dim=1).numpy()
    save("cls_data", **res)


SECTIONS = dict(cls_data=sec_cls_data, baseline_cls=sec_baseline_cls, baseline_train16=sec_baseline_train16, cls=sec_cls, gru=sec_gru, fps=sec_fps, encoder=sec_encoder, head=sec_head, step=sec_step, metrics=sec_metrics,
                collate=sec_collate, dataset=sec_dataset, baseline=sec_baseline)

if __name__ == "__main__":
    install_stubs()
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        print("==", s)
        SECTIONS[s]()
