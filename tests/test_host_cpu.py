"""CPU tests of the host side: collate / dataset / metrics mirrors against the reference's outputs, the C-ABI
symbol table, the parameter tables, checkpoints."""
import ctypes
import os
import pickle
import random
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402


def test_collate_seq_padd_matches_reference(golden, synth):
    g = golden("collate")
    C = sub("pointNet.collate_fns")
    specs = [(61, 2048, 1), (62, 2048, 3), (63, 1500, 5), (64, 3000, 9), (65, 2048, 9)]
    batch = []
    for seed, n, w in specs:
        win = synth.windows(seed, w, n)
        pc = np.ascontiguousarray(win.transpose(1, 2, 0))
        lab = synth.labels_for(win, seed).transpose(1, 0).copy()
        cent = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], 0).astype(np.float32)
        batch.append((pc, lab, f"f{seed}", cent))
    random.seed(5)
    torch.manual_seed(5)
    data, tg, names, cents = C.collate_seq_padd(batch)
    assert list(data.shape) == list(g["data_shape"]) and list(tg.shape) == list(g["tg_shape"])
    assert names == [f"f{s}" for s, _, _ in specs]
    np.testing.assert_array_equal(cents.numpy(), g["cents"])                   # incl. the .view(-1, 9, 2) quirk
    np.testing.assert_array_equal(data[:, ::97].numpy(), g["data_probe"])      # same random draws, same rows
    np.testing.assert_array_equal(tg[:, ::97].numpy(), g["tg_probe"])
    np.testing.assert_allclose(data.double().sum(dim=(1, 2)).numpy(), g["data_sum"], rtol=1e-12)
    np.testing.assert_array_equal(tg.sum(dim=1).numpy(), g["tg_sum"])
    # padded clusters: targets -1, data = the last real cluster
    assert (tg[0, :, 1:] == -1).all() and torch.equal(data[0, :, :, 0], data[0, :, :, 8])


def test_kmeans_dataset_matches_reference(golden, synth, tmp_path):
    g = golden("dataset")
    D = sub("pointNet.datasets")
    raw = synth.kmeans_file_tensor(71, 96, 3)
    torch.save(torch.from_numpy(raw), tmp_path / "kmeans_tile71.pt")
    ds = D.LidarKmeansDataset(str(tmp_path), task="segmentation", number_of_points=2048, files=["tile71.pt"])
    assert len(ds) == 1
    pc, lab, fn, cent = ds[0]
    np.testing.assert_array_equal(pc, g["pc"])
    np.testing.assert_array_equal(lab.numpy(), g["labels"])
    np.testing.assert_array_equal(cent, g["centroids"])
    assert fn.endswith("kmeans_tile71.pt") and pc.shape[0] < 96            # noise rows were dropped


def test_test_dataset_and_labels(synth, tmp_path):
    D = sub("pointNet.datasets")
    U = sub("utils.utils")
    raw = synth.kmeans_file_tensor(72, 50, 1)[:, :, 0]                     # [n, 13]
    with open(tmp_path / "t.pkl", "wb") as f:
        pickle.dump(raw, f)
    ds = D.LidarDataset4Test(str(tmp_path), task="segmentation", number_of_points=2048, files=["t.pkl"], fixed_num_points=False)
    pc, fn = ds[0]
    assert pc.shape == (50, 10)
    np.testing.assert_allclose(pc[:, 0], raw[:, 0] * 2 - 1)
    np.testing.assert_array_equal(pc[:, 9], raw[:, 3])
    labs = U.get_labels([torch.from_numpy(pc).unsqueeze(0)])
    want = np.select([raw[:, 3] == 15, raw[:, 3] == 14, (raw[:, 3] == 3) | (raw[:, 3] == 4), raw[:, 3] == 5], [1, 2, 3, 4], 0)
    np.testing.assert_array_equal(labs[0].numpy(), want)
    c = U.get_cluster_centroid(torch.from_numpy(pc))
    np.testing.assert_allclose(c.numpy(), pc[:, :2].mean(0), rtol=1e-6)


def test_metrics_match_reference(golden, synth):
    g = golden("metrics")
    M = sub("utils.get_metrics")
    U = sub("utils.utils")
    preds = torch.from_numpy(synth.randint(51, (4000,), 0, 5))
    tgt = synth.randint(52, (4000,), 0, 5)
    tgt[synth.uniform01(53, (4000,)) < 0.2] = -1
    p2, t2, keep = U.rm_padding(preds, torch.from_numpy(tgt))
    assert int(keep.sum()) == int(g["n_keep"])
    for c in range(5):
        assert M.get_iou_obj(p2, t2, c) == float(g["ious"][c])
    assert M.get_accuracy(p2, t2, {}, "segmentation")["accuracy"] == float(g["acc"])
    assert np.isnan(M.get_iou_obj(torch.tensor([0, 0, 1, 1]), torch.tensor([0, 1, 1, 1]), 4))


def test_host_augmentation_replays_reference_draws(synth):
    """augment_batch == the reference's draw order (helpers.replay_augment restates train_loop's host legs)."""
    from helpers import replay_augment
    S = sub("pointNet.amp_step")
    pc, tg, cent, _ = synth.sample_batch(9, 3, 64, max_w=4, w_real=[4, 2, 3])
    for train in (False, True):
        want_pc, want_tg = replay_augment(123, pc, tg, train)
        np.random.seed(123)
        x, t = S.augment_batch(torch.from_numpy(pc.copy()), torch.from_numpy(tg.copy()), train)
        np.testing.assert_array_equal(x, want_pc.transpose(0, 3, 1, 2))
        np.testing.assert_array_equal(t, want_tg.transpose(0, 2, 1))


# ---- the C ABI: every symbol the header declares is exported; tables match the Python side -------------------
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ampnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ampnet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = sub("_lib")
    lib = L.lib()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ampnet_hip.h but not exported"
    assert lib.ampnet_abi_version() == L.ABI_VERSION


def test_parameter_tables_match_library(params):
    lib = sub("_lib").lib()
    lib.ampnet_table_name.restype = ctypes.c_char_p
    lib.ampnet_table_numel.restype = ctypes.c_long
    for t, table in enumerate((params.ENC_PARAMS, params.ENC_BUFFERS, params.HEAD_PARAMS, params.HEAD_BUFFERS)):
        assert lib.ampnet_table_count(t) == len(table)
        for i, (name, shape) in enumerate(table.items()):
            assert lib.ampnet_table_name(t, i).decode() == name
            assert lib.ampnet_table_numel(t, i) == params.numel(shape)


def test_cpu_tensors_fail_loudly(synth):
    """No CPU fallback: the HIP path refuses host tensors instead of computing something else."""
    U = sub("utils.utils")
    L = sub("_lib")
    with pytest.raises(L.AmpnetError):
        U.fps_indices(torch.from_numpy(synth.clouds(1, 1, 64)[0]), 8)
    M = sub("pointNet.model.pointnetAtt")
    with pytest.raises(NotImplementedError):
        M.BasePointNet(point_dimension=2, device="cpu")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, device="cpu")
    with pytest.raises(L.AmpnetError):
        enc(torch.zeros(2, 64, 9))


def test_checkpoint_roundtrip_keys(tmp_path, params):
    M = sub("pointNet.model.pointnetAtt")
    U = sub("utils.utils")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, device="cpu")
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device="cpu")
    o1 = torch.optim.Adam(enc.parameters(), lr=1e-3)
    o2 = torch.optim.Adam(att.parameters(), lr=1e-3)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        U.save_checkpoint_segmen_model("t", "segmentation", 3, 0, enc, att, o1, o2, 0.5, 32, 1e-3, 2048, None)
        ck = torch.load("pointNet/checkpoints/model_t.pth", weights_only=True)
    finally:
        os.chdir(cwd)
    assert set(ck) == {"base_pointnet", "segmen_net", "opt_pointnet", "opt_segmen", "task", "batch_size", "lr",
                       "number_of_points", "epoch", "epochs_since_improvement", "accuracy"}
    enc2 = M.BasePointNet(point_dimension=3, return_local_features=True, device="cpu")
    enc2.load_state_dict(ck["base_pointnet"])                  # strict: every reference key present, nothing extra
    assert torch.equal(enc2.conv_6.weight, enc.conv_6.weight)


def test_pickled_inputs_need_an_explicit_opt_in(tmp_path, monkeypatch):
    """a file torch's restricted unpickler rejects must not silently fall through to pickle.load."""
    import pickle
    SL = sub("_safe_load")
    arr = np.arange(12, dtype=np.float32).reshape(3, 4)
    p_np = tmp_path / "a.pkl"
    with open(p_np, "wb") as f:
        pickle.dump(arr, f)
    assert np.array_equal(SL.load_numpy_pickle(str(p_np)), arr)          # plain numpy pickles load, nothing else is resolved

    class Evil:
        def __reduce__(self):
            return (os.path.join, ("executed", "payload"))
    p_bad = tmp_path / "b.pkl"
    with open(p_bad, "wb") as f:
        pickle.dump([Evil()], f)
    monkeypatch.delenv("AMPNET_ALLOW_PICKLE", raising=False)
    for loader in (SL.load_numpy_pickle, SL.load_tensor_list):
        with pytest.raises(Exception) as e:
            loader(str(p_bad))
        assert "AMPNET_ALLOW_PICKLE" in str(e.value)
    assert SL.load_tensor_list(str(p_bad), allow_pickle=True) == ["executed/payload"]      # the explicit opt-in
    p_t = tmp_path / "c.pkl"
    torch.save([torch.ones(2, 3)], p_t)
    assert torch.equal(SL.load_tensor_list(str(p_t))[0], torch.ones(2, 3))


def test_lidar_dataset_and_inference_dataset_match_reference(golden, synth, tmp_path):
    """pointNet/datasets.py:9-142 (LidarDataset) and :518-565 (LidarInferenceDataset) on seeded pickled samples, numpy RNG seeded as in
    tests/golden/make_golden.py::sec_cls_data: the resampling draws, the seven feature columns, both label kinds, the class counts."""
    import pickle
    from helpers import cls_sample_array
    g = golden("cls_data")
    D = sub("pointNet.datasets")
    files = ["pc_81.pkl", "tower_82.pkl", "tower_83.pkl"]
    for f, n, seed in zip(files, (300, 150, 200), (81, 82, 83)):
        with open(tmp_path / f, "wb") as fh:
            pickle.dump(cls_sample_array(synth, seed, n), fh)
    for task in ("classification", "segmentation"):
        for cs in (False, True):
            ds = D.LidarDataset(str(tmp_path), task=task, number_of_points=200, files=files, fixed_num_points=True, c_sample=cs)
            np.random.seed(17)
            for i in range(len(files)):
                pc, lab, fn = ds[i]
                tag = f"{task[:3]}_{int(cs)}_{i}"
                assert isinstance(pc, np.ndarray) and pc.dtype == np.float32 and pc.shape[1] == 7
                assert np.array_equal(pc, g["ds_pc_" + tag]), tag
                assert np.array_equal(np.asarray(lab), g["ds_lab_" + tag]), tag
                assert fn == str(tmp_path / files[i])
            assert [len(ds), ds.len_towers, ds.len_landscape] == g[f"ds_counts_{task[:3]}_{int(cs)}"].tolist()
    inf = D.LidarInferenceDataset(str(tmp_path), files=files, c_sample=True)
    for i in range(len(files)):
        pc, fn = inf[i]
        assert torch.is_tensor(pc) and pc.dtype == torch.float32
        assert np.array_equal(pc.numpy(), g[f"inf_pc_{i}"])


def test_collate_cls_padd_matches_reference(golden, synth):
    """pointNet/collate_fns.py:58-113 with the python / torch RNGs seeded as in make_golden.py::sec_cls_data."""
    import random
    g = golden("cls_data")
    C = sub("pointNet.collate_fns")
    batch = []
    for seed, n, w in [(91, 2048, 1), (92, 1500, 4), (93, 3000, 9)]:
        win = synth.windows(seed, w, n)
        pc = np.ascontiguousarray(win.transpose(1, 2, 0))
        lab = synth.labels_for(win, seed).transpose(1, 0).copy()
        cent = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], 0).astype(np.float32)
        batch.append((pc, [seed % 2], f"f{seed}", cent, lab))
    random.seed(6); torch.manual_seed(6)
    data, tg, names, cents, seg = C.collate_cls_padd(batch)
    assert list(data.shape) == g["c_data_shape"].tolist() and names == ["f91", "f92", "f93"]
    assert np.array_equal(tg.numpy(), g["c_tg"])
    assert np.array_equal(cents.numpy(), g["c_cents"])
    assert np.allclose(data.double().sum(dim=(1, 2)).numpy(), g["c_data_sum"], rtol=0, atol=0)
    assert np.array_equal(data[:, ::97, :, :].numpy(), g["c_data_probe"])
    assert np.array_equal(seg[:, ::97, :].numpy(), g["c_seg_probe"])
    assert np.array_equal(seg.sum(dim=1).numpy(), g["c_seg_sum"])
    assert seg.dtype == torch.int64 and (seg[0, :, 1:] == -1).all()


def test_collate_seq_ragged_is_collate_seq_padd_unpadded(golden, synth):
    """collate_seq_ragged (the package's loader-side collate: no resampling / padding copies on the host) makes collate_seq_padd's random
    draws in its order and carries exactly its information: RaggedBatch.to_padded() rebuilds the padded tensors bit for bit -- checked
    against collate_seq_padd run with the same seeds AND against the reference's own output (tests/golden/collate.npz)."""
    g = golden("collate")
    C = sub("pointNet.collate_fns")
    batch = []
    for seed, n, w in [(61, 2048, 1), (62, 2048, 3), (63, 1500, 5), (64, 3000, 9), (65, 2048, 9)]:
        win = synth.windows(seed, w, n)
        pc = np.ascontiguousarray(win.transpose(1, 2, 0))
        lab = synth.labels_for(win, seed).transpose(1, 0).copy()
        cent = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], 0).astype(np.float32)
        batch.append((pc, lab, f"f{seed}", cent))
    random.seed(5); torch.manual_seed(5)
    data, tg, names, cents = C.collate_seq_padd(batch)
    random.seed(5); torch.manual_seed(5)
    rb, none, names2, cents2 = C.collate_seq_ragged(batch)
    assert none is None and names2 == names and torch.equal(cents2, cents)
    assert rb.pts.dtype == torch.float32 and rb.lab.dtype == torch.int8 and rb.idx.dtype == torch.int32 and tuple(rb.meta.shape) == (5, 4)
    assert rb.pts.numel() == sum(n * 9 * w for _, n, w in [(0, 2048, 1), (0, 2048, 3), (0, 1500, 5), (0, 3000, 9), (0, 2048, 9)])
    d2, t2 = rb.to_padded()
    assert torch.equal(d2, data) and torch.equal(t2, tg)
    assert np.array_equal(d2[:, ::97, :, :].numpy(), g["data_probe"]) and np.array_equal(t2[:, ::97, :].numpy(), g["tg_probe"])
    assert np.array_equal(cents2.numpy(), g["cents"])
    # a DataLoader moves it like a tensor batch: pickling (worker -> main), to()
    rb2 = pickle.loads(pickle.dumps(rb))
    assert torch.equal(rb2.pts, rb.pts) and torch.equal(rb2.idx, rb.idx) and len(rb2) == 5


# ---- libampnet_host.so: the per-sample host work of the loader (include/ampnet_host.h) ---------------------------------------------------
def _host_declared_symbols():
    text = open(os.path.join(ROOT, "include", "ampnet_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ampnet_host_[a-z0-9_]+)\s*\(", text)))


def _host_lib(monkeypatch, mode):
    H = sub("_hostlib")
    monkeypatch.setenv("AMPNET_HOST_LOADER", mode)
    monkeypatch.setattr(H, "_tried", False)
    monkeypatch.setattr(H, "_lib", None)
    return H


def test_host_library_exports_every_declared_symbol(monkeypatch):
    H = _host_lib(monkeypatch, "native")
    lib = H.lib()
    assert lib is not None, "libampnet_host.so is not built: python __graft_entry__.py build"
    names = _host_declared_symbols()
    assert len(names) == 3
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ampnet_host.h but not exported"
    assert lib.ampnet_host_abi_version() == H.ABI_VERSION


def _awkward_sample(synth, seed, n, w):
    """A kmeans file tensor with every kind of class code the label / noise rules distinguish."""
    raw = synth.kmeans_file_tensor(seed, n, w)
    rng = np.random.default_rng(seed)
    odd = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 30, 31, 32, 255, 256, -1, 14.5, 2.0000002, np.nan, np.inf, 1e9], dtype=np.float32)
    rows = rng.choice(n, size=min(n, 3 * len(odd)), replace=False)
    for k, r in enumerate(rows):
        raw[r, 3, rng.integers(0, w)] = odd[k % len(odd)]
    return raw


@pytest.mark.parametrize("n,w", [(96, 3), (700, 9), (2048, 5), (64, 1), (33, 2), (1, 4)])
def test_host_kmeans_sample_equals_the_numpy_statement(synth, tmp_path, monkeypatch, n, w):
    """LidarKmeansDataset.__getitem__ through ampnet_host_kmeans_sample_f32 = the numpy statement of the reference's steps, bit for bit
    (points, labels, centroids), on class codes of every kind; w == 1 takes the numpy path in both (numpy sums a column pairwise)."""
    D = sub("pointNet.datasets")
    raw = _awkward_sample(synth, 500 + n + w, n, w)
    torch.save(torch.from_numpy(raw), tmp_path / "kmeans_a.pt")
    got = {}
    for mode in ("numpy", "native"):
        H = _host_lib(monkeypatch, mode)
        assert (H.lib() is not None) == (mode == "native")
        got[mode] = D.LidarKmeansDataset(str(tmp_path), task="segmentation", number_of_points=2048, files=["a.pt"])[0]
    a, b = got["numpy"], got["native"]
    assert a[0].dtype == b[0].dtype == np.float32 and a[1].dtype == b[1].dtype == torch.int64
    np.testing.assert_array_equal(a[0], b[0])
    assert torch.equal(a[1], b[1])
    np.testing.assert_array_equal(a[3], b[3])
    assert a[3].dtype == b[3].dtype and a[0].shape[0] < n or n == 1 or not np.isin(raw[:, 3, :], D.NOISE_CLASSES).any()


def test_load_pt_array_reads_what_torch_load_reads(synth, tmp_path):
    SL = sub("_safe_load")
    raw = synth.kmeans_file_tensor(9, 50, 4)
    t = torch.from_numpy(raw)
    torch.save(t, tmp_path / "plain.pt")
    torch.save(t[5:40], tmp_path / "view.pt")                        # a view into a larger storage (storage offset != 0)
    torch.save(t.transpose(0, 2), tmp_path / "strided.pt")           # not contiguous: the general loader takes it
    torch.save(t, tmp_path / "legacy.pt", _use_new_zipfile_serialization=False)
    for name, want in (("plain", t), ("view", t[5:40]), ("strided", t.transpose(0, 2)), ("legacy", t)):
        got = SL.load_pt_array(str(tmp_path / (name + ".pt")))
        np.testing.assert_array_equal(np.asarray(got), want.numpy())
    dtype, size, off = SL.pt_tensor_header(str(tmp_path / "view.pt"))
    assert dtype == "float32" and size == (35, 13, 4) and off > 0
    with pytest.raises(Exception):
        SL.pt_tensor_header(str(tmp_path / "strided.pt"))
    torch.save({"a": t}, tmp_path / "dict.pt")
    with pytest.raises(Exception):
        SL.pt_tensor_header(str(tmp_path / "dict.pt"))


def test_lazy_samples_collate_to_the_same_ragged_batch(synth, tmp_path, monkeypatch):
    """LidarKmeansDataset(lazy=True) + collate_seq_ragged (samples read by ampnet_host_kmeans_file_ragged_f32 straight into the batch) =
    the eager numpy samples through the same collate: same bytes, same random draws.  The set holds a w == 1 sample and a legacy file (both
    come back eagerly from the lazy dataset) and samples above and below 2048 surviving rows."""
    D, C = sub("pointNet.datasets"), sub("pointNet.collate_fns")
    shapes = [(2500, 5), (2048, 9), (900, 1), (1500, 2), (2300, 7), (2048, 3)]
    files = []
    for k, (n, w) in enumerate(shapes):
        raw = _awkward_sample(synth, 900 + k, n, w) if k != 1 else synth.kmeans_file_tensor(901, n, w, noise_frac=0.0)
        torch.save(torch.from_numpy(raw), tmp_path / f"kmeans_s{k}.pt", _use_new_zipfile_serialization=(k != 4))
        files.append(f"s{k}.pt")
    out = {}
    for tag, mode, lazy in (("numpy", "numpy", False), ("eager", "native", False), ("lazy", "native", True)):
        _host_lib(monkeypatch, mode)
        ds = D.LidarKmeansDataset(str(tmp_path), task="segmentation", number_of_points=2048, files=files, lazy=lazy)
        items = [ds[i] for i in range(len(files))]
        if tag == "lazy":
            kinds = [isinstance(it[0], D.LazyKmeansSample) for it in items]
            assert kinds == [True, True, False, True, False, True]
        torch.manual_seed(3)
        random.seed(3)
        out[tag] = C.collate_seq_ragged(items)
    ref = out["numpy"]
    for tag in ("eager", "lazy"):
        rb, _, names, cents = out[tag]
        for k in ("pts", "lab", "idx", "meta"):
            assert torch.equal(getattr(rb, k), getattr(ref[0], k)), (tag, k)
        assert names == ref[2] and torch.equal(cents, ref[3])
    data, targets = out["lazy"][0].to_padded()
    assert data.shape == (len(files), 2048, 9, 9) and targets.shape == (len(files), 2048, 9)


def test_start_workers_keeps_every_epoch_complete(synth, tmp_path):
    """amp_train.start_workers forks the persistent workers before the epoch loop asks for its iterator; the batches prefetched meanwhile are
    dropped, the epochs that follow are complete (every sample exactly once, len(loader) batches), and the worker processes are the same
    ones from epoch to epoch (no fork after start-up: DESIGN.md section 5, loader-fed epoch)."""
    D, C, A = sub("pointNet.datasets"), sub("pointNet.collate_fns"), sub("pointNet.amp_train")
    files = []
    for k in range(12):
        torch.save(torch.from_numpy(synth.kmeans_file_tensor(40 + k, 300 + 7 * k, 2 + k % 3, noise_frac=0.0)), tmp_path / f"kmeans_f{k}.pt")
        files.append(f"f{k}.pt")
    ds = D.LidarKmeansDataset(str(tmp_path), task="segmentation", number_of_points=2048, files=files, lazy=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=True, num_workers=2, drop_last=True, collate_fn=C.collate_seq_ragged,
                                         persistent_workers=True)
    A.start_workers(loader)
    pids = None
    for epoch in range(2):
        names, nb = [], 0
        for rb, _, fn, cents in loader:
            assert len(rb) == 4 and cents.shape == (4, 9, 2)
            names += fn
            nb += 1
        assert nb == len(loader) == 3 and sorted(names) == sorted(ds.paths_files)
        now = sorted(w.pid for w in loader._iterator._workers)
        assert pids is None or now == pids
        pids = now
    A.start_workers(torch.utils.data.DataLoader(ds, batch_size=4, num_workers=0))       # no workers: nothing to do
