"""End-to-end on the GPU through the drop-in drivers: train_att on a synthetic dataset in the reference's on-disk
formats (loss goes down, checkpoint written with the reference's keys), then test() on that checkpoint, whose
predictions must give the SAME per-file accuracy / IoU as the oracle run on the same checkpoint."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import sub                           # noqa: E402
from oracle import ampnet_oracle as O              # noqa: E402

pytestmark = pytest.mark.gpu


def test_train_then_test_cli(synth, tmp_path):
    paths = synth.write_dataset(str(tmp_path), n_train=8, n_val=4, n_test=2, n_points=2048, seed=900)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        np.random.seed(0)
        torch.manual_seed(0)
        hist = sub("pointNet.amp_train").train_att('segmentation', paths["data"], paths["lists"], str(tmp_path / "out"), 2048, 4, 3, 1e-3,
                                                   number_of_workers=0)
        assert len(hist) == 3 and hist[-1][0]['loss'] < hist[0][0]['loss']          # it learns the synthetic rule
        cks = sorted(os.listdir("pointNet/checkpoints"))
        assert cks
        ck_path = os.path.join("pointNet/checkpoints", cks[-1])
        res = sub("pointNet.amp_test").test(paths["data"], str(tmp_path / "res"), 2048, 0, ck_path, paths["lists"], cluster_dir=paths["clusters"])
        assert os.path.exists(tmp_path / "res" / "IoU-results-v2.csv")
        # identical metrics from the oracle on the same checkpoint and clusters
        ck = torch.load(ck_path, map_location="cpu", weights_only=True)
        ep = {k: v.float() for k, v in ck["base_pointnet"].items() if "num_batches" not in k}
        hp = {k: v.float() for k, v in ck["segmen_net"].items() if "num_batches" not in k}
        accs, ious = [], {k: [] for k in ("bckg", "tower", "cables", "low_veg", "high_veg")}
        for name in open(os.path.join(paths["lists"], "test_seg_files.txt")).read().split():
            stem = name.split(".")[0]
            clusters = torch.load(os.path.join(paths["clusters"], stem + "_clusters_list.pkl"), weights_only=True)
            cent = torch.load(os.path.join(paths["clusters"], stem + "_centroids.pkl"), weights_only=True)
            lo, gl = [], []
            for c in clusters:
                l, g, _ = O.encoder(ep, ep, c[None, :, :9], train=False)
                lo.append(l)
                gl.append(g)
            logits = O.head(hp, hp, torch.stack(gl, 0), torch.cat(lo, 1), cent[None], [c.shape[0] for c in clusters], None, False)
            preds = O.predictions(logits).reshape(-1).numpy()
            tg = torch.cat(sub("utils.utils").get_labels([c.clone() for c in clusters])).numpy()
            accs.append(O.accuracy(preds, tg))
            # per-file IoU of every class PRESENT in the file's targets (test_pointnet_att_segmen.py:192-219)
            for c, k in enumerate(ious):
                if (tg == c).any():
                    ious[k].append(O.iou_obj(preds, tg, c))
        assert abs(res["accuracy"] - float(np.mean(accs))) < 2e-4      # a handful of fp32 argmax ties at most
        # the CSV row's per-class IoU (mean over the files that hold the class) and mIoU (mean of the five class means, :256-258):
        # the oracle's, up to the same handful of ties
        for k, v in ious.items():
            want = float(np.mean(v)) if v else float("nan")
            assert (np.isnan(want) and np.isnan(res["iou"][k])) or abs(res["iou"][k] - want) < 1e-3, (k, res["iou"][k], want)
        want_miou = float(np.mean([np.mean(ious[k]) for k in ("tower", "low_veg", "high_veg", "bckg", "cables")]))
        assert (np.isnan(want_miou) and np.isnan(res["mean_iou"])) or abs(res["mean_iou"] - want_miou) < 1e-3, (res["mean_iou"], want_miou)
    finally:
        os.chdir(cwd)


def test_gru_train_then_test_cli(synth, tmp_path):
    """The GRU variant's drivers (pointNet/rnn/train_pointnetGRU.py, test_pointnet_gru_segmen.py): train_gru on the synthetic dataset,
    then test() -- which clusters every file in situ with the on-device constrained k-means, as the reference does with KMeansConstrained --
    must report the accuracy the oracle gets on the same checkpoint and the same clusters."""
    paths = synth.write_dataset(str(tmp_path), n_train=8, n_val=4, n_test=2, n_points=2048, seed=901)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        np.random.seed(0)
        torch.manual_seed(0)
        G = sub("pointNet.gru_train")
        hist = G.train_gru('segmentation', paths["data"], paths["lists"], str(tmp_path / "out"), 2048, 9, 4, 3, 1e-3, number_of_workers=0)
        assert len(hist) == 3 and hist[-1][0]['loss'] < hist[0][0]['loss']
        cks = sorted(os.listdir("pointNet/checkpoints"))
        assert cks
        ck_path = os.path.join("pointNet/checkpoints", cks[-1])
        ck = torch.load(ck_path, map_location="cpu", weights_only=True)
        assert set(ck["segmen_net"].keys()) >= {"gru_global.weight_ih_l0", "gru_global.weight_hh_l0", "gru_global.bias_ih_l0", "gru_global.bias_hh_l0"}
        res = G.test(paths["data"], str(tmp_path / "res" / "preds"), 2048, 0, ck_path, paths["lists"])
        assert os.path.exists(tmp_path / "res" / "IoU-results-v2.csv")
        ep = {k: v.float() for k, v in ck["base_pointnet"].items() if "num_batches" not in k}
        hp = {k: v.float() for k, v in ck["segmen_net"].items() if "num_batches" not in k}
        U = sub("utils.utils")
        ds = sub("pointNet.datasets").LidarDataset4Test(paths["data"], task='segmentation', number_of_points=2048,
                                                        files=open(os.path.join(paths["lists"], "test_seg_files.txt")).read().split(),
                                                        fixed_num_points=False)
        accs = []
        for i in range(len(ds)):
            pc, _ = ds[i]
            clusters, _ = U.kmeans_clustering(torch.as_tensor(pc)[None], n_points=2048, get_centroids=True, max_clusters=G.MAX_CLUSTERS)
            lo, gl = [], []
            for c in clusters:
                l, g, _ = O.encoder(ep, ep, c[None, :, :9].float(), train=False)
                lo.append(l)
                gl.append(g)
            logits = O.gru_head(hp, hp, torch.stack(gl, 1), torch.cat(lo, 1), [c.shape[0] for c in clusters], False)
            preds = O.predictions(logits).reshape(-1).numpy()
            tg = torch.cat(U.get_labels([c.clone() for c in clusters])).numpy()
            accs.append(O.accuracy(preds, tg))
        assert abs(res["accuracy"] - float(np.mean(accs))) < 2e-4
    finally:
        os.chdir(cwd)
