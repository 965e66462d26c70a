"""Datasets of the AMP-Net path with the reference's class names, constructor arguments and return values
(pointNet/datasets.py:9-142 LidarDataset, :145-292 LidarDatasetExpanded, :295-460 LidarKmeansDataset, :463-515 LidarDataset4Test,
:518-565 LidarInferenceDataset).  CPU only (DataLoader workers).

On-disk formats (unchanged): `kmeans_<name>.pt` = torch tensor [n, >=10, w] with columns x, y, HAG, class, I, R, G,
B, NIR, NDVI, ... (data_proc/3_kmeans.py:116); test files = pickled numpy [n, >=10] rows with the same columns."""
import os

import numpy as np
import torch
from torch.utils import data

from .. import _hostlib
from .._safe_load import load_numpy_pickle, load_pt_array, pt_tensor_header

NOISE_CLASSES = (30, 7, 2, 8, 13, 14)     # datasets.py:339-350, deleted in this order


def segmentation_labels(codes):
    """ASPRS class codes -> {0 background, 1 tower (15), 2 lines (14), 3 low/medium vegetation (3, 4), 4 high vegetation (5)}
    (datasets.py:449-458; utils/utils.py:562-570)."""
    codes = torch.as_tensor(codes)
    if codes.is_floating_point() or codes.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8, torch.int8):
        # one table look-up instead of five masked assignments (codes outside 0 .. 255 and fractional codes are background, as before:
        # only the exact values 15, 14, 3, 4, 5 map to a class)
        ci = codes.to(torch.int64)
        exact = (ci.to(codes.dtype) == codes) & (ci >= 0) & (ci <= 255)
        return torch.where(exact, _LABEL_LUT[ci.clamp(0, 255)], torch.zeros((), dtype=torch.long))
    lab = torch.zeros(codes.shape, dtype=torch.long)
    lab[codes == 15] = 1
    lab[codes == 14] = 2
    lab[(codes == 3) | (codes == 4)] = 3
    lab[codes == 5] = 4
    return lab


_LABEL_LUT = torch.zeros(256, dtype=torch.long)
_LABEL_LUT[15], _LABEL_LUT[14], _LABEL_LUT[3], _LABEL_LUT[4], _LABEL_LUT[5] = 1, 2, 3, 3, 4


class LazyKmeansSample:
    """A kmeans_<name>.pt sample that has not been read yet: where its [n, feats, w] float32 record sits in the file.
    LidarKmeansDataset(lazy=True) hands these to collate_fns.collate_seq_ragged, which has libampnet_host.so read, filter and relabel
    each one straight into its slice of the batch (one pass, no per-sample arrays)."""
    __slots__ = ("path", "n", "feats", "w", "offset", "want_centroids")

    def __init__(self, path, n, feats, w, offset, want_centroids):
        self.path, self.n, self.feats, self.w, self.offset, self.want_centroids = path, n, feats, w, offset, want_centroids


class LidarKmeansDataset(data.Dataset):
    NUM_CLASSIFICATION_CLASSES = 2
    POINT_DIMENSION = 2

    def __init__(self, dataset_folder, task='classification', number_of_points=None, files=None, fixed_num_points=True,
                 c_sample=False, sort_kmeans=False, get_centroids=True, lazy=False):
        # lazy (not in the reference): __getitem__ returns (LazyKmeansSample, None, filename, None) for collate_seq_ragged to fill in;
        # samples the host library cannot take (not float32 [n, >= 10, 2 .. 64], library not built) come back eagerly as usual
        self.lazy = lazy
        self.dataset_folder = dataset_folder
        self.task = task
        self.n_points = number_of_points
        self.files = [f.split('.')[0] for f in files]
        self.sort_kmeans = sort_kmeans
        self.get_centroids = get_centroids
        self.classes_mapping = {}
        self.constrained_sampling = c_sample
        self.paths_files = [os.path.join(self.dataset_folder, 'kmeans_' + f + '.pt') for f in self.files]

    def __len__(self):
        return len(self.paths_files)

    def __getitem__(self, index):
        """-> (pc [n', 9, w] float32 ndarray, labels [n', w] LongTensor, filename, centroids [2, w] ndarray)
        for task == 'segmentation' (the AMP-Net path)."""
        filename = self.paths_files[index]
        if self.task != 'segmentation':
            raise NotImplementedError("only the segmentation task is on the AMP-Net hot path")
        host = _hostlib.lib()
        if self.lazy and host is not None:
            try:
                dtype, size, offset = pt_tensor_header(filename)
                if dtype == "float32" and len(size) == 3 and size[1] >= 10 and 2 <= size[2] <= 64:
                    return LazyKmeansSample(filename, size[0], size[1], size[2], offset, self.get_centroids), None, filename, None
            except Exception:                        # noqa: BLE001 -- not a plain saved tensor: the general loader below decides
                pass
        pc = load_pt_array(filename)
        if host is not None and pc.dtype == np.float32 and pc.ndim == 3 and pc.shape[1] >= 10 and 2 <= pc.shape[2] <= 64:
            # one pass over the sample in libampnet_host.so (include/ampnet_host.h): the same values as the numpy statement below
            # (w == 1 stays there: numpy sums a length-n column pairwise, a [n, w >= 2] view row by row)
            pc = np.ascontiguousarray(pc)
            n, feats, w = pc.shape
            out = np.empty((n, 9, w), dtype=np.float32)
            labels = torch.empty((n, w), dtype=torch.int64)
            cent = np.empty((2, w), dtype=np.float32) if self.get_centroids else None
            kept = host.ampnet_host_kmeans_sample_f32(pc.ctypes.data, n, feats, w, out.ctypes.data, labels.data_ptr(),
                                                      cent.ctypes.data if cent is not None else None)
            if kept < 0:
                raise RuntimeError(f"{filename}: ampnet_host_kmeans_sample_f32 refused shape {pc.shape}")
            return out[:kept], labels[:kept], filename, cent
        # a point ROW is dropped from every cluster as soon as one cluster carries a noise code in it
        # (np.delete on axis 0 with the row indices of np.where over [n, w]; datasets.py:339-350)
        # (the reference deletes code by code; the rows that survive all six passes are the rows that carry none of the codes in any
        # cluster, in their original order: one mask instead of six np.where / np.delete copies of the whole array)
        codes = pc[:, 3, :]
        drop = np.zeros(codes.shape[0], dtype=bool)
        for code in NOISE_CLASSES:
            drop |= (codes == code).any(axis=1)
        if drop.any():
            pc = pc[~drop]
        labels = segmentation_labels(pc[:, 3, :])
        pc = np.concatenate((pc[:, :3, :], pc[:, 4:10, :]), axis=1)
        xy = pc[:, :2, :]
        xy *= 2                                # x, y <- 2 v - 1 in place (the same two float32 roundings as v * 2 - 1)
        xy -= 1
        centroids = np.stack([pc[:, 0, :].mean(0), pc[:, 1, :].mean(0)], axis=0) if self.get_centroids else None
        return pc, labels, filename, centroids


class LidarDataset4Test(data.Dataset):
    NUM_CLASSIFICATION_CLASSES = 2
    POINT_DIMENSION = 2

    def __init__(self, dataset_folder, task='classification', number_of_points=None, files=None, fixed_num_points=True,
                 c_sample=False, allow_pickle=None):
        self.dataset_folder = dataset_folder
        self.task = task
        self.n_points = number_of_points
        self.files = files
        self.allow_pickle = allow_pickle
        self.fixed_num_points = fixed_num_points
        self.classes_mapping = {}
        self.constrained_sampling = c_sample
        self.paths_files = [os.path.join(self.dataset_folder, f) for f in self.files]

    def __len__(self):
        return len(self.paths_files)

    def __getitem__(self, index):
        """-> (pc [n, 10] float32 ndarray: x, y, HAG, I, R, G, B, NIR, NDVI, class code; filename).
        The reference unpickles the file (pickle.load, datasets.py:499-502); here a restricted unpickler reads plain numpy
        array pickles and anything else needs allow_pickle=True / AMPNET_ALLOW_PICKLE=1 (_safe_load.py)."""
        filename = self.paths_files[index]
        pc = np.asarray(load_numpy_pickle(filename, self.allow_pickle), dtype=np.float32)
        pc = np.concatenate((pc[:, :3], pc[:, 4:10], pc[:, 3:4]), axis=1)
        pc[:, 0] = pc[:, 0] * 2 - 1
        pc[:, 1] = pc[:, 1] * 2 - 1
        return pc, filename


class LidarDatasetExpanded(data.Dataset):
    """Single-window dataset of the baseline PointNet (pointNet/datasets.py:145-292): a file = pickled numpy [n, >= 10] rows
    (x, y, HAG, class, I, R, G, B, NIR, NDVI, ...); noise classes removed, resampled to number_of_points (numpy RNG: choice when more,
    duplicates when fewer), labels from column 3, features = columns 0-2, 4-9 with x, y <- 2 v - 1.
    __getitem__ -> (pc [n, 9] float32 tensor, labels [n] LongTensor, filename) for task 'segmentation'."""
    NUM_CLASSIFICATION_CLASSES = 2
    POINT_DIMENSION = 2

    def __init__(self, dataset_folder, task='classification', number_of_points=None, files=None, fixed_num_points=True, allow_pickle=None):
        self.dataset_folder = dataset_folder
        self.task = task
        self.n_points = number_of_points
        self.files = files
        self.fixed_num_points = fixed_num_points
        self.allow_pickle = allow_pickle
        self.classes_mapping = {}
        self.paths_files = [os.path.join(self.dataset_folder, f) for f in self.files]

    def __len__(self):
        return len(self.paths_files)

    def __getitem__(self, index):
        filename = self.paths_files[index]
        pc = np.asarray(load_numpy_pickle(filename, self.allow_pickle), dtype=np.float32)
        for code in NOISE_CLASSES:
            pc = pc[pc[:, 3] != code]
        if self.fixed_num_points and pc.shape[0] > self.n_points:
            pc = pc[np.random.choice(pc.shape[0], self.n_points), :]
        if self.fixed_num_points and 0 < pc.shape[0] < self.n_points:
            extra = np.random.randint(0, pc.shape[0], self.n_points - pc.shape[0])
            pc = np.concatenate([pc, pc[extra, :]], axis=0)
        if self.task != 'segmentation':
            raise NotImplementedError("only the segmentation task is built (SURVEY.md section 8: classification is out of scope)")
        labels = segmentation_labels(pc[:, 3])
        pc = torch.from_numpy(np.concatenate((pc[:, :3], pc[:, 4:10]), axis=1))
        pc[:, 0] = pc[:, 0] * 2 - 1
        pc[:, 1] = pc[:, 1] * 2 - 1
        return pc, labels, filename


class LidarDataset(data.Dataset):
    """Single-window samples of the tower / no-tower classifier (pointNet/datasets.py:9-142): a file = pickled numpy [n, >= 10] rows
    (x, y, HAG, class, I, R, G, B, NIR, NDVI, sampling flag); the class of a SAMPLE is in its file name ('pc_' -> 0, 'tower_' -> 1).
    __getitem__ -> (pc [n', 7] float32 ndarray = columns x, y, HAG, I, G, B, NDVI (:60), labels, filename); labels = the sample's class
    (task 'classification') or the per-point segmentation labels from column 3 (task 'segmentation', LongTensor [n']).
    Resampling to number_of_points uses numpy's global RNG like the reference (np.random.choice WITH replacement when there are more
    points, np.random.randint duplicates appended when there are fewer, :78-88), so a seeded run draws the same rows."""
    NUM_CLASSIFICATION_CLASSES = 2
    POINT_DIMENSION = 2

    def __init__(self, dataset_folder, task='classification', number_of_points=None, number_of_windows=None, files=None,
                 fixed_num_points=True, c_sample=False, allow_pickle=None):
        self.dataset_folder = dataset_folder
        self.task = task
        self.n_points = number_of_points
        self.n_windows = number_of_windows
        self.files = files
        self.fixed_num_points = fixed_num_points
        self.allow_pickle = allow_pickle
        self.classes_mapping = {}
        self.constrained_sampling = c_sample
        self.paths_files = [os.path.join(self.dataset_folder, f) for f in self.files]
        self._init_mapping()

    def __len__(self):
        return len(self.paths_files)

    def _init_mapping(self):
        for f in self.files:
            if 'pc_' in f:
                self.classes_mapping[f] = 0
            elif 'tower_' in f:
                self.classes_mapping[f] = 1
        self.len_towers = sum(v == 1 for v in self.classes_mapping.values())
        self.len_landscape = sum(v == 0 for v in self.classes_mapping.values())

    def __getitem__(self, index):
        filename = self.paths_files[index]
        pc = self.prepare_data(filename, self.n_points, fixed_num_points=self.fixed_num_points,
                               constrained_sample=self.constrained_sampling, allow_pickle=self.allow_pickle)
        labels = self.get_labels(pc, self.classes_mapping[self.files[index]], self.task)
        pc = pc.numpy()
        return np.concatenate((pc[:, :3], pc[:, 4:5], pc[:, 6:8], pc[:, 9:10]), axis=1), labels, filename

    @staticmethod
    def prepare_data(point_file, number_of_points=None, fixed_num_points=True, constrained_sample=False, allow_pickle=None):
        pc = np.asarray(load_numpy_pickle(point_file, allow_pickle)).astype(np.float32)
        if constrained_sample:
            pc = pc[pc[:, 10] == 1]                                     # rows flagged for sampling
        if fixed_num_points and pc.shape[0] > number_of_points:
            pc = pc[np.random.choice(pc.shape[0], number_of_points), :]
        elif fixed_num_points and pc.shape[0] < number_of_points:
            extra = np.random.randint(0, pc.shape[0], number_of_points - pc.shape[0])
            pc = np.concatenate([pc, pc[extra, :]], axis=0)
        return torch.from_numpy(pc)

    @staticmethod
    def get_labels(pointcloud, point_cloud_class, task='classification'):
        """'segmentation': {0 background, 1 tower (15), 2 lines (14), 3 low / medium vegetation (3, 4), 4 high vegetation (5)} per point
        from column 3 (:118-131); 'classification': the sample's class."""
        if task == 'segmentation':
            return segmentation_labels(pointcloud[:, 3])
        if task == 'classification':
            return point_cloud_class
        raise ValueError(f"task {task!r}: 'classification' or 'segmentation'")        # (the reference leaves `labels` unbound here)


class LidarInferenceDataset(data.Dataset):
    """Raw samples for inference (pointNet/datasets.py:518-565): __getitem__ -> (pc [n', >= 10] float32 tensor with every column of
    the file, filename); with c_sample only the rows whose flag column 10 is 1."""
    NUM_CLASSIFICATION_CLASSES = 2
    POINT_DIMENSION = 2

    def __init__(self, dataset_folder, task='classification', files=None, c_sample=False, allow_pickle=None):
        self.dataset_folder = dataset_folder
        self.task = task
        self.files = files
        self.allow_pickle = allow_pickle
        self.classes_mapping = {}
        self.constrained_sampling = c_sample
        self.paths_files = [os.path.join(self.dataset_folder, f) for f in self.files]

    def __len__(self):
        return len(self.paths_files)

    def __getitem__(self, index):
        filename = self.paths_files[index]
        return self.prepare_data(filename, constrained_sample=self.constrained_sampling, allow_pickle=self.allow_pickle), filename

    @staticmethod
    def prepare_data(point_file, constrained_sample=False, allow_pickle=None):
        pc = np.asarray(load_numpy_pickle(point_file, allow_pickle)).astype(np.float32)
        if constrained_sample:
            pc = pc[pc[:, 10] == 1]
        return torch.from_numpy(pc)
