"""Loaders for the reference's pickled on-disk formats that execute nothing from the file unless the caller opts in.

The reference reads its test samples with pickle.load (pointNet/datasets.py:499-502) and its cluster lists with
pickle.load too (test_pointnet_att_segmen.py:140-143; written by utils/utils.py:526-533).  Unpickling runs whatever
the file names, so here:
  * numpy arrays are read by a restricted unpickler that resolves ONLY the globals numpy's own array pickles use;
  * lists of tensors are read with torch.load(weights_only=True);
  * full unpickling is an explicit opt-in: allow_pickle=True or AMPNET_ALLOW_PICKLE=1 (only for files you produced).
"""
import os
import pickle

import torch

from ._lib import AmpnetError

_NUMPY_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}


class _NumpyOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _NUMPY_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"global {module}.{name} is not part of a plain numpy array pickle")


def pickle_allowed(allow_pickle=None):
    return bool(allow_pickle) if allow_pickle is not None else os.environ.get("AMPNET_ALLOW_PICKLE") == "1"


def load_numpy_pickle(path, allow_pickle=None):
    """A pickled numpy array (the reference's test-sample format) without executing anything else from the file."""
    try:
        with open(path, "rb") as f:
            return _NumpyOnlyUnpickler(f).load()
    except pickle.UnpicklingError as e:
        if pickle_allowed(allow_pickle):
            with open(path, "rb") as f:
                return pickle.load(f)
        raise AmpnetError(f"{path}: not a plain numpy array pickle ({e}). Re-save it with pickle.dump(ndarray), or opt in to "
                          "full unpickling of a file you produced yourself with allow_pickle=True / AMPNET_ALLOW_PICKLE=1.") from e


def load_tensor_list(path, allow_pickle=None):
    """A list of tensors / a tensor (cluster lists, centroids).  torch.save files load with the restricted unpickler; files the
    reference wrote with pickle.dump need the opt-in."""
    try:
        return torch.load(path, weights_only=True)
    except pickle.UnpicklingError as e:
        if pickle_allowed(allow_pickle):
            with open(path, "rb") as f:
                return pickle.load(f)
        raise AmpnetError(f"{path}: torch.load(weights_only=True) refused the file ({e}). Re-save the clusters with torch.save(list_of_tensors, path), "
                          "or opt in to full unpickling of a file you produced yourself with allow_pickle=True / AMPNET_ALLOW_PICKLE=1.") from e
