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


# ---- torch.save'd single tensors (the kmeans_<name>.pt training samples) -----------------------------------------------------------------
# torch.load(weights_only=True) runs a pure-Python unpickler: 0.6 ms per 460 KB sample, a third of what a DataLoader worker spends on it.
# A single saved tensor needs three globals; the C unpickler restricted to exactly those (nothing from the file is imported or called:
# the storage class becomes a dtype name, the rebuild function a tuple) + one read of the storage record is 10x cheaper and as safe.
_PT_STORAGE_DTYPES = {"FloatStorage": "float32", "DoubleStorage": "float64", "LongStorage": "int64", "IntStorage": "int32",
                      "ShortStorage": "int16", "ByteStorage": "uint8", "CharStorage": "int8"}


def _pt_rebuild_tensor(storage, storage_offset, size, stride, *unused):
    return ("tensor", storage, int(storage_offset), tuple(int(v) for v in size), tuple(int(v) for v in stride))


class _SingleTensorUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _pt_rebuild_tensor
        if module == "torch" and name in _PT_STORAGE_DTYPES:
            return _PT_STORAGE_DTYPES[name]
        if module == "collections" and name == "OrderedDict":
            import collections
            return collections.OrderedDict
        raise pickle.UnpicklingError(f"global {module}.{name} is not part of a plain saved tensor")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], str)):
            raise pickle.UnpicklingError("unexpected persistent id")
        return ("storage", pid[1], str(pid[2]), int(pid[4]))          # dtype name, record key, element count


def pt_tensor_header(path):
    """(numpy dtype name, shape, byte offset of the first element in the file) of a torch.save'd single contiguous tensor; raises
    pickle.UnpicklingError for anything else.  Nothing from the file is executed and the data is not read."""
    import io
    reader = torch._C.PyTorchFileReader(path)
    desc = _SingleTensorUnpickler(io.BytesIO(reader.get_record("data.pkl"))).load()
    if not (isinstance(desc, tuple) and len(desc) == 5 and desc[0] == "tensor"):
        raise pickle.UnpicklingError("not a single tensor")
    _, (_, dtype, key, numel), off, size, stride = desc
    want, acc = [], 1
    for d in reversed(size):
        want.append(acc)
        acc *= d
    if tuple(reversed(want)) != stride and acc > 0:
        raise pickle.UnpicklingError("not contiguous")
    if off < 0 or off + acc > numel:
        raise pickle.UnpicklingError("view outside its storage")
    if reader.has_record("byteorder") and reader.get_record("byteorder") != b"little":
        raise pickle.UnpicklingError("big-endian file")
    itemsize = {"float32": 4, "float64": 8, "int64": 8, "int32": 4, "int16": 2, "uint8": 1, "int8": 1}[dtype]
    return dtype, size, reader.get_record_offset("data/" + key) + off * itemsize


def _load_pt_array_fast(path):
    import numpy as np
    dtype, size, start = pt_tensor_header(path)
    count = 1
    for d in size:
        count *= d
    return np.fromfile(path, dtype=np.dtype(dtype), count=count, offset=start).reshape(size)


def load_pt_array(path):
    """A torch.save'd tensor as a numpy array (read once; nothing from the file is executed).  Anything the fast reader does not
    recognise -- legacy non-zip files, several objects, non-contiguous views -- goes through torch.load(weights_only=True)."""
    import numpy as np
    try:
        return _load_pt_array_fast(path)
    except Exception:                                    # noqa: BLE001 -- any surprise means "take the general loader"
        pass
    try:                                                 # the file mapped, not read: the rows that survive are copied once by the caller
        t = torch.load(path, map_location=torch.device("cpu"), weights_only=True, mmap=True)
    except (RuntimeError, ValueError, TypeError):        # legacy (non-zip) torch.save files cannot be mapped
        t = torch.load(path, map_location=torch.device("cpu"), weights_only=True)
    return np.asarray(t)
