"""ctypes binding of libampnet_host.so (include/ampnet_host.h): the per-sample host work of the input pipeline, no HIP -- safe to load
in DataLoader workers.  `AMPNET_HOST_LOADER=numpy` keeps the numpy statement of the same steps (pointNet/datasets.py), which is also what
runs when the library has not been built; both produce the same bits (tests/test_host_cpu.py)."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libampnet_host.so")
ABI_VERSION = 1
_lib = None
_tried = False


def lib():
    """The loaded host library, or None (not built / disabled / wrong ABI)."""
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("AMPNET_HOST_LOADER", "native") == "numpy" or not os.path.exists(LIB_PATH):
        return None
    try:
        l = ctypes.CDLL(LIB_PATH)
        l.ampnet_host_abi_version.restype = ctypes.c_int
        if l.ampnet_host_abi_version() != ABI_VERSION:
            return None
        l.ampnet_host_kmeans_sample_f32.restype = ctypes.c_long
        l.ampnet_host_kmeans_sample_f32.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p]
        l.ampnet_host_kmeans_file_ragged_f32.restype = ctypes.c_long
        l.ampnet_host_kmeans_file_ragged_f32.argtypes = [ctypes.c_char_p, ctypes.c_longlong, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                        ctypes.c_void_p, ctypes.c_void_p]
        _lib = l
    except OSError:
        _lib = None
    return _lib
