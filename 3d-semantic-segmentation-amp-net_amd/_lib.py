"""ctypes binding of libampnet_hip.so (include/ampnet_hip.h).

There is no CPU fallback: if the library is missing, fails to load, or reports an error, the caller
gets an exception.  torch is imported first so that the HIP runtime torch already mapped
(its bundled libamdhip64, SONAME libamdhip64.so.7) is the one this library binds to -- two HIP runtimes in
one process would not share device pointers.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMPNET_LIB_PATH") or os.path.join(_HERE, "libampnet_hip.so")   # the override is for A/B runs of two builds
ABI_VERSION = 4

_lib = None


class AmpnetError(RuntimeError):
    pass


def _hip_runtimes_mapped():
    seen = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    seen.add(line.split()[-1])
    except OSError:
        pass
    return seen


def lib():
    """The loaded library (loads on first use; raises AmpnetError when it cannot)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmpnetError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                          "There is no CPU fallback for the HIP path.")
    try:
        l = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise AmpnetError(f"cannot load {LIB_PATH}: {e}") from e
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise AmpnetError(f"two HIP runtimes mapped in one process: {sorted(rts)}")
    l.ampnet_abi_version.restype = ctypes.c_int
    l.ampnet_last_error.restype = ctypes.c_char_p
    v = l.ampnet_abi_version()
    if v != ABI_VERSION and not (os.environ.get("AMPNET_LIB_PATH") and v < ABI_VERSION):
        # (an OLDER build named explicitly through AMPNET_LIB_PATH is accepted: same-box A/B runs against a previous round's library,
        # tools/ab_lib.sh -- entry points added since then are simply absent from it)
        raise AmpnetError(f"libampnet_hip.so ABI {v} != expected {ABI_VERSION}: rebuild")
    _lib = l
    return l


PRECISIONS = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "bf16_train": 2, "bf16_store": 3,
              "f32x3": 4, "fp32_split": 4}


def set_matrix_precision(mode):
    """'fp32' (default, exact fp32 products: the mode the parity figures hold in), 'bf16' (the forward per-point layers
    round their MFMA operands to bf16, fp32 accumulation) 'bf16_train' (forward and the fused backward of those layers) or 'bf16_store' (bf16_train + the
    activations kept for the backward stored as bf16) or 'f32x3' (fp32 results from the bf16 matrix pipe: three-term bf16 split of
    both operands of the MFMA-bound layers, six exact partial products, fp32 accumulation; include/ampnet_hip.h: ampnet_set_matrix_precision)."""
    if mode not in PRECISIONS:
        raise AmpnetError(f"unknown matrix precision {mode!r}: one of {sorted(PRECISIONS)}")
    check(lib().ampnet_set_matrix_precision(PRECISIONS[mode]), "ampnet_set_matrix_precision")


def get_matrix_precision():
    return {0: "fp32", 1: "bf16", 2: "bf16_train", 3: "bf16_store", 4: "f32x3"}[lib().ampnet_get_matrix_precision()]


def check(rc, what):
    if rc != 0:
        msg = lib().ampnet_last_error()
        raise AmpnetError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_contiguous():
        raise AmpnetError("tensor handed to the C ABI must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """Current torch HIP stream as a void* for the C ABI."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t, name):
    if not t.is_cuda:
        raise AmpnetError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
