import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "3d-semantic-segmentation-amp-net_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sub(name):
    """import <package>.<name> (the package directory name is not a python identifier)."""
    return importlib.import_module(PKG + ("." + name if name else ""))


@pytest.fixture(scope="session")
def synth():
    return sub("synthetic")


@pytest.fixture(scope="session")
def params():
    return sub("params")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    return load


# ---- matrix precision of the fp32 parity tests ---------------------------------------------------------------------------------
# The modules below hold the parity tests of the fp32 path that reach the per-point layers (pw_gemm / the fused backward).  Every
# GPU test in them runs twice: in AMPNET_PRECISION_F32 (exact fp32 MFMA) and in AMPNET_PRECISION_F32_SPLIT ("f32x3": fp32 results from
# six bf16 MFMAs per product, include/ampnet_hip.h) -- with the SAME bars, which is what admits the split mode as an fp32 path.
# AMPNET_TEST_PRECISION=fp32|f32x3 restricts the run to one of them (development).
SPLIT_PARITY_MODULES = {"test_forward_gpu", "test_backward_gpu", "test_step_gpu", "test_fullsize_gpu", "test_edge_shapes_gpu",
                        "test_inference_gpu", "test_gru_gpu", "test_cls_gpu", "test_cli_gpu"}


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in SPLIT_PARITY_MODULES and metafunc.definition.get_closest_marker("gpu") is not None:
        only = os.environ.get("AMPNET_TEST_PRECISION")
        metafunc.parametrize("matrix_precision_mode", [only] if only else ["fp32", "f32x3"], indirect=True)


@pytest.fixture(autouse=True)
def matrix_precision_mode(request):
    """Sets the library's matrix precision for a parametrised parity test and puts fp32 back afterwards; every other test (and every
    CPU test: they never load the library) passes straight through."""
    mode = getattr(request, "param", None)
    if mode is None:
        yield "fp32"
        return
    L = sub("_lib")
    L.set_matrix_precision(mode)
    try:
        yield mode
    finally:
        L.set_matrix_precision("fp32")
