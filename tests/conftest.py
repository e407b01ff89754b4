"""Shared fixtures. `-m "not gpu"` covers the oracle, the host logic and the ABI surface; `-m gpu` runs the parity
tests proper through libltxhip.so on a real MI355X."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: kernels outside the product library; needs the -DLTX_EXPERIMENTS build (LTX_LIB) and a GPU")


def _ensure_built():
    so = os.path.join(ROOT, "ltx-video-swift-mlx_amd", "csrc", "build", "libltxhip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g

        g.build()


@pytest.fixture(scope="session")
def ltx():
    _ensure_built()
    mod = importlib.import_module("ltx-video-swift-mlx_amd")
    sys.modules.setdefault("ltx_amd", mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    import ltx_oracle

    return ltx_oracle


@pytest.fixture(scope="session")
def gpu_ctx(ltx):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    ctx = ltx.Context(0)
    yield ctx
    ctx.close()
