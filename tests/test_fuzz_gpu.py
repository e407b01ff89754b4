"""A short seeded run of tools/fuzz_kernels.py: random shapes of the gated-residual GEMM (ragged M and N, 1..40 K-tiles: every
path of the residual prefetch) and of the attention dispatcher against f32 references, and (round 5) of the conv3d launcher against torch's conv3d on integer
data. The tool itself runs longer sweeps."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes(gpu_ctx):
    spec = importlib.util.spec_from_file_location("fuzz_kernels", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(gpu_ctx, 16, 11, verbose=False) == 0
    assert mod.run_conv(gpu_ctx, 12, 5, verbose=False) == 0   # conv3d launcher, integer-exact against torch (half of the cases: tall tiles)
