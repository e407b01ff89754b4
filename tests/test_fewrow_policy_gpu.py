"""Few-row launches (config 1: 256x256x9 -> 128 tokens; LTXTransformerBlock.swift:187-232 is the graph being run): the weight loads'
cache policy is an A/B hook of the library (LTX_B_NT, read once per process) and must not change a bit of the forward - one forward per
setting, each in a process of its own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hash(env_extra, F, H, W, layers):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "forward_hash.py"), str(F), str(H), str(W), str(layers)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    return [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("HASH")][0]


@pytest.mark.parametrize("F,H,W", [(2, 8, 8), (1, 5, 7)])  # 128 tokens (config 1); 35 tokens (ragged against every tile)
def test_non_temporal_weight_loads_keep_the_bits(F, H, W):
    nt = _hash({}, F, H, W, 6)
    assert nt == _hash({"LTX_B_NT": "0"}, F, H, W, 6)
    assert nt == _hash({"LTX_B_NT": "1"}, F, H, W, 6)
