"""The pinning kit (tools/make_pinning_case.py): a seed-defined case written with numpy only, its per-step
diagnostics from the oracle in the reference's own --profile format (LTXPipeline.swift:945-951), and the same lines from libltxhip.so.
CPU: the kit is deterministic and its tiny instance reproduces the committed fixture (tests/golden/pinning_tiny.json), the file it writes is
standard safetensors. GPU: the HIP path run on the kit's files agrees with the oracle's lines to the fourth decimal."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "make_pinning_case.py")
TINY = ["--layers", "2", "--heads", "2", "--caption", "128", "--width", "64", "--height", "64", "--frames", "9", "--text-keys", "24", "--seed", "42"]


def test_kit_is_deterministic_and_matches_the_committed_fixture(tmp_path):
    r = subprocess.run([sys.executable, TOOL, "--out", str(tmp_path)] + TINY, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = json.load(open(tmp_path / "expected_oracle.json"))
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "pinning_tiny.json")))
    assert got["latent"] == want["latent"] == [2, 2, 2] and got["sigmas"] == want["sigmas"]
    assert np.allclose(got["step_stats"], want["step_stats"], rtol=0, atol=2e-6)
    lines = open(tmp_path / "expected_oracle.txt").read().splitlines()
    assert len(lines) == 9 and lines[0].startswith("  Step 0: σ=1.0000→0.9927, vel mean=") and lines[7].startswith("  Step 7: σ=0.1000→0.0000")
    # the weight file is plain safetensors with bf16 tensors under the unified checkpoint's names
    from safetensors.torch import load_file
    import torch

    t = load_file(str(tmp_path / "ltx_transformer.safetensors"))
    assert t["model.diffusion_model.proj_in.weight"].dtype == torch.bfloat16 and len(t) == 65
    c = load_file(str(tmp_path / "case.safetensors"))
    assert tuple(c["noise"].shape) == (1, 128, 2, 2, 2) and tuple(c["prompt_embeddings"].shape) == (1, 24, 128)


@pytest.mark.gpu
def test_hip_path_reproduces_the_kit_lines(tmp_path):
    r = subprocess.run([sys.executable, TOOL, "--out", str(tmp_path), "--hip"] + TINY, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.load(open(tmp_path / "expected_oracle.json"))
    h = json.load(open(tmp_path / "expected_hip.json"))
    err = np.abs(np.asarray(o["step_stats"]) - np.asarray(h["step_stats"]))
    print("pinning kit, tiny case: max |HIP - oracle| per column", err.max(0))
    assert err.max() <= 2e-3, err.max(0)
    assert abs(o["final_mean"] - h["final_mean"]) <= 2e-3 and abs(o["final_std"] - h["final_std"]) <= 5e-3
