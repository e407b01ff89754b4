"""Text-embedding connector: HIP path vs the oracle restatement of LTXTextEncoder.swift.

Tolerances: the reference keeps bf16 activations between ops while the HIP path keeps the residual stream and q/k in
f32 until their consumers, so agreement is to bf16 resolution: normalised concat within 1 bf16 ulp, feature extractor and
final context rel-L2 <= 2e-2 / cosine >= 0.9995; the register plan (an index permutation) is exact."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIM, HEADS, LAYERS, REGS, STATES = 256, 2, 2, 8, 5


def bits(x, oracle):
    return oracle.f32_to_bf16_bits(np.asarray(x, np.float32))


@pytest.fixture(scope="module")
def conn(ltx, oracle, gpu_ctx, tmp_path_factory):
    from safetensors.torch import save_file

    w = oracle.synth_connector_weights(dim=DIM, heads=HEADS, layers=LAYERS, registers=REGS, states=STATES, seed=3)
    d = tmp_path_factory.mktemp("connector")
    path = d / "unified.safetensors"
    tensors = {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16) for k, v in oracle.connector_file_keys(w).items()}
    # a unified file also carries DiT / audio tensors the connector loader must skip
    tensors["model.diffusion_model.transformer_blocks.0.attn1.to_q.weight"] = torch.zeros(4, 4, dtype=torch.bfloat16)
    tensors["model.diffusion_model.audio_embeddings_connector.learnable_registers"] = torch.zeros(2, 2, dtype=torch.bfloat16)
    save_file(tensors, str(path))
    cfg = ltx.connector_config(dim=DIM, heads=HEADS, layers=LAYERS, registers=REGS, states=STATES)
    gpu_ctx.connector_load(path, cfg)
    rep = gpu_ctx.load_report()
    assert rep["missing"] == 0 and rep["unmatched"] == 0 and rep["loaded"] == len(w), rep
    yield w
    gpu_ctx.connector_unload()


def _run(gpu_ctx, oracle, hs, am, padding_right=False):
    S, B, T, D = hs.shape
    h = torch.from_numpy(bits(hs, oracle).astype(np.int16)).cuda().view(torch.bfloat16)
    m = torch.from_numpy(am.astype(np.int32)).cuda()
    out = torch.empty((B, T, D), device="cuda", dtype=torch.bfloat16)
    nc = torch.empty((B, T, D * S), device="cuda", dtype=torch.bfloat16)
    fe = torch.empty((B, T, D), device="cuda", dtype=torch.bfloat16)
    reg = torch.empty((B, T, D), device="cuda", dtype=torch.float32)
    gpu_ctx.connector_encode_dev(h, m, out, padding_right=padding_right, taps=(nc, fe, reg))
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), nc.float().cpu().numpy(), fe.float().cpu().numpy(), reg.cpu().numpy()


def _rel(a, b):
    return np.linalg.norm(a - b) / max(1e-9, np.linalg.norm(b))


def _cos(a, b):
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))


@pytest.mark.parametrize("B,T,valid", [(1, 16, [11]), (2, 32, [32, 5]), (1, 64, [1])])
def test_connector_parity(gpu_ctx, oracle, conn, B, T, valid):
    w = conn
    rng = np.random.default_rng(B * 100 + T)
    hs = oracle.bf16_round((rng.standard_normal((STATES, B, T, DIM)) * 2.5 + 0.3).astype(np.float32))
    am = np.zeros((B, T), np.int32)
    for b, n in enumerate(valid):
        am[b, T - n:] = 1  # left padding (reference default)
    ref, ref_mask, inter = oracle.connector_encode(w, hs, am, heads=HEADS, layers=LAYERS, return_intermediates=True)
    got, nc, fe, reg = _run(gpu_ctx, oracle, hs, am)
    # normalised concat: same f32 formula, one bf16 rounding; allow 1 ulp where the f32 statistics differ in the last bit
    d = np.abs(nc - inter["norm_concat"])
    assert d.max() <= 2.0 ** -6 * max(1.0, np.abs(inter["norm_concat"]).max()), d.max()
    assert (d > 0).mean() < 0.02
    assert np.all(nc[am == 0] == 0)
    assert _rel(fe, inter["fe"]) <= 1e-2 and _cos(fe, inter["fe"]) >= 0.9999
    # register replacement is a row permutation + register rows: compare against the oracle plan applied to OUR fe rows
    plan = oracle.replace_padded_with_registers(fe, am.astype(bool), w["embeddings_connector.learnable_registers"].astype(np.float32))
    assert np.array_equal(reg, plan)
    assert _rel(got, ref) <= 2e-2 and _cos(got, ref) >= 0.9995, (_rel(got, ref), _cos(got, ref))
    assert ref_mask.all()


def test_connector_host_entry_and_mask(gpu_ctx, oracle, conn):
    w = conn
    rng = np.random.default_rng(9)
    B, T = 1, 24
    hs = oracle.bf16_round(rng.standard_normal((STATES, B, T, DIM)).astype(np.float32))
    am = np.zeros((B, T), np.int32)
    am[0, 10:] = 1
    out_bits, om = gpu_ctx.connector_encode(bits(hs, oracle), am)
    assert om.shape == (B, T) and om.all()
    got = oracle.bf16_bits_to_f32(out_bits)
    ref, _ = oracle.connector_encode(w, hs, am, heads=HEADS, layers=LAYERS)
    assert _rel(got, ref) <= 2e-2


def test_connector_right_padding_follows_reference(gpu_ctx, oracle, conn):
    """With right padding the reference's register plan keeps padded rows in the tail (LTXTextEncoder.swift:440-462 assumes
    left padding); the port reproduces that behaviour instead of 'fixing' it."""
    w = conn
    rng = np.random.default_rng(4)
    B, T = 1, 16
    hs = oracle.bf16_round(rng.standard_normal((STATES, B, T, DIM)).astype(np.float32))
    am = np.zeros((B, T), np.int32)
    am[0, :6] = 1
    ref, _ = oracle.connector_encode(w, hs, am, padding_side="right", heads=HEADS, layers=LAYERS)
    got, _, _, _ = _run(gpu_ctx, oracle, hs, am, padding_right=True)
    assert _rel(got, ref) <= 2e-2


def test_connector_errors(ltx, gpu_ctx, oracle, conn):
    h = torch.zeros((STATES, 1, 12, DIM), device="cuda", dtype=torch.bfloat16)
    m = torch.ones((1, 12), device="cuda", dtype=torch.int32)
    out = torch.empty((1, 12, DIM), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ltx.LTXError) as e:  # 12 % 8 != 0: fatalError in the reference
        gpu_ctx.connector_encode_dev(h, m, out)
    assert "divisible by numLearnableRegisters" in str(e.value)


def test_connector_feeds_the_dit(ltx, oracle, gpu_ctx, conn):
    """The connector's output is exactly what ltx_dit_forward takes as context (shape/dtype contract)."""
    rng = np.random.default_rng(2)
    B, T = 1, 16
    hs = oracle.bf16_round(rng.standard_normal((STATES, B, T, DIM)).astype(np.float32))
    am = np.ones((B, T), np.int32)
    out_bits, om = gpu_ctx.connector_encode(bits(hs, oracle), am)
    cfg = ltx.default_transformer_config(num_layers=1, num_attention_heads=2, cross_attention_dim=256, caption_channels=DIM)
    gpu_ctx.dit_init_synthetic(cfg, seed=7)
    lat = bits(rng.standard_normal((1, 2 * 2 * 2, 128)).astype(np.float32), oracle)
    vel = gpu_ctx.dit_forward(lat, out_bits, np.array([0.5], np.float32), om, 2, 2, 2)
    assert vel.shape == (1, 8, 128) and np.isfinite(vel).all()
    gpu_ctx.dit_unload()


def test_connector_vs_golden_fixture(ltx, oracle, gpu_ctx, tmp_path):
    """HIP path vs the committed golden vector (tests/golden/connector_tiny.npz: oracle output cross-checked against an
    independent torch implementation by make_golden.py)."""
    import os
    from safetensors.torch import save_file

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "connector_tiny.npz"))
    dim, heads, layers, regs, states = (int(g[k]) for k in ("dim", "heads", "layers", "registers", "states"))
    w = oracle.synth_connector_weights(dim=dim, heads=heads, layers=layers, registers=regs, states=states, seed=int(g["seed"]))
    path = tmp_path / "c.safetensors"
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16) for k, v in oracle.connector_file_keys(w, unified=False).items()},
              str(path))
    gpu_ctx.connector_load(path, ltx.connector_config(dim=dim, heads=heads, layers=layers, registers=regs, states=states))
    out_bits, om = gpu_ctx.connector_encode(bits(g["hidden"], oracle), g["mask"])
    gpu_ctx.connector_unload()
    got = oracle.bf16_bits_to_f32(out_bits)
    assert om.all() and _rel(got, g["context"]) <= 2e-2 and _cos(got, g["context"]) >= 0.9995
