"""CPU tests: the oracle's ops cross-checked against an independent implementation (torch CPU), since the reference
itself cannot run here. Also pins the oracle to the committed golden fixtures."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_bf16_round_matches_torch(oracle):
    x = np.random.default_rng(0).standard_normal(100000).astype(np.float32) * 37.0
    assert np.array_equal(oracle.bf16_round(x), torch.from_numpy(x).to(torch.bfloat16).float().numpy())


def test_rms_layer_norm(oracle):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((7, 512)).astype(np.float32) * 3
    w = rng.standard_normal(512).astype(np.float32)
    ref = Fn.rms_norm(torch.from_numpy(x), (512,), torch.from_numpy(w), eps=1e-6).numpy()
    assert np.allclose(oracle.rms_norm(x, w, 1e-6), ref, rtol=1e-5, atol=1e-6)
    ref = Fn.layer_norm(torch.from_numpy(x), (512,), eps=1e-6).numpy()
    assert np.allclose(oracle.layer_norm(x, 1e-6), ref, rtol=1e-5, atol=1e-5)


def test_activations(oracle):
    x = torch.linspace(-6, 6, 1001)
    assert np.allclose(oracle.gelu_tanh(x.numpy()), Fn.gelu(x, approximate="tanh").numpy(), atol=1e-6)
    assert np.allclose(oracle.silu(x.numpy()), Fn.silu(x).numpy(), atol=1e-6)


def test_sdpa(oracle):
    rng = np.random.default_rng(2)
    B, H, Tq, Tk = 2, 3, 37, 53
    q, k, v = (rng.standard_normal((B, t, H * 128)).astype(np.float32) for t in (Tq, Tk, Tk))
    bias = ((rng.random((B, Tk)) > 0.7) * -10000.0).astype(np.float32)
    got = oracle.sdpa(q, k, v, H, 1 / math.sqrt(128), bias)
    tq, tk, tv = (torch.from_numpy(a).reshape(B, -1, H, 128).transpose(1, 2) for a in (q, k, v))
    ref = Fn.scaled_dot_product_attention(tq, tk, tv, attn_mask=torch.from_numpy(bias)[:, None, None, :], scale=1 / math.sqrt(128))
    assert np.allclose(got, ref.transpose(1, 2).reshape(B, Tq, H * 128).numpy(), atol=2e-5)


def test_split_rope_is_rotation(oracle):
    F, H, W, heads = 2, 3, 4, 2
    cos, sin = oracle.rope_tables(F, H, W, dim=heads * 128, num_heads=heads)
    x = np.random.default_rng(3).standard_normal((1, F * H * W, heads * 128)).astype(np.float32)
    y = oracle.apply_split_rope(x, cos, sin, heads)
    # norms of (a_i, b_i) pairs are preserved; identity slots (first 2 of head 0) untouched
    xr, yr = x.reshape(1, -1, heads, 2, 64), y.reshape(1, -1, heads, 2, 64)
    assert np.allclose((xr ** 2).sum(3), (yr ** 2).sum(3), rtol=1e-5)
    assert np.array_equal(xr[:, :, 0, :, :2], yr[:, :, 0, :, :2])


@pytest.mark.parametrize("causal", [False, True])
def test_conv3d_full(oracle, causal):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 8, 3, 5, 6)).astype(np.float32)
    w = rng.standard_normal((12, 8, 3, 3, 3)).astype(np.float32)
    b = rng.standard_normal(12).astype(np.float32)
    xt = torch.from_numpy(x)
    xp = Fn.pad(xt.reshape(1, 8 * 3, 5, 6), (1, 1, 1, 1), mode="reflect").reshape(1, 8, 3, 7, 8)
    xp = torch.cat([xp[:, :, :1]] * 2 + [xp], 2) if causal else torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
    ref = Fn.conv3d(xp, torch.from_numpy(w), torch.from_numpy(b)).numpy()
    assert np.allclose(oracle.conv3d_full(x, w, b, causal), ref, atol=1e-4)
    assert np.allclose(oracle.conv3d_full_einsum(x, w, b, causal), ref, atol=1e-4)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape,cout", [((2, 8, 1, 2, 2), 4), ((1, 16, 3, 5, 6), 8), ((1, 64, 4, 16, 24), 96), ((1, 32, 2, 40, 36), 5)])
def test_conv3d_blas_form_equals_the_einsum_form(oracle, shape, cout, causal):
    """conv3d_full (one in-place sgemm per tap on the flattened padded grid, threaded padding) against the first restatement
    (27 patch copies + einsum): same taps in the same order, so only the BLAS-internal summation order differs."""
    rng = np.random.default_rng(sum(shape) + cout)
    x = rng.standard_normal(shape).astype(np.float32)
    w = rng.standard_normal((cout, shape[1], 3, 3, 3)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    a, c = oracle.conv3d_full_einsum(x, w, b, causal), oracle.conv3d_full(x, w, b, causal)
    assert a.shape == c.shape and float(np.abs(a - c).max()) <= 2e-6 * float(np.abs(a).max())
    assert np.array_equal(oracle.conv3d_full(x, w, None, causal), oracle.conv3d_full(x, w, np.zeros(cout, np.float32), causal))


@pytest.mark.parametrize("per_frame", [False, True])
def test_conv_nd_zero_and_causal_zero_vs_torch(oracle, per_frame):
    """The zero-padded convs of the latent upscaler (SpatialUpscaler.swift:78-92,139-145) and of the VAE encoder
    (VideoConvolution.swift:238-347), BLAS-speed forms, against torch conv3d / conv2d."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 6, 3, 5, 7)).astype(np.float32)
    b = rng.standard_normal(10).astype(np.float32)
    if per_frame:
        w = rng.standard_normal((10, 6, 3, 3)).astype(np.float32)
        ref = torch.stack([Fn.conv2d(torch.from_numpy(x[:, :, f]), torch.from_numpy(w), torch.from_numpy(b), padding=1) for f in range(3)], 2).numpy()
        assert np.allclose(oracle.conv_nd_zero(x, w, b), ref, atol=1e-4)
        return
    w = rng.standard_normal((10, 6, 3, 3, 3)).astype(np.float32)
    ref = Fn.conv3d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1).numpy()
    assert np.allclose(oracle.conv_nd_zero(x, w, b), ref, atol=1e-4)
    for causal in (True, False):
        xt = torch.from_numpy(x)
        xp = Fn.pad(xt, (1, 1, 1, 1, 0, 0))
        xp = torch.cat([xp[:, :, :1]] * 2 + [xp], 2) if causal else torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
        ref = Fn.conv3d(xp, torch.from_numpy(w), torch.from_numpy(b)).numpy()
        assert np.allclose(oracle.conv3d_causal_zero(x, w, b, causal), ref, atol=1e-4)


def test_threaded_passes_do_not_change_a_bit(oracle):
    """The oracle runs its row-wise passes (norms, GELU, RoPE, softmax per head, PixelNorm + SiLU) slab by slab on a thread pool above a
    size threshold. Forcing the slab path on small inputs must reproduce the inline results exactly, for the DiT and for the decoder."""
    rng = np.random.default_rng(3)
    cfg = oracle.DiTConfig(num_layers=2, num_heads=4, caption_channels=64)
    w = oracle.synth_dit_weights(cfg, seed=5)
    F, H, W, S = 3, 8, 12, 40
    lat = oracle.bf16_round(rng.standard_normal((2, F * H * W, 128)).astype(np.float32))
    cx = oracle.bf16_round(rng.standard_normal((2, S, 64)).astype(np.float32))
    mask = (rng.random((2, S)) > 0.2).astype(np.int32)
    mask[:, 0] = 1
    wv = oracle.synth_vae_weights(channels=(64, 32, 16, 8), seed=9)
    vlat = rng.standard_normal((1, 128, 2, 3, 4)).astype(np.float32)
    keep = oracle._PMAP_MIN
    try:
        oracle._PMAP_MIN = 1 << 40
        a = oracle.dit_forward(w, cfg, lat, cx, np.array([0.7, 0.3], np.float32), mask, F, H, W)
        va = oracle.decode_video(wv, vlat, channels=(64, 32, 16, 8))
        oracle._PMAP_MIN = 16
        b = oracle.dit_forward(w, cfg, lat, cx, np.array([0.7, 0.3], np.float32), mask, F, H, W)
        vb = oracle.decode_video(wv, vlat, channels=(64, 32, 16, 8))
    finally:
        oracle._PMAP_MIN = keep
    assert np.array_equal(a, b) and np.array_equal(va, vb)


def test_depth_to_space_and_unpatchify_index_maps(oracle):
    # D2S: out[c, 2t+dt, 2h+dh, 2w+dw] = in[((c*2+dt)*2+dh)*2+dw, t, h, w]  (VideoDecoder.swift:201-213)
    x = np.arange(16 * 2 * 3 * 4, dtype=np.float32).reshape(1, 16, 2, 3, 4)
    y = oracle.depth_to_space(x, 2)
    for c, t, h, w, dt, dh, dw in [(0, 0, 0, 0, 0, 0, 0), (1, 1, 2, 3, 1, 0, 1), (0, 1, 1, 2, 0, 1, 1)]:
        assert y[0, c, 2 * t + dt, 2 * h + dh, 2 * w + dw] == x[0, ((c * 2 + dt) * 2 + dh) * 2 + dw, t, h, w]
    # unpatchify: channel (c*4+a)*4+b -> (H index 4h+b, W index 4w+a)  (VideoDecoder.swift:257-275)
    z = np.arange(48 * 2 * 2 * 3, dtype=np.float32).reshape(1, 48, 2, 2, 3)
    u = oracle.vae_unpatchify(z, 4)
    for c, a, b, t, h, w in [(0, 0, 0, 0, 0, 0), (2, 3, 1, 1, 1, 2), (1, 0, 2, 0, 1, 1)]:
        assert u[0, c, t, 4 * h + b, 4 * w + a] == z[0, (c * 4 + a) * 4 + b, t, h, w]
    assert u.shape == (1, 3, 2, 8, 12)


def test_guidance_and_adain(oracle):
    rng = np.random.default_rng(5)
    u, c = (rng.standard_normal((1, 8, 2, 3, 4)).astype(np.float32) for _ in range(2))
    v = oracle.apply_cfg(u, c, 4.0)
    assert np.allclose(v, u + 4.0 * (c - u), atol=1e-5)
    r = oracle.guidance_rescale(v, c, 1.0)
    assert abs(r.std() - c.std()) < 1e-3 * c.std()
    ref = rng.standard_normal((1, 8, 1, 2, 2)).astype(np.float32) * 3 + 1
    a = oracle.adain_filter_latent(v, ref)
    assert np.allclose(a.mean((2, 3, 4)), ref.mean((2, 3, 4)), atol=1e-4)
    assert np.allclose(a.std((2, 3, 4)), ref.std((2, 3, 4)), rtol=1e-4)


def test_euler_step(oracle):
    x = np.array([1.0, -2.0], np.float32)
    v = np.array([0.5, 0.25], np.float32)
    assert np.allclose(oracle.euler_step(x, v, 0.8, 0.0), x - 0.8 * v)
    # x + (sigma_next - sigma) * v
    assert np.allclose(oracle.euler_step(x, v, 0.8, 0.3), x + (0.3 - 0.8) * v, atol=1e-6)


def test_oracle_matches_golden_fixtures(oracle):
    """Fixtures are produced by tests/golden/make_golden.py (committed with them): regression pin for the oracle."""
    g = np.load(os.path.join(GOLD, "dit_tiny.npz"))
    ocfg = oracle.DiTConfig(num_layers=int(g["layers"]), num_heads=int(g["heads"]), caption_channels=int(g["caption"]))
    w = oracle.synth_dit_weights(ocfg, seed=int(g["seed"]))
    F, H, W = (int(v) for v in g["fhw"])
    vel = oracle.dit_forward(w, ocfg, g["latent"], g["context"], g["ts"], g["mask"], F, H, W)
    assert np.allclose(vel, g["velocity"], rtol=1e-4, atol=1e-4)
    c = np.load(os.path.join(GOLD, "conv3d_small.npz"))
    assert np.allclose(oracle.conv3d_full(c["x"], c["w"], c["b"]), c["y"], atol=1e-4)
    assert np.allclose(oracle.conv3d_full_einsum(c["x"], c["w"], c["b"]), c["y"], atol=1e-4)
    k = np.load(os.path.join(GOLD, "connector_tiny.npz"))
    wk = oracle.synth_connector_weights(dim=int(k["dim"]), heads=int(k["heads"]), layers=int(k["layers"]), registers=int(k["registers"]),
                                        states=int(k["states"]), seed=int(k["seed"]))
    ctxv, om = oracle.connector_encode(wk, k["hidden"], k["mask"], heads=int(k["heads"]), layers=int(k["layers"]))
    assert np.allclose(ctxv, k["context"], atol=1e-6) and om.all()
    e = np.load(os.path.join(GOLD, "vae_encoder_tiny.npz"))
    we = oracle.synth_vae_encoder_weights(base=int(e["base"]), seed=int(e["seed"]))
    assert np.allclose(oracle.vae_encode(we, e["pixels"], base=int(e["base"])), e["latent"], rtol=1e-4, atol=1e-5)


def test_denoise_text_cache_is_bit_neutral(oracle):
    """oracle.denoise keeps the caption projection and the cross-attention K / V of a text context across the steps of a loop (they do
    not depend on the step; the GPU suite's longest tests are bounded by the oracle's host time). The loop must equal, bit for bit, the same
    steps done by hand through dit_forward without a cache - CFG pair included (two contexts, two caches)."""
    cfg = oracle.DiTConfig(num_layers=2, num_heads=2, caption_channels=64)
    w = oracle.synth_dit_weights(cfg, seed=3)
    rng = np.random.default_rng(0)
    F, H, W, S = 1, 2, 3, 5
    lat = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    pos = oracle.bf16_round(rng.standard_normal((1, S, 64)).astype(np.float32))
    neg = oracle.bf16_round(rng.standard_normal((1, S, 64)).astype(np.float32))
    sig = oracle.sigmas(False, 3, F * H * W)
    got = oracle.denoise(w, cfg, lat * sig[0], sig, pos, None, F, H, W, cfg_scale=3.0, neg_context=neg)
    x = lat * sig[0]
    for st in range(len(sig) - 1):
        tok = oracle.bf16_round(oracle.patchify(x))
        ts = np.array([sig[st]], np.float32)
        vp = oracle.unpatchify(oracle.dit_forward(w, cfg, tok, pos, ts, None, F, H, W), F, H, W).astype(np.float32)
        vn = oracle.unpatchify(oracle.dit_forward(w, cfg, tok, neg, ts, None, F, H, W), F, H, W).astype(np.float32)
        x = oracle.euler_step(x, oracle.apply_cfg(vn, vp, 3.0), float(sig[st]), float(sig[st + 1]))
    assert np.array_equal(got, x)
