"""Denoise-loop parity (patchify -> DiT -> unpatchify -> CFG / rescale / STG / GE -> Euler), libltxhip.so vs oracle.

The loop amplifies the per-forward bf16 deviation step by step; tolerance on the final latent after the full
schedule: rel-L2 <= 3e-2 (8 steps, reduced-depth model). Index logic (token order, [neg,pos] batch order, progress
callback sequence) is exact.
"""
import numpy as np
import pytest

from test_dit_gpu import rel_l2, small_cfg, write_dit_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(ltx, oracle, gpu_ctx, tmp_path_factory):
    cfg, ocfg = small_cfg(ltx, oracle, heads=4, layers=3, caption=256)
    w = oracle.synth_dit_weights(ocfg, seed=21)
    path = tmp_path_factory.mktemp("dn") / "dit.safetensors"
    write_dit_file(oracle, w, path)
    gpu_ctx.dit_load(path, cfg)
    return cfg, ocfg, w


def _inputs(oracle, ocfg, F, H, W, S, seed, nb=1):
    rng = np.random.default_rng(seed)
    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    ctx = oracle.bf16_round(rng.standard_normal((nb, S, ocfg.caption_channels)).astype(np.float32))
    return noise, ctx


def test_denoise_distilled_no_cfg(ltx, oracle, gpu_ctx, model):
    cfg, ocfg, w = model
    F, H, W, S = 2, 4, 6, 40
    noise, ctx = _inputs(oracle, ocfg, F, H, W, S, 1)
    sig = ltx.sigmas(True, 8, F * H * W)
    lat0 = noise * sig[0]
    seen = []
    got = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx), None, F, H, W, on_progress=lambda s, t, sg, u: seen.append((s, t, sg)))
    ref = oracle.denoise(w, ocfg, lat0, sig, ctx, None, F, H, W)
    assert [s for s, _, _ in seen] == list(range(8)) and all(t == 8 for _, t, _ in seen)
    assert np.allclose([sg for _, _, sg in seen], sig[:8])
    assert rel_l2(got, ref) <= 3e-2, rel_l2(got, ref)


def test_denoise_cfg_rescale_stg_ge(ltx, oracle, gpu_ctx, model):
    """dev-style schedule with CFG 4.0 ([neg,pos] batch), guidance rescale 0.7, STG on block 1, GE momentum."""
    cfg, ocfg, w = model
    F, H, W, S = 1, 4, 4, 24
    noise, ctx2 = _inputs(oracle, ocfg, F, H, W, S, 2, nb=2)
    rng = np.random.default_rng(5)
    mask = (rng.random((2, S)) > 0.2).astype(np.int32)
    mask[:, 0] = 1
    sig = ltx.sigmas(False, 4, F * H * W)
    lat0 = noise * sig[0]
    got = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx2), mask, F, H, W, cfg_scale=4.0, guidance_rescale=0.7,
                          stg_scale=1.0, stg_blocks=(1,), ge_gamma=0.5)
    ref = oracle.denoise(w, ocfg, lat0, sig, ctx2[1:2], mask[1:2], F, H, W, cfg_scale=4.0, rescale=0.7, stg_scale=1.0,
                         stg_blocks=(1,), ge_gamma=0.5, neg_context=ctx2[0:1], neg_mask=mask[0:1])
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) <= 5e-2, rel_l2(got, ref)


def test_denoise_single_step_bitwise_repeatable(ltx, oracle, gpu_ctx, model):
    cfg, ocfg, w = model
    F, H, W, S = 2, 3, 5, 17
    noise, ctx = _inputs(oracle, ocfg, F, H, W, S, 3)
    sig = np.array([0.9, 0.5], np.float32)
    a = gpu_ctx.denoise(noise, sig, ltx.f32_to_bf16_bits(ctx), None, F, H, W)
    b = gpu_ctx.denoise(noise, sig, ltx.f32_to_bf16_bits(ctx), None, F, H, W)
    assert np.array_equal(a, b)
    ref = oracle.denoise(w, ocfg, noise, sig, ctx, None, F, H, W)
    assert rel_l2(a, ref) <= 1e-2
