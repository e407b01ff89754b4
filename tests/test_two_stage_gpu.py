"""R23: latent upscaler (zero-padded conv3d/conv2d + GroupNorm + pixel shuffle), AdaIN, and the two-stage glue
(generateVideoTwoStage) through the C ABI vs the oracle. The upscaler runs f32 in the reference; the HIP path feeds
bf16 activations/weights to the MFMA: rel-L2 <= 3e-2 on the upscaled latent."""
import json

import numpy as np
import pytest
import torch

from test_dit_gpu import rel_l2, small_cfg, write_dit_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vae_stats_ctx(ltx, oracle, tmp_path_factory):
    """A context with a VAE whose only non-trivial parameters are the per-channel statistics (enough for the
    upscaler's denormalise/renormalise)."""
    from safetensors.numpy import save_file

    rng = np.random.default_rng(0)
    mean = (0.2 * rng.standard_normal(128)).astype(np.float32)
    std = (1.0 + 0.3 * rng.random(128)).astype(np.float32)
    d = tmp_path_factory.mktemp("vstats")
    save_file({"latents_mean": mean, "latents_std": std}, str(d / "vae.safetensors"))
    c = ltx.Context(0)
    c.vae_load(d / "vae.safetensors")
    yield c, mean, std
    c.close()


def test_upscale_latent_parity(ltx, oracle, vae_stats_ctx, tmp_path):
    from safetensors.numpy import save_file

    ctx, mean, std = vae_stats_ctx
    wu = oracle.synth_upscaler_weights(mid=128, seed=3)
    path = tmp_path / "upscaler.safetensors"
    extra = dict(wu)
    extra["upsampler.blur_down.kernel"] = np.ones((1, 1, 5, 5), np.float32)  # must be skipped
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in extra.items()}, str(path))
    ctx.upscaler_load(path)
    rep = ctx.load_report()
    assert rep["missing"] == 0 and rep["unmatched"] == 0
    rng = np.random.default_rng(1)
    lat = rng.standard_normal((1, 128, 2, 3, 4)).astype(np.float32)
    got = ctx.upscale_latent(lat)
    assert got.shape == (1, 128, 2, 6, 8)
    ref = oracle.upsample_latents(wu, lat, mean, std)
    assert rel_l2(got, ref) <= 3e-2, rel_l2(got, ref)


def test_adain_parity(ltx, oracle, vae_stats_ctx):
    ctx, _, _ = vae_stats_ctx
    rng = np.random.default_rng(2)
    lat = (rng.standard_normal((1, 128, 2, 6, 8)) * 1.7 + 0.3).astype(np.float32)
    ref_lat = rng.standard_normal((1, 128, 2, 3, 4)).astype(np.float32)
    for factor in (1.0, 0.4):
        got = ctx.adain_filter_latent(lat, ref_lat, factor)
        ref = oracle.adain_filter_latent(lat, ref_lat, factor)
        assert np.abs(got - ref).max() <= 1e-4


def test_two_stage_latent_parity(ltx, oracle, vae_stats_ctx, tmp_path):
    """stage 1 (half res) -> upscale -> AdaIN -> re-noise (explicit noise) -> 3-step stage 2."""
    from safetensors.numpy import save_file

    ctx, mean, std = vae_stats_ctx
    cfg, ocfg = small_cfg(ltx, oracle, heads=2, layers=2, caption=128)
    w = oracle.synth_dit_weights(ocfg, seed=12)
    write_dit_file(oracle, w, tmp_path / "dit.safetensors")
    ctx.dit_load(tmp_path / "dit.safetensors", cfg)
    wu = oracle.synth_upscaler_weights(mid=64, seed=4)
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in wu.items()}, str(tmp_path / "up.safetensors"))
    ctx.upscaler_load(tmp_path / "up.safetensors")
    width, height, frames = 128, 128, 9  # stage 1 at 64x64 -> latent 2x2x2, stage 2 latent 2x4x4
    rng = np.random.default_rng(6)
    n1 = rng.standard_normal((1, 128, 2, 2, 2)).astype(np.float32)
    n2 = rng.standard_normal((1, 128, 2, 4, 4)).astype(np.float32)
    context = oracle.bf16_round(rng.standard_normal((1, 16, 128)).astype(np.float32))
    got = ctx.generate_two_stage(n1, n2, ltx.f32_to_bf16_bits(context), None, width, height, frames, decode=False)
    ref = oracle.two_stage_latent(w, ocfg, wu, mean, std, n1, n2, context, None, width, height, frames)
    assert got.shape == ref.shape == (1, 128, 2, 4, 4)
    assert rel_l2(got, ref) <= 5e-2, rel_l2(got, ref)
    with pytest.raises(ltx.LTXError):  # two-stage needs %64 dimensions (LTXPipeline.swift:2443)
        ctx.generate_two_stage(n1, n2, ltx.f32_to_bf16_bits(context), None, 96, 128, frames, decode=False)
