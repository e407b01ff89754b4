"""Parity at the reference's DEPTH and at the headline SIZE (round-3 verdict, item 1): the stated tolerances, until now shown on
two-layer cuts, on small latents or block by block, are checked here on the whole thing.

  (a) ltx_denoise_dev, ALL 48 layers, D = 4096, distilled 8-step schedule, config 1's 2x8x8 latent (128 tokens, 256 text keys)
      vs oracle.denoise: final latent rel-L2 <= 1e-2, cos >= 0.999 - the end-to-end bound of DESIGN.md section 2
      (LTXPipeline.swift:800-956).
  (b) one 48-layer forward at the HEADLINE shape (T = 1536 = 4x16x24, S = 1024, masked) vs oracle.dit_forward: rel-L2 <= 2e-2,
      cos >= 0.9995 (the bound the 2- and 8-layer cuts are held to; LTXTransformer.swift:235-486).
  (c) in tests/test_config4_and_full_res_parity_gpu.py (it owns the full-size VAE fixture): the whole 768x512x25 decode.

The oracle gets the very weights the library holds (ltx_dit_export_param); they are fetched once and kept on the host as f32 (52 GB:
the GPU box allows 270 GiB) so that the eight steps of (a) and the forward of (b) do not re-export them. Host time on the GPU box:
about a minute of export, then BLAS.
"""
import time

import numpy as np
import pytest
import torch

from test_dit_gpu import rel_l2
from test_full_width_parity_gpu import _cos, _dev_bf16, _forward

pytestmark = pytest.mark.gpu


class HostWeights(dict):
    """Module key -> f32 array; exported from the library on first access and kept."""

    def __init__(self, ctx, shapes):
        super().__init__()
        self.ctx, self.shapes = ctx, shapes

    def __missing__(self, key):
        self[key] = v = self.ctx.dit_export_param(key).reshape(self.shapes[key])
        return v

    def __contains__(self, key):
        return key in self.shapes


@pytest.fixture(scope="module")
def full48_host(ltx, oracle):
    """The reference architecture (48 layers, 32 heads x 128, caption 3840) with on-device synthetic weights (seed 1234: what
    bench.py runs) and their host mirror for the oracle."""
    ctx = ltx.Context(0)
    cfg = ltx.default_transformer_config()
    ctx.dit_init_synthetic(cfg, seed=1234)
    ocfg = oracle.DiTConfig()
    yield ctx, cfg, ocfg, HostWeights(ctx, oracle.dit_param_shapes(ocfg))
    ctx.close()


@pytest.fixture(scope="module")
def headline_loop(ltx, oracle, full48_host):
    """BASELINE configs[1]'s WHOLE loop through the oracle ONCE (4 min of host BLAS on the GPU box; the driver's GPU step has 900 s):
    distilled 8-step schedule, all 48 blocks, T = 1536 (4x16x24), 1024 text keys, a tenth masked. Keeps the library's final latent, the
    oracle's, and the oracle's forward of step 0 (input tokens + raw transformer output) for the single-forward test."""
    ctx, cfg, ocfg, w = full48_host
    assert ltx.latent_shape(768, 512, 25) == (4, 16, 24)
    F, H, W, S = 4, 16, 24, 1024
    rng = np.random.default_rng(88)
    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
    mask = (rng.random((1, S)) > 0.1).astype(np.int32)
    mask[:, 0] = 1
    sig = ltx.sigmas(True, 8, F * H * W)
    lat0 = noise * sig[0]
    latd = torch.from_numpy(lat0).cuda()
    ctx.denoise_dev(latd, sig, _dev_bf16(cx), torch.from_numpy(mask).cuda(), F, H, W, ctx_version=79)
    got = latd.cpu().numpy()
    t0 = time.time()
    forwards = []
    ref = oracle.denoise(w, ocfg, lat0, sig, cx, mask, F, H, W, velocity_tokens=forwards)
    assert len(forwards) == 8 and [f[0] for f in forwards] == list(range(8))
    return dict(got=got, ref=ref, step0=forwards[0], sigma0=float(sig[0]), cx=cx, mask=mask, dims=(F, H, W, S), oracle_s=time.time() - t0)


def test_48_layer_forward_at_the_headline_shape_vs_oracle(ltx, oracle, full48_host, headline_loop):
    """(b) BASELINE configs[1]'s forward: 768x512x25 -> 1536 tokens, 1024 text keys, a tenth of them masked, all 48 blocks - the oracle's
    forward of the loop's first step (round 5: shared with the loop test below instead of a second 30 s oracle run), the library's
    forward on the same tokens at the same sigma."""
    ctx, cfg, ocfg, w = full48_host
    F, H, W, S = headline_loop["dims"]
    _, tok, ref = headline_loop["step0"]
    got = _forward(ctx, tok, headline_loop["cx"], headline_loop["sigma0"], headline_loop["mask"], F, H, W, version=79)
    r, c = rel_l2(got, ref), _cos(got, ref)
    print(f"full width, 48 blocks, T={F * H * W}, S={S}, masked: rel-L2 {r:.3e}, cos {c:.6f}")
    assert np.isfinite(got).all() and r <= 2e-2 and c >= 0.9995, (r, c)


def test_48_layer_eight_step_denoise_vs_oracle(ltx, oracle, full48_host):
    """(a) The stated end-to-end tolerance at the reference's depth: 8 Euler steps x 48 blocks compound the per-block bf16 deviation."""
    ctx, cfg, ocfg, w = full48_host
    assert ltx.latent_shape(256, 256, 9) == (2, 8, 8)
    F, H, W, S = 2, 8, 8, 256
    rng = np.random.default_rng(8)
    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
    sig = ltx.sigmas(True, 8, F * H * W)
    assert len(sig) == 9
    lat0 = noise * sig[0]
    latd = torch.from_numpy(lat0).cuda()
    ctx.denoise_dev(latd, sig, _dev_bf16(cx), None, F, H, W, ctx_version=78)
    got = latd.cpu().numpy()
    t0 = time.time()
    ref = oracle.denoise(w, ocfg, lat0, sig, cx, None, F, H, W)
    r, c = rel_l2(got, ref), _cos(got, ref)
    print(f"full width, 48 blocks, 8-step denoise: rel-L2 {r:.3e}, cos {c:.6f} (oracle {time.time() - t0:.0f} s)")
    assert np.isfinite(got).all() and r <= 1e-2 and c >= 0.999, (r, c)


def test_48_layer_eight_step_denoise_at_the_headline_shape_vs_oracle(headline_loop):
    """Round-4 verdict, item 1(c): BASELINE configs[1]'s WHOLE loop - distilled 8-step schedule, all 48 blocks, T = 1536 (4x16x24),
    1024 text keys, a tenth masked - vs oracle.denoise. This is the launch shape that takes the 192x256 split-K of the FFN's second
    GEMM with its bf16 partials and the bf16 q|k store (two roundings the reference does not have): their
    compounding over eight Euler steps is observed here. Bound: the end-to-end one, 1e-2 / 0.999 (LTXPipeline.swift:800-956)."""
    got, ref = headline_loop["got"], headline_loop["ref"]
    r, c = rel_l2(got, ref), _cos(got, ref)
    print(f"full width, 48 blocks, 8-step denoise at T=1536, S=1024 masked: rel-L2 {r:.3e}, cos {c:.6f} (oracle {headline_loop['oracle_s']:.0f} s)")
    assert np.isfinite(got).all() and r <= 1e-2 and c >= 0.999, (r, c)


def test_full_depth_48_layers_config1_vs_oracle(ltx, oracle, full48_host):
    """BASELINE configs[0]: 256x256x9 -> latent 2x8x8 = 128 tokens through ALL 48 blocks of the reference architecture."""
    ctx, cfg, ocfg, w = full48_host
    assert ltx.latent_shape(256, 256, 9) == (2, 8, 8)
    F, H, W, S = 2, 8, 8, 256
    T = F * H * W
    rng = np.random.default_rng(1)
    lat = oracle.bf16_round(rng.standard_normal((1, T, 128)).astype(np.float32))
    cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
    got = _forward(ctx, lat, cx, 0.725, None, F, H, W)
    again = _forward(ctx, lat, cx, 0.725, None, F, H, W)
    assert np.array_equal(got, again)  # the split-K path of this launch shape is deterministic
    ref = oracle.dit_forward(w, ocfg, lat, cx, np.array([0.725], np.float32), None, F, H, W)
    r, c = rel_l2(got, ref), _cos(got, ref)
    print(f"full width, 48 blocks, T={T}, S={S}: rel-L2 {r:.3e}, cos {c:.6f}")
    assert np.isfinite(got).all() and r <= 3e-2 and c >= 0.999, (r, c)


def test_48_layer_cfg_loop_vs_oracle(ltx, oracle, full48_host):
    """BASELINE configs[2]'s loop body at the reference's depth: dev schedule, CFG 4.0 on a [negative, positive] pair with masked text keys,
    guidance rescale 0.7 (LTXPipeline.swift:820-865, LatentUtils.swift:131-183) - three steps through all 48 blocks on config 1's latent.
    CFG multiplies the difference of two forwards by 4, so the per-forward deviation (2.4e-3 .. 2.9e-3) is amplified: bound 5e-2, the one the
    reduced-width CFG test states."""
    ctx, cfg, ocfg, w = full48_host
    F, H, W, S = 2, 8, 8, 64
    rng = np.random.default_rng(21)
    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    ctx2 = oracle.bf16_round(rng.standard_normal((2, S, 3840)).astype(np.float32))   # [negative, positive]
    mask = (rng.random((2, S)) > 0.2).astype(np.int32)
    mask[:, 0] = 1
    sig = ltx.sigmas(False, 3, F * H * W)
    lat0 = noise * sig[0]
    got = ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx2), mask, F, H, W, cfg_scale=4.0, guidance_rescale=0.7)
    t0 = time.time()
    ref = oracle.denoise(w, ocfg, lat0, sig, ctx2[1:2], mask[1:2], F, H, W, cfg_scale=4.0, rescale=0.7, neg_context=ctx2[0:1], neg_mask=mask[0:1])
    r, c = rel_l2(got, ref), _cos(got, ref)
    print(f"full width, 48 blocks, 3-step CFG 4.0 + rescale loop: rel-L2 {r:.3e}, cos {c:.6f} (oracle {time.time() - t0:.0f} s)")
    assert np.isfinite(got).all() and r <= 5e-2 and c >= 0.999, (r, c)


def test_48_layer_image_to_video_loop_vs_oracle(ltx, oracle, full48_host):
    """Image-to-video at the reference's depth (LTXPipeline.swift:2191-2401): frame 0 holds the image latent re-noised per step, its
    tokens carry timestep 0 through adaLN (per-token timestep groups in the gated-residual epilogues at D = 4096), the Euler step skips
    it - distilled 8-step schedule, all 48 blocks, a 3x8x8 latent."""
    ctx, cfg, ocfg, w = full48_host
    F, H, W, S = 3, 8, 8, 64
    rng = np.random.default_rng(33)
    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
    cond = rng.standard_normal((1, 128, 1, H, W)).astype(np.float32)
    sig = ltx.sigmas(True, 8, F * H * W)
    cnoise = rng.standard_normal((len(sig) - 1, 128, 1, H, W)).astype(np.float32)
    lat0 = noise * sig[0]
    kw = dict(cond_latent=cond, image_cond_noise_scale=0.15, cond_noise=cnoise)
    got = ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(cx), None, F, H, W, **kw)
    t0 = time.time()
    ref = oracle.denoise(w, ocfg, lat0, sig, cx, None, F, H, W, **kw)
    assert np.array_equal(got[:, :, 0], ref[:, :, 0])  # frame 0 is never stepped: its last re-noised value, exactly
    r, c = rel_l2(got[:, :, 1:], ref[:, :, 1:]), _cos(got[:, :, 1:], ref[:, :, 1:])
    print(f"full width, 48 blocks, 8-step image-to-video loop: frames 1+ rel-L2 {r:.3e}, cos {c:.6f} (oracle {time.time() - t0:.0f} s)")
    assert np.isfinite(got).all() and r <= 1e-2 and c >= 0.999, (r, c)
