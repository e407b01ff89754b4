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


def test_denoise_step_diagnostics_match_the_oracle(ltx, oracle, gpu_ctx, model):
    """ltx_denoise_options.step_stats: per step the mean / population std of the guided velocity and of the latent after the Euler update -
    what the reference logs under --profile (LTXPipeline.swift:945-951, four decimals) - against the same statistics of oracle.denoise, with
    CFG + rescale on; asking for them must not change the latent."""
    cfg, ocfg, w = model
    F, H, W, S = 2, 4, 6, 24
    noise, ctx2 = _inputs(oracle, ocfg, F, H, W, S, 41, nb=2)
    sig = ltx.sigmas(False, 5, F * H * W)
    lat0 = noise * sig[0]
    stats = np.zeros((len(sig) - 1, 4), np.float32)
    got = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx2), None, F, H, W, cfg_scale=3.0, guidance_rescale=0.5, step_stats=stats)
    plain = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx2), None, F, H, W, cfg_scale=3.0, guidance_rescale=0.5)
    assert np.array_equal(got, plain)
    ref_stats = []
    ref = oracle.denoise(w, ocfg, lat0, sig, ctx2[1:2], None, F, H, W, cfg_scale=3.0, rescale=0.5, neg_context=ctx2[0:1], step_stats=ref_stats)
    ref_stats = np.asarray(ref_stats, np.float32)
    assert stats.shape == ref_stats.shape and np.isfinite(stats).all() and (stats[:, 1] > 0).all() and (stats[:, 3] > 0).all()
    # the last row's latent statistics are those of the returned latent, exactly as numpy computes them in f32-safe precision
    assert abs(float(got.mean(dtype=np.float64)) - stats[-1, 2]) <= 1e-5 and abs(float(got.std(dtype=np.float64)) - stats[-1, 3]) <= 1e-5
    err = np.abs(stats - ref_stats)
    print("step diagnostics, max abs difference to the oracle per column (vel mean, vel std, latent mean, latent std):", err.max(0))
    assert (err[:, 0] <= 2e-3).all() and (err[:, 2] <= 2e-3).all()
    assert (err[:, 1] <= 2e-2 * ref_stats[:, 1]).all() and (err[:, 3] <= 1e-2 * ref_stats[:, 3]).all()
    assert rel_l2(got, ref) <= 5e-2


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


# ---------------------------------------------------------------------------------------------------------------
# image-to-video conditioning - per-token timesteps, frame-0 slice Euler, re-noised frame 0
# ---------------------------------------------------------------------------------------------------------------
def test_dit_forward_per_token_timesteps(ltx, oracle, gpu_ctx, model):
    """prepareTimestep with [B,T] sigmas (LTXTransformer.swift:105-124): frame-0 tokens at 0, the rest at sigma; a row of
    equal per-token values must reproduce the single-timestep forward exactly (same kernels, same tables)."""
    cfg, ocfg, w = model
    F, H, W, S = 3, 2, 4, 24
    T = F * H * W
    rng = np.random.default_rng(31)
    lat = oracle.bf16_round(rng.standard_normal((1, T, 128)).astype(np.float32))
    ctx = oracle.bf16_round(rng.standard_normal((1, S, ocfg.caption_channels)).astype(np.float32))
    lb, cb = ltx.f32_to_bf16_bits(lat), ltx.f32_to_bf16_bits(ctx)
    ts = np.full((1, T), 0.6, np.float32)
    uniform = gpu_ctx.dit_forward_tokens(lb, cb, ts, None, F, H, W)
    single = gpu_ctx.dit_forward(lb, cb, np.array([0.6], np.float32), None, F, H, W)
    assert np.array_equal(uniform, single)
    ts[:, :H * W] = 0.0
    got = gpu_ctx.dit_forward_tokens(lb, cb, ts, None, F, H, W)
    ref = oracle.dit_forward(w, ocfg, lat, ctx, ts, None, F, H, W)
    assert rel_l2(got, ref) <= 2e-2, rel_l2(got, ref)
    assert rel_l2(got, single) > 5e-2  # the conditioning really changes the result
    # three distinct values, B = 2 (CFG batch): 2 x 3 groups
    lat2 = np.concatenate([lat, lat]); ctx2 = np.concatenate([ctx, ctx[:, ::-1]])
    ts2 = np.stack([ts[0], np.where(np.arange(T) % 3 == 0, 0.25, 0.9).astype(np.float32)])
    ts2[1, 0] = 0.0
    got2 = gpu_ctx.dit_forward_tokens(ltx.f32_to_bf16_bits(lat2), ltx.f32_to_bf16_bits(ctx2), ts2, None, F, H, W)
    ref2 = oracle.dit_forward(w, ocfg, lat2, ctx2, ts2, None, F, H, W)
    assert rel_l2(got2, ref2) <= 2e-2
    with pytest.raises(ltx.LTXError):  # more distinct (batch, value) pairs than the timestep path holds
        gpu_ctx.dit_forward_tokens(lb, cb, np.linspace(0, 1, T, dtype=np.float32)[None], None, F, H, W)


@pytest.mark.parametrize("use_cfg,noise_scale", [(False, 0.0), (False, 0.15), (True, 0.15)])
def test_denoise_image_to_video(ltx, oracle, gpu_ctx, model, use_cfg, noise_scale):
    """denoise(...) with conditioningMask / conditionedLatent (LTXPipeline.swift:2191-2401)."""
    cfg, ocfg, w = model
    F, H, W, S = 3, 4, 4, 24
    nb = 2 if use_cfg else 1
    noise, ctx = _inputs(oracle, ocfg, F, H, W, S, 11, nb=nb)
    rng = np.random.default_rng(12)
    cond = rng.standard_normal((1, 128, 1, H, W)).astype(np.float32)
    sig = ltx.sigmas(True, 8, F * H * W) if not use_cfg else ltx.sigmas(False, 4, F * H * W)
    cnoise = rng.standard_normal((len(sig) - 1, 128, 1, H, W)).astype(np.float32)
    lat0 = noise * sig[0]
    kw = dict(cond_latent=cond, image_cond_noise_scale=noise_scale, cond_noise=cnoise if noise_scale > 0 else None)
    if use_cfg:
        got = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx), None, F, H, W, cfg_scale=3.0, **kw)
        ref = oracle.denoise(w, ocfg, lat0, sig, ctx[1:2], None, F, H, W, cfg_scale=3.0, neg_context=ctx[0:1], **kw)
    else:
        got = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx), None, F, H, W, **kw)
        ref = oracle.denoise(w, ocfg, lat0, sig, ctx, None, F, H, W, **kw)
    # frame 0 is never stepped: it holds the image latent, or its last re-noised value - exact either way
    assert np.array_equal(got[:, :, 0], ref[:, :, 0])
    if noise_scale == 0:
        assert np.array_equal(got[:, :, 0:1], cond)
    assert rel_l2(got[:, :, 1:], ref[:, :, 1:]) <= 3e-2, rel_l2(got[:, :, 1:], ref[:, :, 1:])
    t2v = gpu_ctx.denoise(lat0, sig, ltx.f32_to_bf16_bits(ctx), None, F, H, W, **({"cfg_scale": 3.0} if use_cfg else {}))
    assert rel_l2(got[:, :, 1:], t2v[:, :, 1:]) > 1e-2  # conditioning is not a no-op


def test_context_cache_keys_do_not_collide_across_modes(ltx, oracle, gpu_ctx, model):
    """The projected text context is cached under keys derived from the caller's version per pass kind (batched [neg,pos], negative
    alone, positive alone). A caller that reuses SMALL version numbers across prompts and modes (plain, CFG, CFG+STG) must never be
    served another prompt's entry: every cached run must equal the same run with ctx_version = 0 (recompute), bit for bit."""
    import torch

    cfg, ocfg, w = model
    F, H, W, S = 1, 4, 4, 24
    rng = np.random.default_rng(77)
    sig = ltx.sigmas(False, 2, F * H * W)

    def run(ctxs, version, **kw):
        lat = torch.from_numpy(np.ascontiguousarray(noise * sig[0])).cuda()
        c = torch.from_numpy(ltx.f32_to_bf16_bits(ctxs).astype(np.int16)).cuda().view(torch.bfloat16)
        gpu_ctx.denoise_dev(lat, sig, c, None, F, H, W, ctx_version=version, **kw)
        torch.cuda.synchronize()
        return lat.cpu().numpy()

    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    prompts = [oracle.bf16_round(rng.standard_normal((2, S, ocfg.caption_channels)).astype(np.float32)) for _ in range(3)]
    modes = [dict(), dict(cfg_scale=3.0), dict(cfg_scale=3.0, stg_scale=0.7, stg_blocks=(1,)), dict(stg_scale=0.7, stg_blocks=(1,))]
    want = {}
    for p, ctxs in enumerate(prompts):
        for m, kw in enumerate(modes):
            c = ctxs if "cfg_scale" in kw else ctxs[1:2]
            want[(p, m)] = run(c, 0, **kw)
    # now with caching: one small version per prompt, chosen so that ad-hoc per-mode keys (version * 4 + {0, 1, 2} for the CFG / STG
    # passes, the bare version for the plain pass) WOULD meet: prompt 0 = 1, prompt 1 = 6 = 1 * 4 + 2, prompt 2 = 5 = 1 * 4 + 1
    versions = [1, 6, 5]
    order = [(0, 2), (1, 0), (2, 0), (0, 1), (1, 3), (2, 2), (0, 0), (1, 1), (2, 1), (0, 3), (1, 2), (2, 3), (1, 0), (0, 2)]
    for p, m in order:
        kw = modes[m]
        c = prompts[p] if "cfg_scale" in kw else prompts[p][1:2]
        got = run(c, versions[p], **kw)
        assert np.array_equal(got, want[(p, m)]), (p, m)


def test_denoise_options_struct_size_is_honoured(ltx, oracle, gpu_ctx, model):
    """ABI revision 2 (round-4 advice): ltx_denoise_options starts with the size the CALLER compiled. 0 (a revision-1 host, or a struct
    nobody initialised) is refused with a message that names the fix; a size that stops before `step_stats` (a host built before that
    field existed) runs and the library does NOT touch the bytes behind the declared size - here a poisoned pointer that would fault if
    it were written through."""
    import ctypes as C

    from importlib import import_module

    lib_mod = import_module("ltx-video-swift-mlx_amd._lib")
    cfg, ocfg, w = model
    F, H, W, S = 1, 4, 4, 16
    noise, cx = _inputs(oracle, ocfg, F, H, W, S, 9)
    sig = np.ascontiguousarray(ltx.sigmas(True, 8, F * H * W), dtype=np.float32)
    lat0 = np.ascontiguousarray(noise * sig[0], dtype=np.float32)
    bits = np.ascontiguousarray(ltx.f32_to_bf16_bits(cx))
    cb = lib_mod.PROGRESS_CB(lambda s, t, sg, u: None)

    def call(size, step_stats=None):
        lat = lat0.copy()
        o = lib_mod.DenoiseOptions(size, 1.0, 0.0, 0.0, None, 0, 0.0, None, 0.0, None, 0, step_stats)
        rc = ltx.lib.ltx_denoise(gpu_ctx._h, lat.ctypes.data_as(C.c_void_p), F, H, W, sig.ctypes.data_as(C.POINTER(C.c_float)), len(sig),
                                 bits.ctypes.data_as(C.c_void_p), None, S, C.byref(o), cb, None)
        return rc, lat

    rc, _ = call(0)
    assert rc == 2 and "struct_size" in ltx.lib.ltx_last_error(gpu_ctx._h).decode()
    full = C.sizeof(lib_mod.DenoiseOptions)
    rc, ref = call(full)
    assert rc == 0
    short = lib_mod.DenoiseOptions.step_stats.offset   # the revision-2 core: everything up to and including `shard`
    rc, got = call(short, step_stats=C.c_void_p(0x10))  # behind the declared size: must not be read
    assert rc == 0 and np.array_equal(got, ref)
    stats = np.zeros((len(sig) - 1, 4), np.float32)
    rc, got = call(full, step_stats=stats.ctypes.data_as(C.c_void_p))
    assert rc == 0 and np.array_equal(got, ref) and np.abs(stats[:, 1]).min() > 0
