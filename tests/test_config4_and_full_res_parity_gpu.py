"""The parity holes the round-2 verdict named ("Next round" item 1), closed at the sizes the BASELINE configurations run:

  (a) config 4's decode: latent [1,128,4,32,48] -> (25,1024,1536,3). The 128-channel stage is 25x256x384 = 2.46 M rows (a
      1.26 GB f32 stream, a 17 GB virtual im2col matrix: the largest offsets any BASELINE workload forms). Finite, in [0,1],
      whole-clip tile == untiled bit for bit, the last output rows depend on the last latent positions, and ONE 128 -> 128 conv
      at (25,256,384) integer-exact against torch conv3d (VideoConvolution.swift:202-348).
  (b) one res-block of every decoder stage at the HEADLINE resolution (768x512x25: 128 channels at 25x128x192 with the fused
      PixelNorm + SiLU epilogues, 256 at 13x64x96 with the split-K tile window, 512 at 7x32x48, 1024 at 4x16x24) through
      ltx_vae_res_block_dev against oracle.vae_res_block (VideoDecoder.swift:75-131).
  (c) fuseLoRA at the distilled LoRA's real rank 384 on a D = 4096 2-layer model against oracle.lora_fuse
      (LoRAAdapter.swift:64-166, ModelDownloader.swift:473-476), plain and on a qint8 model (dequant -> merge -> requant).
  (d) full width at T = 1536, S = 1024 with EIGHT layers against the oracle (two in test_full_width_parity_gpu.py).
"""
import json

import numpy as np
import pytest
import torch

from test_dit_gpu import rel_l2
from test_full_width_parity_gpu import DeviceWeights, _cos, _dev_bf16, _forward
from test_lora_quant_gpu import make_lora
from test_vae_gpu import relayout

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------
# (a) config 4 decode
# ---------------------------------------------------------------------------------------------------------------
def test_config4_vae_decode_1536x1024x25(ltx):
    ctx = ltx.Context(0)
    try:
        ctx.vae_init_synthetic(seed=77)
        Fl, Hl, Wl = 4, 32, 48
        assert ltx.latent_shape(1536, 1024, 25) == (Fl, Hl, Wl)
        lat = torch.empty((1, 128, Fl, Hl, Wl), dtype=torch.float32, device="cuda")
        ctx.op_fill_normal_f32(lat, seed=45)
        nf = 8 * (Fl - 1) + 1
        a = torch.full((nf, Hl * 32, Wl * 32, 3), float("nan"), dtype=torch.float32, device="cuda")
        assert ctx.vae_decode_dev(lat, Fl, Hl, Wl, a) == 25
        torch.cuda.synchronize()
        assert a.shape == (25, 1024, 1536, 3)
        assert bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
        assert float(a.std()) > 1e-3  # not a constant image
        b = torch.empty_like(a)
        assert ctx.vae_decode_dev(lat, Fl, Hl, Wl, b, tile=Fl, overlap=1) == 25  # one tile that covers the clip
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        # the far end of the 1.26 GB streams inside: the last output rows must depend on the LAST latent positions (and only they)
        lat2 = lat.clone()
        lat2[:, :, -1, -1, -1] += 1.0
        ctx.vae_decode_dev(lat2, Fl, Hl, Wl, b)
        torch.cuda.synchronize()
        assert not torch.equal(a[-1, -32:, -32:], b[-1, -32:, -32:]) and torch.equal(a[0, :32, :32], b[0, :32, :32])
    finally:
        ctx.close()


def test_conv3d_128ch_at_25x256x384_integer_exact(gpu_ctx):
    """The 128-channel stage of config 4: 2 457 600 output rows (12 800 tiles of 192), a 629 MB bf16 input, a 1.26 GB f32 output,
    row x K offsets of the virtual im2col matrix up to 2 457 600 x 3456 x 2 B = 17 GB: every index product must be 64-bit.
    Small integers: any summation order gives the same f32, so equality with torch's conv3d is exact."""
    import torch.nn.functional as F_

    F, H, W, C = 25, 256, 384, 128
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randint(-2, 3, (1, C, F, H, W), generator=g, device="cuda", dtype=torch.int8).float()
    w = torch.randint(-2, 3, (C, C, 3, 3, 3), generator=g, device="cuda", dtype=torch.int8).float()
    b = torch.randint(-4, 5, (C,), generator=g, device="cuda", dtype=torch.int8).float()
    xd = x[0].permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    wd = torch.from_numpy(relayout(w.cpu().numpy())).to(torch.bfloat16).cuda()
    out = torch.full((F, H, W, C), float("nan"), device="cuda")
    gpu_ctx.op_conv3d(xd, wd, b, out)
    torch.cuda.synchronize()
    assert out.numel() * 4 > 2 ** 30 and out.shape[0] * out.shape[1] * out.shape[2] * 27 * C * 2 > 2 ** 33
    # reference in frame slabs (keeps torch's workspace small): output frame f reads input frames f-1..f+1, replicated at the ends
    xp = F_.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
    xp = torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
    for f0 in range(0, F, 5):
        ref = F_.conv3d(xp[:, :, f0:f0 + 7].double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
        assert torch.equal(out[f0:f0 + 5], ref), (f0, float((out[f0:f0 + 5] - ref).abs().max()))


# ---------------------------------------------------------------------------------------------------------------
# (b) one res-block per decoder stage at the headline resolution
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vae_from_oracle_weights(ltx, oracle, tmp_path_factory):
    from safetensors.torch import save_file

    w = oracle.synth_vae_weights(seed=5, timestep_conditioning=True)   # the file carries the time embedders; config.json switches them off
    d = tmp_path_factory.mktemp("vae_full")
    path = d / "diffusion_pytorch_model.safetensors"
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16 if v.ndim == 5 else torch.float32)
               for k, v in oracle.vae_file_keys(w).items()}, str(path))
    (d / "config.json").write_text(json.dumps({"timestep_conditioning": False}))
    ctx = ltx.Context(0)
    ctx.vae_load(path)
    rep = ctx.load_report()
    assert rep["unmatched"] == 0 and rep["missing"] == 0, rep
    yield ctx, w
    ctx.close()


def _res_block_prefix(oracle, w, group, block):
    """module prefix of res-block `block` of up-block group `group` in the oracle's weight dict (mapVAEWeights naming)"""
    p = f"up_blocks_{2 * group}.res_blocks.{block}."
    C = oracle.VAE_CHANNELS[group]
    assert w[p + "conv1.conv.weight"].shape[:2] == (C, C)
    return p, C


@pytest.mark.parametrize("group,block,F,H,W", [(3, 0, 25, 128, 192), (3, 4, 25, 128, 192), (2, 1, 13, 64, 96), (1, 2, 7, 32, 48), (0, 3, 4, 16, 24)])
def test_vae_res_block_at_headline_resolution_vs_oracle(ltx, oracle, vae_from_oracle_weights, group, block, F, H, W):
    """Stage shapes of 768x512x25. Tolerance: bf16 conv inputs x bf16 weights with f32 accumulation against the
    oracle's f32 activations: rel-L2 <= 1e-2 on the block's output, <= 3e-2 on what the block ADDS to the stream."""
    ctx, w = vae_from_oracle_weights
    p, C = _res_block_prefix(oracle, w, group, block)
    rng = np.random.default_rng(100 * group + block)
    x = rng.standard_normal((1, C, F, H, W), dtype=np.float32)
    xd = torch.from_numpy(np.ascontiguousarray(x[0].transpose(1, 2, 3, 0))).cuda()
    ctx.vae_res_block_dev(group, block, xd, F, H, W)
    torch.cuda.synchronize()
    got = xd.cpu().numpy()
    ref = oracle.vae_res_block(w, p, x)[0].transpose(1, 2, 3, 0)
    r_out = rel_l2(got, ref)
    x_cl = x[0].transpose(1, 2, 3, 0)
    r_delta = rel_l2(got - x_cl, ref - x_cl)
    print(f"res-block group {group} ({C} ch) block {block} at {F}x{H}x{W}: rel-L2 out {r_out:.3e}, delta {r_delta:.3e}")
    assert np.isfinite(got).all() and r_out <= 1e-2 and r_delta <= 3e-2, (r_out, r_delta)
    assert float(np.linalg.norm(ref - x_cl) / np.linalg.norm(x_cl)) > 1e-2, "the block adds too little to prove anything"


@pytest.mark.parametrize("group,F,H,W", [(2, 13, 64, 96), (1, 7, 32, 48), (0, 4, 16, 24), (2, 3, 6, 10)])
def test_vae_upsampler_at_headline_resolution_vs_oracle(ltx, oracle, vae_from_oracle_weights, group, F, H, W):
    """The three depth-to-space upsamplers at the stage shapes of 768x512x25 through ltx_vae_upsample_dev - the persistent
    halo-staged conv with round 4's depth-to-space epilogue (groups 2 and 1), the ring kernel with the same epilogue (group 0: W < 48) -
    and one small, ragged shape that takes the general epilogue, vs oracle.vae_upsample (VideoDecoder.swift:201-251). Index logic
    (sub-position order, first-frame drop, channel tiling of the residual) is checked exactly on a second pass with the conv zeroed
    out of the comparison: out - D2S-residual must be what the conv alone gives for a zero stream, i.e. the bias pattern."""
    ctx, w = vae_from_oracle_weights
    C = oracle.VAE_CHANNELS[group]
    p = f"up_blocks_{2 * group + 1}."
    rng = np.random.default_rng(40 + group + H)
    x = rng.standard_normal((1, C, F, H, W), dtype=np.float32)
    xd = torch.from_numpy(np.ascontiguousarray(x[0].transpose(1, 2, 3, 0))).cuda()
    out = torch.empty((2 * F - 1, 2 * H, 2 * W, C // 2), dtype=torch.float32, device="cuda")
    ctx.vae_upsample_dev(group, xd, F, H, W, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = oracle.vae_upsample(w, p, x)[0].transpose(1, 2, 3, 0)
    assert got.shape == ref.shape
    r = rel_l2(got, ref)
    # the residual term alone, exactly: a zero stream leaves conv(0) + bias = bias, so got0 - bias == 0 and got - got0 - (ref residual) is the conv
    res = oracle.depth_to_space(x, C // 8)[:, :, 1:]
    res = np.concatenate([res] * 4, axis=1)[0].transpose(1, 2, 3, 0)
    conv_got, conv_ref = got - res, ref - res
    rc = rel_l2(conv_got, conv_ref)
    print(f"upsampler group {group} ({C} -> {C // 2} ch) at {F}x{H}x{W}: rel-L2 out {r:.3e}, conv part {rc:.3e}")
    assert np.isfinite(got).all() and r <= 1e-2 and rc <= 2e-2, (r, rc)
    z = torch.zeros_like(xd)
    out0 = torch.empty_like(out)
    ctx.vae_upsample_dev(group, z, F, H, W, out0)
    torch.cuda.synchronize()
    bias = w[p + "conv.conv.bias"].astype(np.float32)                      # file order n = c * 8 + sub
    ref0 = oracle.vae_upsample(w, p, np.zeros_like(x))[0].transpose(1, 2, 3, 0)
    assert np.array_equal(out0.cpu().numpy(), ref0), "bias / sub-position / first-frame-drop pattern of a zero stream"
    assert len(np.unique(bias)) > 8


def test_vae_whole_decode_768x512x25_vs_oracle(ltx, oracle, vae_from_oracle_weights):
    """The WHOLE headline decode in one comparison (round-3 verdict 1c): [1,128,4,16,24] -> (25,512,768,3), all 42 convs chained -
    the halo-staged kernel, the split-K tile windows, the three fused depth-to-space stores, the fused PixelNorm + SiLU epilogues and
    the permuted conv_out rows - against oracle.decode_video (VideoDecoder.swift:358-449) on every one of the 29.5 M values.
    Bounds: the ones test_vae_gpu.py states (<= 2e-2 abs on [0,1] frames, rel-L2 <= 3e-2 about the mean)."""
    import time

    ctx, w = vae_from_oracle_weights
    rng = np.random.default_rng(45)
    lat = rng.standard_normal((1, 128, 4, 16, 24)).astype(np.float32)
    got = ctx.vae_decode(lat)
    assert got.shape == (25, 512, 768, 3) and np.isfinite(got).all()
    t0 = time.time()
    ref = np.clip((oracle.decode_video(w, lat, return_raw=True) + 1) / 2, 0, 1)
    err = float(np.abs(got - ref).max())
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref - ref.mean()))
    sat = float(np.mean((ref <= 0) | (ref >= 1)))
    print(f"whole decode 768x512x25: max abs {err:.3e}, rel-L2 {rel:.3e}, clipped share {sat:.3f} (oracle {time.time() - t0:.0f} s)")
    assert sat < 0.5, "the synthetic decoder saturates: the comparison proves too little"
    assert err <= 2e-2 and rel <= 3e-2, (err, rel)


# ---------------------------------------------------------------------------------------------------------------
# (c) rank-384 LoRA on a D = 4096 model
# ---------------------------------------------------------------------------------------------------------------
def _materialise(ctx, shapes):
    return {k: ctx.dit_export_param(k).reshape(s) for k, s in shapes.items()}


@pytest.mark.parametrize("quant", [False, True])
def test_fuse_lora_rank_384_at_full_width(ltx, oracle, tmp_path, quant):
    from safetensors.numpy import save_file

    cfg = ltx.default_transformer_config(num_layers=2)
    ocfg = oracle.DiTConfig(num_layers=2)
    lora, n_layers = make_lora(oracle, ocfg, rank=384, seed=11)
    lpath = tmp_path / "lora384.safetensors"
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in lora.items()}, str(lpath))
    F, H, W, S = 2, 8, 8, 128
    rng = np.random.default_rng(5)
    lat = oracle.bf16_round(rng.standard_normal((1, F * H * W, 128)).astype(np.float32))
    cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
    ctx = ltx.Context(0)
    try:
        ctx.dit_init_synthetic(cfg, seed=21)
        w = _materialise(ctx, oracle.dit_param_shapes(ocfg))  # bf16 values, before any quantisation
        if quant:
            ctx.dit_quantize(8)
        assert ctx.fuse_lora(lpath, scale=0.7) == n_layers
        got = _forward(ctx, lat, cx, 0.725, None, F, H, W)
        base = oracle.quantize_dit_weights(w, 8) if quant else w
        wf, nf = oracle.lora_fuse(base, lora, scale=0.7)
        assert nf == n_layers
        if quant:  # dequant -> merge -> requant (LoRAAdapter.swift:104-131)
            for k in {oracle.map_lora_key(k.split(".lora_")[0]) for k in lora if ".lora_" in k}:
                if k in wf:
                    wf[k] = oracle.fake_quant(wf[k], 8)
        else:
            # the merged weights themselves: bf16(W + bf16(bf16(up @ down) * eff)). The 384-term f32 sums are accumulated in another
            # order on the MFMA than in numpy, so a bf16 rounding of a sum may fall the other way: at most one bf16 ulp, on few elements
            k = "transformer_blocks.1.attn1.to_v.weight"
            got_w, ref_w = ctx.dit_export_param(k).reshape(4096, 4096), wf[k]
            diff = np.abs(got_w - ref_w)
            ulp = np.maximum(np.abs(ref_w), 2.0 ** -10) * 2.0 ** -7
            frac = float((diff > 0).mean())
            print(f"merged to_v: {frac:.2e} of the elements differ from the oracle rule, max {float((diff / ulp).max()):.2f} bf16 ulp")
            assert (diff <= ulp).all() and frac <= 2e-2, (frac, float((diff / ulp).max()))
            assert not np.array_equal(got_w, w[k]), "the LoRA did not change to_v"
        ts = np.array([0.725], np.float32)
        ref = oracle.dit_forward(wf, ocfg, lat, cx, ts, None, F, H, W)
        ref0 = oracle.dit_forward(base, ocfg, lat, cx, ts, None, F, H, W)
        r = rel_l2(got, ref)
        print(f"rank-384 LoRA at D=4096 ({'qint8' if quant else 'bf16'}): rel-L2 {r:.3e}; LoRA moves the output by {rel_l2(ref, ref0):.3e}")
        assert r <= (3e-2 if quant else 2e-2), r
        assert rel_l2(ref, ref0) > 5e-2, "LoRA too weak to prove anything"
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------------
# (d) eight layers at config 2's shape
# ---------------------------------------------------------------------------------------------------------------
def test_full_width_eight_blocks_config2_shape_vs_oracle(ltx, oracle):
    ctx = ltx.Context(0)
    try:
        cfg = ltx.default_transformer_config(num_layers=8)
        ctx.dit_init_synthetic(cfg, seed=1234)
        ocfg = oracle.DiTConfig(num_layers=8)
        w = DeviceWeights(ctx, oracle.dit_param_shapes(ocfg))
        F, H, W, S = 4, 16, 24, 1024
        T = F * H * W
        rng = np.random.default_rng(12)
        lat = oracle.bf16_round(rng.standard_normal((1, T, 128)).astype(np.float32))
        cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
        mask = (rng.random((1, S)) > 0.1).astype(np.int32)
        mask[:, 0] = 1
        got = _forward(ctx, lat, cx, 0.421875, mask, F, H, W)
        ref = oracle.dit_forward(w, ocfg, lat, cx, np.array([0.421875], np.float32), mask, F, H, W)
        r, c = rel_l2(got, ref), _cos(got, ref)
        print(f"full width, 8 blocks, T={T}, S={S}, masked: rel-L2 {r:.3e}, cos {c:.6f}")
        assert np.isfinite(got).all() and r <= 2e-2 and c >= 0.9995, (r, c)
    finally:
        ctx.close()


def test_context_cache_keys_do_not_collide_on_high_bits(ltx):
    """versions that differ only in bits 61-63 used to share a cache entry ((ver << 2) + kind): two different contexts
    under such versions must give different outputs, and a raw forward must never be served from a denoise pass's entry."""
    ctx = ltx.Context(0)
    try:
        cfg = ltx.default_transformer_config(num_layers=1, num_attention_heads=4, cross_attention_dim=512, caption_channels=256)
        ctx.dit_init_synthetic(cfg, seed=3)
        F, H, W, S = 1, 4, 4, 32
        T = F * H * W
        lat = torch.empty((1, T, 128), dtype=torch.bfloat16, device="cuda")
        ctx.op_fill_normal_bf16(lat, seed=1)
        c1 = torch.empty((1, S, 256), dtype=torch.bfloat16, device="cuda")
        c2 = torch.empty_like(c1)
        ctx.op_fill_normal_bf16(c1, seed=2)
        ctx.op_fill_normal_bf16(c2, seed=3)
        ts = torch.full((1,), 0.5, dtype=torch.float32, device="cuda")
        outs = []
        for c, ver in ((c1, 5), (c2, 5 | (1 << 61)), (c2, 5 | (1 << 63)), (c2, 0)):
            v = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
            ctx.dit_forward_dev(lat, c, ts, None, F, H, W, v, ctx_version=ver, mask_all_ones=True)
            outs.append(v)
        torch.cuda.synchronize()
        assert not torch.equal(outs[0], outs[3])
        assert torch.equal(outs[1], outs[3]) and torch.equal(outs[2], outs[3])
        # a denoise step under version 5 caches its passes under kinds of their own: the raw forward above stays valid, and a raw
        # forward with a NEW context under the version a denoise call used is recomputed, not served from the loop's entry
        latent = torch.randn((1, 128, F, H, W), device="cuda")
        sig = ltx.sigmas(True, 8, T)
        ctx.denoise_dev(latent, sig[:2], c2, None, F, H, W, ctx_version=9, mask_all_ones=True)
        v = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
        ctx.dit_forward_dev(lat, c1, ts, None, F, H, W, v, ctx_version=9, mask_all_ones=True)
        torch.cuda.synchronize()
        assert torch.equal(v, outs[0])
    finally:
        ctx.close()


def test_conv_kernel_variants_agree_bit_for_bit_per_epilogue(ltx, oracle, vae_from_oracle_weights):
    """Round-4 advice: the persistent halo kernel requests the next tile's operands from inside the epilogue, so every epilogue variant
    must be exercised with the persistent walk forced (more tiles than CUs) against one workgroup per tile. With the launcher's switches
    behind ltx_ctx_set_option (round 5) that is one process: `conv_persist` 0 / 1 (and `conv_stagger` 0 / 1) must not move a bit, on the
    192-row kernel (`conv_tall` = 0) for: residual + fused PixelNorm output (128 channels), plain residual with a split-K tail window (256
    channels), the depth-to-space store of an upsampler (d2s == 1), and the whole decode, which ends in conv_out's un-patchify store
    (d2s == 3) and has partial last tiles. The tall kernel (`conv_tall` = 3: forced) has no per-tile form; its staggered twin is compared too."""
    ctx, w = vae_from_oracle_weights
    rng = np.random.default_rng(7)

    def res_block(group, block, F, H, W):
        C = oracle.VAE_CHANNELS[group]
        x = torch.from_numpy(rng.standard_normal((F, H, W, C), dtype=np.float32)).cuda()

        def run(**opts):
            y = x.clone()
            with ctx.options(**opts):
                ctx.vae_res_block_dev(group, block, y, F, H, W)
                torch.cuda.synchronize()
            return y

        return run

    def upsample(group, F, H, W):
        C = oracle.VAE_CHANNELS[group]
        x = torch.from_numpy(rng.standard_normal((F, H, W, C), dtype=np.float32)).cuda()

        def run(**opts):
            out = torch.empty((2 * F - 1, 2 * H, 2 * W, C // 2), dtype=torch.float32, device="cuda")
            with ctx.options(**opts):
                ctx.vae_upsample_dev(group, x, F, H, W, out)
                torch.cuda.synchronize()
            return out

        return run

    def decode(F, H, W):
        lat = torch.from_numpy(rng.standard_normal((1, 128, F, H, W), dtype=np.float32)).cuda()

        def run(**opts):
            frames = torch.empty((8 * (F - 1) + 1, H * 32, W * 32, 3), dtype=torch.float32, device="cuda")
            with ctx.options(**opts):
                ctx.vae_decode_dev(lat, F, H, W, frames)
                torch.cuda.synchronize()
            return frames

        return run

    cases = {"res-block 128 ch (PixelNorm + residual epilogue)": res_block(3, 1, 5, 64, 192),
             "res-block 256 ch (plain residual, tail window)": res_block(2, 0, 13, 64, 96),
             "upsampler 256 ch (depth-to-space store)": upsample(2, 5, 32, 96),
             "whole decode 2x8x12 (un-patchify store, partial tiles)": decode(2, 8, 12)}
    for name, run in cases.items():
        base = run(conv_tall=0)
        assert bool(torch.isfinite(base).all()), name
        for opts in ({"conv_persist": 0}, {"conv_stagger": 0}, {"conv_persist": 0, "conv_stagger": 0}):
            assert torch.equal(run(conv_tall=0, **opts), base), (name, opts)
        tall = run(conv_tall=3)   # 3: wherever the shape allows (the default keeps launches of at most half a round on 192-row tiles)
        assert torch.equal(run(conv_tall=3, conv_stagger=0), tall), name
        r = float((tall - base).norm() / base.norm())
        assert r <= 2e-3, (name, r)  # the tall kernel sums its K-tiles in another order: rounding only
    # the 128-channel stage's first PixelNorm in the upsampler conv's epilogue (round 5) or as the row pass it replaced: the same values
    # up to the order of the 128-term sum of squares and the bf16 rounding of the result
    dec = cases["whole decode 2x8x12 (un-patchify store, partial tiles)"]
    fused, separate = dec(), dec(conv_d2s_pn=0)
    assert float((fused - separate).abs().max()) <= 2e-2 and float((fused - separate).norm() / separate.norm()) <= 2e-3
