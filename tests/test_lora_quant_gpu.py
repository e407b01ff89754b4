"""R21 (qint8/int4 on-the-fly quantisation) and R22 (LoRA fusion) through the C ABI vs the oracle."""
import numpy as np
import pytest

from test_dit_gpu import rel_l2, small_cfg, write_dit_file

pytestmark = pytest.mark.gpu


def _setup(ltx, oracle, tmp_path, seed=31):
    cfg, ocfg = small_cfg(ltx, oracle, heads=2, layers=2, caption=128)
    w = oracle.synth_dit_weights(ocfg, seed=seed)
    path = tmp_path / "dit.safetensors"
    write_dit_file(oracle, w, path)
    rng = np.random.default_rng(seed)
    F, H, W, S = 2, 3, 4, 19
    latent = oracle.bf16_round(rng.standard_normal((1, F * H * W, 128)).astype(np.float32))
    context = oracle.bf16_round(rng.standard_normal((1, S, 128)).astype(np.float32))
    ts = np.array([0.8], np.float32)
    return cfg, ocfg, w, path, (latent, context, ts, F, H, W)


def _fwd(ltx, ctx, inp):
    latent, context, ts, F, H, W = inp
    return ctx.dit_forward(ltx.f32_to_bf16_bits(latent), ltx.f32_to_bf16_bits(context), ts, None, F, H, W)


def make_lora(oracle, ocfg, rank=16, seed=9):
    """ComfyUI-style file keys; mixes lora_down/up with lora_A/B naming and alpha / no-alpha layers."""
    rng = np.random.default_rng(seed)
    D = ocfg.dim
    lora = {}
    layers = []
    for i in range(ocfg.num_layers):
        p = f"diffusion_model.transformer_blocks.{i}."
        layers += [(p + "attn1.to_q", D, D), (p + "attn1.to_k", D, D), (p + "attn1.to_v", D, D), (p + "attn1.to_out.0", D, D),
                   (p + "attn2.to_q", D, D), (p + "attn2.to_out.0", D, D),
                   (p + "ff.net.0.proj", 4 * D, D), (p + "ff.net.2", D, 4 * D)]
    for n, (base, out, inn) in enumerate(layers):
        down = oracle.bf16_round((rng.standard_normal((rank, inn)) / np.sqrt(inn)).astype(np.float32))
        up = oracle.bf16_round((rng.standard_normal((out, rank)) * 0.05).astype(np.float32))
        if n % 3 == 0:
            lora[base + ".lora_A.weight"], lora[base + ".lora_B.weight"] = down, up
        else:
            lora[base + ".lora_down.weight"], lora[base + ".lora_up.weight"] = down, up
        if n % 2 == 0:
            lora[base + ".alpha"] = np.array(8.0, np.float32)
    lora["diffusion_model.not_a_layer.lora_down.weight"] = np.zeros((rank, 64), np.float32)
    lora["diffusion_model.not_a_layer.lora_up.weight"] = np.zeros((64, rank), np.float32)
    return lora, len(layers)


def test_lora_fuse_parity(ltx, oracle, tmp_path):
    from safetensors.numpy import save_file

    cfg, ocfg, w, path, inp = _setup(ltx, oracle, tmp_path)
    lora, n_layers = make_lora(oracle, ocfg)
    lpath = tmp_path / "lora.safetensors"
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in lora.items()}, str(lpath))
    ctx = ltx.Context(0)
    ctx.dit_load(path, cfg)
    base = _fwd(ltx, ctx, inp)
    n = ctx.fuse_lora(lpath, scale=0.8)
    assert n == n_layers  # the unmapped layer is skipped, not an error (LoRAAdapter.swift:136-139)
    got = _fwd(ltx, ctx, inp)
    wf, nf = oracle.lora_fuse(w, lora, scale=0.8)
    assert nf == n_layers
    ref = oracle.dit_forward(wf, ocfg, *inp[:3], None, *inp[3:])
    ref0 = oracle.dit_forward(w, ocfg, *inp[:3], None, *inp[3:])
    assert rel_l2(got, ref) <= 2e-2
    assert rel_l2(base, ref0) <= 2e-2
    assert rel_l2(ref, ref0) > 5e-2, "LoRA too weak to prove anything"
    with pytest.raises(ltx.LTXError) as e:
        ctx.fuse_lora(tmp_path / "missing.safetensors")
    assert e.value.case == "fileNotFound"
    bad = {"diffusion_model.transformer_blocks.0.attn1.to_q.lora_down.weight": np.zeros((16, 100), np.float32),
           "diffusion_model.transformer_blocks.0.attn1.to_q.lora_up.weight": np.zeros((ocfg.dim, 16), np.float32)}
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(ltx.LTXError) as e:
        ctx.fuse_lora(tmp_path / "bad.safetensors")
    assert e.value.case == "invalidLoRA"
    ctx.close()


@pytest.mark.parametrize("bits,tol", [(8, 3e-2), (4, 8e-2)])
def test_quantized_forward_parity(ltx, oracle, tmp_path, bits, tol):
    cfg, ocfg, w, path, inp = _setup(ltx, oracle, tmp_path, seed=40 + bits)
    ctx = ltx.Context(0)
    ctx.dit_load(path, cfg, quant_bits=bits, group_size=64)
    got = _fwd(ltx, ctx, inp)
    wq = oracle.quantize_dit_weights(w, bits)
    ref = oracle.dit_forward(wq, ocfg, *inp[:3], None, *inp[3:])
    ref16 = oracle.dit_forward(w, ocfg, *inp[:3], None, *inp[3:])
    assert rel_l2(got, ref) <= tol, rel_l2(got, ref)
    # the quantised model is closer to the quantised oracle than to the bf16 one (int4: clearly so)
    if bits == 4:
        assert rel_l2(got, ref) < rel_l2(got, ref16)
    with pytest.raises(ltx.LTXError) as e:
        ctx.dit_load(path, cfg, quant_bits=3)
    assert e.value.case == "invalidConfiguration"
    ctx.close()


def test_lora_on_quantized_model(ltx, oracle, tmp_path):
    """dequant -> merge -> requant (LoRAAdapter.swift:104-131)."""
    from safetensors.numpy import save_file

    cfg, ocfg, w, path, inp = _setup(ltx, oracle, tmp_path, seed=77)
    lora, n_layers = make_lora(oracle, ocfg, seed=5)
    lpath = tmp_path / "lora.safetensors"
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in lora.items()}, str(lpath))
    ctx = ltx.Context(0)
    ctx.dit_load(path, cfg, quant_bits=8)
    assert ctx.fuse_lora(lpath, 1.0) == n_layers
    got = _fwd(ltx, ctx, inp)
    wq = oracle.quantize_dit_weights(w, 8)
    wf, _ = oracle.lora_fuse(wq, lora, 1.0)
    fused_keys = {oracle.map_lora_key(k.split(".lora_")[0]) for k in lora if ".lora_" in k}
    for k in fused_keys:
        if k in wf:
            wf[k] = oracle.fake_quant(wf[k], 8)
    ref = oracle.dit_forward(wf, ocfg, *inp[:3], None, *inp[3:])
    assert rel_l2(got, ref) <= 3e-2
    ctx.close()


@pytest.mark.parametrize("bits", [8, 4])
def test_quantised_storage_is_real_and_matches_the_oracle_rule(ltx, oracle, tmp_path, bits):
    """ltx_dit_quantize keeps codes + bf16 group scale / bias in HBM and RELEASES the bf16 weights (LTXQuantizationConfig.swift:19-62:
    the reference holds 8 / 4-bit codes, not de-quantised copies). Checked through the ABI: (1) free device memory grows by about
    (bf16 bytes - code bytes) when a synthetic model is quantised; (2) every exported weight equals bf16(oracle.fake_quant(w)) of the
    exported bf16 weight - bit for bit, including the to_q / to_k row views of the fused matrix and 4-bit nibble packing."""
    import torch

    cfg = ltx.default_transformer_config(num_layers=2, num_attention_heads=8, cross_attention_dim=1024, caption_channels=256)
    ctx = ltx.Context(0)
    try:
        ctx.dit_init_synthetic(cfg, seed=5)
        keys = ["transformer_blocks.1.attn1.to_q.weight", "transformer_blocks.1.attn1.to_k.weight", "transformer_blocks.0.ff.project_in.proj.weight",
                "transformer_blocks.1.ff.project_out.weight", "patchify_proj.weight", "adaln_single.linear.weight", "proj_out.weight"]
        shapes = {"patchify_proj.weight": (1024, 128), "adaln_single.linear.weight": (6144, 1024), "proj_out.weight": (128, 1024),
                  "transformer_blocks.0.ff.project_in.proj.weight": (4096, 1024), "transformer_blocks.1.ff.project_out.weight": (1024, 4096)}
        before = {k: ctx.dit_export_param(k) for k in keys}
        n_weights = sum(int(np.prod(s)) for k, s in oracle.dit_param_shapes(oracle.DiTConfig(num_layers=2, num_heads=8, caption_channels=256)).items()
                        if k.endswith(".weight") and len(s) == 2)
        mem0 = ctx.dit_memory_info()
        assert mem0["quantised_weights"] == 0 and mem0["scratch"] == 0 and n_weights * 2 <= mem0["bf16_weights"] <= n_weights * 2 + 64 * 1024
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        ctx.dit_quantize(bits)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        # (1) by the library's own books: the bf16 arena is gone, what is resident is codes + scales / biases + one scratch matrix
        mem1 = ctx.dit_memory_info()
        codes = n_weights * bits // 8 + n_weights // 64 * 4
        assert mem1["bf16_weights"] == 0, mem1
        assert codes <= mem1["quantised_weights"] <= codes + 64 * 1024 and mem1["scratch"] == 6144 * 1024 * 2, mem1   # largest Linear: adaln_single.linear
        assert mem1["other"] == mem0["other"]
        # (the device-wide free-memory figure is printed, not asserted: in a fresh process it moves by exactly the difference above,
        # +27 MB at 8 bits - tools/memprobe.py - but inside a pytest process the runtime's own pools shift it by tens of MB)
        expect = n_weights * 2 - codes - 6144 * 1024 * 2
        print(f"quantise to {bits} bits: free device memory moved by {free1 - free0} B; the arenas shrank by {expect} B")
        for k in keys:
            w16 = before[k].reshape(shapes.get(k, (1024, 1024)))
            want = oracle.bf16_round(oracle.fake_quant(w16, bits))
            got = ctx.dit_export_param(k).reshape(w16.shape)
            assert np.array_equal(got, want), (k, float(np.abs(got - want).max()))
        with pytest.raises(ltx.LTXError):
            ctx.dit_quantize(bits)   # already quantised
    finally:
        ctx.close()


@pytest.mark.parametrize("cfg", [29])
@pytest.mark.parametrize("M,N,K,split", [(128, 256, 4096, 4), (128, 4096, 4096, 4), (24, 192, 256, 1), (200, 512, 1024, 2), (1, 64, 256, 1),
                                         (128, 1024, 16384, 1), (77, 260, 512, 2), (128, 128, 64 * 37, 1), (100, 68, 64 * 5, 5), (77, 320, 256 * 9, 3),
                                         (128, 64, 256, 1), (5, 128, 256 * 6, 2)])
def test_q8_gemm_dequantises_in_its_b_stage(ltx, oracle, gpu_ctx, M, N, K, split, cfg):
    """SURVEY K11 / LTXQuantizationConfig.swift:19-62: a few-row GEMM on a quantised Linear reads the 8-bit codes themselves and
    de-quantises them in its B stage (w' = bf16(q * scale + bias)). Against (a) the same kernel fed from the scratch matrix that
    every many-row launch uses: bit-identical; (b) an f32 matmul with the oracle's reconstruction of the weights: accumulation
    order only. Integer activations and random codes; ragged N (260, 68) and M (77), K split or not, every ring-slot rotation (1..256
    K-tiles per workgroup). cfg 30: the few-row kernel (operand rings of their own; M <= 128), cfg 29: the 128x64 ring kernel."""
    import torch

    if cfg == 30 and not (M <= 128 and N % 64 == 0 and K % 256 == 0 and (K // 256) % split == 0):
        pytest.skip("the few-row kernel takes M <= 128, N % 64 == 0 and whole 256-wide macro-tiles of K per split")

    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    codes = rng.integers(0, 256, (N, K)).astype(np.uint8)
    scales = oracle.bf16_round((rng.random((N, K // 64)) * 0.01 + 0.002).astype(np.float32))
    biases = oracle.bf16_round((-rng.random((N, K // 64)) * 1.0).astype(np.float32))
    bias = rng.integers(-5, 6, (N,)).astype(np.float32)
    w = oracle.bf16_round(codes.reshape(N, K // 64, 64).astype(np.float32) * scales[:, :, None] + biases[:, :, None]).reshape(N, K)
    to_bf = lambda a: torch.from_numpy(ltx.f32_to_bf16_bits(a).astype(np.int16)).cuda().view(torch.bfloat16)
    Ad, sd, bd = to_bf(A), to_bf(scales), to_bf(biases)
    cd = torch.from_numpy(codes).cuda()
    biasd = torch.from_numpy(bias).cuda()
    o1 = torch.full((M, N), float("nan"), device="cuda")
    o2 = torch.full((M, N), float("nan"), device="cuda")
    gpu_ctx.op_gemm_q8(Ad, cd, sd, bd, biasd, o1, split_k=split, tile_cfg=cfg)
    gpu_ctx.op_gemm_q8(Ad, cd, sd, bd, biasd, o2, split_k=split, via_scratch=True, tile_cfg=cfg)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2), float((o1 - o2).abs().max())
    ref = A.astype(np.float64) @ w.astype(np.float64).T + bias
    got = o1.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


def test_quantised_few_row_forward_is_the_scratch_path_forward(ltx, oracle, tmp_path):
    """The whole forward at few tokens (every Linear's codes de-quantised in the GEMMs) against the same model with option "qb_off" = 1
    semantics, i.e. against a many-row-style scratch de-quantisation: here checked through the oracle - both must meet the 8-bit
    tolerance - and through determinism (two forwards bit-identical)."""
    cfg, ocfg, w, path, inp = _setup(ltx, oracle, tmp_path, seed=61)
    ctx = ltx.Context(0)
    try:
        ctx.dit_load(path, cfg, quant_bits=8, group_size=64)
        a = _fwd(ltx, ctx, inp)
        b = _fwd(ltx, ctx, inp)
        assert np.array_equal(a, b)
        ref = oracle.dit_forward(oracle.quantize_dit_weights(w, 8), ocfg, *inp[:3], None, *inp[3:])
        assert rel_l2(a, ref) <= 3e-2, rel_l2(a, ref)
    finally:
        ctx.close()
