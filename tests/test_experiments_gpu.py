"""Kernels that were built, measured and NOT selected: the generated-assembly one-wave-per-SIMD GEMMs (tile_cfg
71-74), the phased 8-wave GEMM (41/42) and the plain-HIP layout reference of the 48-query attention kernel (option "attn_impl" = 3). They
are not in the product library; these tests need the experiments build:

    make -C ltx-video-swift-mlx_amd/csrc EXPERIMENTS=1
    LTX_LIB=ltx-video-swift-mlx_amd/csrc/build_exp/libltxhip_exp.so python -m pytest tests -m experiments

They carry the `experiments` marker (not `gpu`), so `-m gpu` measures the shipped path only; without the experiments build or
without a GPU they skip.
"""
import math

import numpy as np
import pytest
import torch

from test_kernels_gpu import as_f32, dev_bf16, dev_f32

pytestmark = [pytest.mark.experiments, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")]


@pytest.fixture(autouse=True)
def _need_experiments_build(ltx):
    if not ltx._lib.HAS_EXPERIMENTS:
        pytest.skip("needs the experiments build (LTX_LIB=.../build_exp/libltxhip_exp.so)")


@pytest.mark.parametrize("cfg", [71, 72, 73, 74])
@pytest.mark.parametrize("M,N,K,reps", [(192, 256, 64, 1), (192, 256, 128, 1), (384, 512, 448, 1), (192, 768, 7 * 64, 1), (576, 256, 13 * 64, 2),
                                        (1536, 2048, 4096, 4), (768, 4096, 1024, 3)])
def test_gemm_assembly_kernel_integer_exact(gpu_ctx, cfg, M, N, K, reps):
    """One-wave-per-SIMD kernels with the generated assembly main loop (tile 192x256 = cfg 71, 192x128 = cfg 72: three register
    sets, six-tile loop body; cfg 73: 192x128 with an LDS-DMA ring of four slots, four-tile loop body; cfg 74: the same with the B fragments loaded straight to registers; all left after any tile): bit-exact on integer data for 1, 2, 7, 13 and 64 K-tiles - every exit point of the loop body and both LDS slots -
    with bias, an f32 and a bf16 output through its own epilogue; the large shapes are repeated with fresh operands to screen the
    register-set / LDS-slot rotation for races (a stale or early-read tile shows as a wrong integer)."""
    for r in range(reps):
        rng = np.random.default_rng(M + N + K + cfg + 1000 * r)
        A = rng.integers(-3, 4, (M, K)).astype(np.float32)
        B = rng.integers(-3, 4, (N, K)).astype(np.float32)
        bias = rng.integers(-5, 6, (N,)).astype(np.float32)
        out = torch.empty((M, N), device="cuda")
        outb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), tile_cfg=cfg, out_f32=out, out_bf16=outb)
        torch.cuda.synchronize()
        ref = A @ B.T + bias
        got = as_f32(out)
        assert np.array_equal(got, ref), f"rep {r}: {np.count_nonzero(got != ref)} wrong, max diff {np.abs(got - ref).max()}"
        assert np.array_equal(as_f32(outb), torch.from_numpy(ref).to(torch.bfloat16).float().numpy())


def test_gemm_assembly_kernel_epilogue_matches_ring_kernel(gpu_ctx, ltx):
    """GELU-tanh + bias + bf16 output (the FFN's first GEMM) and the refusal of shapes the assembly kernel does not take."""
    rng = np.random.default_rng(9)
    M, N, K = 384, 512, 256
    A = dev_bf16(rng.standard_normal((M, K)))
    B = dev_bf16(rng.standard_normal((N, K)) * 0.1)
    bias = dev_f32(rng.standard_normal((N,)))
    o1 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm(A, B, bias, act=1, tile_cfg=21, out_bf16=o1)
    gpu_ctx.op_gemm(A, B, bias, act=1, tile_cfg=71, out_bf16=o2)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)  # same products, same f32 accumulation order per K-tile, same epilogue arithmetic
    with pytest.raises(ltx.LTXError):
        gpu_ctx.op_gemm(A[:100], B, bias, tile_cfg=71, out_bf16=o2[:100])


@pytest.mark.parametrize("cfg", [41, 42, 31])  # 31: the ring tile on 32x32x16 MFMAs (2 x 4 waves of 96 x 32), same staging as the product ring
@pytest.mark.parametrize("M,N,K,reps", [(192, 256, 128, 1), (256, 256, 192, 1), (100, 60, 320, 1), (500, 700, 256, 2),
                                        (777, 1000, 448, 2), (1536, 1024, 4096, 6), (1536, 2048, 1024, 6), (3000, 768, 2112, 3)])
def test_gemm_pingpong_integer_exact(gpu_ctx, cfg, M, N, K, reps):
    """Phased 8-wave kernel (two wave groups half a phase apart, 2-slot LDS, counted vmcnt): bit-exact on integer
    data for 2/3/even/odd K-tile counts and ragged edges; the large shapes are repeated with fresh operands to
    screen the staging schedule for LDS races (a stale or early-read tile shows as a wrong integer)."""
    for r in range(reps):
        rng = np.random.default_rng(M + N + K + cfg + 1000 * r)
        A = rng.integers(-3, 4, (M, K)).astype(np.float32)
        B = rng.integers(-3, 4, (N, K)).astype(np.float32)
        out = torch.empty((M, N), device="cuda")
        gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), None, tile_cfg=cfg, out_f32=out)
        torch.cuda.synchronize()
        ref = A @ B.T
        got = as_f32(out)
        assert np.array_equal(got, ref), f"rep {r}: {np.count_nonzero(got != ref)} wrong, max diff {np.abs(got - ref).max()}"


def test_attention_layout_reference_kernel(ltx, gpu_ctx):
    """attn_fwd_kernel_w48_ref (plain HIP): pins the LDS images / operand mapping the assembly kernel uses - delta softmax, exact V rows."""
    from test_kernels_gpu import _attn_inputs

    ltx.set_option("attn_impl", 3)
    try:
        B, H, Tq, Tk = 1, 2, 192, 512
        D = H * 128
        rng = np.random.default_rng(0)
        sel = rng.integers(0, Tk, (H, Tq))
        q = np.zeros((B, Tq, D), np.float32)
        k = np.zeros((B, Tk, D), np.float32)
        codes = rng.choice([-1.0, 1.0], (H, Tk, 128)).astype(np.float32)
        for h in range(H):
            k[0, :, h * 128:(h + 1) * 128] = codes[h]
            q[0, :, h * 128:(h + 1) * 128] = codes[h][sel[h]] * 8.0
        v = rng.integers(-8, 9, (B, Tk, D)).astype(np.float32)
        qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q, k=k, v=v)
        o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
        gpu_ctx.op_attention(qd, kd, vt, None, H, o, 1.0 / math.sqrt(128.0))
        torch.cuda.synchronize()
        got = as_f32(o)
        for h in range(H):
            assert np.abs(got[0, :, h * 128:(h + 1) * 128] - v[0, sel[h], h * 128:(h + 1) * 128]).max() <= 1e-2
    finally:
        ltx.set_option("attn_impl", 0)


@pytest.mark.parametrize("M,N,K,split", [(128, 64, 64, 1), (128, 4096, 4096, 1), (128, 4096, 4096, 4), (77, 200, 1024, 2), (1, 64, 256, 1),
                                         (128, 8192, 4096, 2), (33, 1000, 16384, 8), (128, 128, 320, 1), (100, 4096, 128, 1)])
def test_gemm_weight_streaming_kernel_integer_exact(gpu_ctx, M, N, K, split):
    """tile_cfg 90: 128 x 64 tile, the four waves split the K-tiles and read their fragments straight from global memory in a
    permuted k order (16 consecutive k per lane), partial tiles summed through LDS in wave order, K split over workgroups through
    the workspace. Bit-exact on integer data for ragged M / N, 1 ... 256 K-tiles (waves with 0, 1, odd and even tile counts), with
    bias, f32 + bf16 outputs."""
    rng = np.random.default_rng(M + N + K + split)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    B = rng.integers(-3, 4, (N, K)).astype(np.float32)
    bias = rng.integers(-5, 6, (N,)).astype(np.float32)
    out = torch.empty((M, N), device="cuda")
    outb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), tile_cfg=(split * 100 if split > 1 else 0) + 90, out_f32=out, out_bf16=outb)
    torch.cuda.synchronize()
    ref = A @ B.T + bias
    assert np.array_equal(as_f32(out), ref), f"{np.count_nonzero(as_f32(out) != ref)} wrong"
    assert np.array_equal(as_f32(outb), torch.from_numpy(ref).to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("M,N,K", [(128, 128, 256), (128, 64, 512), (100, 192, 256 * 3), (128, 16384, 4096), (128, 4096, 256 * 5), (17, 64, 256 * 7), (128, 256, 16384)])
def test_gemm_few_row_kernel_integer_exact(gpu_ctx, M, N, K):
    """tile_cfg 30: the few-row kernel (activation ring of 3 slots staged by waves 0-1, weight ring of 12 slots staged by waves 2-3,
    each role on its own counted vmcnt, one barrier per K-tile; weights in macro-tiles of four K-tiles): bit-exact on integer data for
    1 .. 64 macro-tiles (shorter and longer than both rings), ragged M, bias + f32 and bf16 outputs through the ring kernel's epilogue."""
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    B = rng.integers(-3, 4, (N, K)).astype(np.float32)
    bias = rng.integers(-5, 6, (N,)).astype(np.float32)
    out = torch.full((M, N), float("nan"), device="cuda")
    outb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), tile_cfg=30, out_f32=out, out_bf16=outb)
    torch.cuda.synchronize()
    ref = A @ B.T + bias
    assert np.array_equal(as_f32(out), ref)
    assert np.array_equal(as_f32(outb), torch.from_numpy(ref).to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("M,N,K,S", [(128, 4096, 4096, 4), (128, 256, 16384, 8), (77, 320, 256 * 9, 3), (1, 64, 256 * 2, 2)])
def test_gemm_few_row_kernel_split_k_integer_exact(gpu_ctx, M, N, K, S):
    rng = np.random.default_rng(M + N + K + S)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    B = rng.integers(-3, 4, (N, K)).astype(np.float32)
    bias = rng.integers(-5, 6, (N,)).astype(np.float32)
    out = torch.empty((M, N), device="cuda")
    gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), tile_cfg=S * 100 + 30, out_f32=out)
    torch.cuda.synchronize()
    assert np.array_equal(as_f32(out), A @ B.T + bias)


@pytest.mark.parametrize("M,N,K,split", [(128, 256, 4096, 4), (128, 4096, 4096, 4), (128, 1024, 16384, 1), (77, 320, 256 * 9, 3), (128, 64, 256, 1),
                                         (5, 128, 256 * 6, 2)])
def test_q8_few_row_kernel_equals_the_scratch_path(ltx, oracle, gpu_ctx, M, N, K, split):
    """The few-row kernel's de-quantising instance (codes in macro-tiles of 4 rows x 256 B, staged three macro-tiles ahead) against the
    same kernel fed from the scratch matrix: bit-identical."""
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    codes = rng.integers(0, 256, (N, K)).astype(np.uint8)
    scales = oracle.bf16_round((rng.random((N, K // 64)) * 0.01 + 0.002).astype(np.float32))
    biases = oracle.bf16_round((-rng.random((N, K // 64)) * 1.0).astype(np.float32))
    to_bf = lambda a: torch.from_numpy(ltx.f32_to_bf16_bits(a).astype(np.int16)).cuda().view(torch.bfloat16)
    Ad, sd, bd = to_bf(A), to_bf(scales), to_bf(biases)
    cd = torch.from_numpy(codes).cuda()
    biasd = torch.from_numpy(rng.integers(-5, 6, (N,)).astype(np.float32)).cuda()
    o1 = torch.full((M, N), float("nan"), device="cuda")
    o2 = torch.full((M, N), float("nan"), device="cuda")
    o3 = torch.full((M, N), float("nan"), device="cuda")
    gpu_ctx.op_gemm_q8(Ad, cd, sd, bd, biasd, o1, split_k=split, tile_cfg=30)
    gpu_ctx.op_gemm_q8(Ad, cd, sd, bd, biasd, o2, split_k=split, via_scratch=True, tile_cfg=30)
    gpu_ctx.op_gemm_q8(Ad, cd, sd, bd, biasd, o3, split_k=split, tile_cfg=29)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and bool(torch.isfinite(o1).all())
    assert float((o1 - o3).abs().max()) <= 1e-3 * max(1.0, float(o3.abs().max()))
