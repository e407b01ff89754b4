"""Parity of each hand-written gfx950 kernel against the oracle / an f32 reference, through the C ABI.

Tolerances (stated per test): GEMM and attention take bf16 inputs and accumulate in f32, so against an f32
reference on the SAME bf16-rounded inputs the only differences are accumulation order (GEMM: rel 1e-5) and, for
attention, the bf16 rounding of P before the PV product (abs 2e-2 on O(1) outputs). Integer-valued inputs make the
GEMM exact and catch any fragment-layout error bit for bit.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev_bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).cuda()


def dev_f32(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def as_f32(t):
    return t.float().cpu().numpy()


@pytest.mark.parametrize("cfg", [0, 1, 3, 4, 21, 23, 25])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (105, 200, 128), (1, 128, 64), (577, 130, 320), (200, 136, 64), (384, 256, 1024), (130, 260, 448), (64, 128, 576), (300, 192, 256)])
def test_gemm_integer_exact(gpu_ctx, cfg, M, N, K):
    """Small-integer operands: every product and partial sum is exact in f32 -> result must be bit-exact."""
    rng = np.random.default_rng(M * 7 + N * 3 + K + cfg)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    B = rng.integers(-3, 4, (N, K)).astype(np.float32)
    bias = rng.integers(-5, 6, (N,)).astype(np.float32)
    out = torch.full((M, N + (4 - N % 4) % 4), -777.0, device="cuda")
    gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), act=0, tile_cfg=cfg, out_f32=out)
    torch.cuda.synchronize()
    ref = A @ B.T + bias
    got = as_f32(out)[:, :N]
    assert np.array_equal(got, ref), f"max diff {np.abs(got - ref).max()}"
    if out.shape[1] > N:  # padding columns untouched
        assert np.all(as_f32(out)[:, N:] == -777.0)


@pytest.mark.parametrize("M,N,K,S,cfg", [(200, 136, 2048, 4, 21), (64, 512, 4096, 8, 21), (300, 192, 1728, 3, 25), (1536, 1024, 3456, 4, 21)])
def test_gemm_split_k_integer_exact(gpu_ctx, M, N, K, S, cfg):
    """Deterministic split-K (grid.y workgroups per tile, f32 partials, fixed-order finish pass with the epilogue): bit-exact
    on integer data, uneven K-tile shares (27 tiles over 3 splits, 54 over 4), bias + bf16 mirror through the finish pass."""
    rng = np.random.default_rng(M + N + K + S)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    B = rng.integers(-3, 4, (N, K)).astype(np.float32)
    bias = rng.integers(-5, 6, (N,)).astype(np.float32)
    out = torch.empty((M, N), device="cuda")
    outb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), tile_cfg=S * 100 + cfg, out_f32=out, out_bf16=outb)
    torch.cuda.synchronize()
    ref = A @ B.T + bias
    assert np.array_equal(as_f32(out), ref)
    assert np.array_equal(as_f32(outb), torch.from_numpy(ref).to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("M,N,K,reps", [(192, 256, 64, 1), (192, 256, 128, 1), (192, 256, 192, 1), (384, 512, 448, 1), (192, 768, 7 * 64, 1),
                                        (576, 256, 13 * 64, 2), (1536, 2048, 4096, 4), (768, 4096, 1024, 3), (1536, 8192, 4096, 2)])
def test_gemm_192x256_kernel_integer_exact(gpu_ctx, M, N, K, reps):
    """tile_cfg 75 (tools/gen_gemm_asm_dtl.py): 192x256 tile, one wave per SIMD, two LDS slots, both k-steps' fragments in registers,
    LDS-DMA of tile t+2 into the slot of tile t. Bit-exact on integer data for 1, 2, 3, 7, 13, 16 and 64 K-tiles (both exits of the
    two-tile loop body, either slot last) with bias, an f32 and a bf16 output; the large shapes are repeated with fresh operands
    to screen the slot reuse for races (a stale or early-read tile shows as a wrong integer)."""
    for r in range(reps):
        rng = np.random.default_rng(M + N + K + 75 + 1000 * r)
        A = rng.integers(-3, 4, (M, K)).astype(np.float32)
        B = rng.integers(-3, 4, (N, K)).astype(np.float32)
        bias = rng.integers(-5, 6, (N,)).astype(np.float32)
        out = torch.empty((M, N), device="cuda")
        outb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        gpu_ctx.op_gemm(dev_bf16(A), dev_bf16(B), dev_f32(bias), tile_cfg=75, out_f32=out, out_bf16=outb)
        torch.cuda.synchronize()
        ref = A @ B.T + bias
        got = as_f32(out)
        assert np.array_equal(got, ref), f"rep {r}: {np.count_nonzero(got != ref)} wrong, max diff {np.abs(got - ref).max()}"
        assert np.array_equal(as_f32(outb), torch.from_numpy(ref).to(torch.bfloat16).float().numpy())


def test_gemm_192x256_kernel_epilogue_matches_ring_kernel(gpu_ctx, ltx):
    """GELU-tanh + bias + bf16 output (the FFN's first GEMM) and an f32 output with bias (the fused q|k projection): same products,
    same f32 accumulation order per K-tile, same epilogue arithmetic as the ring kernel -> bit-identical; shapes it does not take
    are refused."""
    rng = np.random.default_rng(9)
    M, N, K = 384, 512, 256
    A = dev_bf16(rng.standard_normal((M, K)))
    B = dev_bf16(rng.standard_normal((N, K)) * 0.1)
    bias = dev_f32(rng.standard_normal((N,)))
    o1 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm(A, B, bias, act=1, tile_cfg=21, out_bf16=o1)
    gpu_ctx.op_gemm(A, B, bias, act=1, tile_cfg=75, out_bf16=o2)
    f1 = torch.empty((M, N), device="cuda")
    f2 = torch.empty((M, N), device="cuda")
    gpu_ctx.op_gemm(A, B, bias, tile_cfg=21, out_f32=f1)
    gpu_ctx.op_gemm(A, B, bias, tile_cfg=75, out_f32=f2)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(f1, f2)
    with pytest.raises(ltx.LTXError):
        gpu_ctx.op_gemm(A[:100], B, bias, tile_cfg=75, out_bf16=o2[:100])


@pytest.mark.parametrize("M,N,K,act", [(1536, 512, 4096, 0), (300, 256, 1024, 1), (128, 4096, 256, 2)])
def test_gemm_random_vs_f32(gpu_ctx, M, N, K, act):
    rng = np.random.default_rng(11)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,)).astype(np.float32)
    Ad, Bd = dev_bf16(A), dev_bf16(B)
    out = torch.empty((M, N), device="cuda")
    outb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm(Ad, Bd, dev_f32(bias), act=act, out_f32=out, out_bf16=outb)
    torch.cuda.synchronize()
    ref = Ad.float().cpu() @ Bd.float().cpu().T + torch.from_numpy(bias)
    if act == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    elif act == 2:
        ref = torch.nn.functional.silu(ref)
    ref = ref.numpy()
    got = as_f32(out)
    # f32 accumulation order only
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()) + 1e-5
    gb = as_f32(outb)
    assert np.abs(gb - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-3


def test_gemm_gated_residual(gpu_ctx):
    rng = np.random.default_rng(5)
    M, N, K = 200, 256, 512
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,)).astype(np.float32)
    gate = rng.standard_normal((N,)).astype(np.float32)
    x0 = rng.standard_normal((M, N)).astype(np.float32)
    Ad, Bd = dev_bf16(A), dev_bf16(B)
    x = dev_f32(x0)
    mirror = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_gemm_gated_residual(Ad, Bd, dev_f32(bias), dev_f32(gate), 1.0, x, mirror)
    torch.cuda.synchronize()
    ref = x0 + gate[None] * ((Ad.float().cpu() @ Bd.float().cpu().T).numpy() + bias[None])
    assert np.abs(as_f32(x) - ref).max() <= 5e-5
    assert np.abs(as_f32(mirror) - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-3
    # scalar gate (cross-attention scale path)
    x = dev_f32(x0)
    gpu_ctx.op_gemm_gated_residual(Ad, Bd, dev_f32(bias), None, 0.5, x, None)
    torch.cuda.synchronize()
    ref = x0 + 0.5 * ((Ad.float().cpu() @ Bd.float().cpu().T).numpy() + bias[None])
    assert np.abs(as_f32(x) - ref).max() <= 5e-5


@pytest.mark.parametrize("tokens,N,K", [(1536, 512, 1024), (1000, 384, 2048), (777, 256, 512), (70, 256, 4096), (128, 512, 4096), (301, 130 * 2, 256)])
def test_value_projection_transposed(gpu_ctx, tokens, N, K):
    """V^T = (X.Wv^T + b)^T in the layout the attention op takes. Both routes of the helper: tokens as GEMM rows with the transposed
    epilogue store (enough tiles: integer-exact, ragged 4-row groups at the end) and the swapped-operand launch with a K split (few
    tokens). Padding columns of V^T stay untouched."""
    rng = np.random.default_rng(tokens + N + K)
    X = rng.integers(-3, 4, (tokens, K)).astype(np.float32)
    W = rng.integers(-3, 4, (N, K)).astype(np.float32)
    b = rng.integers(-8, 9, (N,)).astype(np.float32)
    ldvt = ((tokens + 63) // 64) * 64
    vt = torch.full((N, ldvt), 7.0, device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_value_projection_t(dev_bf16(X), dev_bf16(W), dev_f32(b), vt)
    torch.cuda.synchronize()
    ref = (X @ W.T + b[None]).T
    got = as_f32(vt)
    want = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()  # integer sums are exact in f32; one rounding to bf16
    assert np.array_equal(got[:, :tokens], want)
    assert np.all(got[:, tokens:] == 7.0)


@pytest.mark.parametrize("M,in_act", [(1, 0), (2, 2), (5, 2)])
def test_gemv(gpu_ctx, M, in_act):
    rng = np.random.default_rng(3)
    N, K = 300, 1024
    a = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,)).astype(np.float32)
    Wd = dev_bf16(W)
    out = torch.empty((M, N), device="cuda")
    gpu_ctx.op_gemv(dev_f32(a), Wd, dev_f32(bias), out, in_act)
    torch.cuda.synchronize()
    ain = a / (1 + np.exp(-a)) if in_act == 2 else a
    ref = ain.astype(np.float64) @ Wd.float().cpu().numpy().astype(np.float64).T + bias
    assert np.abs(as_f32(out) - ref).max() <= 1e-4


def _attn_ref(q, k, v, H, bias, scale):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    qh = q.reshape(B, Tq, H, 128).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, H, 128).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, H, 128).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, Tq, D)


@pytest.mark.parametrize("B,H,Tq,Tk,use_bias", [(1, 2, 128, 128, False), (2, 2, 105, 77, True), (1, 4, 1536, 1536, False),
                                                 (1, 3, 300, 1024, True), (1, 1, 32, 64, False), (1, 1, 1, 3, True),
                                                 (1, 2, 193, 33, False), (1, 1, 384, 97, True), (2, 1, 191, 1000, False)])
def test_attention_vs_f32(gpu_ctx, B, H, Tq, Tk, use_bias):
    rng = np.random.default_rng(B * 1000 + Tq + Tk)
    D = H * 128
    q = dev_bf16(rng.standard_normal((B, Tq, D)))
    k = dev_bf16(rng.standard_normal((B, Tk, D)))
    v = dev_bf16(rng.standard_normal((B, Tk, D)))
    bias = None
    if use_bias:
        m = (rng.random((B, Tk)) > 0.3).astype(np.float32)
        m[:, 0] = 1
        bias = dev_f32((1 - m) * -10000.0)
    ldvt = ((Tk + 63) // 64) * 64
    vt = torch.zeros((B, D, ldvt), device="cuda", dtype=torch.bfloat16)
    vt[:, :, :Tk] = v.transpose(1, 2)
    o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(128.0)
    gpu_ctx.op_attention(q, k, vt, bias, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), H, None if bias is None else bias.cpu(), scale).numpy()
    got = as_f32(o)
    err = np.abs(got - ref).max()
    assert err <= 2e-2, f"max abs err {err}"
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel <= 1e-2, f"rel l2 {rel}"


def test_attention_late_maximum(gpu_ctx):
    """Online-softmax rescale under stress: score magnitudes grow with the key index, so the running maximum moves in
    EVERY key tile including the last, while every third query has flat scores whose maximum settles in the first tile."""
    B, H, Tq, Tk = 1, 2, 200, 448
    D = H * 128
    rng = np.random.default_rng(5)
    q = rng.standard_normal((B, Tq, D)).astype(np.float32)
    k = rng.standard_normal((B, Tk, D)).astype(np.float32)
    ramp = np.linspace(0.2, 3.0, Tk).astype(np.float32)
    k *= ramp[None, :, None]
    q[:, ::3] *= 0.05  # every third query: flat scores, maximum settles early
    v = rng.standard_normal((B, Tk, D)).astype(np.float32)
    qd, kd, vd = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    vt = torch.zeros((B, D, 448), device="cuda", dtype=torch.bfloat16)
    vt[:, :, :Tk] = vd.transpose(1, 2)
    o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(128.0)
    gpu_ctx.op_attention(qd, kd, vt, None, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, scale).numpy()
    got = as_f32(o)
    assert np.abs(got - ref).max() <= 3e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.fixture
def attn_impl(ltx):
    """Force one attention kernel through the launcher's A/B switch (option "attn_impl": 1 = 4-wave, 2 = ping-pong, 3 = plain-HIP layout
    reference of the 48-query kernel, 4 = its assembly main loop; 0 = the launcher's grid-fill choice)."""
    def force(impl):
        ltx.set_option("attn_impl", 0 if impl is None else int(impl))
    yield force
    ltx.set_option("attn_impl", 0)


def _attn_inputs(rng, B, H, Tq, Tk, q=None, k=None, v=None):
    D = H * 128
    q = rng.standard_normal((B, Tq, D)).astype(np.float32) if q is None else q
    k = rng.standard_normal((B, Tk, D)).astype(np.float32) if k is None else k
    v = rng.standard_normal((B, Tk, D)).astype(np.float32) if v is None else v
    qd, kd, vd = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    ldvt = ((Tk + 63) // 64) * 64
    vt = torch.zeros((B, D, ldvt), device="cuda", dtype=torch.bfloat16)
    vt[:, :, :Tk] = vd.transpose(1, 2)
    return qd, kd, vd, vt


@pytest.mark.parametrize("impl", [1, 2, 4, None])
@pytest.mark.parametrize("B,H,Tq,Tk", [(1, 2, 192, 256), (2, 3, 384, 512), (1, 4, 1536, 1024), (1, 1, 576, 1536)])
def test_attention_every_kernel_vs_f32(gpu_ctx, attn_impl, impl, B, H, Tq, Tk):
    """Shapes every kernel takes (Tq % 192 == 0, Tk % 256 == 0, no mask): each of them against the f32 reference, same bounds as
    test_attention_vs_f32. The assembly kernel normalises P by a reference maximum that is raised only when a score exceeds it by
    2^8 - a different rounding of the bf16 P than the running-maximum kernels, not a different result."""
    attn_impl(impl)
    rng = np.random.default_rng(B * 7 + H + Tq + Tk)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk)
    o = torch.empty((B, Tq, H * 128), device="cuda", dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(128.0)
    gpu_ctx.op_attention(qd, kd, vt, None, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, scale).numpy()
    got = as_f32(o)
    assert np.abs(got - ref).max() <= 2e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.parametrize("B,H,Tq,Tk", [(1, 2, 128, 128), (2, 2, 105, 77), (1, 1, 32, 64), (1, 1, 1, 3), (1, 2, 193, 33), (2, 1, 191, 1000),
                                       (1, 2, 1024, 1024), (1, 1, 500, 321), (1, 3, 384, 640), (1, 1, 200, 65), (1, 1, 50, 255)])
def test_attention_assembly_kernel_any_shape(gpu_ctx, attn_impl, B, H, Tq, Tk):
    """The assembly kernel on sizes that are NOT multiples of its 192-query / 64-key tiles: query rows past Tq must not be stored
    (guard band checked), keys past Tk must not contribute (1, 2, 3 and 5 ... 16 key tiles: every exit point of the unrolled loop,
    a single ragged tile, a ragged last tile after full ones)."""
    attn_impl(4)
    rng = np.random.default_rng(B * 5 + H + Tq * 3 + Tk)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk)
    D = H * 128
    obuf = torch.full((B, Tq + 8, D), 7.0, device="cuda", dtype=torch.bfloat16)  # 8 guard rows after the last batch element only
    o = obuf[:, :Tq]
    scale = 1.0 / math.sqrt(128.0)
    if B == 1:
        gpu_ctx.op_attention(qd, kd, vt, None, H, o, scale)
    else:
        o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
        gpu_ctx.op_attention(qd, kd, vt, None, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, scale).numpy()
    got = as_f32(o)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2
    if B == 1:
        assert (as_f32(obuf[:, Tq:]) == 7.0).all(), "rows past Tq were written"


@pytest.mark.parametrize("impl", [1, 2, 4, None])
@pytest.mark.parametrize("B,H,Tq,Tk,masked", [(1, 2, 192, 256, False), (1, 4, 1536, 1024, False), (1, 2, 1536, 1536, False), (2, 2, 105, 77, True),
                                              (1, 1, 500, 321, False), (1, 3, 300, 1024, True), (2, 1, 1536, 1024, True)])
def test_attention_prescaled_query(gpu_ctx, attn_impl, impl, B, H, Tq, Tk, masked):
    """q_prescaled (scale <= 0 at the ABI): Q carries (1/sqrt(128)) * log2(e), rounded to bf16 ONCE by its producer (the DiT's
    q-norm + RoPE pass), and the scores are base-2 exponents - the assembly kernel then runs its stream without the 48 multiplies
    per key tile, the other kernels run with scale = ln 2. Reference: f32 softmax of the SAME bf16 Q' with scale ln 2 (identical
    math); the masked case includes a fully masked batch element (bias in the reference's post-scale units)."""
    if impl == 2 and masked:
        pytest.skip("the ping-pong kernel is not used for masked launches")
    attn_impl(impl)
    rng = np.random.default_rng(B * 13 + H + Tq + Tk)
    c = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
    D = H * 128
    q = (rng.standard_normal((B, Tq, D)) * 1.5).astype(np.float32)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q * np.float32(c))
    bias = None
    if masked:
        m = (rng.random((B, Tk)) > 0.3).astype(np.float32)
        m[:, 0] = 1
        m[0, :] = 0  # every key masked: softmax(s - 10000) = softmax(s)
        bias = dev_f32((1 - m) * -10000.0)
    o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_attention(qd, kd, vt, bias, H, o, 0.0)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None if bias is None else bias.cpu(), math.log(2.0)).numpy()
    got = as_f32(o)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.parametrize("impl", [1, 4])
@pytest.mark.parametrize("B,H,Tq,Tk,all_masked", [(1, 2, 192, 256, False), (2, 2, 105, 77, False), (1, 3, 300, 1024, False), (1, 1, 384, 97, False),
                                                  (2, 1, 1536, 1024, True), (1, 1, 1, 3, False), (1, 2, 200, 4096, False)])
def test_attention_masked_every_kernel(gpu_ctx, attn_impl, impl, B, H, Tq, Tk, all_masked):
    """Additive key mask ((1-m) * -10000, LTXTransformer.swift:141-156) on the 4-wave kernel and on the masked assembly variant (bias
    vector in LDS, added to the scores in raw units before the reference maximum is taken). all_masked: batch element 0 has EVERY
    key masked - the CFG null-mask case: softmax(s - 10000) = softmax(s) must come out, not 0/0."""
    attn_impl(impl)
    rng = np.random.default_rng(B * 11 + H + Tq + Tk)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk)
    m = (rng.random((B, Tk)) > 0.3).astype(np.float32)
    m[:, 0] = 1
    if all_masked:
        m[0, :] = 0
    bias = dev_f32((1 - m) * -10000.0)
    o = torch.empty((B, Tq, H * 128), device="cuda", dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(128.0)
    gpu_ctx.op_attention(qd, kd, vt, bias, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, bias.cpu(), scale).numpy()
    got = as_f32(o)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.parametrize("impl", [1, 2, 4])
@pytest.mark.parametrize("Tk", [64, 256, 1000])
def test_attention_scores_far_below_zero(gpu_ctx, attn_impl, impl, Tk):
    """Every score of every row around -2560 (c*s = -326): exp2 of the unshifted score underflows to 0 and 2^+326 overflows, so
    the very first tile has to set the reference maximum (regression: the assembly kernel once relied on the 2^8 threshold test
    for the first tile as well and returned 0/0 here)."""
    attn_impl(impl)
    rng = np.random.default_rng(Tk)
    Tq, H = 192, 1
    k = (1.0 + 0.05 * rng.standard_normal((1, Tk, 128))).astype(np.float32)
    q = (-20.0 + 0.05 * rng.standard_normal((1, Tq, 128))).astype(np.float32)
    qd, kd, vd, vt = _attn_inputs(rng, 1, H, Tq, Tk, q=q, k=k)
    o = torch.empty((1, Tq, 128), device="cuda", dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(128.0)
    gpu_ctx.op_attention(qd, kd, vt, None, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, scale).numpy()
    got = as_f32(o)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2


@pytest.mark.parametrize("impl", [1, 2, 4])
def test_attention_reference_maximum_stress(gpu_ctx, attn_impl, impl):
    """The assembly kernel's rare path under stress, the other kernels beside it: key magnitudes grow with the key index (the
    maximum moves in every tile, by more than the 2^8 threshold several times), every third query has flat scores (its reference
    settles in the first tile and must never be rescaled wrongly), and one query block sees a single huge late key (delta)."""
    attn_impl(impl)
    B, H, Tq, Tk = 1, 2, 384, 768
    D = H * 128
    rng = np.random.default_rng(5)
    q = rng.standard_normal((B, Tq, D)).astype(np.float32)
    k = rng.standard_normal((B, Tk, D)).astype(np.float32)
    k *= np.linspace(0.2, 6.0, Tk).astype(np.float32)[None, :, None]
    q[:, ::3] *= 0.05
    k[0, 700] = 0.0
    k[0, 700, :128] = q[0, 17, :128] * 4.0  # head 0, query 17: one dominant key in the last tiles
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q, k=k)
    o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(128.0)
    gpu_ctx.op_attention(qd, kd, vt, None, H, o, scale)
    torch.cuda.synchronize()
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, scale).numpy()
    got = as_f32(o)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 3e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.parametrize("impl", [1, 2, 4])
@pytest.mark.parametrize("Tq,Tk", [(1536, 1536), (1536, 1024), (192, 1536)])
def test_attention_is_repeatable_with_cold_caches(gpu_ctx, attn_impl, impl, Tq, Tk):
    """Two different inputs launched alternately, 120 times each, the caches flushed with 1 GB of other traffic every eighth launch:
    all outputs of one input must be bit-identical (a ring slot read before its LDS-DMA has landed would hold the OTHER input's
    keys). Note: the one race of this kind found so far - a prologue that waited for tile 0 only while the first loop step already
    reads the keys of tile 1 - did NOT show up in this isolated form; inside the 48-layer forward it broke bit-repeatability
    every other run, which is what tests/test_full_size_properties_gpu.py repeats six times for."""
    attn_impl(impl)
    rng = np.random.default_rng(Tq + Tk + impl)
    H = 32
    sets = [_attn_inputs(rng, 1, H, Tq, Tk) for _ in range(2)]
    scale = 1.0 / math.sqrt(128.0)
    trash = torch.empty((512 * 2 ** 20,), device="cuda", dtype=torch.bfloat16)
    first = [None, None]
    o = [torch.empty((1, Tq, H * 128), device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    diff = torch.zeros((), device="cuda", dtype=torch.int64)
    for i in range(240):
        qd, kd, vd, vt = sets[i & 1]
        if i % 8 == 0:
            trash.fill_(float(i))
        gpu_ctx.op_attention(qd, kd, vt, None, H, o[i & 1], scale)
        if first[i & 1] is None:
            first[i & 1] = o[i & 1].clone()
        else:
            diff += (o[i & 1] != first[i & 1]).sum()
    torch.cuda.synchronize()
    assert int(diff.item()) == 0


@pytest.mark.parametrize("impl", [4])
def test_attention_integer_layout_48_query_kernels(gpu_ctx, attn_impl, impl):
    """Delta softmax pins the 16x16x32 key permutation, both swizzles and the O store of the 48-query kernels: O must equal the
    selected key's V row exactly (integer V, bf16-exact)."""
    attn_impl(impl)
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


def test_attention_integer_layout(gpu_ctx):
    """One-hot style check that pins the key permutation / Vt layout exactly: with q.k = big for exactly one key per
    query, softmax is a delta and O must equal that key's V row (bf16-exact)."""
    B, H, Tq, Tk = 1, 2, 160, 200
    D = H * 128
    rng = np.random.default_rng(0)
    sel = rng.integers(0, Tk, (H, Tq))
    q = np.zeros((B, Tq, D), np.float32)
    k = np.zeros((B, Tk, D), np.float32)
    # keys: distinct +-1 codes over 128 dims (Tk <= 2^7 would be needed for binary; use random +-1, near-orthogonal)
    codes = rng.choice([-1.0, 1.0], (H, Tk, 128)).astype(np.float32)
    for h in range(H):
        k[0, :, h * 128:(h + 1) * 128] = codes[h]
        q[0, :, h * 128:(h + 1) * 128] = codes[h][sel[h]] * 8.0  # q.k_sel = 1024, others ~ N(0, 8*sqrt(128)=90)
    v = rng.integers(-8, 9, (B, Tk, D)).astype(np.float32)
    ldvt = ((Tk + 63) // 64) * 64
    vt = torch.zeros((B, D, ldvt), device="cuda", dtype=torch.bfloat16)
    vt[:, :, :Tk] = dev_bf16(v).transpose(1, 2)
    o = torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_attention(dev_bf16(q), dev_bf16(k), vt, None, H, o, 1.0 / math.sqrt(128.0))
    torch.cuda.synchronize()
    got = as_f32(o)
    for h in range(H):
        ref = v[0, sel[h], h * 128:(h + 1) * 128]
        assert np.abs(got[0, :, h * 128:(h + 1) * 128] - ref).max() <= 1e-2


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("rows,D", [(37, 512), (128, 4096), (5, 1024)])
def test_norm_mod(gpu_ctx, oracle, kind, rows, D):
    rng = np.random.default_rng(rows + D + kind)
    x = (rng.standard_normal((rows, D)) * 3 + 0.5).astype(np.float32)
    scale = (0.1 * rng.standard_normal((D,))).astype(np.float32)
    shift = (0.1 * rng.standard_normal((D,))).astype(np.float32)
    out = torch.empty((rows, D), device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_norm_mod(dev_f32(x), dev_f32(scale), dev_f32(shift), out, norm_kind=kind, eps=1e-6)
    torch.cuda.synchronize()
    n = oracle.rms_norm(x, None, 1e-6) if kind == 0 else oracle.layer_norm(x, 1e-6)
    ref = n * (1 + scale) + shift
    assert np.abs(as_f32(out) - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-4
    # rounding point of block 0: norm rounded to bf16 before modulation
    if kind == 0:
        gpu_ctx.op_norm_mod(dev_f32(x), dev_f32(scale), dev_f32(shift), out, norm_kind=0, eps=1e-6, round_norm_bf16=True)
        torch.cuda.synchronize()
        ref2 = oracle.bf16_round(n) * (1 + scale) + shift
        assert np.abs(as_f32(out) - ref2).max() <= 2 ** -7 * np.abs(ref2).max() + 1e-4


@pytest.mark.parametrize("use_rope", [True, False])
def test_qknorm_rope(gpu_ctx, oracle, ltx, use_rope):
    F, H, W, heads = 2, 3, 5, 4
    D = heads * 128
    T = F * H * W
    B = 2
    rng = np.random.default_rng(9)
    x = rng.standard_normal((B * T, 2 * D)).astype(np.float32)  # q | k fused buffer, use the second half
    w = (1 + 0.1 * rng.standard_normal((D,))).astype(np.float32)
    cfg = ltx.default_transformer_config(num_attention_heads=heads, cross_attention_dim=D)
    cos, sin = ltx.rope_tables(cfg, F, H, W)
    xd = dev_f32(x)
    out = torch.empty((B * T, D), device="cuda", dtype=torch.bfloat16)
    xin = xd[:, D:]
    wd, cd, sd = dev_f32(w), dev_f32(cos), dev_f32(sin)  # keep alive across the call
    # strided view: pass base pointer of the second half with ldx = 2D
    rc = ltx.lib.ltx_op_qknorm_rope(gpu_ctx._h, xin.data_ptr(), 2 * D, wd.data_ptr(),
                                    cd.data_ptr() if use_rope else None,
                                    sd.data_ptr() if use_rope else None, T, B * T, D, 1e-6, out.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    ref = oracle.rms_norm(x[:, D:], w, 1e-6).reshape(B, T, D)
    if use_rope:
        ref = oracle.apply_split_rope(ref, cos, sin, heads)
    ref = ref.reshape(B * T, D)
    assert np.abs(as_f32(out) - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-4


def test_gemm_row_split_of_a_ragged_last_round(gpu_ctx):
    """Round 4: a launch whose 192x256 tiling ends in a mostly empty round (T = 9984: 52 x 16 tiles = 3.25 rounds of 256 CUs) is split by rows
    into whole rounds on the 192x256 kernel and a tail that takes the launcher's choice for its own shape. Exact on integer data against torch,
    rows of the head bit-equal to the unsplit 192x256 launch, the gated-residual form (in place on the f32 stream, bf16 mirror) included."""
    torch.manual_seed(5)
    M, N, K = 9984, 4096, 256
    A = torch.randint(-2, 3, (M, K), device="cuda").to(torch.bfloat16)
    B = torch.randint(-2, 3, (N, K), device="cuda").to(torch.bfloat16)
    bias = torch.randint(-4, 5, (N,), device="cuda").float()
    ref = A.float() @ B.float().T + bias
    out = torch.full((M, N), float("nan"), device="cuda")
    gpu_ctx.op_gemm(A, B, bias, out_f32=out)                    # the launcher's choice: head + tail
    whole = torch.full((M, N), float("nan"), device="cuda")
    gpu_ctx.op_gemm(A, B, bias, tile_cfg=75, out_f32=whole)     # one 192x256 launch, 3.25 rounds
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(whole, ref)
    Ar = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Br = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    o1 = torch.empty((M, N), device="cuda")
    o2 = torch.empty((M, N), device="cuda")
    gpu_ctx.op_gemm(Ar, Br, bias, out_f32=o1)
    gpu_ctx.op_gemm(Ar, Br, bias, tile_cfg=75, out_f32=o2)
    torch.cuda.synchronize()
    assert torch.equal(o1[:9216], o2[:9216])                    # 48 row tiles = three whole rounds: the same kernel, the same bits
    assert float((o1[9216:] - o2[9216:]).abs().max()) <= 1e-4 * float(o2.abs().max())
    # gated residual, in place, with the bf16 mirror
    x = torch.randint(-3, 4, (M, N), device="cuda").float()
    gate = torch.randint(-2, 3, (1, N), device="cuda").float()
    want = x + gate * ref
    mirror = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    gpu_ctx.op_gemm_gated_residual(A, B, bias, gate, 1.0, x, mirror)
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(mirror.float(), want.to(torch.bfloat16).float())


@pytest.mark.parametrize("splits,K", [(2, 1024), (3, 448), (2, 64 * 5)])
def test_gemm_192x256_split_k(gpu_ctx, splits, K):
    """Round 4: deterministic split-K on the 192x256 kernel (grid.y = K range, raw partial tiles to the workspace, the ring kernels' finish
    pass applies the epilogue): exact on integer data (uneven K ranges included), bias + GELU applied once, bit-repeatable."""
    torch.manual_seed(7)
    M, N = 384, 1024
    A = torch.randint(-2, 3, (M, K), device="cuda").to(torch.bfloat16)
    B = torch.randint(-2, 3, (N, K), device="cuda").to(torch.bfloat16)
    bias = torch.randint(-4, 5, (N,), device="cuda").float()
    ref = A.float() @ B.float().T + bias
    out = torch.full((M, N), float("nan"), device="cuda")
    mir = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    gpu_ctx.op_gemm(A, B, bias, tile_cfg=splits * 100 + 75, out_f32=out, out_bf16=mir)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(mir.float(), ref.to(torch.bfloat16).float())
    Ar = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Br = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    o1 = torch.empty((M, N), device="cuda")
    o2 = torch.empty((M, N), device="cuda")
    o3 = torch.empty((M, N), device="cuda")
    gpu_ctx.op_gemm(Ar, Br, bias, act=1, tile_cfg=splits * 100 + 75, out_f32=o1)
    gpu_ctx.op_gemm(Ar, Br, bias, act=1, tile_cfg=splits * 100 + 75, out_f32=o2)
    gpu_ctx.op_gemm(Ar, Br, bias, act=1, tile_cfg=75, out_f32=o3)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    assert float((o1 - o3).abs().max()) <= 1e-4 * float(o3.abs().max()) + 1e-5


@pytest.mark.parametrize("M,K", [(1536, 16384), (1536, 4096), (384, 16384)])
def test_gated_residual_with_the_next_norm_on_its_finish_pass(gpu_ctx, M, K):
    """Round 4: the FFN's second GEMM at 1536 tokens ends in a split-K finish pass, and the next block's adaLN pass rides on it
    (NormAfter, gemm.h). The pair launched the DiT graph's way and as two launches give the same bits on the f32 stream and on the bf16
    norm rows; launches without a finish pass (K = 4096; 384 rows: the ring kernel's own split) take the two-launch path inside the
    launcher. Against torch: the update within the bf16-product tolerance, the norm rows within bf16 rounding of torch's norm of x."""
    torch.manual_seed(11)
    N = 4096
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    gate = torch.randn(1, N, device="cuda") * 0.5
    scale = torch.randn(N, device="cuda") * 0.2
    shift = torch.randn(N, device="cuda") * 0.2
    x0 = torch.randn(M, N, device="cuda")
    xa, xb_ = x0.clone(), x0.clone()
    na = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    nb = torch.empty_like(na)
    gpu_ctx.op_gemm_gated_residual_norm(A, B, bias, gate, 1.0, xa, scale, shift, na, fused=True)
    gpu_ctx.op_gemm_gated_residual_norm(A, B, bias, gate, 1.0, xb_, scale, shift, nb, fused=False)
    torch.cuda.synchronize()
    assert torch.equal(xa, xb_)
    assert torch.equal(na.view(torch.int16), nb.view(torch.int16))
    prod = A.float() @ B.float().T + bias
    want = x0 + gate * prod
    # the K ranges' partial tiles cross the workspace as bf16 (GemmArgs::split_bf16): two roundings of at most 2^-9 of a partial's
    # magnitude each, scaled by the gate - the rounding the reference's bf16 Linear applies to the whole product
    assert float((xa - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 2 ** -8 * float((gate * prod).abs().max())
    nrm = xa * torch.rsqrt((xa * xa).mean(dim=1, keepdim=True) + 1e-6) * (1 + scale) + shift
    assert float((na.float() - nrm).abs().max()) <= 2 ** -7 * float(nrm.abs().max())


@pytest.mark.parametrize("in_act", [0, 2])
def test_gemv_rows_of_a_batch_are_computed_alike(gpu_ctx, in_act):
    """The timestep / adaLN path runs this kernel with one row per batch element: identical rows must give identical bits (left to
    FMA contraction, the compiler fused the accumulate for some rows and not for others; the full-size batch-consistency test saw the
    1-ulp difference 48 layers later as 1.5e-2)."""
    torch.manual_seed(0)
    for K, N in ((256, 4096), (4096, 24576)):
        a1 = torch.randn(1, K, device="cuda")
        a = a1.repeat(4, 1).contiguous()
        Wt = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        out = torch.empty((4, N), device="cuda")
        gpu_ctx.op_gemv(a, Wt, b, out, in_act=in_act)
        torch.cuda.synchronize()
        for r in range(1, 4):
            assert torch.equal(out[0], out[r]), (K, N, r)
