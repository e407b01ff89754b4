"""The 32x32x16 attention stream (attn_fwd_kernel_x32_asm, tools/gen_attn_x32.py; option "attn_impl" = 5) against an f32 softmax reference.

It takes unmasked launches with prescaled Q (scale <= 0 at the ABI: Q carries (1/sqrt(128)) * log2(e), LTXAttention.swift:192-214
with the scale folded into the q-norm + RoPE pass) and whole 64-key tiles; any number of query rows. What is new against the
48-query 16x16 kernel and therefore tested here: the A / B key halves folded into per-wave fragment addresses, the query block that
a wave pair shares by keys (two partial (m, l, O) combined through LDS in the epilogue), fragment reads that run ahead across
the step boundary, the loop left after any tile count, rows past Tq dropped by the O descriptor.

Status: parity-green on MI355X, measured 4-8 % slower than the 16x16x32 stream (the part clocks lower under 32x32x16 MFMAs), so the
kernel lives in the -DLTX_EXPERIMENTS library only: run with LTX_LIB=.../build_exp/libltxhip_exp.so and -m experiments.
"""
import math

import numpy as np
import pytest
import torch

from test_kernels_gpu import _attn_inputs, _attn_ref, as_f32, dev_bf16

pytestmark = [pytest.mark.experiments, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")]


@pytest.fixture(autouse=True)
def _need_experiments_build(ltx):
    if not ltx._lib.HAS_EXPERIMENTS:
        pytest.skip("needs the experiments build (LTX_LIB=.../build_exp/libltxhip_exp.so)")

C = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
LN2 = math.log(2.0)


@pytest.fixture
def x32(ltx):
    ltx.set_option("attn_impl", 5)
    yield
    ltx.set_option("attn_impl", 0)


def _run(ctx, qd, kd, vt, H, Tq):
    o = torch.full((qd.shape[0], Tq, H * 128), 7.0, device="cuda", dtype=torch.bfloat16)
    ctx.op_attention(qd, kd, vt, None, H, o, 0.0)
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize("B,H,Tq,Tk", [(1, 2, 192, 256), (2, 3, 384, 512), (1, 4, 1536, 1024), (1, 1, 576, 1536), (1, 2, 105, 64),
                                        (1, 1, 1, 64), (2, 1, 191, 1024), (1, 2, 193, 128), (1, 1, 500, 320), (1, 1, 192, 192),
                                        (1, 3, 160, 384), (1, 1, 129, 448)])
def test_x32_vs_f32(gpu_ctx, x32, B, H, Tq, Tk):
    """Tile counts 1..24 (the four-step loop body is left after any tile), ragged query counts (the shared block partly or wholly
    past Tq), several heads and batch elements."""
    rng = np.random.default_rng(B * 13 + H + Tq + Tk)
    q = (rng.standard_normal((B, Tq, H * 128)) * 1.5).astype(np.float32) * np.float32(C)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q)
    got = as_f32(_run(gpu_ctx, qd, kd, vt, H, Tq))
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, LN2).numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2, np.abs(got - ref).max()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


def test_x32_rows_past_tq_are_not_written(gpu_ctx, x32):
    rng = np.random.default_rng(3)
    B, H, Tq, Tk = 1, 2, 200, 256
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk)
    obuf = torch.full((B, Tq + 184, H * 128), 7.0, device="cuda", dtype=torch.bfloat16)
    gpu_ctx.op_attention(qd, kd, vt, None, H, obuf[:, :Tq], 0.0)
    torch.cuda.synchronize()
    assert (as_f32(obuf[:, Tq:]) == 7.0).all(), "rows past Tq were written"
    assert not (as_f32(obuf[:, :Tq]) == 7.0).all()


def test_x32_integer_layout(gpu_ctx, x32):
    """Delta softmax pins the K-row permutation, both swizzles, the A / B key halves, the shared block's key split and the O store:
    O must equal the selected key's V row exactly (integer V, bf16-exact), for selections spread over every key position."""
    B, H, Tq, Tk = 1, 2, 384, 512
    D = H * 128
    rng = np.random.default_rng(0)
    sel = rng.integers(0, Tk, (H, Tq))
    sel[:, :Tk // 2] = np.arange(Tk // 2)[None] * 2 % Tk      # every second key position at least once ...
    sel[1, :Tk // 2] = (np.arange(Tk // 2) * 2 + 1) % Tk       # ... and the odd ones in the other head
    q = np.zeros((B, Tq, D), np.float32)
    k = np.zeros((B, Tk, D), np.float32)
    codes = rng.choice([-1.0, 1.0], (H, Tk, 128)).astype(np.float32)
    for h in range(H):
        k[0, :, h * 128:(h + 1) * 128] = codes[h]
        q[0, :, h * 128:(h + 1) * 128] = codes[h][sel[h]] * 0.5   # q.k_sel = 64 (base-2 exponent), the others ~ N(0, 5.7)
    v = rng.integers(-8, 9, (B, Tk, D)).astype(np.float32)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q, k=k, v=v)
    got = as_f32(_run(gpu_ctx, qd, kd, vt, H, Tq))
    for h in range(H):
        assert np.abs(got[0, :, h * 128:(h + 1) * 128] - v[0, sel[h], h * 128:(h + 1) * 128]).max() <= 1e-2, h


@pytest.mark.parametrize("Tk", [64, 256, 1024])
def test_x32_scores_far_below_zero(gpu_ctx, x32, Tk):
    """Every score around -326 (base 2): the first tile has to set the reference maximum, for the own and the shared block alike."""
    rng = np.random.default_rng(Tk)
    Tq, H = 192, 1
    k = (1.0 + 0.05 * rng.standard_normal((1, Tk, 128))).astype(np.float32)
    q = (-20.0 + 0.05 * rng.standard_normal((1, Tq, 128))).astype(np.float32) * np.float32(C)
    qd, kd, vd, vt = _attn_inputs(rng, 1, H, Tq, Tk, q=q, k=k)
    got = as_f32(_run(gpu_ctx, qd, kd, vt, H, Tq))
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, LN2).numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2


def test_x32_reference_maximum_stress(gpu_ctx, x32):
    """The rare path: key magnitudes grow with the key index (the maximum moves in every tile, by more than 2^8 several times),
    every third query has flat scores, one query sees a single huge late key. The two halves of a shared block raise their
    references independently (different key subsets), so the epilogue combine has to reconcile different maxima."""
    B, H, Tq, Tk = 1, 2, 384, 768
    D = H * 128
    rng = np.random.default_rng(5)
    q = rng.standard_normal((B, Tq, D)).astype(np.float32)
    k = rng.standard_normal((B, Tk, D)).astype(np.float32)
    k *= np.linspace(0.2, 6.0, Tk).astype(np.float32)[None, :, None]
    q[:, ::3] *= 0.05
    k[0, 700] = 0.0
    k[0, 700, :128] = q[0, 17, :128] * 4.0
    k[0, 731] = 0.0
    k[0, 731, :128] = q[0, 150, :128] * 4.0    # query 150 sits in a shared block (rows 128..191 of the first workgroup)
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q * np.float32(C), k=k)
    got = as_f32(_run(gpu_ctx, qd, kd, vt, H, Tq))
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, LN2).numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 3e-2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


def test_x32_shared_block_halves_with_disjoint_maxima(gpu_ctx, x32):
    """All the weight of a shared-block query sits in ONE key half (even 32-key groups only, or odd only): one wave of the pair
    holds a partial sum that is 2^-100 of the other's; the combine must not produce NaN / inf and must pick the right half."""
    B, H, Tq, Tk = 1, 1, 192, 256
    rng = np.random.default_rng(9)
    q = rng.standard_normal((B, Tq, 128)).astype(np.float32)
    k = rng.standard_normal((B, Tk, 128)).astype(np.float32) * 0.05
    for qi, key in ((130, 5), (131, 40), (170, 100), (171, 70), (10, 33)):
        k[0, key] = q[0, qi] / np.linalg.norm(q[0, qi]) * 12.0   # q.k ~ 12 |q| ~ 135 before the prescale
    qd, kd, vd, vt = _attn_inputs(rng, B, H, Tq, Tk, q=q * np.float32(C) * 8.0, k=k)
    got = as_f32(_run(gpu_ctx, qd, kd, vt, H, Tq))
    ref = _attn_ref(qd.float().cpu(), kd.float().cpu(), vd.float().cpu(), H, None, LN2).numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 3e-2


@pytest.mark.parametrize("Tq,Tk", [(1536, 1536), (1536, 1024), (192, 1536)])
def test_x32_is_repeatable_with_cold_caches(gpu_ctx, x32, Tq, Tk):
    """Two inputs launched alternately, the caches flushed every eighth launch: all outputs of one input bit-identical (a ring slot
    read before its LDS-DMA landed would hold the OTHER input's keys; the static proof is tests/test_asm_wait_coverage.py)."""
    rng = np.random.default_rng(Tq + Tk)
    H = 32
    sets = [_attn_inputs(rng, 1, H, Tq, Tk) for _ in range(2)]
    trash = torch.empty((512 * 2 ** 20,), device="cuda", dtype=torch.bfloat16)
    first = [None, None]
    o = [torch.empty((1, Tq, H * 128), device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    diff = torch.zeros((), device="cuda", dtype=torch.int64)
    for i in range(240):
        qd, kd, vd, vt = sets[i & 1]
        if i % 8 == 0:
            trash.fill_(float(i))
        gpu_ctx.op_attention(qd, kd, vt, None, H, o[i & 1], 0.0)
        if first[i & 1] is None:
            first[i & 1] = o[i & 1].clone()
        else:
            diff += (o[i & 1] != first[i & 1]).sum()
    torch.cuda.synchronize()
    assert int(diff.item()) == 0


def test_x32_refuses_what_it_does_not_take(gpu_ctx, x32, ltx):
    rng = np.random.default_rng(1)
    qd, kd, vd, vt = _attn_inputs(rng, 1, 1, 192, 100)
    o = torch.empty((1, 192, 128), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ltx.LTXError):
        gpu_ctx.op_attention(qd, kd, vt, None, 1, o, 0.0)   # ragged key count
    qd, kd, vd, vt = _attn_inputs(rng, 1, 1, 192, 128)
    with pytest.raises(ltx.LTXError):
        gpu_ctx.op_attention(qd, kd, vt, None, 1, o, 1.0 / math.sqrt(128.0))   # Q not prescaled
