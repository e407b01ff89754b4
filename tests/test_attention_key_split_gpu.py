"""Key-split attention launches (few queries against many keys: config 1's cross-attention is 128 queries x 1024 text keys,
LTXAttention.swift:209 is the call being replaced): the launcher divides the keys over up to 8 workgroups per (query block, head) and a
combine pass weights the per-range softmaxes together. Checked against an f32 softmax over ALL keys - the split must not be visible."""
import importlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
LOG2E = 1.4426950408889634


@pytest.fixture(scope="module")
def ctx():
    c = ltx.Context(0)
    yield c
    c.close()


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16)


def _ref(q, k, v, H, bias, scale):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    qh = q.double().reshape(B, Tq, H, 128).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, Tk, H, 128).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, Tk, H, 128).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias.double()[:, None, None, :]
    return (torch.softmax(s, dim=-1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, D).float().numpy()


def _run(ctx, B, H, Tq, Tk, bias, prescaled, q, k, v):
    D = H * 128
    scale = 1.0 / math.sqrt(128.0)
    qd = _bf16(q * (scale * LOG2E)) if prescaled else _bf16(q)
    kd, vd = _bf16(k), _bf16(v)
    vt = torch.zeros((B, D, (Tk + 63) // 64 * 64), device="cuda", dtype=torch.bfloat16)
    vt[:, :, :Tk] = vd.transpose(1, 2)
    guard = torch.full((B, Tq + 8, D), 7.0, device="cuda", dtype=torch.bfloat16)
    o = guard[:, :Tq] if B == 1 else torch.empty((B, Tq, D), device="cuda", dtype=torch.bfloat16)
    bd = None if bias is None else torch.from_numpy(bias).cuda()
    ctx.op_attention(qd, kd, vt, bd, H, o, -1.0 if prescaled else scale)
    torch.cuda.synchronize()
    qe = qd.float().cpu() / (scale * LOG2E) if prescaled else qd.float().cpu()
    ref = _ref(qe, kd.float().cpu(), vd.float().cpu(), H, None if bias is None else torch.from_numpy(bias), scale)
    got = o.float().cpu().numpy()
    if B == 1:
        assert (guard[:, Tq:].float().cpu().numpy() == 7.0).all(), "rows past Tq were written"
    return got, ref


SHAPES = [(1, 32, 128, 1024), (1, 4, 128, 1024), (1, 2, 100, 1000), (2, 3, 192, 700), (1, 1, 35, 257), (1, 8, 384, 4096), (1, 2, 1, 512)]


@pytest.mark.parametrize("prescaled", [True, False])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("B,H,Tq,Tk", SHAPES)
def test_key_split_matches_the_softmax_over_all_keys(ctx, B, H, Tq, Tk, masked, prescaled):
    assert ctx.attention_key_splits(B, H, Tq, Tk) > 1, "shape list: every entry must take the split path"
    rng = np.random.default_rng(B * 31 + H * 7 + Tq + Tk + 2 * masked + prescaled)
    D = H * 128
    q, k, v = (rng.standard_normal((B, t, D)).astype(np.float32) for t in (Tq, Tk, Tk))
    bias = None
    if masked:
        m = (rng.random((B, Tk)) > 0.3).astype(np.float32)
        m[:, 0] = 1
        bias = ((1 - m) * -10000.0).astype(np.float32)
    got, ref = _run(ctx, B, H, Tq, Tk, bias, prescaled, q, k, v)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2, np.abs(got - ref).max()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2


def test_key_ranges_that_are_masked_out_entirely_carry_no_weight(ctx):
    """The reference's text mask leaves a prefix of valid keys: every range but the first is fully masked (-10000). Their softmaxes are
    uniform over masked keys; the combine weights must send them to exactly nothing."""
    B, H, Tq, Tk = 1, 4, 128, 1024
    assert ctx.attention_key_splits(B, H, Tq, Tk) == 8
    rng = np.random.default_rng(11)
    D = H * 128
    q, k, v = (rng.standard_normal((B, t, D)).astype(np.float32) for t in (Tq, Tk, Tk))
    v[:, 100:] = 1e3  # anything leaking from a masked range would be seen at once
    bias = np.zeros((B, Tk), np.float32)
    bias[:, 100:] = -10000.0
    got, ref = _run(ctx, B, H, Tq, Tk, bias, True, q, k, v)
    assert np.abs(got - ref).max() <= 2e-2, np.abs(got - ref).max()


def test_key_ranges_with_maxima_far_apart(ctx):
    """Score magnitudes grow with the key index: the last range holds every query's maximum and the first ranges' denominators are
    2^-100 and less of it - their weights underflow to zero, they must not turn into NaN or win by rounding."""
    B, H, Tq, Tk = 1, 2, 128, 1024
    rng = np.random.default_rng(5)
    D = H * 128
    q, k, v = (rng.standard_normal((B, t, D)).astype(np.float32) for t in (Tq, Tk, Tk))
    k *= np.linspace(0.2, 6.0, Tk).astype(np.float32)[None, :, None]
    q[:, ::3] *= 0.05  # every third query: flat scores, every range matters
    for prescaled in (True, False):
        got, ref = _run(ctx, B, H, Tq, Tk, None, prescaled, q, k, v)
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= 3e-2, np.abs(got - ref).max()
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-2
