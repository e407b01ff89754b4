"""CPU oracle for the LTX-2 denoise + VAE-decode hot path.  *** TEST INFRASTRUCTURE - NOT THE PRODUCT ***

This file is a numpy restatement of the reference's algorithm (VincentGourbin/ltx-video-swift-mlx, Swift + MLX)
for the functions listed in SURVEY.md section 8(a). Each function cites the reference file:line it follows.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the product path
(`libltxhip.so`) never does and has no CPU fallback.

PARITY UNPINNED: the reference ships no golden vectors, fixtures or numeric tests for this path (its only test
asserts a version string: Tests/LTXVideoTests/LTXVideoTests.swift:9-11) and it cannot be built or run here (Swift +
Apple MLX, macOS only; arithmetic lives in the un-vendored third-party package mlx-swift exact 0.30.6,
Package.swift:21). This oracle is therefore pinned only by
  (1) hand-derived integer / scalar known-answer vectors from the reference source (tests/test_host_logic.py),
  (2) op-level cross-checks against an independent implementation (torch CPU) - tests/golden/make_golden.py,
not by outputs of the reference itself.

dtype model: the reference's effective precision (inferred from MLX promotion rules, not
observable here) is f32 activations x bf16-rounded weights, with bf16 storage only for the patchify projection,
the caption projection and the cross-attention K/V. `bf16_round` marks those rounding points.
"""
import math
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

F32 = np.float32

# Host threads for the position-wise passes of the full-size runs (numpy ufuncs release the GIL; BLAS has its own pool). Values do
# not depend on it: `_pmap` only ever splits an axis along which `fn` is position-wise.
_HOST_THREADS = max(1, min(32, int(os.environ.get("LTX_ORACLE_THREADS", os.cpu_count() or 1))))
_pool = None
# BLAS threads of the DiT's dense products. numpy's bundled OpenBLAS starts 64 threads on a many-core host and is 2.5x SLOWER with them
# than with 16 at these shapes (GPU box, 256 cores: 1536 x 4096 x 4096 f32 at 1.4 TFLOP/s with 64 threads, 3.4 with 16, 5.1 with 32 -
# tools/host_blas_probe.py); the conv path's one large sgemm per tap is fastest at the library default and is left alone.
_DIT_BLAS_THREADS = max(1, min(int(os.environ.get("LTX_ORACLE_BLAS_THREADS", 16)), os.cpu_count() or 1))
_CONV_BLAS_THREADS = max(1, min(int(os.environ.get("LTX_ORACLE_CONV_BLAS_THREADS", 64)), os.cpu_count() or 1))
try:
    from threadpoolctl import ThreadpoolController as _TPC
    _tpc = None

    def _dit_blas():
        global _tpc
        if _tpc is None:
            _tpc = _TPC()
        return _tpc.limit(limits=_DIT_BLAS_THREADS, user_api="blas")

    def _conv_blas():
        """the conv path's one tall sgemm per tap: scipy's OpenBLAS starts a thread per core (256 on the GPU box: 0.7 TFLOP/s on the
        128-channel tap shape) and is fastest with 64 (5.7 TFLOP/s; tools/host_blas_probe.py)"""
        global _tpc
        if _tpc is None:
            _tpc = _TPC()
        return _tpc.limit(limits=_CONV_BLAS_THREADS, user_api="blas")
except Exception:  # noqa: BLE001 - threadpoolctl missing: the library default
    import contextlib

    def _dit_blas():
        return contextlib.nullcontext()

    def _conv_blas():
        return contextlib.nullcontext()
_in_worker = threading.local()  # a slab function that itself calls `_pmap` runs that inner call inline (no nested submission)


_PMAP_MIN = 1 << 22  # elements below which a pass runs inline (tests lower it to exercise the slab path on small inputs)


def _pmap(fn, x, axis, out_dtype=F32, with_slice=False):
    """out = fn(x), computed slab by slab along `axis` on a thread pool. `fn` must map a slab to the same slab of the result
    (same shape), i.e. be position-wise along `axis`; with_slice: fn(slab, slice) also gets the slab's index range (for operands that
    are indexed by the same axis, e.g. rotary tables by token). Small inputs run inline."""
    global _pool
    n = x.shape[axis]
    k = min(_HOST_THREADS, n)
    if k <= 1 or x.size < _PMAP_MIN or getattr(_in_worker, "on", False):
        return np.asarray(fn(x, slice(0, n)) if with_slice else fn(x), dtype=out_dtype)
    if _pool is None:
        _pool = ThreadPoolExecutor(_HOST_THREADS)
    out = np.empty(x.shape, out_dtype)
    cuts = [n * i // k for i in range(k + 1)]

    def run(i):
        sl = [slice(None)] * x.ndim
        sl[axis] = slice(cuts[i], cuts[i + 1])
        _in_worker.on = True
        try:
            out[tuple(sl)] = fn(x[tuple(sl)], sl[axis]) if with_slice else fn(x[tuple(sl)])
        finally:
            _in_worker.on = False

    list(_pool.map(run, range(k)))
    return out


# ---------------------------------------------------------------------------------------------------------------
# bf16 helpers
# ---------------------------------------------------------------------------------------------------------------
def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16).reshape(x.shape)


def bf16_bits_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32).reshape(b.shape)


def bf16_round(x):
    """Round f32 values to the nearest bf16 (ties to even), returned as f32."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# ---------------------------------------------------------------------------------------------------------------
# R1: configuration validation and latent shapes (LTXConfig.swift:310-361, VideoLatentShape.swift:34-48,95-111)
# ---------------------------------------------------------------------------------------------------------------
def validate_generation_config(width, height, num_frames, num_steps, cfg_scale, two_stage=False):
    """Returns None when valid, else the reference's error message (LTXError.invalidConfiguration)."""
    if width % 32 != 0:
        return f"Width must be divisible by 32, got {width}"
    if height % 32 != 0:
        return f"Height must be divisible by 32, got {height}"
    if (num_frames - 1) % 8 != 0:
        return f"Number of frames must be 8n + 1 (e.g., 9, 17, 25, ..., 121), got {num_frames}"
    if not (64 <= width <= 2048):
        return f"Width must be between 64 and 2048, got {width}"
    if not (64 <= height <= 2048):
        return f"Height must be between 64 and 2048, got {height}"
    if not (9 <= num_frames <= 257):
        return f"Number of frames must be between 9 and 257, got {num_frames}"
    if not (1 <= num_steps <= 100):
        return f"Number of steps must be between 1 and 100, got {num_steps}"
    if not (1.0 <= cfg_scale <= 20.0):
        return f"CFG scale must be between 1.0 and 20.0, got {cfg_scale:g}"
    if two_stage and (width % 64 != 0 or height % 64 != 0):  # LTXPipeline.swift:2443
        return f"Two-stage generation requires width and height divisible by 64, got {width}x{height}"
    return None


def latent_shape(width, height, num_frames):
    return (num_frames - 1) // 8 + 1, height // 32, width // 32


# ---------------------------------------------------------------------------------------------------------------
# R3: sigma schedules (LTXScheduler.swift:10-36,74-182), f32 scalar arithmetic without FMA
# ---------------------------------------------------------------------------------------------------------------
DISTILLED_SIGMA_VALUES = [1.0, 0.99375, 0.9875, 0.98125, 0.975, 0.909375, 0.725, 0.421875, 0.0]
STAGE_2_DISTILLED_SIGMA_VALUES = [0.909375, 0.725, 0.421875, 0.0]


def _expf(x):
    """Correctly rounded f32 exp (what glibc/Apple libm expf return for these arguments); numpy's SIMD float32 exp is
    1 ulp off for mu = 2.05 and mu = 0.629, which moves two sigmas by 1-2 ulp."""
    return F32(math.exp(float(x)))


def sigmas(distilled, num_steps, token_count=None, max_shift=2.05, base_shift=0.95, terminal=0.1):
    one = F32(1.0)
    max_shift, base_shift, terminal = F32(max_shift), F32(base_shift), F32(terminal)
    x1, x2 = F32(1024), F32(4096)
    mm = F32(F32(max_shift - base_shift) / F32(x2 - x1))
    b = F32(base_shift - F32(mm * x1))
    if distilled:
        s = [F32(v) for v in DISTILLED_SIGMA_VALUES if v > 0]
        if token_count:
            clamped = min(int(token_count), 4096)
            mu = F32(F32(F32(clamped) * mm) + b)
            exp_mu = _expf(mu)
            s = [v if (v == 0 or v == one) else F32(exp_mu / F32(exp_mu + F32(F32(one / v) - one))) for v in s]
            last_om = F32(one - s[-1])
            if last_om > 0:
                sf = F32(last_om / F32(one - terminal))
                s = [F32(0) if v == 0 else F32(one - F32(F32(one - v) / sf)) for v in s]
        s.append(F32(0.0))
        return np.array(s, dtype=F32)
    tc = min(int(token_count) if token_count else 4096, 4096)
    s = [F32(one - F32(F32(i) / F32(num_steps))) for i in range(num_steps + 1)]
    shift = F32(F32(F32(tc) * mm) + b)
    exp_shift = _expf(shift)
    s = [F32(0) if v == 0 else F32(exp_shift / F32(exp_shift + F32(F32(one / v) - one))) for v in s]
    if num_steps > 0:
        om = [F32(one - v) for v in s]
        sf = F32(om[num_steps - 1] / F32(one - terminal))
        # num_steps == 1: om[0] == 0, so the stretch is 0 / 0 and sigma[0] becomes NaN - exactly what the reference's Float arithmetic
        # yields (LTXScheduler.swift:159-174 has no guard); kept, without numpy's warning.
        with np.errstate(invalid="ignore", divide="ignore"):
            s = [F32(0) if v == 0 else F32(one - F32(om[i] / sf)) for i, v in enumerate(s)]
    return np.array(s, dtype=F32)


# ---------------------------------------------------------------------------------------------------------------
# R4: patchify / unpatchify (LatentUtils.swift:20-54)
# ---------------------------------------------------------------------------------------------------------------
def patchify(latent):
    b, c, f, h, w = latent.shape
    return latent.transpose(0, 2, 3, 4, 1).reshape(b, f * h * w, c)


def unpatchify(x, f, h, w):
    b, t, c = x.shape
    return x.reshape(b, f, h, w, c).transpose(0, 4, 1, 2, 3)


# ---------------------------------------------------------------------------------------------------------------
# R8 + R9: position grid and double-precision split-RoPE tables (LTXRoPE.swift:552-610, 375-527)
# ---------------------------------------------------------------------------------------------------------------
def position_grid(frames, height, width, fps=24.0):
    """(3, T) f32 pixel-space mid coordinates; temporal with causal fix, divided by fps."""
    ts, ss, fps = F32(8), F32(32), F32(fps)
    tcv = []
    for i in range(frames):
        fi = F32(i)
        start = max(F32(F32(fi * ts) + F32(1 - ts)), F32(0))
        end = max(F32(F32(F32(fi + 1) * ts) + F32(1 - ts)), F32(0))
        tcv.append(F32(F32(F32(start + end) / F32(2)) / fps))
    hc = [F32(F32(F32(i) * ss) + F32(ss / F32(2))) for i in range(height)]
    wc = [F32(F32(F32(i) * ss) + F32(ss / F32(2))) for i in range(width)]
    t = np.broadcast_to(np.array(tcv, F32)[:, None, None], (frames, height, width)).reshape(-1)
    h = np.broadcast_to(np.array(hc, F32)[None, :, None], (frames, height, width)).reshape(-1)
    w = np.broadcast_to(np.array(wc, F32)[None, None, :], (frames, height, width)).reshape(-1)
    return np.stack([t, h, w], 0).astype(F32)


_ROPE_CACHE = {}


def rope_tables(frames, height, width, dim=4096, num_heads=32, theta=10000.0, max_pos=(20, 2048, 2048)):
    """cos, sin as [T][dim/2] f32 (== the reference's [B,H,T,64] tensors flattened per token: head h owns columns
    h*64 .. h*64+63). The tables depend on the latent shape only: the last one computed is kept (read-only), so the eight steps of a
    denoise loop do not each spend a second in math.cos / math.sin (the reference caches them per shape too, LTXTransformer.swift:30-31)."""
    key = (frames, height, width, dim, num_heads, float(theta), tuple(max_pos))
    hit = _ROPE_CACHE.get(key)
    if hit is not None:
        return hit
    cos, sin = _rope_tables(frames, height, width, dim, num_heads, theta, max_pos)
    cos.setflags(write=False)
    sin.setflags(write=False)
    _ROPE_CACHE.clear()
    _ROPE_CACHE[key] = (cos, sin)
    return cos, sin


def _rope_tables(frames, height, width, dim, num_heads, theta, max_pos):
    grid = position_grid(frames, height, width).astype(np.float64)  # (3, T)
    n_pos = 3
    n_elem = 2 * n_pos
    n_idx = max(1, dim // n_elem)
    log_start = math.log(1.0) / math.log(theta)
    log_end = math.log(theta) / math.log(theta)
    idx = np.array([math.pow(theta, log_start + (log_end - log_start) * i / (n_idx - 1) if n_idx > 1 else log_start)
                    * (math.pi / 2.0) for i in range(n_idx)], dtype=np.float64)
    frac = grid / np.array(max_pos, dtype=np.float64)[:, None]
    scaled = frac * 2.0 - 1.0  # (3, T)
    freqs = idx[None, :, None] * scaled.T[:, None, :]  # (T, n_idx, 3): index fi*3 + d
    freqs = freqs.reshape(grid.shape[1], n_idx * n_pos)
    cosv = np.array([[math.cos(v) for v in row] for row in freqs], dtype=np.float64)
    sinv = np.array([[math.sin(v) for v in row] for row in freqs], dtype=np.float64)
    pad = max(0, dim // 2 - n_idx * n_pos)
    T = grid.shape[1]
    cos = np.concatenate([np.ones((T, pad), F32), cosv.astype(F32)], axis=1)
    sin = np.concatenate([np.zeros((T, pad), F32), sinv.astype(F32)], axis=1)
    assert cos.shape[1] == dim // 2 and (dim // 2) % num_heads == 0
    return cos, sin


def apply_split_rope(x, cos, sin, num_heads):
    """applySplitRotaryEmb (LTXRoPE.swift:84-149). x [B,T,H*D] f32; cos/sin [T, H*D/2]."""
    b, t, hd = x.shape
    d = hd // num_heads

    def f(xs, ts):  # a slab of tokens and its rows of the tables
        n = xs.shape[1]
        xh = xs.reshape(b, n, num_heads, 2, d // 2).astype(F32, copy=False)
        c = cos[ts].reshape(1, n, num_heads, d // 2)
        s_ = sin[ts].reshape(1, n, num_heads, d // 2)
        first, second = xh[:, :, :, 0], xh[:, :, :, 1]
        return np.stack([first * c - second * s_, second * c + first * s_], axis=3).reshape(b, n, hd)

    return _pmap(f, x, 1, with_slice=True)


# ---------------------------------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------------------------------
def _tokens(fn, x):
    """fn over the token axis (second to last) of a [.., T, D] array on the thread pool; fn is row-wise (reductions over D only)."""
    return _pmap(fn, x, x.ndim - 2) if x.ndim >= 2 else np.asarray(fn(x), dtype=F32)


def linear(x, w, b=None):
    """Linear with weights [out,in] (MLXNN.Linear): f32 accumulate."""
    with _dit_blas():
        y = x.astype(F32, copy=False) @ w.astype(F32, copy=False).T
    if b is not None:
        y += b.astype(F32, copy=False)
    return y


def rms_norm(x, weight=None, eps=1e-6):
    """MLXFast.rmsNorm (LTXAttention.swift:12-33): x * rsqrt(mean(x^2) + eps) * w over the last axis."""
    wf = None if weight is None else weight.astype(F32)

    def f(v):
        v = v.astype(F32, copy=False)
        ms = np.mean(v.astype(np.float64) ** 2, axis=-1, keepdims=True)
        y = v * (1.0 / np.sqrt(ms + eps)).astype(F32)
        if wf is not None:
            y = y * wf
        return y.astype(F32, copy=False)

    return _tokens(f, x)


def layer_norm(x, eps=1e-6):
    """LayerNorm(affine: false) (LTXTransformer.swift:96)."""
    def f(v):
        x64 = v.astype(np.float64)
        mu = x64.mean(-1, keepdims=True)
        var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
        return ((x64 - mu) / np.sqrt(var + eps)).astype(F32)

    return _tokens(f, x)


def gelu_tanh(x):
    """MLXNN.geluApproximate (LTXFeedForward.swift:13-17)."""
    def f(v):
        v = v.astype(F32, copy=False)
        return (F32(0.5) * v * (F32(1.0) + np.tanh(F32(0.7978845608028654) * (v + F32(0.044715) * v * v * v)))).astype(F32, copy=False)

    return _tokens(f, x)


def silu(x):
    x = x.astype(F32, copy=False)
    return (x / (F32(1.0) + np.exp(-x))).astype(F32, copy=False)


def sdpa(q, k, v, num_heads, scale, bias=None):
    """MLXFast.scaledDotProductAttention on [B,T,H*D] inputs (LTXAttention.swift:192-214). bias [B,S] additive.
    Per batch element: the scores of a group of heads as one batched product, the softmax of the group on the thread pool (it is
    head-wise), then P.V batched - the same products and the same element-wise steps as a loop over single heads."""
    b, tq, hd = q.shape
    tk = k.shape[1]
    d = hd // num_heads
    qh = q.reshape(b, tq, num_heads, d).transpose(0, 2, 1, 3).astype(F32)
    kh = k.reshape(b, tk, num_heads, d).transpose(0, 2, 1, 3).astype(F32)
    vh = v.reshape(b, tk, num_heads, d).transpose(0, 2, 1, 3).astype(F32)
    out = np.empty((b, num_heads, tq, d), F32)
    grp = max(1, min(num_heads, (1 << 28) // max(1, tq * tk)))  # heads per group: at most 1 GiB of f32 scores at a time
    for bi in range(b):
        bias_row = None if bias is None else bias[bi][None, None, :].astype(F32)

        def softmax(sc):
            sc = sc * F32(scale)
            if bias_row is not None:
                sc = sc + bias_row
            sc = sc - sc.max(-1, keepdims=True)
            p = np.exp(sc)
            return p / p.sum(-1, keepdims=True)

        for h0 in range(0, num_heads, grp):
            h1 = min(num_heads, h0 + grp)
            with _dit_blas():
                sc = np.matmul(qh[bi, h0:h1], kh[bi, h0:h1].transpose(0, 2, 1))
            p = _pmap(softmax, sc, 0)
            with _dit_blas():
                np.matmul(p, vh[bi, h0:h1], out=out[bi, h0:h1])
    return out.transpose(0, 2, 1, 3).reshape(b, tq, hd)


def timestep_embedding(t, dim=256):
    """getTimestepEmbedding (LTXTimestepEmbedding.swift:17-54): [cos, sin] order, f32."""
    half = dim // 2
    freqs = np.exp(-F32(math.log(10000.0)) * (np.arange(half, dtype=F32) / F32(half))).astype(F32)
    args = np.asarray(t, F32).reshape(-1, 1) * freqs[None, :]
    return np.concatenate([np.cos(args), np.sin(args)], -1).astype(F32)


# ---------------------------------------------------------------------------------------------------------------
# R5-R14: the DiT (LTXTransformer.swift:235-486). `w` maps module keys to f32 arrays holding
# bf16-rounded values.
# ---------------------------------------------------------------------------------------------------------------
class DiTConfig:
    def __init__(self, num_layers=48, num_heads=32, head_dim=128, in_channels=128, out_channels=128,
                 caption_channels=3840, rope_theta=10000.0, max_pos=(20, 2048, 2048), timestep_mult=1000.0,
                 norm_eps=1e-6):
        self.num_layers, self.num_heads, self.head_dim = num_layers, num_heads, head_dim
        self.in_channels, self.out_channels, self.caption_channels = in_channels, out_channels, caption_channels
        self.rope_theta, self.max_pos, self.timestep_mult, self.norm_eps = rope_theta, max_pos, timestep_mult, norm_eps

    @property
    def dim(self):
        return self.num_heads * self.head_dim


def dit_param_shapes(cfg):
    """Module keys and shapes expected after key mapping (LTXTransformer.swift:34-101)."""
    D = cfg.dim
    s = {
        "patchify_proj.weight": (D, cfg.in_channels), "patchify_proj.bias": (D,),
        "adaln_single.emb.linear_1.weight": (D, 256), "adaln_single.emb.linear_1.bias": (D,),
        "adaln_single.emb.linear_2.weight": (D, D), "adaln_single.emb.linear_2.bias": (D,),
        "adaln_single.linear.weight": (6 * D, D), "adaln_single.linear.bias": (6 * D,),
        "caption_projection.linear_1.weight": (D, cfg.caption_channels), "caption_projection.linear_1.bias": (D,),
        "caption_projection.linear_2.weight": (D, D), "caption_projection.linear_2.bias": (D,),
        "scale_shift_table": (2, D),
        "proj_out.weight": (cfg.out_channels, D), "proj_out.bias": (cfg.out_channels,),
    }
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        for a in ("attn1", "attn2"):
            for l in ("to_q", "to_k", "to_v", "to_out"):
                s[p + f"{a}.{l}.weight"] = (D, D)
                s[p + f"{a}.{l}.bias"] = (D,)
            s[p + f"{a}.q_norm.weight"] = (D,)
            s[p + f"{a}.k_norm.weight"] = (D,)
        s[p + "ff.project_in.proj.weight"] = (4 * D, D)
        s[p + "ff.project_in.proj.bias"] = (4 * D,)
        s[p + "ff.project_out.weight"] = (D, 4 * D)
        s[p + "ff.project_out.bias"] = (D,)
        s[p + "scale_shift_table"] = (6, D)
    return s


def attention(w, prefix, x, cfg, context=None, bias=None, rope=None, kv_bf16=False, kv_cache=None):
    """LTXAttention.callAsFunction (LTXAttention.swift:160-218). kv_cache (a dict owned by the caller, cross-attention only): the keys
    and values of a text context do not depend on the step, so a denoise loop computes them once per block - the same arrays every step
    (the reference recomputes them; the result is the same, bit for bit)."""
    H = cfg.num_heads
    ctx = x if context is None else context
    q = linear(x, w[prefix + "to_q.weight"], w[prefix + "to_q.bias"])
    q = rms_norm(q, w[prefix + "q_norm.weight"], cfg.norm_eps)
    hit = kv_cache.get(prefix) if (kv_cache is not None and context is not None) else None
    if hit is not None:
        k, v = hit
    else:
        k = linear(ctx, w[prefix + "to_k.weight"], w[prefix + "to_k.bias"])
        v = linear(ctx, w[prefix + "to_v.weight"], w[prefix + "to_v.bias"])
        if kv_bf16:  # cross-attention K/V come out of bf16 x bf16 Linears in the reference
            k, v = bf16_round(k), bf16_round(v)
        k = rms_norm(k, w[prefix + "k_norm.weight"], cfg.norm_eps)
        if kv_bf16:
            k = bf16_round(k)
        if kv_cache is not None and context is not None:
            kv_cache[prefix] = (k, v)
    if rope is not None:
        q = apply_split_rope(q, rope[0], rope[1], H)
        k = apply_split_rope(k, rope[0], rope[1], H)
    o = sdpa(q, k, v, H, 1.0 / math.sqrt(cfg.head_dim), bias)
    return linear(o, w[prefix + "to_out.weight"], w[prefix + "to_out.bias"])


def feed_forward(w, prefix, x):
    """LTXFeedForward (LTXFeedForward.swift:35-52)."""
    h = gelu_tanh(linear(x, w[prefix + "project_in.proj.weight"], w[prefix + "project_in.proj.bias"]))
    return linear(h, w[prefix + "project_out.weight"], w[prefix + "project_out.bias"])


def transformer_block(w, i, x, ctx, temb, cfg, rope, bias, cross_scale=1.0, skip_attn=False, skip_ff=False,
                      first_norm_bf16=False, kv_cache=None):
    """BasicTransformerBlock.callAsFunction (LTXTransformerBlock.swift:187-232). temb [B,1,6,D] or [B,T,6,D]."""
    p = f"transformer_blocks.{i}."
    ada = w[p + "scale_shift_table"][None, None].astype(F32) + temb  # [B,1,6,D]
    shift_msa, scale_msa, gate_msa = ada[:, :, 0], ada[:, :, 1], ada[:, :, 2]
    shift_mlp, scale_mlp, gate_mlp = ada[:, :, 3], ada[:, :, 4], ada[:, :, 5]
    if not skip_attn:
        n = rms_norm(x, None, cfg.norm_eps)
        if first_norm_bf16:  # block 0 normalises a bf16 stream: MLXFast.rmsNorm returns bf16 there
            n = bf16_round(n)
        n = n * (F32(1) + scale_msa) + shift_msa
        x = x + attention(w, p + "attn1.", n, cfg, rope=rope) * gate_msa
    cross = attention(w, p + "attn2.", x, cfg, context=ctx, bias=bias, kv_bf16=True, kv_cache=kv_cache)
    if cross_scale != 1.0:
        cross = cross * F32(cross_scale)
    x = x + cross
    if not skip_ff:
        n = rms_norm(x, None, cfg.norm_eps) * (F32(1) + scale_mlp) + shift_mlp
        x = x + feed_forward(w, p + "ff.", n) * gate_mlp
    return x.astype(F32)


def caption_projection(w, context):
    """PixArtAlphaTextProjection on a bf16 context (LTXTimestepEmbedding.swift:131-152): bf16 at every stage."""
    h = bf16_round(linear(context, w["caption_projection.linear_1.weight"], w["caption_projection.linear_1.bias"]))
    h = bf16_round(gelu_tanh(h))
    return bf16_round(linear(h, w["caption_projection.linear_2.weight"], w["caption_projection.linear_2.bias"]))


def mask_to_bias(mask):
    """prepareAttentionMask (LTXTransformer.swift:141-156): (1-m)*-10000."""
    if mask is None:
        return None
    return ((F32(1) - mask.astype(F32)) * F32(-10000.0)).astype(F32)


def dit_forward(w, cfg, latent, context, timesteps, mask, F, H, W, cross_scale=None, stg_blocks=(), skip_ff_blocks=(),
                num_layers=None, text_cache=None):
    """LTXTransformer.callAsFunction (LTXTransformer.swift:235-486).
    latent [B,T,C] (bf16-representable f32), context [B,S,Cc] (bf16-representable), timesteps [B] sigma, or [B,T]
    per-token sigmas (image-to-video: prepareTimestep flattens them and reshapes to [B,T,6,D], LTXTransformer.swift:105-124)."""
    D = cfg.dim
    B = latent.shape[0]
    x = bf16_round(linear(latent, w["patchify_proj.weight"], w["patchify_proj.bias"]))
    ts_arr = np.asarray(timesteps, F32)
    n_tok = 1 if ts_arr.ndim == 1 else ts_arr.shape[1]
    t = ts_arr.reshape(-1) * F32(cfg.timestep_mult)
    e = timestep_embedding(t, 256)
    e = linear(e, w["adaln_single.emb.linear_1.weight"], w["adaln_single.emb.linear_1.bias"])
    emb_ts = linear(silu(e), w["adaln_single.emb.linear_2.weight"], w["adaln_single.emb.linear_2.bias"])  # [B*n,D]
    ada = linear(silu(emb_ts), w["adaln_single.linear.weight"], w["adaln_single.linear.bias"])  # [B*n,6D]
    temb = ada.reshape(B, n_tok, 6, D)
    # text_cache (a dict owned by the caller, one per text context): the caption projection and the cross-attention K / V of every block
    # are computed on the first forward and reused by the later steps of a denoise loop
    if text_cache is not None and "ctx" in text_cache:
        ctx = text_cache["ctx"]
    else:
        ctx = caption_projection(w, context).reshape(B, -1, D)
        if text_cache is not None:
            text_cache["ctx"] = ctx
    bias = mask_to_bias(mask)
    rope = rope_tables(F, H, W, D, cfg.num_heads, cfg.rope_theta, cfg.max_pos)
    L = cfg.num_layers if num_layers is None else num_layers
    for i in range(L):
        cs = 1.0 if cross_scale is None else cross_scale
        x = transformer_block(w, i, x, ctx, temb, cfg, rope, bias, cs, skip_attn=(i in stg_blocks),
                              skip_ff=(i in skip_ff_blocks), first_norm_bf16=(i == 0), kv_cache=text_cache)
    # processOutput (LTXTransformer.swift:208-224)
    ss = w["scale_shift_table"][None, None].astype(F32) + emb_ts.reshape(B, n_tok, 1, D)
    shift, scale = ss[:, :, 0], ss[:, :, 1]
    out = layer_norm(x, cfg.norm_eps) * (F32(1) + scale) + shift
    return linear(out, w["proj_out.weight"], w["proj_out.bias"])


# ---------------------------------------------------------------------------------------------------------------
# R15-R17: guidance and the Euler step
# ---------------------------------------------------------------------------------------------------------------
def apply_cfg(uncond, cond, scale):
    """LatentUtils.swift:131-141."""
    return (cond + F32(scale - 1.0) * (cond - uncond)).astype(F32)


def guidance_rescale(cfg_out, cond_out, phi):
    """LatentUtils.swift:164-183 (population variance over C,F,H,W; eps inside the sqrt)."""
    if phi <= 0:
        return cfg_out
    ax = tuple(range(1, cfg_out.ndim))
    cfg_std = np.sqrt(cfg_out.astype(np.float64).var(axis=ax, keepdims=True) + 1e-8)
    cond_std = np.sqrt(cond_out.astype(np.float64).var(axis=ax, keepdims=True) + 1e-8)
    rescaled = cfg_out * (cond_std / cfg_std).astype(F32)
    return (F32(phi) * rescaled + F32(1.0 - phi) * cfg_out).astype(F32)


def adain_filter_latent(latent, reference, factor=1.0):
    """LatentUtils.swift:201-227."""
    if factor <= 0:
        return latent
    ax = (2, 3, 4)
    lm, ls = latent.mean(ax, keepdims=True), np.sqrt(latent.astype(np.float64).var(ax, keepdims=True)).astype(F32)
    rm, rs = reference.mean(ax, keepdims=True), np.sqrt(reference.astype(np.float64).var(ax, keepdims=True)).astype(F32)
    res = (latent - lm) / (ls + F32(1e-8)) * rs + rm
    if factor >= 1.0:
        return res.astype(F32)
    return (F32(factor) * res + F32(1 - factor) * latent).astype(F32)


def euler_step(latent, velocity, sigma, sigma_next):
    """LTXScheduler.step (LTXScheduler.swift:305-327), latent f32."""
    s = F32(sigma)
    den = (latent.astype(F32) - s * velocity.astype(F32)).astype(F32)
    if sigma_next > 0:
        return (den + F32(sigma_next) * (latent.astype(F32) - den) / s).astype(F32)
    return den


def denoise(w, cfg, latent, sigmas_, context, mask, F, H, W, cfg_scale=1.0, rescale=0.0, stg_scale=0.0,
            stg_blocks=(29,), ge_gamma=0.0, neg_context=None, neg_mask=None, num_layers=None, cond_latent=None,
            image_cond_noise_scale=0.0, cond_noise=None, step_stats=None, velocity_tokens=None):
    """generateVideo's loop (LTXPipeline.swift:800-956), T2V. latent [1,C,F,H,W] f32 already scaled by sigmas[0].
    Image-to-video (denoise(...) :2191-2401 with conditioningMask / conditionedLatent): cond_latent [1,C,1,H,W] is the encoded
    image; frame 0 is that latent (:2092-2094), optionally re-noised per step with cond_noise[step] * scale * sigma^2
    (:2225-2229), its tokens carry timestep 0 (:2237-2252) and the Euler step skips it (:2344-2357).
    velocity_tokens (test hook): a list that receives (step, token input [1,T,C], raw transformer output [1,T,C]) of every forward, so that
    a test can hold ONE forward of the loop against the HIP forward on the same input without a second oracle run."""
    prev_v = None
    text_caches = {}  # per text context (positive / negative): caption projection + cross-attention K / V, computed at the first step
    i2v = cond_latent is not None
    if i2v:
        latent = latent.copy()
        latent[:, :, 0:1] = cond_latent
    for step in range(len(sigmas_) - 1):
        sg, sn = float(sigmas_[step]), float(sigmas_[step + 1])
        if i2v and image_cond_noise_scale > 0 and sg > 0 and cond_noise is not None:
            latent[:, :, 0:1] = cond_latent + F32(image_cond_noise_scale) * cond_noise[step].astype(F32) * F32(sg * sg)
        tok = bf16_round(patchify(latent))
        if i2v:
            cm = np.zeros((1, F * H * W), F32)
            cm[:, :H * W] = 1
            ts = (F32(sg) * (F32(1) - cm)).astype(F32)  # (1, T) per-token
        else:
            ts = np.array([sg], F32)

        def fwd(c, m, **kw):
            v = dit_forward(w, cfg, tok, c, ts, m, F, H, W, num_layers=num_layers, text_cache=text_caches.setdefault(id(c), {}), **kw)
            if velocity_tokens is not None:
                velocity_tokens.append((step, tok, v))
            return unpatchify(v, F, H, W).astype(F32)

        if cfg_scale > 1.0:
            v_pos, v_neg = fwd(context, mask), fwd(neg_context, neg_mask)
            v = apply_cfg(v_neg, v_pos, cfg_scale)
            if rescale > 0:
                v = guidance_rescale(v, v_pos, rescale)
        else:
            v = fwd(context, mask)
        if stg_scale > 0:
            vp = fwd(context, mask, stg_blocks=tuple(stg_blocks))
            v = v + F32(stg_scale) * (v - vp)
        if ge_gamma > 0 and prev_v is not None:
            v = F32(ge_gamma) * (v - prev_v) + prev_v
        prev_v = v
        if i2v:
            latent = np.concatenate([latent[:, :, 0:1], euler_step(latent[:, :, 1:], v[:, :, 1:], sg, sn)], axis=2)
        else:
            latent = euler_step(latent, v, sg, sn)
        if step_stats is not None:  # the reference's --profile line (LTXPipeline.swift:945-951): mean and sqrt(population variance)
            step_stats.append((float(v.mean(dtype=np.float64)), float(v.std(dtype=np.float64)), float(latent.mean(dtype=np.float64)),
                               float(latent.std(dtype=np.float64))))
    return latent


# ---------------------------------------------------------------------------------------------------------------
# R18-R19: VAE decoder (VideoDecoder.swift, VideoConvolution.swift:202-348). Tensors are [B,C,F,H,W] f32.
# ---------------------------------------------------------------------------------------------------------------
def _pad_reflect_hw_replicate_t(x, causal):
    """Conv3dFull's padding (VideoConvolution.swift:268-305) on a [B,C,T,H,W] tensor: reflect 1 in H and W, then replicate in T
    (1 + 1, or 2 + 0 when causal)."""
    h, wd = x.shape[3], x.shape[4]
    xp = np.concatenate([x[:, :, :, 1:2], x, x[:, :, :, h - 2:h - 1]], axis=3)
    xp = np.concatenate([xp[:, :, :, :, 1:2], xp, xp[:, :, :, :, wd - 2:wd - 1]], axis=4)
    if causal:
        return np.concatenate([xp[:, :, :1]] * 2 + [xp], axis=2)
    return np.concatenate([xp[:, :, :1], xp, xp[:, :, -1:]], axis=2)


def conv3d_full_einsum(x, weight, bias, causal=False):
    """Conv3dFull, first restatement (27 strided patch copies + einsum; ~10 GFLOP/s): kept as the cross-check of `conv3d_full`
    (tests/test_oracle_vs_torch.py asserts the two agree on the golden fixture)."""
    b, c, t, h, wd = x.shape
    o = weight.shape[0]
    xp = _pad_reflect_hw_replicate_t(x, causal)
    out = np.zeros((b, o, t, h, wd), F32)
    wf = weight.astype(F32)
    for kt in range(3):
        for kh in range(3):
            for kw in range(3):
                patch = xp[:, :, kt:kt + t, kh:kh + h, kw:kw + wd].reshape(b, c, -1).astype(F32)
                out += np.einsum("oc,bcn->bon", wf[:, :, kt, kh, kw], patch, optimize=True).reshape(b, o, t, h, wd)
    if bias is not None:
        out += bias.astype(F32).reshape(1, -1, 1, 1, 1)
    return out


def _conv_taps_blas(xp, wt, bias):
    """The tap loop shared by the BLAS-speed conv forms. xp: the PADDED input, channels last, [T + kT - 1][H + 2][W + 2][C] f32 contiguous;
    wt: [kT][3][3][C][O] f32 contiguous. On the padded grid flattened to rows [(T+kT-1)(H+2)(W+2)][C] a tap is a constant row offset, so
    each tap is one zero-copy `rows[off:off+n] @ W_tap` sgemm accumulated in place (beta = 1, taps in (kt, kh, kw) order); rows that fall
    on padding positions are computed and dropped. Returns [O][T][H][W]."""
    from scipy.linalg.blas import sgemm
    global _pool
    kt_n = wt.shape[0]
    tp, hp, wp, c = xp.shape
    t, h, wd, o = tp - kt_n + 1, hp - 2, wp - 2, wt.shape[4]
    n = (t - 1) * hp * wp + (h - 1) * wp + wd                                    # last output row + 1 on the padded grid
    rows = xp.reshape(-1, c)
    acc = np.empty((t * hp * wp, o), F32)
    acc[:] = 0.0 if bias is None else bias.astype(F32)[None, :]
    with _conv_blas():
        for kt in range(kt_n):
            for kh in range(3):
                for kw in range(3):
                    off = (kt * hp + kh) * wp + kw
                    # acc[:n] += rows[off:off+n] @ wt[kt,kh,kw], as the column-major product acc^T += W^T @ rows^T (all three
                    # operands are Fortran-contiguous views, so the BLAS call works in place)
                    r = sgemm(1.0, wt[kt, kh, kw].T, rows[off:off + n].T, beta=1.0, c=acc[:n].T, overwrite_c=1)
                    assert np.shares_memory(r, acc)
    a4 = acc.reshape(t, hp, wp, o)
    out = np.empty((o, t, h, wd), F32)
    if _pool is None:
        _pool = ThreadPoolExecutor(_HOST_THREADS)

    def take(f):
        out[:, f] = a4[f, :h, :wd].transpose(2, 0, 1)

    list(_pool.map(take, range(t)))
    return out


def _taps_first(weight):
    """(O, I, kT, 3, 3) -> [kT][3][3][I][O] f32 contiguous (the tap loop's W_tap operands), one tap per pool task."""
    global _pool
    o, i, kt_n = weight.shape[0], weight.shape[1], weight.shape[2]
    wt = np.empty((kt_n, 3, 3, i, o), F32)
    if _pool is None:
        _pool = ThreadPoolExecutor(_HOST_THREADS)

    def one(j):
        kt, r = divmod(j, 9)
        kh, kw = divmod(r, 3)
        wt[kt, kh, kw] = weight[:, :, kt, kh, kw].T

    list(_pool.map(one, range(kt_n * 9)))
    return wt


def _pad_channels_last(x1, t_src, mode):
    """[C][T][H][W] -> padded channels-last [len(t_src)][H+2][W+2][C] f32: frame i of the result is source frame t_src[i] (None = a zero
    frame); `mode` pads H and W by one: "reflect" or "zero". Frames are filled on the thread pool."""
    global _pool
    c, t, h, wd = x1.shape
    xp = np.zeros((len(t_src), h + 2, wd + 2, c), F32) if mode == "zero" else np.empty((len(t_src), h + 2, wd + 2, c), F32)
    if _pool is None:
        _pool = ThreadPoolExecutor(_HOST_THREADS)

    def fill(pf):
        src = t_src[pf]
        if src is None:
            return
        xp[pf, 1:h + 1, 1:wd + 1] = x1[:, src].transpose(1, 2, 0)
        if mode == "reflect":
            xp[pf, 0, 1:wd + 1] = xp[pf, 2, 1:wd + 1]
            xp[pf, h + 1, 1:wd + 1] = xp[pf, h - 1, 1:wd + 1]
            xp[pf, :, 0] = xp[pf, :, 2]
            xp[pf, :, wd + 1] = xp[pf, :, wd - 1]

    list(_pool.map(fill, range(len(t_src))))
    return xp


def conv3d_full(x, weight, bias, causal=False):
    """Conv3dFull (VideoConvolution.swift:202-348): reflect pad H/W by 1, replicate pad T (1+1, or 2+0 causal), 27 taps, bias,
    f32 accumulation. BLAS-speed form (`_conv_taps_blas`); same tap order (kt, kh, kw) as `conv3d_full_einsum`."""
    b, c, t, h, wd = x.shape
    wt = _taps_first(weight)
    t_src = [min(max(pf - (2 if causal else 1), 0), t - 1) for pf in range(t + 2)]
    out = np.empty((b, weight.shape[0], t, h, wd), F32)
    for bi in range(b):
        out[bi] = _conv_taps_blas(_pad_channels_last(x[bi], t_src, "reflect"), wt, bias)
    return out


def pixel_norm(x, eps=1e-8):
    """vaePixelNorm (VideoDecoder.swift:29-32)."""
    def f(v):
        ms = np.mean(v.astype(np.float64) ** 2, axis=1, keepdims=True)
        return (v / np.sqrt(ms + eps)).astype(F32)
    return _pmap(f, x, x.ndim - 2) if x.ndim >= 4 else f(x)


def vae_res_block(w, p, x, time_emb=None):
    """VAEResBlock3d (VideoDecoder.swift:75-131)."""
    sst = w[p + "scale_shift_table"].astype(F32)  # [4,C]
    if time_emb is not None:
        sst = sst[None] + time_emb.reshape(x.shape[0], 4, -1)
    else:
        sst = sst[None]
    r = lambda v: v.reshape(v.shape[0], -1, 1, 1, 1)
    shift1, scale1, shift2, scale2 = r(sst[:, 0]), r(sst[:, 1] + 1), r(sst[:, 2]), r(sst[:, 3] + 1)
    h = _pmap(lambda v: silu(pixel_norm(v) * scale1 + shift1), x, 3)   # position-wise: slabs along H
    h = conv3d_full(h, w[p + "conv1.conv.weight"], w[p + "conv1.conv.bias"])
    h = _pmap(lambda v: silu(pixel_norm(v) * scale2 + shift2), h, 3)
    h = conv3d_full(h, w[p + "conv2.conv.weight"], w[p + "conv2.conv.bias"])
    return (h + x).astype(F32)


def depth_to_space(x, c_out):
    """VAEDepthToSpaceUpsample3d.depthToSpace (VideoDecoder.swift:201-213), factor (2,2,2)."""
    b, _, t, h, w = x.shape
    o = x.reshape(b, c_out, 2, 2, 2, t, h, w).transpose(0, 1, 5, 2, 6, 3, 7, 4)
    return o.reshape(b, c_out, t * 2, h * 2, w * 2)


def vae_upsample(w, p, x):
    """VAEDepthToSpaceUpsample3d.callAsFunction (VideoDecoder.swift:215-251)."""
    c_in = x.shape[1]
    res = depth_to_space(x, c_in // 8)[:, :, 1:]
    res = np.concatenate([res] * 4, axis=1)
    h = conv3d_full(x, w[p + "conv.conv.weight"], w[p + "conv.conv.bias"])
    h = depth_to_space(h, c_in // 2)[:, :, 1:]
    return (h + res).astype(F32)


def vae_unpatchify(x, p=4):
    """unpatchify (VideoDecoder.swift:257-275): channel = (c*4+a)*4+b, a -> W offset, b -> H offset."""
    b, cp, t, h, w = x.shape
    c = cp // (p * p)
    o = x.reshape(b, c, 1, p, p, t, h, w).transpose(0, 1, 5, 2, 6, 4, 7, 3)
    return o.reshape(b, c, t, h * p, w * p)


VAE_CHANNELS = (1024, 512, 256, 128)


def vae_param_shapes(channels=VAE_CHANNELS, latent_channels=128, timestep_conditioning=False):
    """Module keys after mapVAEWeights (VideoDecoder.swift:302-356)."""
    s = {"conv_in.conv.weight": (channels[0], latent_channels, 3, 3, 3), "conv_in.conv.bias": (channels[0],),
         "conv_out.conv.weight": (48, channels[3], 3, 3, 3), "conv_out.conv.bias": (48,),
         "last_scale_shift_table": (2, channels[3]), "mean_of_means": (latent_channels,), "std_of_means": (latent_channels,)}
    for gi, c in enumerate(channels):
        g = f"up_blocks_{2 * gi}."
        for r in range(5):
            for cv in ("conv1", "conv2"):
                s[g + f"res_blocks.{r}.{cv}.conv.weight"] = (c, c, 3, 3, 3)
                s[g + f"res_blocks.{r}.{cv}.conv.bias"] = (c,)
            s[g + f"res_blocks.{r}.scale_shift_table"] = (4, c)
        if gi < 3:
            u = f"up_blocks_{2 * gi + 1}."
            s[u + "conv.conv.weight"] = (4 * c, c, 3, 3, 3)
            s[u + "conv.conv.bias"] = (4 * c,)
        if timestep_conditioning:
            t = g + "time_embedder.timestep_embedder."
            s[t + "linear_1.weight"], s[t + "linear_1.bias"] = (256, 256), (256,)
            s[t + "linear_2.weight"], s[t + "linear_2.bias"] = (4 * c, 256), (4 * c,)
    if timestep_conditioning:
        t = "last_time_embedder.timestep_embedder."
        s[t + "linear_1.weight"], s[t + "linear_1.bias"] = (256, 256), (256,)
        s[t + "linear_2.weight"], s[t + "linear_2.bias"] = (2 * channels[3], 256), (2 * channels[3],)
        s["timestep_scale_multiplier"] = ()
    return s


def vae_time_embed(w, prefix, t_scaled):
    """VAETimestepEmbedder on getTimestepEmbedding (VideoDecoder.swift:13-52)."""
    e = timestep_embedding(np.asarray(t_scaled, F32).reshape(-1), 256)
    h = silu(linear(e, w[prefix + "timestep_embedder.linear_1.weight"], w[prefix + "timestep_embedder.linear_1.bias"]))
    return linear(h, w[prefix + "timestep_embedder.linear_2.weight"], w[prefix + "timestep_embedder.linear_2.bias"])


def vae_decode_raw(w, latent, channels=VAE_CHANNELS, timestep=None, noise=None):
    """VideoDecoder.callAsFunction (VideoDecoder.swift:358-449) -> [B,3,F,H,W]. With `timestep` the explicit `noise`
    replaces the reference's MLXRandom.normal draw (:369)."""
    x = latent.astype(F32)
    scaled = None
    if timestep is not None:
        x = (noise.astype(F32) * F32(0.025) + (F32(1.0) - F32(0.025)) * x).astype(F32)
        scaled = np.full((x.shape[0],), timestep, F32) * F32(w["timestep_scale_multiplier"])
    x = (x * w["std_of_means"].astype(F32).reshape(1, -1, 1, 1, 1)
         + w["mean_of_means"].astype(F32).reshape(1, -1, 1, 1, 1)).astype(F32)
    x = conv3d_full(x, w["conv_in.conv.weight"], w["conv_in.conv.bias"])
    for gi in range(4):
        te = vae_time_embed(w, f"up_blocks_{2 * gi}.time_embedder.", scaled) if scaled is not None else None
        for r in range(5):
            x = vae_res_block(w, f"up_blocks_{2 * gi}.res_blocks.{r}.", x, te)
        if gi < 3:
            x = vae_upsample(w, f"up_blocks_{2 * gi + 1}.", x)
    lsst = w["last_scale_shift_table"].astype(F32)[None]
    if scaled is not None:
        lsst = lsst + vae_time_embed(w, "last_time_embedder.", scaled).reshape(x.shape[0], 2, -1)
    sc, sh = (lsst[:, 1] + 1).reshape(lsst.shape[0], -1, 1, 1, 1), lsst[:, 0].reshape(lsst.shape[0], -1, 1, 1, 1)
    x = _pmap(lambda v: silu(pixel_norm(v) * sc + sh), x, 3)
    x = conv3d_full(x, w["conv_out.conv.weight"], w["conv_out.conv.bias"])
    return vae_unpatchify(x, 4)


def vae_tile_plan(latent_frames, tile, overlap):
    """decodeWithTemporalTiling chunk walk + blended frame count (VideoDecoder.swift:517-592)."""
    if not (tile > 0 and latent_frames > tile):
        return [(0, latent_frames)], 8 * (latent_frames - 1) + 1
    stride = tile - overlap
    tiles, start = [], 0
    while start < latent_frames:
        end = min(start + tile, latent_frames)
        tiles.append((start, end))
        if end >= latent_frames:
            break
        start += stride
    po = 8 * overlap
    total = 8 * (tiles[0][1] - tiles[0][0] - 1) + 1
    for s, e in tiles[1:]:
        nxt = 8 * (e - s - 1) + 1
        total = total + nxt - po if (0 < po < total and po < nxt) else total + nxt
    return tiles, total


def decode_video(w, latent, tile=0, overlap=1, channels=VAE_CHANNELS, timestep=None, noise=None, return_raw=False):
    """decodeVideo (VideoDecoder.swift:466-602) -> (F,H,W,3) f32 in [0,1]."""
    tiles, _ = vae_tile_plan(latent.shape[2], tile, overlap)
    chunks = [vae_decode_raw(w, latent[:, :, s:e], channels, timestep, None if noise is None else noise[:, :, s:e])
              for s, e in tiles]
    result = chunks[0]
    po = 8 * overlap
    for nxt in chunks[1:]:
        rf, nf = result.shape[2], nxt.shape[2]
        if 0 < po < rf and po < nf:
            wts = (np.arange(po, dtype=F32) / F32(po)).reshape(1, 1, po, 1, 1)
            blended = result[:, :, rf - po:] * (1 - wts) + nxt[:, :, :po] * wts
            result = np.concatenate([result[:, :, :rf - po], blended, nxt[:, :, po:]], axis=2)
        else:
            result = np.concatenate([result, nxt], axis=2)
    if return_raw:
        return result[0].transpose(1, 2, 3, 0).astype(F32)
    frames = np.clip((result + 1.0) / 2.0, 0.0, 1.0)[0]
    return frames.transpose(1, 2, 3, 0).astype(F32)


# ---------------------------------------------------------------------------------------------------------------
# R20: weight-key mapping (ModelDownloader.swift:605-639,756-899)
# ---------------------------------------------------------------------------------------------------------------
def map_transformer_key(key):
    if key.endswith(".weight_scale") or key.endswith(".input_scale"):
        return None
    if "audio" in key or key.startswith("vocoder") or "av_ca_" in key:
        return None
    pre = "model.diffusion_model."
    if not key.startswith(pre):
        return None
    if key.startswith(pre + "video_embeddings_connector.") or key.startswith(pre + "audio_embeddings_connector."):
        return None
    k = key[len(pre):]
    if (k.startswith("audio_") or ".audio_" in k or k.startswith("av_cross_attn_") or "video_to_audio" in k
            or "video_a2v" in k or "a2v_ca" in k or "scale_shift_table_a2v" in k):
        return None
    if k.startswith("proj_in."):
        k = "patchify_proj." + k[len("proj_in."):]
    if k.startswith("time_embed.emb.timestep_embedder."):
        k = "adaln_single.emb." + k[len("time_embed.emb.timestep_embedder."):]
    elif k.startswith("time_embed.linear."):
        k = "adaln_single." + k[len("time_embed."):]
    elif k.startswith("adaln_single.emb.timestep_embedder."):
        k = "adaln_single.emb." + k[len("adaln_single.emb.timestep_embedder."):]
    k = k.replace(".emb.timestep_embedder.", ".emb.")
    k = k.replace(".norm_q.", ".q_norm.").replace(".norm_k.", ".k_norm.")
    k = k.replace(".to_out.0.", ".to_out.")
    k = k.replace("ff.net.0.proj.", "ff.project_in.proj.").replace("ff.net.2.", "ff.project_out.")
    return k


def map_vae_key(key):
    if key.startswith("vae."):
        key = key[4:]
    if key.startswith("encoder."):
        return None
    if "per_channel_statistics" in key:
        base = key.split(".")[-1]
        return {"mean-of-means": "mean_of_means", "std-of-means": "std_of_means"}.get(base)
    if key == "latents_mean":
        return "mean_of_means"
    if key == "latents_std":
        return "std_of_means"
    k = key[len("decoder."):] if key.startswith("decoder.") else key
    if k.startswith("mid_block."):
        k = "up_blocks_0." + k[len("mid_block."):]
    else:
        for i in range(3):
            up, rs = f"up_blocks.{i}.upsamplers.0.", f"up_blocks.{i}.resnets."
            if k.startswith(up):
                k = f"up_blocks_{2 * i + 1}." + k[len(up):]
                break
            if k.startswith(rs):
                k = f"up_blocks_{2 * i + 2}.resnets." + k[len(rs):]
                break
    for i in range(7):
        src = f"up_blocks.{i}."
        if k.startswith(src):
            k = f"up_blocks_{i}." + k[len(src):]
            break
    return k.replace(".resnets.", ".res_blocks.")


def map_lora_key(key):
    """LoRAKeyMapper.loraKeyToModelKey (LoRALoader.swift:209-243)."""
    k = key[len("diffusion_model."):] if key.startswith("diffusion_model.") else key
    k = k.replace(".emb.timestep_embedder.", ".emb.").replace(".to_out.0", ".to_out")
    k = k.replace(".ff.net.0.proj", ".ff.project_in.proj").replace(".ff.net.2", ".ff.project_out")
    return k + ".weight"


# ---------------------------------------------------------------------------------------------------------------
# synthetic parameters for tests: deterministic, bf16-rounded
# ---------------------------------------------------------------------------------------------------------------
def synth_dit_weights(cfg, seed=1234):
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in dit_param_shapes(cfg).items():
        if k.endswith("_norm.weight"):
            v = 1.0 + 0.02 * rng.standard_normal(shp)
        elif k.endswith(".bias"):
            v = 0.01 * rng.standard_normal(shp)
        elif k.endswith("scale_shift_table"):
            v = 0.05 * rng.standard_normal(shp)
        else:
            v = rng.standard_normal(shp) / math.sqrt(shp[-1])
        w[k] = bf16_round(v.astype(F32))
    return w


def synth_vae_weights(channels=VAE_CHANNELS, latent_channels=128, seed=77, timestep_conditioning=False):
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in vae_param_shapes(channels, latent_channels, timestep_conditioning).items():
        if k == "timestep_scale_multiplier":
            w[k] = np.array(1000.0, F32)
            continue
        if "timestep_embedder" in k and k.endswith(".weight"):
            w[k] = bf16_round((rng.standard_normal(shp, dtype=F32) / F32(16.0)))
            continue
        if k == "mean_of_means":
            v = 0.1 * rng.standard_normal(shp)
        elif k == "std_of_means":
            v = 1.0 + 0.1 * rng.random(shp)
        elif k.endswith(".bias"):
            v = 0.01 * rng.standard_normal(shp)
        elif "scale_shift_table" in k:
            v = 0.05 * rng.standard_normal(shp)
        else:
            fan_in = shp[1] * 27
            v = rng.standard_normal(shp, dtype=F32) / F32(math.sqrt(fan_in))
        w[k] = bf16_round(np.asarray(v, dtype=F32))
    return w


def dit_file_keys(w):
    """Module-key dict -> file-key dict in the unified checkpoint's naming (inverse of map_transformer_key)."""
    out = {}
    for k, v in w.items():
        fk = k
        fk = fk.replace("ff.project_in.proj.", "ff.net.0.proj.").replace("ff.project_out.", "ff.net.2.")
        fk = fk.replace(".to_out.", ".to_out.0.").replace(".q_norm.", ".norm_q.").replace(".k_norm.", ".norm_k.")
        if fk.startswith("adaln_single.emb."):
            fk = "adaln_single.emb.timestep_embedder." + fk[len("adaln_single.emb."):]
        if fk.startswith("patchify_proj."):
            fk = "proj_in." + fk[len("patchify_proj."):]
        out["model.diffusion_model." + fk] = v
    return out


def vae_file_keys(w):
    """Module-key dict -> Diffusers-style standalone VAE file keys (inverse of map_vae_key)."""
    out = {}
    for k, v in w.items():
        if k == "mean_of_means":
            out["latents_mean"] = v
            continue
        if k == "std_of_means":
            out["latents_std"] = v
            continue
        fk = k.replace(".res_blocks.", ".resnets.")
        if fk.startswith("up_blocks_0."):
            fk = "mid_block." + fk[len("up_blocks_0."):]
        else:
            for i in range(3):
                if fk.startswith(f"up_blocks_{2 * i + 1}."):
                    fk = f"up_blocks.{i}.upsamplers.0." + fk[len(f"up_blocks_{2 * i + 1}."):]
                    break
                if fk.startswith(f"up_blocks_{2 * i + 2}.resnets."):
                    fk = f"up_blocks.{i}.resnets." + fk[len(f"up_blocks_{2 * i + 2}.resnets."):]
                    break
        out["decoder." + fk] = v
    return out


# ---------------------------------------------------------------------------------------------------------------
# R22: LoRA fusion (LoRALoader.swift:63-111,162-178; LoRAAdapter.swift:64-166)
# ---------------------------------------------------------------------------------------------------------------
def lora_fuse(w, lora, scale=1.0):
    """`lora` maps file keys (e.g. 'diffusion_model.transformer_blocks.0.attn1.to_q.lora_down.weight') to arrays.
    Returns (new weight dict, number of fused layers). Arithmetic in bf16 as MLX does for bf16 LoRA files."""
    out = dict(w)
    fused = 0
    for key, down in lora.items():
        if "lora_down" in key:
            up_key = key.replace("lora_down", "lora_up")
            base = key.replace(".lora_down.weight", "").replace(".lora_down", "")
        elif "lora_A" in key:
            up_key = key.replace("lora_A", "lora_B")
            base = key.replace(".lora_A.weight", "").replace(".lora_A", "")
        else:
            continue
        if up_key not in lora:
            continue
        up = lora[up_key]
        rank = down.shape[0]
        eff = F32(scale)
        if base + ".alpha" in lora:
            eff = F32(scale) * (F32(lora[base + ".alpha"]) / F32(rank))
        mk = map_lora_key(base)
        if mk not in out:
            continue
        delta = bf16_round(bf16_round(up).astype(F32) @ bf16_round(down).astype(F32))  # bf16 matmul, f32 accumulate
        delta = bf16_round(delta * eff)
        out[mk] = bf16_round(out[mk].astype(F32) + delta)
        fused += 1
    return out, fused


# ---------------------------------------------------------------------------------------------------------------
# R21: affine group quantisation of every Linear (LTXQuantizationConfig.swift:19-62; MLXNN.quantize, group 64).
# The rounding rule is MLX's (third party, unverifiable here); this is the documented affine min/max rule.
# ---------------------------------------------------------------------------------------------------------------
def fake_quant(wm, bits, group=64):
    wm = wm.astype(F32)
    o, i = wm.shape
    g = wm.reshape(-1, group)
    n_bins = F32((1 << bits) - 1)
    w_max, w_min = g.max(1, keepdims=True), g.min(1, keepdims=True)
    mask = np.abs(w_min) > np.abs(w_max)
    scale = np.maximum((w_max - w_min) / n_bins, F32(1e-7)).astype(F32)
    scale = np.where(mask, scale, -scale)
    edge = np.where(mask, w_min, w_max)
    q0 = np.rint(edge / scale)
    scale = np.where(q0 != 0, edge / np.where(q0 != 0, q0, 1), scale).astype(F32)
    bias = np.where(q0 == 0, F32(0), edge).astype(F32)
    scale, bias = bf16_round(scale), bf16_round(bias)
    q = np.clip(np.rint((g - bias) / scale), 0, n_bins)
    return (q * scale + bias).astype(F32).reshape(o, i)


def quantize_dit_weights(w, bits, group=64):
    out = dict(w)
    for k, v in w.items():
        if k.endswith(".weight") and v.ndim == 2:  # every Linear (norm weights are vectors, tables are not '.weight')
            out[k] = fake_quant(v, bits, group)
    return out


# ---------------------------------------------------------------------------------------------------------------
# R23: latent spatial upscaler + two-stage glue (SpatialUpscaler.swift:14-258,352-379; LTXPipeline.swift:2588-2647)
# ---------------------------------------------------------------------------------------------------------------
def conv_nd_zero(x, weight, bias):
    """MLXNN.Conv3d/Conv2d with padding=1 (zeros), stride 1. x [B,C,F,H,W]; weight (O,I,3,3,3) or (O,I,3,3) (per frame).
    BLAS-speed form (`_conv_taps_blas`, as conv3d_full; the einsum form it replaced is `conv3d_full_einsum`'s twin and was checked
    against torch by tests/test_oracle_vs_torch.py, as this one is)."""
    b, c, t, h, w = x.shape
    if weight.ndim == 4:
        weight = weight[:, :, None]  # kT = 1
        t_src = list(range(t))
    else:
        t_src = [None] + list(range(t)) + [None]
    wt = _taps_first(weight)
    out = np.empty((b, weight.shape[0], t, h, w), F32)
    for bi in range(b):
        out[bi] = _conv_taps_blas(_pad_channels_last(x[bi].astype(F32, copy=False), t_src, "zero"), wt, bias)
    return out


def group_norm3d(x, weight, bias, groups=32, eps=1e-5):
    """UpscalerGroupNorm3D (SpatialUpscaler.swift:14-58): stats over (F,H,W, C/G), population variance. x [B,C,F,H,W]."""
    b, c = x.shape[:2]
    g = x.astype(np.float64).reshape(b, groups, -1)
    mu = g.mean(-1, keepdims=True)
    var = ((g - mu) ** 2).mean(-1, keepdims=True)
    y = ((g - mu) / np.sqrt(var + eps)).reshape(x.shape)
    return (y * weight.astype(np.float64).reshape(1, -1, 1, 1, 1) + bias.astype(np.float64).reshape(1, -1, 1, 1, 1)).astype(F32)


def upscaler_param_shapes(mid=1024, in_ch=128):
    s = {"initial_conv.weight": (mid, in_ch, 3, 3, 3), "initial_conv.bias": (mid,),
         "initial_norm.weight": (mid,), "initial_norm.bias": (mid,),
         "upsampler.conv.weight": (4 * mid, mid, 3, 3), "upsampler.conv.bias": (4 * mid,),
         "final_conv.weight": (in_ch, mid, 3, 3, 3), "final_conv.bias": (in_ch,)}
    for stage in ("res_blocks", "post_upsample_res_blocks"):
        for i in range(4):
            for cv, nm in (("conv1", "norm1"), ("conv2", "norm2")):
                s[f"{stage}.{i}.{cv}.weight"] = (mid, mid, 3, 3, 3)
                s[f"{stage}.{i}.{cv}.bias"] = (mid,)
                s[f"{stage}.{i}.{nm}.weight"] = (mid,)
                s[f"{stage}.{i}.{nm}.bias"] = (mid,)
    return s


def synth_upscaler_weights(mid=128, seed=55):
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in upscaler_param_shapes(mid).items():
        if "norm" in k and k.endswith(".weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shp)
        elif k.endswith(".bias"):
            v = 0.05 * rng.standard_normal(shp)
        else:
            v = rng.standard_normal(shp) / math.sqrt(np.prod(shp[1:]))
        w[k] = bf16_round(np.asarray(v, F32))
    return w


def upscaler_forward(w, x):
    """SpatialUpscaler.callAsFunction (SpatialUpscaler.swift:226-258). x [B,128,F,H,W] -> [B,128,F,2H,2W]."""
    def res_block(p, h):
        r = h
        h = silu(group_norm3d(conv_nd_zero(h, w[p + "conv1.weight"], w[p + "conv1.bias"]), w[p + "norm1.weight"], w[p + "norm1.bias"]))
        h = group_norm3d(conv_nd_zero(h, w[p + "conv2.weight"], w[p + "conv2.bias"]), w[p + "norm2.weight"], w[p + "norm2.bias"])
        return silu(h + r)

    h = conv_nd_zero(x, w["initial_conv.weight"], w["initial_conv.bias"])
    h = silu(group_norm3d(h, w["initial_norm.weight"], w["initial_norm.bias"]))
    for i in range(4):
        h = res_block(f"res_blocks.{i}.", h)
    h = conv_nd_zero(h, w["upsampler.conv.weight"], w["upsampler.conv.bias"])  # [B,4C,F,H,W]
    b, c4, f, hh, ww = h.shape
    c = c4 // 4
    # pixelShuffle2DNHWC: channel = c*4 + i*2 + j -> (2h+i, 2w+j) (SpatialUpscaler.swift:116-131)
    h = h.reshape(b, c, 2, 2, f, hh, ww).transpose(0, 1, 4, 5, 2, 6, 3).reshape(b, c, f, hh * 2, ww * 2)
    for i in range(4):
        h = res_block(f"post_upsample_res_blocks.{i}.", h)
    return conv_nd_zero(h, w["final_conv.weight"], w["final_conv.bias"])


def upsample_latents(wu, latent, mean, std):
    """upsampleLatents (SpatialUpscaler.swift:352-379)."""
    m5, s5 = mean.astype(F32).reshape(1, -1, 1, 1, 1), std.astype(F32).reshape(1, -1, 1, 1, 1)
    x = upscaler_forward(wu, latent.astype(F32) * s5 + m5)
    return ((x - m5) / s5).astype(F32)


def two_stage_latent(w, cfg, wu, mean, std, noise1, noise2, context, mask, width, height, num_frames, num_layers=None):
    """generateVideoTwoStage, distilled T2V (LTXPipeline.swift:2420-2741) up to the final latent."""
    F1, H1, W1 = latent_shape(width // 2, height // 2, num_frames)
    F2, H2, W2 = latent_shape(width, height, num_frames)
    sig1 = sigmas(True, 8, F1 * H1 * W1)
    lat = denoise(w, cfg, noise1.astype(F32) * sig1[0], sig1, context, mask, F1, H1, W1, num_layers=num_layers)
    stage1 = lat
    lat = upsample_latents(wu, lat, mean, std)
    lat = adain_filter_latent(lat, stage1)
    s2 = np.array(STAGE_2_DISTILLED_SIGMA_VALUES, F32)
    lat = (s2[0] * noise2.astype(F32) + (F32(1.0) - s2[0]) * lat).astype(F32)
    return denoise(w, cfg, lat, s2, context, mask, F2, H2, W2, num_layers=num_layers)


# ---------------------------------------------------------------------------------------------------------------
# text-embedding connector (the step immediately before the denoise loop). Restates
# Models/TextEncoder/LTXTextEncoder.swift: norm_and_concat (:62-122), GemmaFeaturesExtractor (:126-187),
# ConnectorAttention (:197-269), BasicTransformerBlock1D (:316-371), Embeddings1DConnector (:375-522),
# encodeFromHiddenStates (:574-643). The Gemma-3 language model itself stays out of scope: its 49 hidden states are
# the INPUT here. Activations are bf16 in the reference (the hidden states arrive as bf16); every op output is rounded
# to bf16 below where MLX would store bf16. PARITY UNPINNED like the rest of this file.
# ---------------------------------------------------------------------------------------------------------------
CONNECTOR_DIM = 3840
CONNECTOR_HEADS = 30
CONNECTOR_LAYERS = 2
CONNECTOR_REGISTERS = 128
GEMMA_STATES = 49


def rope_tables_1d(seq_len, dim=CONNECTOR_DIM, num_heads=CONNECTOR_HEADS, theta=10000.0, max_pos=4096):
    """precomputeFreqsCisDoublePrecision with a [1,1,T] grid of integer positions (LTXTextEncoder.swift:482-497,
    LTXRoPE.swift:375-490): one position dim -> dim/2 frequencies, no padding. cos/sin [T][dim/2] f32 (head h owns
    columns h*64..), before the connector's cast to bf16."""
    n_idx = max(1, dim // 2)
    log_start = math.log(1.0) / math.log(theta)
    log_end = math.log(theta) / math.log(theta)
    idx = [math.pow(theta, log_start + (log_end - log_start) * i / (n_idx - 1) if n_idx > 1 else log_start) * (math.pi / 2.0)
           for i in range(n_idx)]
    cos = np.empty((seq_len, n_idx), F32)
    sin = np.empty((seq_len, n_idx), F32)
    for t in range(seq_len):
        scaled = (float(np.float32(t)) / float(max_pos)) * 2.0 - 1.0
        for i in range(n_idx):
            v = idx[i] * scaled
            cos[t, i] = math.cos(v)
            sin[t, i] = math.sin(v)
    return cos, sin


def norm_and_concat(stacked, seq_lens, padding_side="left", eps=1e-6):
    """normAndConcatPaddedBatch (LTXTextEncoder.swift:62-122). stacked [B,T,D,L] (bf16 values); returns [B,T,D*L] bf16
    values, zero at padded positions; statistics per (batch, layer) over the valid tokens, in f32."""
    x = stacked.astype(F32)
    b, t, d, nl = x.shape
    idx = np.arange(t)[None, :]
    sl = np.asarray(seq_lens).reshape(b, 1)
    mask = (idx < sl) if padding_side == "right" else (idx >= (t - sl))
    m4 = mask[:, :, None, None]
    masked = np.where(m4, x, F32(0))
    denom = (sl.astype(F32) * F32(d)).reshape(b, 1, 1, 1) + F32(eps)
    mean = masked.sum(axis=(1, 2), keepdims=True, dtype=np.float64).astype(F32) / denom
    xmin = np.where(m4, x, F32(np.inf)).min(axis=(1, 2), keepdims=True)
    xmax = np.where(m4, x, F32(-np.inf)).max(axis=(1, 2), keepdims=True)
    rng = xmax - xmin
    normed = bf16_round(F32(8.0) * (x - mean) / (rng + F32(eps)))
    normed = normed.reshape(b, t, d * nl)
    return np.where(mask[:, :, None], normed, F32(0)).astype(F32)


def connector_param_shapes(dim=CONNECTOR_DIM, layers=CONNECTOR_LAYERS, registers=CONNECTOR_REGISTERS, states=GEMMA_STATES):
    sh = {"feature_extractor.aggregate_embed.weight": (dim, dim * states),
          "embeddings_connector.learnable_registers": (registers, dim)}
    for i in range(layers):
        p = f"embeddings_connector.transformer_1d_blocks.{i}."
        for n in ("to_q", "to_k", "to_v", "to_out"):
            sh[p + f"attn1.{n}.weight"] = (dim, dim)
            sh[p + f"attn1.{n}.bias"] = (dim,)
        sh[p + "attn1.q_norm.weight"] = (dim,)
        sh[p + "attn1.k_norm.weight"] = (dim,)
        sh[p + "ff.project_in.proj.weight"] = (4 * dim, dim)
        sh[p + "ff.project_in.proj.bias"] = (4 * dim,)
        sh[p + "ff.project_out.weight"] = (dim, 4 * dim)
        sh[p + "ff.project_out.bias"] = (dim,)
    return sh


def synth_connector_weights(dim=CONNECTOR_DIM, heads=CONNECTOR_HEADS, layers=CONNECTOR_LAYERS,
                            registers=CONNECTOR_REGISTERS, states=GEMMA_STATES, seed=91):
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in connector_param_shapes(dim, layers, registers, states).items():
        if k.endswith("learnable_registers"):
            v = rng.uniform(-1.0, 1.0, shp)  # reference initialiser (LTXTextEncoder.swift:421-425)
        elif k.endswith("_norm.weight"):
            v = 1.0 + 0.02 * rng.standard_normal(shp)
        elif k.endswith(".bias"):
            v = 0.01 * rng.standard_normal(shp)
        elif "aggregate_embed" in k:
            v = rng.standard_normal(shp) / math.sqrt(shp[1])
        else:
            v = 0.02 * rng.standard_normal(shp)
        w[k] = bf16_round(v.astype(F32))
    return w


def replace_padded_with_registers(hidden, valid, registers):
    """replacePaddedWithLearnableRegisters (LTXTextEncoder.swift:428-470): valid tokens are moved to the front in order,
    then position p keeps the moved token where reverse(valid)[p] is set and takes register p mod R elsewhere."""
    b, t, d = hidden.shape
    r = registers.shape[0]
    assert t % r == 0, "sequence length must be divisible by the number of registers"
    out = np.empty_like(hidden)
    tiled = np.tile(registers, (t // r, 1))
    for bi in range(b):
        v = valid[bi].astype(np.int64)
        order = np.argsort((1 - v) * t + np.arange(t), kind="stable")
        adjusted = hidden[bi][order]
        flipped = valid[bi][::-1].astype(F32)[:, None]
        out[bi] = bf16_round(flipped * adjusted + (F32(1) - flipped) * tiled)
    return out


def connector_block(w, p, x, heads, cos, sin):
    """BasicTransformerBlock1D (LTXTextEncoder.swift:316-371) with ConnectorAttention (:197-269)."""
    n = bf16_round(rms_norm(x))
    q = bf16_round(linear(n, w[p + "attn1.to_q.weight"], w[p + "attn1.to_q.bias"]))
    k = bf16_round(linear(n, w[p + "attn1.to_k.weight"], w[p + "attn1.to_k.bias"]))
    v = bf16_round(linear(n, w[p + "attn1.to_v.weight"], w[p + "attn1.to_v.bias"]))
    q = bf16_round(rms_norm(q, w[p + "attn1.q_norm.weight"]))
    k = bf16_round(rms_norm(k, w[p + "attn1.k_norm.weight"]))
    q = bf16_round(apply_split_rope(q, cos, sin, heads))
    k = bf16_round(apply_split_rope(k, cos, sin, heads))
    a = bf16_round(sdpa(q, k, v, heads, 1.0 / math.sqrt(q.shape[-1] // heads)))
    x = bf16_round(x + bf16_round(linear(a, w[p + "attn1.to_out.weight"], w[p + "attn1.to_out.bias"])))
    n = bf16_round(rms_norm(x))
    hdn = bf16_round(gelu_tanh(bf16_round(linear(n, w[p + "ff.project_in.proj.weight"], w[p + "ff.project_in.proj.bias"]))))
    return bf16_round(x + bf16_round(linear(hdn, w[p + "ff.project_out.weight"], w[p + "ff.project_out.bias"])))


def connector_encode(w, hidden_states, attention_mask, padding_side="left", heads=CONNECTOR_HEADS, layers=CONNECTOR_LAYERS,
                     return_intermediates=False):
    """encodeFromHiddenStates (LTXTextEncoder.swift:574-643): hidden_states [L][B][T][D] bf16 values, attention_mask [B][T]
    0/1 -> (video_encoding [B,T,D] bf16 values, mask [B,T] int32 all ones)."""
    hs = np.asarray(hidden_states, dtype=F32)
    nl, b, t, d = hs.shape
    stacked = np.moveaxis(hs, 0, -1)  # [B,T,D,L]
    am = np.asarray(attention_mask)
    seq = am.sum(-1).astype(np.int32)
    nc = norm_and_concat(stacked, seq, padding_side)
    enc = bf16_round(linear(nc, w["feature_extractor.aggregate_embed.weight"]))  # f32 matmul, then cast (:180-186)
    valid = am.astype(bool)  # additive mask (m-1)*3.38e38 >= -9000  <=>  m == 1
    x = replace_padded_with_registers(enc, valid, w["embeddings_connector.learnable_registers"].astype(F32))
    cos, sin = rope_tables_1d(t, d, heads)
    cos, sin = bf16_round(cos), bf16_round(sin)  # cast to the input dtype (:498)
    inter = {"norm_concat": nc, "fe": enc, "registers": x}
    for i in range(layers):
        x = connector_block(w, f"embeddings_connector.transformer_1d_blocks.{i}.", x, heads, cos, sin)
    x = bf16_round(rms_norm(x))
    out_mask = np.ones((b, t), np.int32)  # mask cleared after register replacement (:466-468, :622-626)
    if return_intermediates:
        return x, out_mask, inter
    return x, out_mask


def map_text_encoder_key(key):
    """mapTextEncoderWeights + applyConnectorInternalMapping (ModelDownloader.swift:911-968) preceded by the unified-file
    prefix strip of splitUnifiedWeightsDict (:1353-1399). Returns the module key or None (dropped). Audio connector keys
    map too (kept for completeness; this build loads the video connector only)."""
    k = key
    for pre, new in (("model.diffusion_model.video_embeddings_connector.", "video_embeddings_connector."),
                     ("model.diffusion_model.audio_embeddings_connector.", "audio_embeddings_connector."),
                     ("model.diffusion_model.text_embedding_projection.", "text_embedding_projection.")):
        if k.startswith(pre):
            k = new + k[len(pre):]
            break

    def internal(s):
        for a, b_ in (("transformer_blocks.", "transformer_1d_blocks."), (".norm_q.", ".q_norm."), (".norm_k.", ".k_norm."),
                      (".to_out.0.", ".to_out."), (".ff.net.0.proj.", ".ff.project_in.proj."), (".ff.net.2.", ".ff.project_out.")):
            s = s.replace(a, b_)
        return s

    if k.startswith("text_proj_in."):
        return k.replace("text_proj_in.", "feature_extractor.aggregate_embed.")
    if k.startswith("video_connector."):
        return internal(k.replace("video_connector.", "embeddings_connector."))
    if k.startswith("audio_connector."):
        return internal(k.replace("audio_connector.", "audio_embeddings_connector."))
    if k.startswith("text_embedding_projection."):
        return k.replace("text_embedding_projection.", "feature_extractor.")
    if k.startswith("video_embeddings_connector."):
        return internal(k.replace("video_embeddings_connector.", "embeddings_connector."))
    if k.startswith("audio_embeddings_connector."):
        return internal(k)
    return None


def connector_file_keys(w, unified=True):
    """Module keys -> the names a checkpoint carries (inverse of map_text_encoder_key), for loader tests."""
    out = {}
    for k, v in w.items():
        if k.startswith("feature_extractor."):
            fk = ("model.diffusion_model.text_embedding_projection." + k[len("feature_extractor."):]) if unified \
                else k.replace("feature_extractor.aggregate_embed.", "text_proj_in.")
        else:
            s = k[len("embeddings_connector."):]
            for a, b_ in (("transformer_1d_blocks.", "transformer_blocks."), (".q_norm.", ".norm_q."), (".k_norm.", ".norm_k."),
                          (".to_out.", ".to_out.0."), (".ff.project_in.proj.", ".ff.net.0.proj."), (".ff.project_out.", ".ff.net.2.")):
                s = s.replace(a, b_)
            fk = ("model.diffusion_model.video_embeddings_connector." if unified else "video_connector.") + s
        out[fk] = v
    return out


# ---------------------------------------------------------------------------------------------------------------
# SURVEY 8(f) item 3, second half: the VAE *encoder* that turns the conditioning image into the latent of frame 0.
# Restates Models/VAE/VideoEncoder.swift (encoderPatchify :13-34, spaceToDepth :40-68, EncoderResBlock3d :74-100,
# VAESpaceToDepthDownsample3d :124-168, VideoEncoder :211-312) and encodeImage's normalisation (LTXPipeline.swift:1902-1932).
# `base` scales the channel ladder (reference 128 -> 128,256,512,1024,2048) so tests can run a thinner copy.
# ---------------------------------------------------------------------------------------------------------------
ENC_RESNETS = (4, 6, 6, 2)
ENC_FACTORS = ((1, 2, 2), (2, 1, 1), (2, 2, 2), (2, 2, 2))


def conv3d_causal_zero(x, weight, bias, causal=True):
    """CausalConv3dFull with spatialPaddingMode .zeros (VideoConvolution.swift:238-347): zero pad H/W by 1; temporal pad
    = first frame twice in front (causal) or replicate 1+1. BLAS-speed form (`_conv_taps_blas`)."""
    b, c, t, h, wd = x.shape
    t_src = [min(max(pf - (2 if causal else 1), 0), t - 1) for pf in range(t + 2)]
    wt = _taps_first(weight)
    out = np.empty((b, weight.shape[0], t, h, wd), F32)
    for bi in range(b):
        out[bi] = _conv_taps_blas(_pad_channels_last(x[bi].astype(F32, copy=False), t_src, "zero"), wt, bias)
    return out


def encoder_patchify(x):
    """(B,3,T,H,W) -> (B,48,T,H/4,W/4); channel = c*16 + pw*4 + ph (pW before pH, VideoEncoder.swift:25-31)."""
    b, c, t, h, w = x.shape
    o = x.reshape(b, c, t, h // 4, 4, w // 4, 4).transpose(0, 1, 6, 4, 2, 3, 5)
    return o.reshape(b, c * 16, t, h // 4, w // 4)


def space_to_depth(x, factor):
    """spaceToDepth (VideoEncoder.swift:40-68): channel = c*ft*fh*fw + (it*fh + ih)*fw + iw; odd T is padded in FRONT with
    copies of the first frame."""
    ft, fh, fw = factor
    b, c, t, h, w = x.shape
    if t % ft != 0:
        x = np.concatenate([x[:, :, :1]] * (ft - t % ft) + [x], axis=2)
        t = x.shape[2]
    o = x.reshape(b, c, t // ft, ft, h // fh, fh, w // fw, fw).transpose(0, 1, 3, 5, 7, 2, 4, 6)
    return o.reshape(b, c * ft * fh * fw, t // ft, h // fh, w // fw)


def enc_res_block(w, p, x):
    h = silu(pixel_norm(x))
    h = conv3d_causal_zero(h, w[p + "conv1.conv.weight"], w[p + "conv1.conv.bias"])
    h = silu(pixel_norm(h))
    h = conv3d_causal_zero(h, w[p + "conv2.conv.weight"], w[p + "conv2.conv.bias"])
    return (h + x).astype(F32)


def enc_downsample(w, p, x, factor, c_out):
    main = space_to_depth(conv3d_causal_zero(x, w[p + "conv.conv.weight"], w[p + "conv.conv.bias"]), factor)
    res = space_to_depth(x, factor)
    b, cs, t2, h2, w2 = res.shape
    avg = res.reshape(b, c_out, cs // c_out, t2, h2, w2).mean(axis=2, dtype=np.float64).astype(F32)
    return (main + avg).astype(F32)


def vae_encoder_param_shapes(base=128):
    ch = [base, base * 2, base * 4, base * 8, base * 16]
    sh = {"conv_in.conv.weight": (ch[0], 48, 3, 3, 3), "conv_in.conv.bias": (ch[0],),
          "conv_out.conv.weight": (129, ch[4], 3, 3, 3), "conv_out.conv.bias": (129,)}
    for i in range(4):
        for j in range(ENC_RESNETS[i]):
            for cn in ("conv1", "conv2"):
                sh[f"down_blocks_{i}.resnets.resnets.{j}.{cn}.conv.weight"] = (ch[i], ch[i], 3, 3, 3)
                sh[f"down_blocks_{i}.resnets.resnets.{j}.{cn}.conv.bias"] = (ch[i],)
        f = ENC_FACTORS[i]
        co = ch[i + 1] // (f[0] * f[1] * f[2])
        sh[f"down_blocks_{i}.downsamplers.conv.conv.weight"] = (co, ch[i], 3, 3, 3)
        sh[f"down_blocks_{i}.downsamplers.conv.conv.bias"] = (co,)
    for j in range(2):
        for cn in ("conv1", "conv2"):
            sh[f"mid_block.resnets.{j}.{cn}.conv.weight"] = (ch[4], ch[4], 3, 3, 3)
            sh[f"mid_block.resnets.{j}.{cn}.conv.bias"] = (ch[4],)
    return sh


def synth_vae_encoder_weights(base=128, seed=66):
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in vae_encoder_param_shapes(base).items():
        if k.endswith(".bias"):
            v = 0.01 * rng.standard_normal(shp)
        else:
            v = rng.standard_normal(shp) / math.sqrt(27 * shp[1])
        w[k] = bf16_round(v.astype(F32))
    return w


def vae_encode(w, pixels, base=128, causal=True, mean=None, std=None):
    """VideoEncoder.callAsFunction (VideoEncoder.swift:262-311) then encodeImage's (latent - mean_of_means)/std_of_means
    (LTXPipeline.swift:1920-1927) when mean/std are given. pixels [B,3,T,H,W] -> [B,128,T',H/32,W/32]."""
    ch = [base, base * 2, base * 4, base * 8, base * 16]
    h = encoder_patchify(pixels.astype(F32))
    h = conv3d_causal_zero(h, w["conv_in.conv.weight"], w["conv_in.conv.bias"], causal)
    for i in range(4):
        for j in range(ENC_RESNETS[i]):
            h = enc_res_block(w, f"down_blocks_{i}.resnets.resnets.{j}.", h)
        h = enc_downsample(w, f"down_blocks_{i}.downsamplers.", h, ENC_FACTORS[i], ch[i + 1])
    for j in range(2):
        h = enc_res_block(w, f"mid_block.resnets.{j}.", h)
    h = silu(pixel_norm(h))
    h = conv3d_causal_zero(h, w["conv_out.conv.weight"], w["conv_out.conv.bias"], causal)[:, :128]
    if mean is not None:
        h = (h - mean.astype(F32).reshape(1, -1, 1, 1, 1)) / std.astype(F32).reshape(1, -1, 1, 1, 1)
    return h.astype(F32)


def map_vae_encoder_key(key):
    """mapVAEEncoderWeights (ModelDownloader.swift:1222-1283); None for non-encoder tensors."""
    if not key.startswith("encoder."):
        return None
    k = key[len("encoder."):]
    for i in range(4):
        if k.startswith(f"down_blocks.{i}."):
            k = f"down_blocks_{i}." + k[len(f"down_blocks.{i}."):]
            break
    for i in range(4):
        rp = f"down_blocks_{i}.resnets."
        if k.startswith(rp):
            if not k[len(rp):].startswith("resnets."):
                k = rp + "resnets." + k[len(rp):]
            break
    for i in range(4):
        dp = f"down_blocks_{i}.downsamplers.0."
        if k.startswith(dp):
            k = f"down_blocks_{i}.downsamplers." + k[len(dp):]
            break
    return k


def vae_encoder_file_keys(w):
    out = {}
    for k, v in w.items():
        fk = k
        for i in range(4):
            fk = fk.replace(f"down_blocks_{i}.resnets.resnets.", f"down_blocks.{i}.resnets.")
            fk = fk.replace(f"down_blocks_{i}.downsamplers.", f"down_blocks.{i}.downsamplers.0.")
        out["encoder." + fk] = v
    return out


# ---------------------------------------------------------------------------------------------------------------------
# MLX-compatible noise (R2 generateNoise, LatentUtils.swift:69-83: MLXRandom.seed(seed) then
# MLXRandom.normal(shape, float32)). The generator is mlx-swift 0.30.6's (Package.swift:21), which is NOT in the reference tree:
# restated from MLX's published algorithm. Pinned: the threefry2x32-20 hash by the Random123 known-answer vectors
# (tests/test_host_logic.py). Unpinned: the bits -> uniform -> erfinv pipeline (no MLX run is possible here).
# ---------------------------------------------------------------------------------------------------------------------
def threefry2x32(key, c0, c1):
    """key: two uint32; c0, c1: uint32 arrays (counter words). Returns the two output word arrays."""
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, k0 ^ k1 ^ np.uint32(0x1BD11BDA))
    with np.errstate(over="ignore"):
        x0 = np.asarray(c0, np.uint32) + ks[0]
        x1 = np.asarray(c1, np.uint32) + ks[1]
        for i in range(5):
            for r in rot[i & 1]:
                x0 = x0 + x1
                x1 = (x1 << np.uint32(r)) | (x1 >> np.uint32(32 - r))
                x1 = x1 ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0, x1


def mlx_random_bits(key, n):
    """n uint32 words from one key: words i and i + ceil(n/2) come from hash(key, (i, i + ceil(n/2)))."""
    half, odd = n // 2, n % 2
    second = half + odd
    out = np.empty(n, np.uint32)
    i = np.arange(half, dtype=np.uint32)
    a, b = threefry2x32(key, i, i + np.uint32(second))
    out[:half] = a
    out[second:second + half] = b
    if odd:
        a, _ = threefry2x32(key, np.array([half], np.uint32), np.array([0], np.uint32))
        out[half] = a[0]
    return out


def mlx_erfinv(a):
    a = np.asarray(a, np.float32)
    f = np.float32
    t = np.log((f(1.0) - a.astype(np.float64) * a.astype(np.float64)).astype(np.float32))  # fma(a, -a, 1): one rounding

    def poly(cs):
        p = np.full_like(t, f(cs[0]))
        for c in cs[1:]:
            p = (p.astype(np.float64) * t.astype(np.float64) + np.float64(f(c))).astype(np.float32)  # fma: one rounding
        return p

    big = poly((3.03697567e-10, 2.93243101e-8, 1.22150334e-6, 2.84108955e-5, 3.93552968e-4, 3.02698812e-3, 4.83185798e-3,
                -2.64646143e-1, 8.40016484e-1))
    small = poly((5.43877832e-9, 1.43285448e-7, 1.22774793e-6, 1.12963626e-7, -5.61530760e-5, -1.47697632e-4, 2.31468678e-3,
                  1.15392581e-2, -2.32015476e-1, 8.86226892e-1))
    return a * np.where(np.abs(t) > f(6.125), big, small)


def mlx_random_normal(seed, shape, draw_index=0):
    """The draw_index-th keyless MLXRandom.normal(shape) after MLXRandom.seed(seed)."""
    g = (np.uint32((seed >> 32) & 0xFFFFFFFF), np.uint32(seed & 0xFFFFFFFF))
    sub = None
    for _ in range(draw_index + 1):
        w = mlx_random_bits(g, 4)
        g, sub = (w[0], w[1]), (w[2], w[3])
    n = int(np.prod(shape))
    f = np.float32
    u = mlx_random_bits(sub, n).astype(np.float32) / f(4294967295.0)
    u = np.minimum(u, np.nextafter(f(1.0), f(0.0)))
    lo = np.nextafter(f(-1.0), f(0.0))
    u = (f(1.0) - lo) * u + lo
    return (f(np.sqrt(2.0)) * mlx_erfinv(u)).astype(np.float32).reshape(shape)
