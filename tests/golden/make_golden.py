#!/usr/bin/env python3
"""Generates the committed golden fixtures. The reference cannot run here (no Swift/MLX), so the expected outputs
come from the oracle, and the independent cross-check is torch CPU (conv3d via F.conv3d on the padded input; the DiT
fixture is additionally recomputed with a torch re-implementation of the attention/FFN blocks below and must agree; the
text-embedding connector and the VAE encoder fixtures are cross-checked the same way).

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ltx_oracle as o  # noqa: E402


def torch_dit_forward(w, cfg, latent, context, ts, mask, F, H, W):
    """Independent f32 torch implementation (different library, different op decomposition) of the same forward."""
    t = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in w.items()}
    bf = lambda x: x.to(torch.bfloat16).float()
    lin = lambda x, p: Fn.linear(x, t[p + ".weight"], t[p + ".bias"])
    D, Hh = cfg.dim, cfg.num_heads
    x = bf(lin(torch.from_numpy(latent), "patchify_proj"))
    B = x.shape[0]
    tt = torch.from_numpy(np.asarray(ts, np.float32)) * cfg.timestep_mult
    half = 128
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = tt[:, None] * freqs[None]
    e = torch.cat([torch.cos(args), torch.sin(args)], -1)
    emb = lin(Fn.silu(lin(e, "adaln_single.emb.linear_1")), "adaln_single.emb.linear_2")
    ada = lin(Fn.silu(emb), "adaln_single.linear").reshape(B, 1, 6, D)
    c = bf(lin(torch.from_numpy(context), "caption_projection.linear_1"))
    c = bf(lin(bf(Fn.gelu(c, approximate="tanh")), "caption_projection.linear_2"))
    bias = None if mask is None else ((1 - torch.from_numpy(mask).float()) * -10000.0)[:, None, None, :]
    cos, sin = (torch.from_numpy(a) for a in o.rope_tables(F, H, W, D, Hh, cfg.rope_theta, cfg.max_pos))

    def rope(z):
        zh = z.reshape(B, -1, Hh, 2, 64)
        cc, ss = cos.reshape(1, -1, Hh, 64), sin.reshape(1, -1, Hh, 64)
        a, b = zh[:, :, :, 0], zh[:, :, :, 1]
        return torch.stack([a * cc - b * ss, b * cc + a * ss], 3).reshape(B, -1, D)

    def attn(p, xq, ctx=None, bias=None, use_rope=False, kvbf=False):
        src = xq if ctx is None else ctx
        q, k, v = lin(xq, p + "to_q"), lin(src, p + "to_k"), lin(src, p + "to_v")
        if kvbf:
            k, v = bf(k), bf(v)
        q = Fn.rms_norm(q, (D,), t[p + "q_norm.weight"], 1e-6)
        k = Fn.rms_norm(k, (D,), t[p + "k_norm.weight"], 1e-6)
        if kvbf:
            k = bf(k)
        if use_rope:
            q, k = rope(q), rope(k)
        sh = lambda z: z.reshape(B, -1, Hh, 128).transpose(1, 2)
        y = Fn.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=bias, scale=1 / math.sqrt(128))
        return lin(y.transpose(1, 2).reshape(B, -1, D), p + "to_out")

    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        a6 = t[p + "scale_shift_table"][None, None] + ada
        n = Fn.rms_norm(x, (D,), None, 1e-6)
        if i == 0:
            n = bf(n)
        x = x + attn(p + "attn1.", n * (1 + a6[:, :, 1]) + a6[:, :, 0], use_rope=True) * a6[:, :, 2]
        x = x + attn(p + "attn2.", x, ctx=c, bias=bias, kvbf=True)
        n = Fn.rms_norm(x, (D,), None, 1e-6) * (1 + a6[:, :, 4]) + a6[:, :, 3]
        hdn = Fn.gelu(lin(n, p + "ff.project_in.proj"), approximate="tanh")
        x = x + lin(hdn, p + "ff.project_out") * a6[:, :, 5]
    ss = t["scale_shift_table"][None, None] + emb.reshape(B, 1, 1, D)
    out = Fn.layer_norm(x, (D,), eps=1e-6) * (1 + ss[:, :, 1]) + ss[:, :, 0]
    return lin(out, "proj_out").numpy()


def torch_connector(w, hs, am, heads, layers):
    """Independent torch implementation of the connector (F.linear / F.rms_norm / F.scaled_dot_product_attention, rounding to
    bf16 where MLX stores bf16); register replacement and the feature-extractor statistics are written from scratch."""
    t = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in w.items()}
    bf = lambda x: x.to(torch.bfloat16).float()
    x = torch.from_numpy(hs).permute(1, 2, 3, 0)  # [B,T,D,L]
    B, T, D, L = x.shape
    m = torch.from_numpy(am).bool()
    n = m.sum(-1)
    out = torch.zeros(B, T, D * L)
    for b in range(B):
        xv = x[b][m[b]]  # [n,D,L]
        mean = xv.double().sum((0, 1)).float() / (n[b].float() * D + 1e-6)
        rng_ = xv.amax((0, 1)) - xv.amin((0, 1))
        out[b][m[b]] = bf(8.0 * (xv - mean) / (rng_ + 1e-6)).reshape(int(n[b]), D * L)
    enc = bf(Fn.linear(out, t["feature_extractor.aggregate_embed.weight"]))
    reg = t["embeddings_connector.learnable_registers"]
    xs = torch.empty(B, T, D)
    for b in range(B):
        valid = enc[b][m[b]]
        k = valid.shape[0]
        xs[b, :k] = valid  # left padding: reverse(valid) keeps the first k positions
        for p_ in range(k, T):
            xs[b, p_] = reg[p_ % reg.shape[0]]
    cos, sin = (bf(torch.from_numpy(a)) for a in o.rope_tables_1d(T, D, heads))

    def rope(z):
        zh = z.reshape(B, T, heads, 2, 64)
        cc, ss = cos.reshape(1, T, heads, 64), sin.reshape(1, T, heads, 64)
        a, b_ = zh[:, :, :, 0], zh[:, :, :, 1]
        return torch.stack([a * cc - b_ * ss, b_ * cc + a * ss], 3).reshape(B, T, D)

    x = xs
    for i in range(layers):
        p = f"embeddings_connector.transformer_1d_blocks.{i}."
        lin = lambda z, q: bf(Fn.linear(z, t[p + q + ".weight"], t[p + q + ".bias"]))
        nrm = bf(Fn.rms_norm(x, (D,), None, 1e-6))
        q = bf(rope(bf(Fn.rms_norm(lin(nrm, "attn1.to_q"), (D,), t[p + "attn1.q_norm.weight"], 1e-6))))
        k = bf(rope(bf(Fn.rms_norm(lin(nrm, "attn1.to_k"), (D,), t[p + "attn1.k_norm.weight"], 1e-6))))
        v = lin(nrm, "attn1.to_v")
        sh = lambda z: z.reshape(B, T, heads, 128).transpose(1, 2)
        a = bf(Fn.scaled_dot_product_attention(sh(q), sh(k), sh(v), scale=1 / math.sqrt(128)).transpose(1, 2).reshape(B, T, D))
        x = bf(x + lin(a, "attn1.to_out"))
        nrm = bf(Fn.rms_norm(x, (D,), None, 1e-6))
        x = bf(x + lin(bf(Fn.gelu(lin(nrm, "ff.project_in.proj"), approximate="tanh")), "ff.project_out"))
    return bf(Fn.rms_norm(x, (D,), None, 1e-6)).numpy()


def make_connector_fixture():
    dim, heads, layers, regs, states, seed = 256, 2, 2, 8, 5, 17
    w = o.synth_connector_weights(dim=dim, heads=heads, layers=layers, registers=regs, states=states, seed=seed)
    rng = np.random.default_rng(200)
    B, T = 2, 16
    hs = o.bf16_round((rng.standard_normal((states, B, T, dim)) * 2.0 + 0.25).astype(np.float32))
    am = np.zeros((B, T), np.int32)
    am[0, 5:] = 1
    am[1, 12:] = 1
    ctxv, _ = o.connector_encode(w, hs, am, heads=heads, layers=layers)
    tv = torch_connector(w, hs, am, heads, layers)
    rel = np.linalg.norm(ctxv - tv) / np.linalg.norm(tv)
    print("connector oracle vs torch: rel l2", rel)
    assert rel < 1e-2  # both round to bf16 after every op; single-ulp flips of intermediate values are the difference
    np.savez_compressed(os.path.join(HERE, "connector_tiny.npz"), seed=seed, dim=dim, heads=heads, layers=layers, registers=regs,
                        states=states, hidden=hs, mask=am, context=ctxv.astype(np.float32))


def torch_vae_encode(w, px, base):
    """Independent torch implementation: F.conv3d on explicitly padded inputs, space-to-depth through unfold-free reshapes of
    a different axis order than the oracle's."""
    t = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in w.items()}

    def conv(x, p):
        xp = Fn.pad(x, (1, 1, 1, 1, 0, 0))
        xp = torch.cat([xp[:, :, :1], xp[:, :, :1], xp], 2)
        return Fn.conv3d(xp, t[p + ".conv.weight"], t[p + ".conv.bias"])

    pn = lambda x: x / torch.sqrt((x * x).mean(1, keepdim=True) + 1e-8)

    def s2d(x, f):
        ft, fh, fw = f
        if x.shape[2] % ft:
            x = torch.cat([x[:, :, :1]] * (ft - x.shape[2] % ft) + [x], 2)
        b, c, tt, hh, ww = x.shape
        y = torch.empty(b, c * ft * fh * fw, tt // ft, hh // fh, ww // fw)
        for it in range(ft):
            for ih in range(fh):
                for iw in range(fw):
                    y[:, (it * fh + ih) * fw + iw::ft * fh * fw] = x[:, :, it::ft, ih::fh, iw::fw]
        return y

    def res(x, p):
        h = conv(Fn.silu(pn(x)), p + "conv1")
        return conv(Fn.silu(pn(h)), p + "conv2") + x

    x = torch.from_numpy(px)
    b, c, tt, hh, ww = x.shape
    h = torch.empty(b, 48, tt, hh // 4, ww // 4)
    for cc in range(3):
        for pw in range(4):
            for ph in range(4):
                h[:, cc * 16 + pw * 4 + ph] = x[:, cc, :, ph::4, pw::4]
    h = conv(h, "conv_in")
    ch = [base << i for i in range(5)]
    for i in range(4):
        for j in range(o.ENC_RESNETS[i]):
            h = res(h, f"down_blocks_{i}.resnets.resnets.{j}.")
        f = o.ENC_FACTORS[i]
        main = s2d(conv(h, f"down_blocks_{i}.downsamplers.conv"), f)
        r = s2d(h, f)
        g = r.shape[1] // ch[i + 1]
        h = main + r.reshape(r.shape[0], ch[i + 1], g, *r.shape[2:]).mean(2)
    for j in range(2):
        h = res(h, f"mid_block.resnets.{j}.")
    return conv(Fn.silu(pn(h)), "conv_out")[:, :128].numpy()


def make_vae_encoder_fixture():
    base, seed = 64, 23
    w = o.synth_vae_encoder_weights(base=base, seed=seed)
    rng = np.random.default_rng(300)
    px = rng.uniform(-1, 1, (1, 3, 3, 32, 64)).astype(np.float32)
    z = o.vae_encode(w, px, base=base)
    zt = torch_vae_encode(w, px, base)
    err = np.abs(z - zt).max() / np.abs(zt).max()
    print("vae encoder oracle vs torch: max rel err", err)
    assert err < 1e-4
    np.savez_compressed(os.path.join(HERE, "vae_encoder_tiny.npz"), seed=seed, base=base, pixels=px, latent=z.astype(np.float32))


def main():
    seed, layers, heads, caption = 11, 2, 2, 128
    cfg = o.DiTConfig(num_layers=layers, num_heads=heads, caption_channels=caption)
    w = o.synth_dit_weights(cfg, seed=seed)
    rng = np.random.default_rng(100)
    F, H, W, S = 2, 3, 5, 21
    latent = o.bf16_round(rng.standard_normal((1, F * H * W, 128)).astype(np.float32))
    context = o.bf16_round(rng.standard_normal((1, S, caption)).astype(np.float32))
    mask = (rng.random((1, S)) > 0.3).astype(np.int32)
    mask[:, 0] = 1
    ts = np.array([0.725], np.float32)
    vel = o.dit_forward(w, cfg, latent, context, ts, mask, F, H, W)
    vel_t = torch_dit_forward(w, cfg, latent, context, ts, mask, F, H, W)
    err = np.abs(vel - vel_t).max() / np.abs(vel).max()
    print("DiT oracle vs torch: max rel err", err)
    assert err < 2e-4
    np.savez_compressed(os.path.join(HERE, "dit_tiny.npz"), seed=seed, layers=layers, heads=heads, caption=caption,
                        fhw=np.array([F, H, W]), latent=latent, context=context, mask=mask, ts=ts,
                        velocity=vel.astype(np.float32))
    x = o.bf16_round(rng.standard_normal((1, 64, 2, 3, 4)).astype(np.float32))
    cw = o.bf16_round((rng.standard_normal((64, 64, 3, 3, 3)) / math.sqrt(64 * 27)).astype(np.float32))
    cb = o.bf16_round((0.1 * rng.standard_normal(64)).astype(np.float32))
    y = o.conv3d_full(x, cw, cb)
    xt = torch.from_numpy(x)
    xp = Fn.pad(xt.reshape(1, 64 * 2, 3, 4), (1, 1, 1, 1), mode="reflect").reshape(1, 64, 2, 5, 6)
    xp = torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
    yt = Fn.conv3d(xp, torch.from_numpy(cw), torch.from_numpy(cb)).numpy()
    print("conv3d oracle vs torch:", np.abs(y - yt).max())
    assert np.abs(y - yt).max() < 1e-4
    np.savez_compressed(os.path.join(HERE, "conv3d_small.npz"), x=x, w=cw, b=cb, y=y.astype(np.float32))
    make_connector_fixture()
    make_vae_encoder_fixture()
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
