#!/usr/bin/env python3
"""Random-shape sweep of the GEMM epilogues and the attention dispatcher against f32 references (torch on the CPU), and of the conv3d
launcher (192-row halo kernel, tall kernel, ring kernel, tail windows) against torch's conv3d on integer data: a wider net than the fixed
cases of tests/test_kernels_gpu.py / tests/test_vae_gpu.py, same tolerances. Usage: python tools/fuzz_kernels.py [cases] [seed]"""
import importlib
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")


def bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().to(torch.bfloat16)


def f32(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def run(ctx, cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        # ---- gated residual GEMM: x += gate * (A.B^T + bias), ragged M, N (multiple of 4), K (multiple of 64: 1..40 K-tiles)
        M = int(rng.integers(1, 700))
        N = int(rng.integers(1, 200)) * 4
        K = int(rng.integers(1, 41)) * 64
        A = rng.standard_normal((M, K))
        B = rng.standard_normal((N, K)) / math.sqrt(K)
        bias, gate, x0 = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal((M, N))
        Ad, Bd = bf16(A), bf16(B)
        use_gate = bool(rng.integers(0, 2))
        use_mirror = bool(rng.integers(0, 2))
        x = f32(x0)
        mirror = torch.empty((M, N), device="cuda", dtype=torch.bfloat16) if use_mirror else None
        ctx.op_gemm_gated_residual(Ad, Bd, f32(bias), f32(gate) if use_gate else None, 0.75, x, mirror)
        torch.cuda.synchronize()
        g = gate[None] if use_gate else 0.75
        ref = x0 + g * ((Ad.float().cpu() @ Bd.float().cpu().T).numpy() + bias[None])
        err = np.abs(x.cpu().numpy() - ref).max()
        ok = err <= 1e-4
        if use_mirror:
            ok = ok and np.abs(mirror.float().cpu().numpy() - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-3
        if not ok:
            bad += 1
        if verbose:
            print(f"gemm gated M={M:4d} N={N:4d} K={K:5d} gate={use_gate} mirror={use_mirror}: max err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)

        # ---- attention: any Tq, Tk, optional key mask
        Bn, H = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        Tq, Tk = int(rng.integers(1, 900)), int(rng.integers(1, 1300))
        D = H * 128
        q, k, v = (bf16(rng.standard_normal((Bn, t, D))) for t in (Tq, Tk, Tk))
        prescaled = bool(rng.integers(0, 2))  # Q carrying (1 / sqrt(128)) * log2(e), as the DiT's q-norm pass writes it
        if prescaled:
            q = (q.float() * (1.4426950408889634 / math.sqrt(128.0))).to(torch.bfloat16)
        biasv = None
        if rng.integers(0, 2):
            m = (rng.random((Bn, Tk)) > 0.3).astype(np.float32)
            m[:, 0] = 1
            biasv = f32((1 - m) * -10000.0)
        ldvt = ((Tk + 63) // 64) * 64
        vt = torch.zeros((Bn, D, ldvt), device="cuda", dtype=torch.bfloat16)
        vt[:, :, :Tk] = v.transpose(1, 2)
        o = torch.empty((Bn, Tq, D), device="cuda", dtype=torch.bfloat16)
        scale = 1.0 / math.sqrt(128.0)
        ctx.op_attention(q, k, vt, biasv, H, o, -1.0 if prescaled else scale)
        torch.cuda.synchronize()
        qh, kh, vh = (t.float().cpu().reshape(Bn, -1, H, 128).permute(0, 2, 1, 3) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) * (math.log(2.0) if prescaled else scale)
        if biasv is not None:
            s = s + biasv.cpu()[:, None, None, :]
        ref = (torch.softmax(s, dim=-1) @ vh).permute(0, 2, 1, 3).reshape(Bn, Tq, D).numpy()
        got = o.float().cpu().numpy()
        err = np.abs(got - ref).max()
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        ok = err <= 2e-2 and rel <= 1e-2
        if not ok:
            bad += 1
        if verbose:
            print(f"attention B={Bn} H={H} Tq={Tq:4d} Tk={Tk:4d} mask={biasv is not None} prescaled={prescaled} key ranges={ctx.attention_key_splits(Bn, H, Tq, Tk)}: max err {err:.2e} rel {rel:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    return bad


def run_conv(ctx, cases, seed, verbose=True):
    """conv3d launches of random shape on small-integer data against torch's conv3d: EXACT equality (any summation order gives the same
    f32). Half the cases have W in {192, 96} with an even / fourfold H and Cout a multiple of 128, i.e. the tall kernel of conv_halo2.inc
    (two / four image rows per tile, rotating row slots, whole-round launch + tail window); the rest take the 192-row halo kernel, the ring
    kernel and its split-K tail windows (odd H, W = 48, ragged widths, Cout = 48 ...)."""
    import torch.nn.functional as F_

    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        tall = bool(rng.integers(0, 2))
        if tall:
            W = int(rng.choice([192, 96]))
            nrt = 2 if W == 192 else 4
            H = nrt * int(rng.integers(1, 40))
            Cin = int(rng.choice([128, 256])) if W == 96 else int(rng.choice([64, 128, 192, 256]))
            Cout = 128 * int(rng.integers(1, 4))
            F = int(rng.integers(1, 5))
        else:
            W = int(rng.choice([48, 96, 192, 24, 50, 200]))
            H = int(rng.integers(2, 40))
            Cin = 64 * int(rng.integers(1, 5))
            Cout = int(rng.choice([48, 64, 96, 128, 132, 256]))
            F = int(rng.integers(1, 5))
        causal = bool(rng.integers(0, 2))
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 2 ** 31)))
        x = torch.randint(-2, 3, (1, Cin, F, H, W), generator=g, device="cuda", dtype=torch.int8).float()
        w = torch.randint(-2, 3, (Cout, Cin, 3, 3, 3), generator=g, device="cuda", dtype=torch.int8).float()
        b = torch.randint(-4, 5, (Cout,), generator=g, device="cuda", dtype=torch.int8).float()
        xd = x[0].permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
        wd = w.reshape(Cout, Cin, 27).permute(0, 2, 1).contiguous().to(torch.bfloat16)   # [O][27][I], the ABI's conv weight layout
        out = torch.full((F, H, W, Cout), float("nan"), device="cuda")
        with ctx.options(conv_tall=3 if tall else 1):   # 3: the tall kernel whatever the tile count (the launcher keeps small launches on 192-row tiles)
            ctx.op_conv3d(xd, wd, b, out, causal=causal)
            torch.cuda.synchronize()
        xp = F_.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
        xp = torch.cat([xp[:, :, :1], xp[:, :, :1], xp], 2) if causal else torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
        ref = F_.conv3d(xp.double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
        ok = bool(torch.equal(out, ref))
        if not ok:
            bad += 1
        if verbose:
            print(f"conv3d {Cin:3d} -> {Cout:3d} at {F}x{H}x{W} causal={causal}{' (tall tile)' if tall else ''}: {'exact' if ok else 'FAIL max err %.3g' % float((out - ref).abs().max())}", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    the_ctx = ltx.Context(0)
    n_bad = run(the_ctx, n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    n_bad += run_conv(the_ctx, n_cases // 2, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{2 * n_cases + n_cases // 2} cases, {n_bad} failures")
    sys.exit(1 if n_bad else 0)
