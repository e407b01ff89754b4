#!/usr/bin/env python3
"""VAE decode micro-benchmark at 768x512x25 (latent 4x16x24): python tools/bench_vae.py [--iters 3]"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--F", type=int, default=4)
ap.add_argument("--H", type=int, default=16)
ap.add_argument("--W", type=int, default=24)
ap.add_argument("--ab", default=None, help="same-process A/B of one library option, interleaved rounds: --ab conv_stagger=0,1 (bit-equality of the two decodes is checked)")
ap.add_argument("--encode", action="store_true", help="time the VAE encoder on one (32H x 32W) image instead (image-to-video)")
a = ap.parse_args()
ctx = ltx.Context(0)
if a.encode:
    ctx.vae_encoder_init_synthetic(0, seed=66)
    px = torch.rand((1, 3, 1, a.H * 32, a.W * 32), dtype=torch.float32, device="cuda") * 2 - 1
    lat = torch.empty((1, 128, 1, a.H, a.W), dtype=torch.float32, device="cuda")
    ctx.vae_encode_dev(px, lat)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(lat).all())
    t0 = time.perf_counter()
    for _ in range(a.iters):
        ctx.vae_encode_dev(px, lat)
    torch.cuda.synchronize()
    print(f"vae encode {a.W * 32}x{a.H * 32} image: {1e3 * (time.perf_counter() - t0) / a.iters:.3f} ms")
    ctx.close()
    sys.exit(0)
ctx.vae_init_synthetic(seed=77)
lat = torch.empty((1, 128, a.F, a.H, a.W), dtype=torch.float32, device="cuda")
ctx.op_fill_normal_f32(lat, seed=45)
nf = 8 * (a.F - 1) + 1
frames = torch.empty((nf, a.H * 32, a.W * 32, 3), dtype=torch.float32, device="cuda")
ctx.vae_decode_dev(lat, a.F, a.H, a.W, frames)
torch.cuda.synchronize()
if a.ab:
    key, vals = a.ab.split("=")
    vals = [int(v) for v in vals.split(",")]
    outs, times = {}, {v: [] for v in vals}
    for v in vals:
        ctx.set_option(key, v)
        ctx.vae_decode_dev(lat, a.F, a.H, a.W, frames)
        torch.cuda.synchronize()
        outs[v] = frames.clone()
    for r in range(6):
        for v in vals:
            ctx.set_option(key, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ctx.vae_decode_dev(lat, a.F, a.H, a.W, frames)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / a.iters)
    same = all(torch.equal(outs[vals[0]], outs[v]) for v in vals[1:])
    print(f"vae decode, option {key}: " + "   ".join(f"{v}: " + " ".join(f"{t:.3f}" for t in times[v]) + f" (median {sorted(times[v])[len(times[v]) // 2]:.3f} ms)" for v in vals)
          + f"   outputs bit-equal: {same}")
    ctx.close()
    sys.exit(0)
t0 = time.perf_counter()
for _ in range(a.iters):
    ctx.vae_decode_dev(lat, a.F, a.H, a.W, frames)
torch.cuda.synchronize()
print(f"vae decode {1e3 * (time.perf_counter() - t0) / a.iters:.3f} ms")
ctx.close()
