#!/usr/bin/env python3
"""One DiT forward at the other BASELINE.json configurations (full 48-layer architecture, synthetic weights): robustness +
ms per forward. Usage: python tools/bench_dit_shapes.py [config number] [--q8]   (--q8: qint8 transformer, LTXQuantizationConfig)"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
only = argv[0] if argv else None  # e.g. "1" -> only config 1
ctx = ltx.Context(0)
ctx.dit_init_synthetic(None, seed=1234)
if "--q8" in sys.argv:
    ctx.dit_quantize(8)
    if "--qb-off" in sys.argv:
        ctx.set_option("qb_off", 1)
    print("qint8 transformer" + (" (option qb_off: every launch through the scratch matrix)" if "--qb-off" in sys.argv else ""), flush=True)
S = 1024
for name, B, F, H, W in (("config 1  256x256x9   T=128 ", 1, 2, 8, 8), ("config 2  768x512x25  T=1536", 1, 4, 16, 24),
                         ("config 3  CFG pair    T=1536", 2, 4, 16, 24), ("config 4  1536x1024x25 T=6144", 1, 4, 32, 48),
                         ("config 5  768x512x201 T=9984", 1, 26, 16, 24)):
    if only and not name.startswith("config " + only):
        continue
    T = F * H * W
    lat = torch.empty((B, T, 128), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(lat, seed=3)
    c = torch.empty((B, S, 3840), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(c, seed=4)
    ts = torch.full((B,), 0.7, dtype=torch.float32, device="cuda")
    vel = torch.empty((B, T, 128), dtype=torch.float32, device="cuda")
    ctx.dit_forward_dev(lat, c, ts, None, F, H, W, vel, ctx_version=7 + B, mask_all_ones=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(vel).all()), name
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.dit_forward_dev(lat, c, ts, None, F, H, W, vel, ctx_version=7 + B, mask_all_ones=True)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{name}: {ms:9.3f} ms per forward (B={B})", flush=True)
ctx.close()
