#!/usr/bin/env python3
"""Does capturing one DiT forward in a HIP graph shorten it? Same forward, eager launches against graph replays, same process.
Usage: python tools/graph_probe.py"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
ctx = ltx.Context(0)
ctx.dit_init_synthetic(None, seed=1234)
B, F, H, W, S = 1, 4, 16, 24, 1024
T = F * H * W
lat = torch.empty((B, T, 128), dtype=torch.bfloat16, device="cuda")
ctx.op_fill_normal_bf16(lat, seed=3)
c = torch.empty((B, S, 3840), dtype=torch.bfloat16, device="cuda")
ctx.op_fill_normal_bf16(c, seed=4)
ts = torch.full((B,), 0.7, dtype=torch.float32, device="cuda")
vel = torch.empty((B, T, 128), dtype=torch.float32, device="cuda")
vel2 = torch.empty_like(vel)


def fwd(out):
    ctx.dit_forward_dev(lat, c, ts, None, F, H, W, out, ctx_version=9, mask_all_ones=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


fwd(vel)
torch.cuda.synchronize()
eager = timeit(lambda: fwd(vel))
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    fwd(vel2)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
graph = timeit(g.replay)
eager2 = timeit(lambda: fwd(vel))
print(f"eager {eager:.3f} ms  graph {graph:.3f} ms  eager again {eager2:.3f} ms  identical output: {bool(torch.equal(vel, vel2))}")
