#!/usr/bin/env python3
"""Cost of the gated-residual epilogue (x += gate * (A.W^T + b), f32 residual stream + bf16 mirror) against the same product with a
plain f32 store, at the token counts of the BASELINE configurations (N = K = 4096: the attention output projections and, with K = 16384,
ff2). Run with LTX_LIB=<other libltxhip.so> for a same-box A/B.    python tools/bench_gated_gemm.py"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
ctx = ltx.Context(0)
N = 4096
for M, K in ((1536, 4096), (3072, 4096), (6144, 4096), (9984, 4096), (9984, 16384)):
    A = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
    W = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(A, seed=1)
    ctx.op_fill_normal_bf16(W, seed=2, std=0.02)
    bias = torch.zeros(N, dtype=torch.float32, device="cuda")
    gate = torch.full((1, N), 0.01, dtype=torch.float32, device="cuda")
    x = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    mirror = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n

    plain = timed(lambda: ctx.op_gemm(A, W, bias, out_f32=out))
    gv = timed(lambda: ctx.op_gemm_gated_residual(A, W, bias, gate, 1.0, x, mirror))
    gs = timed(lambda: ctx.op_gemm_gated_residual(A, W, bias, None, 0.01, x, mirror))
    rmw = M * N * (4 + 4 + 2) / 1e6
    print(f"M={M:5d} K={K:5d}: plain {plain:7.1f} us  gated(vector) {gv:7.1f} us  gated(scalar) {gs:7.1f} us   "
          f"[epilogue read-modify-write {rmw:.0f} MB = {rmw / 4e3 * 1e3:.0f} us at 4 TB/s]", flush=True)
ctx.close()
