#!/usr/bin/env python3
"""A few launches of this library's GEMM tiles and of the library GEMM (hipBLASLt via torch.matmul) on the wide DiT shapes,
for a rocprofv3 --pmc pass (tools/pmc_clock.py)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
ctx = ltx.Context(0)
for M, N, K in ((1536, 8192, 4096), (1536, 16384, 4096), (1536, 4096, 16384)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Bs = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(6)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for i in range(6):
        ctx.op_gemm(A, Bs[i], None, tile_cfg=21, out_bf16=out)
    for i in range(6):
        ctx.op_gemm(A, Bs[i], None, tile_cfg=1, out_bf16=out)
    for i in range(6):
        ctx.op_gemm(A, Bs[i], None, tile_cfg=41, out_bf16=out)
    for i in range(6):
        torch.matmul(A, Bs[i].t(), out=out)
    torch.cuda.synchronize()
ctx.close()
