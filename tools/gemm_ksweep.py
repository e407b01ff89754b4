"""Time of one GEMM launch against K at fixed M, N: the intercept is the launch's fixed cost (launch gap, pipeline fill, epilogue),
the slope the steady-state cost of a K-tile. Usage: python tools/gemm_ksweep.py [cfg ...]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
ctx = ltx.Context(0)
cfgs = [int(a) for a in sys.argv[1:]] or [21]
M = 1536
for N in (4096, 12288, 16384):
    for cfg in cfgs:
        for out in ("f32", "bf16"):
            ts = {}
            for K in (64, 256, 1024, 2048, 4096, 8192, 16384):
                g = torch.Generator(device="cuda").manual_seed(K)
                A = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
                B = torch.randn((N, K), device="cuda", generator=g).to(torch.bfloat16)
                bias = torch.randn((N,), device="cuda", generator=g)
                o = torch.empty((M, N), device="cuda", dtype=torch.float32 if out == "f32" else torch.bfloat16)
                kw = {"out_f32": o} if out == "f32" else {"out_bf16": o}
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(5):
                    ctx.op_gemm(A, B, bias, tile_cfg=cfg, **kw)
                e0.record()
                for _ in range(40):
                    ctx.op_gemm(A, B, bias, tile_cfg=cfg, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts[K] = e0.elapsed_time(e1) / 40 * 1e3
            slope = (ts[16384] - ts[4096]) / (12288 / 64)
            print(f"N={N} cfg {cfg} out {out}: " + "  ".join(f"K={k}: {t:6.1f}" for k, t in ts.items()) +
                  f"  us | per K-tile {slope:.3f} us, intercept at K=4096 {ts[4096] - 64 * slope:.1f} us", flush=True)
