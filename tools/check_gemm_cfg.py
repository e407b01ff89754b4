"""Bit-exact check of one GEMM tile configuration against another on small-integer data (every product and partial sum is exact in
f32, so any summation order must give identical results), then a timing of both. Usage: python tools/check_gemm_cfg.py 71 [21]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
ctx = ltx.Context(0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 71
ref_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 21


def bf16(x):
    return torch.from_numpy(x.astype(np.float32)).cuda().to(torch.bfloat16)


for (M, N, K) in [(192, 256, 256), (384, 512, 1024), (1536, 4096, 4096), (1536, 8192, 4096), (1536, 16384, 4096), (1536, 4096, 16384)]:
    rng = np.random.default_rng(M + N + K)
    A = bf16(rng.integers(-4, 5, (M, K)))
    B = bf16(rng.integers(-4, 5, (N, K)))
    bias = torch.from_numpy(rng.integers(-8, 9, (N,)).astype(np.float32)).cuda()
    o1 = torch.empty((M, N), device="cuda", dtype=torch.float32)
    o2 = torch.empty((M, N), device="cuda", dtype=torch.float32)
    ctx.op_gemm(A, B, bias, tile_cfg=ref_cfg, out_f32=o1)
    ctx.op_gemm(A, B, bias, tile_cfg=cfg, out_f32=o2)
    torch.cuda.synchronize()
    same = torch.equal(o1, o2)
    nbad = int((o1 != o2).sum().item())
    print(f"{M}x{N}x{K}: cfg {cfg} vs {ref_cfg} bit-identical: {same} (mismatching elements: {nbad})", flush=True)
    if not same:
        idx = (o1 != o2).nonzero()[:5].tolist()
        print("  first mismatches (row, col):", idx, [(o1[i][j].item(), o2[i][j].item()) for i, j in idx])
        sys.exit(1)
    times = {}
    for c in (ref_cfg, cfg, 1):
        if c in (71, 72, 73, 74) and (N % (256 if c == 71 else 128) or M % 192):
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ctx.op_gemm(A, B, bias, tile_cfg=c, out_f32=o2)
        e0.record()
        for _ in range(10):
            ctx.op_gemm(A, B, bias, tile_cfg=c, out_f32=o2)
        e1.record()
        torch.cuda.synchronize()
        times[c] = e0.elapsed_time(e1) / 10
    print("   " + "  ".join(f"cfg{c}: {t * 1e3:7.1f} us {2.0 * M * N * K / t / 1e9:6.0f} TF/s" for c, t in times.items()), flush=True)
print("ok")
