#!/usr/bin/env python3
"""Row-kernel micro-benchmark on the DiT step's shapes: norm + modulation (f32 [T][4096] -> bf16) and q / k norm + RoPE, back to back in a
chain of buffers larger than the L2s (the step's operands come from the previous GEMM's epilogue: Infinity-Cache-warm at best).
Usage: python tools/bench_rows.py [T] [rows]      (rows = 2 | 4: option "norm_rows", the rows per workgroup of norm_mod_rows_kernel; default: the launcher's choice)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    D = 4096
    ctx = ltx.Context(0)
    if len(sys.argv) > 2:
        ctx.set_option("norm_rows", int(sys.argv[2]))
    n = 8
    xs = [torch.randn(T, D, device="cuda") for _ in range(n)]
    outs = [torch.empty(T, D, device="cuda", dtype=torch.bfloat16) for _ in range(n)]
    scale, shift = torch.randn(D, device="cuda") * 0.1, torch.randn(D, device="cuda") * 0.1
    for i in range(n):
        ctx.op_norm_mod(xs[i], scale, shift, outs[i], norm_kind=0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(6):
            for i in range(n):
                ctx.op_norm_mod(xs[i], scale, shift, outs[i], norm_kind=0)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (6 * n))
    print(f"norm_mod T={T} rows/wg option norm_rows={ltx.get_option('norm_rows')} (0 = the launcher's choice): {best:.2f} us per launch, {T * D * 6 / best / 1e6:.2f} TB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
