#!/usr/bin/env python3
"""Few-row GEMM (128 tokens) against the leading dimensions of its operands: does a row stride that is a power of two (every row of a
K-tile in the same L2 / HBM channel set) bound the launch? Weights rotate over enough copies to stay HBM-cold.
Usage: python tools/fewrow_stride.py [--rows 128]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
import ctypes  # noqa: E402

ltx._ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # strided views on purpose (the binding insists on contiguous)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cfg", type=int, default=29)
    ap.add_argument("--splits", default="4,2,1,4", help="K splits of the four shapes")
    args = ap.parse_args()
    ctx = ltx.Context(0)
    M = args.rows
    sp = [int(x) for x in args.splits.split(",")]
    for (N, K), cfg in zip(((4096, 4096), (8192, 4096), (16384, 4096), (4096, 16384)), [(x if x > 1 else 0) * 100 + args.cfg for x in sp]):
        copies = max(2, int(600e6 // (N * K * 2)))
        for pa, pb in ((0, 0), (64, 64)) if args.rows == 128 else ((0, 0),):
            A = torch.randn(M, K + pa, device="cuda").to(torch.bfloat16)[:, :K]
            Ws = [torch.randn(N, K + pb, device="cuda").to(torch.bfloat16)[:, :K] for _ in range(copies)]
            out = torch.empty(M, N, device="cuda", dtype=torch.float32)
            for w in Ws:
                ctx.op_gemm(A, w, tile_cfg=cfg, out_f32=out)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for r in range(3):
                    for w in Ws:
                        ctx.op_gemm(A, w, tile_cfg=cfg, out_f32=out)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / (3 * copies))
            print(f"M={M} N={N:5d} K={K:5d} cfg {cfg}: lda=K+{pa:3d} ldb=K+{pb:3d}  {best:7.2f} us  weights {N * K * 2 / best / 1e6:6.2f} TB/s", flush=True)
            del Ws


if __name__ == "__main__":
    main()
