#!/usr/bin/env python3
"""Micro-benchmark of the GEMM tile configurations on the DiT / VAE shapes (interleaved rounds in one process,
random data - guide rule 24/25). Usage: python tools/bench_gemm.py [--rounds 5]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")

SHAPES = [  # (name, M, N, K)
    ("qk   1536x8192x4096", 1536, 8192, 4096),
    ("vT   4096x1536x4096", 4096, 1536, 4096),
    ("o/q  1536x4096x4096", 1536, 4096, 4096),
    ("ff1  1536x16384x4096", 1536, 16384, 4096),
    ("ff2  1536x4096x16384", 1536, 4096, 16384),
    ("big  6144x4096x4096", 6144, 4096, 4096),
    ("sq   4096x4096x4096", 4096, 4096, 4096),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cfgs", type=str, default="0,1,2,10,11,12,13,14")
    ap.add_argument("--cold", action="store_true", help="cycle through >512 MB of distinct weight matrices (HBM-cold B operand, as in the DiT)")
    ap.add_argument("--f32out", action="store_true")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")]
    ctx = ltx.Context(0)
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        nb = max(1, (640 * 2 ** 20) // (N * K * 2)) if args.cold else 1
        Bs = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(nb)]
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if args.f32out else torch.bfloat16)
        out_lib = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        it = 0
        best = {}
        bad = set()
        for c in cfgs:  # a tile configuration may refuse a shape (e.g. full-tile-only kernels)
            if c >= 0:
                try:
                    ctx.op_gemm(A, Bs[0], None, tile_cfg=c, **({"out_f32": out} if args.f32out else {"out_bf16": out}))
                except Exception:
                    bad.add(c)
        for r in range(args.rounds + 1):
            for c in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    B = Bs[it % nb]
                    it += 1
                    if c in bad:
                        continue
                    if c == -1:  # library GEMM (hipBLASLt through torch) - calibration only, never on the product path
                        torch.matmul(A, B.t(), out=out_lib)
                    elif c == -2:  # automatic tile choice
                        ctx.op_gemm(A, B, None, tile_cfg=-1, **({"out_f32": out} if args.f32out else {"out_bf16": out}))
                    elif args.f32out:
                        ctx.op_gemm(A, B, None, tile_cfg=c, out_f32=out)
                    else:
                        ctx.op_gemm(A, B, None, tile_cfg=c, out_bf16=out)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    ms = e0.elapsed_time(e1) / 5
                    best.setdefault(c, []).append(ms)
        fl = 2.0 * M * N * K
        line = f"{name:24s}"
        for c in cfgs:
            if c in bad:
                line += f" | c{c}:    n/a"
                continue
            v = sorted(best[c])
            med = v[len(v) // 2]
            line += f" | c{c}: {fl / med / 1e9:6.0f}"
        print(line + "  (TFLOP/s, median)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
