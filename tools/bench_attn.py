#!/usr/bin/env python3
"""Attention kernel micro-benchmark on the DiT shapes (self: 1536x1536, cross: 1536x1024 with key mask), random data.
With --once it launches each shape a few times only (for a rocprofv3 --pmc pass).
Usage: python tools/bench_attn.py [--rounds 5] [--once]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")

SHAPES = [("self  T=1536 S=1536", 1, 1536, 1536, False), ("cross T=1536 S=1024", 1, 1536, 1024, False),
          ("cross+mask S=1024", 1, 1536, 1024, True), ("self B=2 T=1536", 2, 1536, 1536, False),
          ("self T=6144 (hi-res)", 1, 6144, 6144, False), ("self T=9984 (201 fr)", 1, 9984, 9984, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--once", action="store_true")
    ap.add_argument("--impls", default="1,2,4,5,0")
    args = ap.parse_args()
    ctx = ltx.Context(0)
    H = 32
    for name, B, T, S, masked in SHAPES:
        Q = (torch.randn(B, T, H * 128, device="cuda") * 0.1275).to(torch.bfloat16)  # (1/sqrt(128)) * log2(e) folded in
        K = torch.randn(B, S, H * 128, device="cuda").to(torch.bfloat16)
        Sp = (S + 63) // 64 * 64
        Vt = torch.randn(B, H * 128, Sp, device="cuda").to(torch.bfloat16)
        bias = None
        if masked:
            bias = torch.zeros(B, S, device="cuda")
            bias[:, S - 100:] = -10000.0
        O = torch.empty(B, T, H * 128, device="cuda", dtype=torch.bfloat16)
        fl = 4.0 * B * H * T * S * 128
        n = 3 if args.once else 20
        line = f"{name:24s}"
        impls = args.impls.split(",")  # 1 = 4-wave, 2 = ping-pong, 4 = 48-query 16x16 assembly, 5 = 32x32 assembly, 0 = the launcher's choice
        best = {i: [] for i in impls}
        bad = set()
        for r in range(1 if args.once else args.rounds + 1):
            for impl in impls:  # interleaved rounds in one process (A/B)
                if impl in bad:
                    continue
                ltx.set_option("attn_impl", int(impl))
                try:
                    ctx.op_attention(Q, K, Vt, bias, H, O, 0.0)  # prescaled Q, as the DiT launches it
                except Exception:
                    bad.add(impl)  # this kernel does not take the shape (e.g. masked launches on the 48-query kernels)
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    ctx.op_attention(Q, K, Vt, bias, H, O, 0.0)  # prescaled Q, as the DiT launches it
                e1.record()
                torch.cuda.synchronize()
                if r > 0 or args.once:
                    best[impl].append(e0.elapsed_time(e1) / n)
        for impl in impls:
            if impl in bad:
                line += f" | impl{impl}:     n/a"
                continue
            v = sorted(best[impl])
            med = v[len(v) // 2]
            line += f" | impl{impl}: {med * 1e3:7.1f} us {fl / med / 1e9:6.0f} TF/s util {fl / med / 1e9 / 2500:.3f}"
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
