#!/usr/bin/env python3
"""A few launches of chosen GEMM tile configurations for a rocprofv3 --pmc pass (counters are read per dispatch).
Usage: rocprofv3 --pmc <counters> --kernel-trace -d out -- python3 tools/pmc_gemm.py --cfgs 21,42"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="21,1,42")
    ap.add_argument("--shapes", default="1536x8192x4096,4096x4096x4096")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    ctx = ltx.Context(0)
    for sh in args.shapes.split(","):
        M, N, K = [int(v) for v in sh.split("x")]
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for c in [int(v) for v in args.cfgs.split(",")]:
            for _ in range(args.reps):
                ctx.op_gemm(A, B, None, tile_cfg=c, out_bf16=out)
        torch.cuda.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
