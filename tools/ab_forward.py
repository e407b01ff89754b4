#!/usr/bin/env python3
"""Same-process, interleaved A/B of one library option (ltx_ctx_set_option) on the 48-layer DiT forward at a BASELINE shape:
    python tools/ab_forward.py gemm_stagger=0,1 [--frames 4 --height 16 --width 24] [--rounds 6] [--iters 5]
Prints the per-round forward times of each setting, their medians, and whether the velocities are bit-equal."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ab")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--height", type=int, default=16)
    ap.add_argument("--width", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", type=int, default=48)
    a = ap.parse_args()
    key, vals = a.ab.split("=")
    vals = [int(v) for v in vals.split(",")]
    ctx = ltx.Context(0)
    ctx.dit_init_synthetic(ltx.default_transformer_config(num_layers=a.layers) if a.layers != 48 else None, seed=1234)
    F, H, W, S = a.frames, a.height, a.width, 1024
    T = F * H * W
    lat = torch.empty((1, T, 128), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(lat, seed=3)
    c = torch.empty((1, S, 3840), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(c, seed=4)
    ts = torch.full((1,), 0.7, dtype=torch.float32, device="cuda")
    vel = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
    outs, times = {}, {v: [] for v in vals}
    for v in vals:
        ctx.set_option(key, v)
        for _ in range(2):
            ctx.dit_forward_dev(lat, c, ts, None, F, H, W, vel, ctx_version=5, mask_all_ones=True)
        torch.cuda.synchronize()
        outs[v] = vel.clone()
    for _ in range(a.rounds):
        for v in vals:
            ctx.set_option(key, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ctx.dit_forward_dev(lat, c, ts, None, F, H, W, vel, ctx_version=5, mask_all_ones=True)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / a.iters)
    same = all(torch.equal(outs[vals[0]], outs[v]) for v in vals[1:])
    print(f"DiT forward T={T}, {a.layers} layers, option {key}: " +
          "   ".join(f"{v}: " + " ".join(f"{t:.3f}" for t in times[v]) + f" (median {sorted(times[v])[len(times[v]) // 2]:.3f} ms)" for v in vals) +
          f"   outputs bit-equal: {same}")
    ctx.close()


if __name__ == "__main__":
    main()
