#!/usr/bin/env python3
"""Prints the SHA-256 of one DiT forward's velocity (full 48-layer architecture, synthetic weights, fixed seeds). Run by GPU tests in
a process of its own per setting of an environment hook the library reads once (LTX_FUSE_FINISH, ...).
Usage: python tests/helpers/forward_hash.py F H W [layers]"""
import hashlib
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")


def main():
    F, H, W = (int(v) for v in sys.argv[1:4])
    layers = int(sys.argv[4]) if len(sys.argv) > 4 else 48
    ctx = ltx.Context(0)
    ctx.dit_init_synthetic(ltx.default_transformer_config(num_layers=layers) if layers != 48 else None, seed=1234)
    T, S = F * H * W, 1024
    lat = torch.empty((1, T, 128), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(lat, seed=3)
    c = torch.empty((1, S, 3840), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(c, seed=4)
    ts = torch.full((1,), 0.7, dtype=torch.float32, device="cuda")
    vel = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
    ctx.dit_forward_dev(lat, c, ts, None, F, H, W, vel, ctx_version=5, mask_all_ones=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(vel).all())
    print("HASH", hashlib.sha256(vel.cpu().numpy().tobytes()).hexdigest(), flush=True)


if __name__ == "__main__":
    main()
