#!/usr/bin/env python3
"""Text-embedding connector at the reference architecture (dim 3840, 30 heads, 2 blocks, 128 registers, 49 hidden
states, T = 1024 tokens): one encodeFromHiddenStates call, synthetic weights / hidden states resident in HBM.
Algorithmic work: feature extractor 2*T*188160*3840 = 1.48 TFLOP + 2 blocks x (8 T D^2 + 4 T^2 D + 16 T D^2) = 0.76 TFLOP;
bytes: 49 hidden states 385 MB read + 385 MB concat written and read + 1.44 GB projection weight.
Usage: python tools/bench_connector.py [--iters 5]"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--T", type=int, default=1024)
ap.add_argument("--valid", type=int, default=37, help="number of real prompt tokens (left padding)")
a = ap.parse_args()
ctx = ltx.Context(0)
cfg = ltx.connector_config()
ctx.connector_init_synthetic(cfg, seed=91)
T, D, S = a.T, cfg.dim, cfg.states
hidden = torch.empty((S, 1, T, D), dtype=torch.bfloat16, device="cuda")
ctx.op_fill_normal_bf16(hidden, seed=5, std=3.0)
mask = torch.zeros((1, T), dtype=torch.int32, device="cuda")
mask[:, T - a.valid:] = 1
out = torch.empty((1, T, D), dtype=torch.bfloat16, device="cuda")
om = torch.empty((1, T), dtype=torch.int32, device="cuda")
ctx.connector_encode_dev(hidden, mask, out, om)
torch.cuda.synchronize()
assert bool(om.all()) and bool(torch.isfinite(out.float()).all())
t0 = time.perf_counter()
for _ in range(a.iters):
    ctx.connector_encode_dev(hidden, mask, out, om)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / a.iters
fl = 2.0 * T * D * S * D + cfg.layers * (8.0 * T * D * D + 4.0 * T * T * D + 16.0 * T * D * D)
by = 2.0 * S * T * D * 2 + S * T * D * 2 + D * S * D * 2.0
print(f"connector encode T={T}: {ms:.3f} ms  ({fl / ms / 1e9:.0f} TFLOP/s model rate, {by / ms / 1e6:.0f} GB/s algorithmic)")
ctx.close()
