#!/usr/bin/env python3
"""How fast is the host's f32 GEMM at the oracle's shapes, and with how many BLAS threads? (`cpu_baseline` in bench.py times the numpy
oracle; its rate depends on this.) numpy's OpenBLAS at 1536 x 4096 x 4096 and 1536 x 16384 x 4096, thread counts via threadpoolctl;
scipy's sgemm (the oracle's conv path) on the 128-channel conv tap shape. Usage: python tools/host_blas_probe.py"""
import os
import time

import numpy as np
from threadpoolctl import threadpool_info, threadpool_limits

print("host cores:", len(os.sched_getaffinity(0)))
for p in threadpool_info():
    print({k: p.get(k) for k in ("user_api", "internal_api", "version", "num_threads", "threading_layer", "architecture")})
rng = np.random.default_rng(0)


def rate(fn, flops, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return flops * reps / (time.perf_counter() - t0) / 1e12


for (M, N, K) in ((1536, 4096, 4096), (1536, 16384, 4096), (1536, 4096, 16384), (128, 4096, 4096)):
    a = rng.standard_normal((M, K), dtype=np.float32)
    b = rng.standard_normal((N, K), dtype=np.float32)
    line = f"numpy  {M}x{N}x{K}:"
    for nt in (8, 16, 32, 64, 128, 256):
        with threadpool_limits(limits=nt, user_api="blas"):
            line += f"  {nt}t {rate(lambda: a @ b.T, 2.0 * M * N * K):.2f}"
    print(line + "  TFLOP/s", flush=True)
try:
    from scipy.linalg.blas import sgemm

    n, c, o = 25 * 130 * 194, 128, 128
    rows = rng.standard_normal((n, c), dtype=np.float32)
    w = rng.standard_normal((c, o), dtype=np.float32)
    acc = np.zeros((n, o), np.float32)
    line = f"scipy sgemm (conv tap, {n} x {c} x {o}, in place):"
    for nt in (8, 16, 32, 64, 128):
        with threadpool_limits(limits=nt, user_api="blas"):
            line += f"  {nt}t {rate(lambda: sgemm(1.0, w.T, rows.T, beta=1.0, c=acc.T, overwrite_c=1), 2.0 * n * c * o):.2f}"
    print(line + "  TFLOP/s", flush=True)
except Exception as e:  # noqa: BLE001
    print("scipy:", e)
try:
    import torch

    a = torch.randn(1536, 4096)
    b = torch.randn(4096, 4096)
    line = f"torch  1536x4096x4096 ({torch.__config__.parallel_info().splitlines()[0]}):"
    for nt in (16, 32, 64, 128):
        torch.set_num_threads(nt)
        line += f"  {nt}t {rate(lambda: a @ b.T, 2.0 * 1536 * 4096 * 4096):.2f}"
    print(line + "  TFLOP/s", flush=True)
except Exception as e:  # noqa: BLE001
    print("torch:", e)
