#!/usr/bin/env python3
"""Where one oracle forward spends its host time (cProfile, top entries by own time): full width, a few layers, the headline shape.
The GPU suite's longest tests are bounded by this host code, not by the GPU. Usage: python tools/oracle_profile.py [layers] [T-frames]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ltx_oracle as oracle  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    H, W, S = 16, 24, 1024
    T = F * H * W
    cfg = oracle.DiTConfig(num_layers=layers)
    t0 = time.time()
    w = oracle.synth_dit_weights(cfg, seed=1)
    print(f"weights for {layers} layers: {time.time() - t0:.1f} s", flush=True)
    rng = np.random.default_rng(0)
    lat = oracle.bf16_round(rng.standard_normal((1, T, 128)).astype(np.float32))
    cx = oracle.bf16_round(rng.standard_normal((1, S, 3840)).astype(np.float32))
    ts = np.array([0.8], np.float32)
    oracle.dit_forward(w, cfg, lat, cx, ts, None, F, H, W)  # warm the pools
    t0 = time.time()
    pr = cProfile.Profile()
    pr.enable()
    oracle.dit_forward(w, cfg, lat, cx, ts, None, F, H, W)
    pr.disable()
    el = time.time() - t0
    print(f"forward, {layers} layers, T={T}, S={S}: {el:.2f} s -> {el / layers * 48:.1f} s per 48-layer step", flush=True)
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
