#!/usr/bin/env python3
"""What the part draws and clocks while the hot path runs: board power, power cap and shader clock sampled from the driver's sysfs nodes
(fallback: `rocm-smi --json`) every ~20 ms while one workload at a time loops on the GPU for a few seconds:

  idle | bare MFMA-heavy GEMM (1536x16384x4096) | N = 4096 GEMM | self-attention T = 1536 | adaLN row pass | whole DiT step | VAE decode

The question it answers ("the part is power-limited under this load"): is the step's clock set by the power cap? If
power sits at the cap while the clock is far below its maximum, removing stall cycles returns clock, not time, and only removing
energy (bytes moved, instructions issued) pays.
Usage: python tools/power_probe.py [--seconds 4]      (prints one table; commit it under profiles/)"""
import argparse
import glob
import importlib
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ltx = importlib.import_module("ltx-video-swift-mlx_amd")


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sampler:
    def __init__(self):
        self.hw = None
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if _read(os.path.join(d, "power1_average")) or _read(os.path.join(d, "power1_input")):
                self.hw = d
                break
        self.dev = os.path.dirname(os.path.dirname(self.hw)) if self.hw else None
        self.mode = "sysfs" if self.hw else "rocm-smi"
        self.samples, self._stop, self._t = [], False, None

    def caps(self):
        if self.hw:
            cap = _read(os.path.join(self.hw, "power1_cap"))
            cmax = _read(os.path.join(self.hw, "power1_cap_max"))
            sclk = _read(os.path.join(self.dev, "pp_dpm_sclk"))
            return {"power_cap_W": int(cap) / 1e6 if cap else None, "power_cap_max_W": int(cmax) / 1e6 if cmax else None,
                    "pp_dpm_sclk": sclk.replace("\n", " | ") if sclk else None}
        try:
            out = subprocess.run(["rocm-smi", "--showmaxpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
            return {"rocm_smi": json.loads(out)}
        except Exception as e:  # noqa: BLE001
            return {"error": repr(e)}

    def one(self):
        if self.hw:
            p = _read(os.path.join(self.hw, "power1_average")) or _read(os.path.join(self.hw, "power1_input"))
            f = _read(os.path.join(self.hw, "freq1_input"))
            return (int(p) / 1e6 if p else None, int(f) / 1e6 if f else None)
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
            d = list(json.loads(out).values())[0]
            p = next((float(v) for k, v in d.items() if "ower" in k and "W" in k), None)
            f = next((float(str(v).strip("()Mhz")) for k, v in d.items() if k.startswith("sclk")), None)
            return (p, f)
        except Exception:  # noqa: BLE001
            return (None, None)

    def start(self):
        self.samples, self._stop = [], False

        def run():
            while not self._stop:
                self.samples.append(self.one())
                time.sleep(0.02 if self.hw else 0.2)

        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def stop(self):
        self._stop = True
        self._t.join()
        ps = [p for p, _ in self.samples if p is not None]
        fs = [f for _, f in self.samples if f is not None]
        # drop the ramp: statistics over the second half of the window
        ps, fs = ps[len(ps) // 2:], fs[len(fs) // 2:]
        avg = lambda v: sum(v) / len(v) if v else float("nan")
        return {"n": len(self.samples), "power_W": avg(ps), "power_max_W": max(ps) if ps else float("nan"),
                "sclk_MHz": avg(fs), "sclk_min_MHz": min(fs) if fs else float("nan")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    args = ap.parse_args()
    ctx = ltx.Context(0)
    smp = Sampler()
    print("sampler:", smp.mode, json.dumps(smp.caps()), flush=True)
    dev = "cuda"
    D, T, S, H = 4096, 1536, 1024, 32

    def rnd(*shape, std=1.0):
        t = torch.empty(shape, dtype=torch.bfloat16, device=dev)
        ctx.op_fill_normal_bf16(t, seed=sum(shape) % 1000, std=std)
        return t

    A = rnd(T, D)
    Wup, Wsq = rnd(4 * D, D, std=0.02), rnd(D, D, std=0.02)
    out_up = torch.empty((T, 4 * D), dtype=torch.bfloat16, device=dev)
    out_sq = torch.empty((T, D), dtype=torch.float32, device=dev)
    Q, K, Vt = rnd(1, T, D, std=0.1275), rnd(1, T, D), rnd(1, D, T)
    O = torch.empty((1, T, D), dtype=torch.bfloat16, device=dev)
    xf = torch.randn(T, D, device=dev)
    sc, sh = torch.randn(1, D, device=dev) * 0.02, torch.randn(1, D, device=dev) * 0.02
    xn = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    cfg = ltx.default_transformer_config()
    ctx.dit_init_synthetic(cfg, seed=1234)
    ctx.vae_init_synthetic(seed=77)
    lat = rnd(1, T, 128)
    cx = rnd(1, S, 3840)
    ts = torch.full((1,), 0.7, dtype=torch.float32, device=dev)
    vel = torch.empty((1, T, 128), dtype=torch.float32, device=dev)
    vlat = torch.randn(1, 128, 4, 16, 24, device=dev)
    frames = torch.empty((25, 512, 768, 3), dtype=torch.float32, device=dev)

    work = [
        ("idle", None, 0),
        ("GEMM 1536x16384x4096 (FFN-up shape, 192x256 kernel)", lambda: ctx.op_gemm(A, Wup, out_bf16=out_up), 2.0 * T * 4 * D * D),
        ("GEMM 1536x4096x4096 (192x128 ring kernel)", lambda: ctx.op_gemm(A, Wsq, out_f32=out_sq), 2.0 * T * D * D),
        ("self-attention T=1536, 32 heads", lambda: ctx.op_attention(Q, K, Vt, None, H, O, 0.0), 4.0 * H * T * T * 128),
        ("adaLN row pass 1536x4096", lambda: ctx.op_norm_mod(xf, sc, sh, xn), 0),
        ("whole DiT forward (48 blocks, T=1536, S=1024)", lambda: ctx.dit_forward_dev(lat, cx, ts, None, 4, 16, 24, vel, ctx_version=5, mask_all_ones=True), 37.7e12),
        ("VAE decode 4x16x24 -> 25x512x768", lambda: ctx.vae_decode_dev(vlat, 4, 16, 24, frames), 12.96e12),
    ]
    print(f"{'workload':58s} {'W avg':>7s} {'W max':>7s} {'sclk avg':>9s} {'sclk min':>9s} {'us/launch':>10s} {'TFLOP/s':>8s}")
    for name, fn, flops in work:
        if fn is None:
            torch.cuda.synchronize()
            smp.start()
            time.sleep(min(args.seconds, 2.0))
            r = smp.stop()
            per = float("nan")
        else:
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            est = (time.perf_counter() - t0) / 3
            n = max(3, int(args.seconds / max(est, 1e-6)))
            smp.start()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fn()
                if i % 64 == 63:
                    torch.cuda.synchronize()  # keep the queue short: the sampler thread needs the GIL now and then
            e1.record()
            torch.cuda.synchronize()
            r = smp.stop()
            per = e0.elapsed_time(e1) / n * 1e3
        tf = flops / per / 1e6 if flops and per == per else float("nan")
        print(f"{name:58s} {r['power_W']:7.0f} {r['power_max_W']:7.0f} {r['sclk_MHz']:9.0f} {r['sclk_min_MHz']:9.0f} {per:10.1f} {tf:8.0f}   ({r['n']} samples)",
              flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
