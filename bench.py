#!/usr/bin/env python3
"""bench.py - DiT denoise steps/sec (+ VAE decode ms) at 768x512x25 distilled on MI355X, one JSON line.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" is one denoise step of the hot path on one sample: patchify+bf16 cast -> 48-block DiT forward -> unpatchify
-> Euler update (reference loop body, LTXPipeline.swift:800-956) with every input already resident in HBM.
Workload = BASELINE.json configs[1]: distilled, 768x512x25 (latent 4x16x24 = 1536 tokens), 1024 text keys, CFG off,
bf16 weights of the full 48-layer architecture (random init, generated on device), synthetic latent/context.
N > 1: the path shards by sample (independent videos / seeds), so every rank runs its own replica of the same
workload with no data-path collective (weak scaling); the only collective is the one-time broadcast of the text
context before the timed region. value = (N * K steps) / max-over-ranks time.

roofline: dominant kernel family = the bf16 MFMA GEMM (gemm_bf16_kernel / gemm_bf16_kernel_v2). achieved = sum of the
GEMM launches' algorithmic FLOPs (2*M*N*K) / sum of their durations over K steps, measured with HIP events recorded
on the launch stream around every GEMM launch (ltx_prof_*) in a second pass of the same K steps right after the timed
region (the event packets cost ~6 % of a step, so `value` comes from the un-instrumented pass). peak = 2500 TFLOP/s
dense bf16. traffic = fabric-side bytes per GEMM launch from the rocprofv3 PMC passes committed under profiles/
(tools/pmc_traffic.py); the algorithmic bytes per launch (A + B + C + residual) average 103 MB, the counters see
~298 MB because every XCD's L2 fetches the whole activation operand once (8 x A).
cpu_baseline: the oracle (numpy restatement of the reference path, kind "port") timed on the host cores for a few
transformer blocks of the same workload and extrapolated to one full step.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, FRAMES, S_TEXT = 768, 512, 25, 1024
PEAK_BF16_TFLOPS = 2500.0
# What the part sustains in a bare v_mfma_f32_16x16x32_bf16 loop on random data (operands in registers, no memory traffic): the clock is held
# down under matrix load. Measured with tools/ubench/mfma_peak.hip (profiles/r01_ubench_mfma_sustained_peak.txt: 1.95-2.07 PFLOP/s).
# Reported beside `frac` for context only; `frac` stays priced against the 2.5 PFLOP/s headline.
SUSTAINED_BF16_TFLOPS = 2000.0


def pmc_traffic():
    """HBM-side bytes per GEMM launch (mean over all GEMM launches of this workload). PMC counters cannot be read from
    inside the process: they come from two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of this same
    command, corrected as MI355X_MICROARCH.md prescribes (KiB units, FETCH_SIZE x2 on gfx950) by tools/pmc_traffic.py and
    committed under profiles/. null when that file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return round(json.load(f)["gemm_all"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def dit_flops_per_step(T, S=1024, D=4096, L=48, B=1):
    """BASELINE.md section 2."""
    per = L * (8 * T * D * D + 4 * T * T * D + 4 * T * D * D + 4 * S * D * D + 4 * T * S * D + 16 * T * D * D)
    per += 2 * T * 128 * D + 2 * S * 3840 * D + 2 * S * D * D + 2 * (256 * D + 7 * D * D) + 2 * T * D * 128
    return B * per


def cpu_baseline(T, S, budget_s=20.0):
    """Oracle (numpy, f32 activations x bf16-rounded weights) on the host cores: time whole transformer blocks at
    the bench workload's shapes, extrapolate to 48 blocks. Bounded to ~budget_s seconds."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ltx_oracle as o

    cfg = o.DiTConfig(num_layers=1)
    D = cfg.dim
    rng = np.random.default_rng(0)
    w = {}
    for k, shp in o.dit_param_shapes(cfg).items():
        if not k.startswith("transformer_blocks.0."):
            continue
        if k.endswith("_norm.weight"):
            w[k] = np.ones(shp, np.float32)
        else:
            w[k] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(0.02))
    x = rng.standard_normal((1, T, D), dtype=np.float32)
    ctx = rng.standard_normal((1, S, D), dtype=np.float32)
    temb = (0.02 * rng.standard_normal((1, 1, 6, D))).astype(np.float32)
    rope = o.rope_tables(4, 16, 24)
    t0 = time.perf_counter()
    nblk = 0
    while True:
        x = o.transformer_block(w, 0, x, ctx, temb, cfg, rope, None)
        nblk += 1
        el = time.perf_counter() - t0
        if el > budget_s or nblk >= 8:
            break
    per_block = el / nblk
    steps_per_s = 1.0 / (per_block * 48)
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info

        nthreads = max([p.get("num_threads", 1) for p in threadpool_info()] or [ncores])
    except Exception:
        nthreads = ncores
    return {"value": steps_per_s, "unit": "steps/s", "cores": int(min(ncores, nthreads)), "kind": "port",
            "sample": f"{nblk} of 48 transformer blocks of one 768x512x25 step (T={T}, S={S}, D=4096) in {el:.1f} s, "
                      f"numpy/BLAS f32 with {nthreads} BLAS threads on {ncores} schedulable host cores, extrapolated x48/{nblk} "
                      f"(head/tail ops excluded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the once-per-prompt legs (text-embedding connector, VAE encoder)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-launch HIP events in the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    ltx = importlib.import_module("ltx-video-swift-mlx_amd")
    F, H, W = ltx.latent_shape(WIDTH, HEIGHT, FRAMES)
    T = F * H * W
    ctx = ltx.Context(local)
    cfg = ltx.default_transformer_config()
    ctx.dit_init_synthetic(cfg, seed=1234)

    dev = torch.device("cuda", local)
    # text context: generated on rank 0 and broadcast once (the path's only exchange; outside the timed region)
    context = torch.empty((1, S_TEXT, cfg.caption_channels), dtype=torch.bfloat16, device=dev)
    if rank == 0:
        ctx.op_fill_normal_bf16(context, seed=43)
    torch.cuda.synchronize()
    if world > 1:
        dist.broadcast(context, src=0)
    mask = torch.ones((1, S_TEXT), dtype=torch.int32, device=dev)
    latent = torch.empty((1, 128, F, H, W), dtype=torch.float32, device=dev)
    ctx.op_fill_normal_f32(latent, seed=42 + rank)  # independent sample per rank
    sig = ltx.sigmas(True, 8, T)

    def step(i):
        j = i % 8
        ctx.denoise_dev(latent, sig[j:j + 2], context, mask, F, H, W, ctx_version=7, mask_all_ones=True)
        if j == 7:  # schedule finished: start the next sample from fresh noise (keeps values in range)
            ctx.op_fill_normal_f32(latent, seed=1000 + i + rank)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # Roofline leg: the SAME K steps again with a HIP-event pair recorded on the launch stream around every GEMM /
    # attention launch. The event packets themselves cost ~6 % of a step (2 x 434 launches), so they are kept out
    # of the region `value` is computed from; the per-launch durations they yield are unaffected by that overhead.
    roofline = None
    extra = {}
    if not args.no_prof:
        ctx.prof_collect(0, reset=True)
        ctx.prof_enable(True)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        el_prof = time.perf_counter() - tp
        g = ctx.prof_collect(0)
        a = ctx.prof_collect(1)
        ctx.prof_enable(False)
        if g["ms"] > 0:
            ach = g["work"] / (g["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel{,_v2}", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(),
                        "sustained_peak_measured": SUSTAINED_BF16_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_BF16_TFLOPS, 4),
                        "launches": g["launches"], "avg_launch_us": round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                        "gemm_ms_per_step": round(g["ms"] / args.steps, 3),
                        "ms_per_step_with_events": round(1e3 * el_prof / args.steps, 3)}
        if a["ms"] > 0:
            extra["attention"] = {"achieved_tflops": round(a["work"] / (a["ms"] * 1e-3) / 1e12, 1),
                                  "mfma_util": round(a["work"] / (a["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "ms_per_step": round(a["ms"] / args.steps, 3), "launches": a["launches"]}

    ms_per_step = 1e3 * el / args.steps
    value = world * args.steps / el
    out = {
        "metric": "DiT denoise steps/sec + VAE decode ms, 768x512x25 distilled",
        "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "distilled 8-step schedule, 768x512x25 -> latent 4x16x24 (1536 tokens), 1024 text keys, "
                               "CFG off, batch 1 per GPU, 48-layer DiT bf16 weights (random init)",
                   "tokens": T, "text_keys": S_TEXT, "parallelism": f"replica x{world} (one sample per GPU)"},
        "dit_tflops_per_step": round(dit_flops_per_step(T) / 1e12, 2),
        "dit_model_tflops_per_s": round(dit_flops_per_step(T) / 1e12 / (el / args.steps), 1),
    }
    if roofline:
        out["roofline"] = roofline
    out.update(extra)

    if rank == 0 and not args.no_vae and hasattr(ctx, "vae_init_synthetic"):
        try:
            out["vae"] = bench_vae(ctx, ltx, torch, dev, F, H, W)
        except Exception as e:  # the DiT line must still be reported
            out["vae"] = {"error": str(e)}
    if rank == 0 and not args.no_aux:
        try:
            out["pre_loop"] = bench_pre_loop(ctx, ltx, torch, dev)
        except Exception as e:
            out["pre_loop"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T, S_TEXT)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def bench_pre_loop(ctx, ltx, torch, dev, iters=3):
    """The once-per-prompt steps in front of the loop, reference architecture, synthetic weights:
    text-embedding connector on 49 x [1,1024,3840] hidden states, and the VAE encoder on one 768x512 image."""
    res = {}
    cfg = ltx.connector_config()
    ctx.connector_init_synthetic(cfg, seed=91)
    T = 1024
    hidden = torch.empty((cfg.states, 1, T, cfg.dim), dtype=torch.bfloat16, device=dev)
    ctx.op_fill_normal_bf16(hidden, seed=5, std=3.0)
    mask = torch.zeros((1, T), dtype=torch.int32, device=dev)
    mask[:, T - 37:] = 1
    outc = torch.empty((1, T, cfg.dim), dtype=torch.bfloat16, device=dev)
    ctx.connector_encode_dev(hidden, mask, outc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ctx.connector_encode_dev(hidden, mask, outc)
    torch.cuda.synchronize()
    res["connector_ms"] = round(1e3 * (time.perf_counter() - t0) / iters, 3)
    ctx.connector_unload()
    del hidden, outc
    ctx.vae_encoder_init_synthetic(0, seed=66)
    px = torch.rand((1, 3, 1, HEIGHT, WIDTH), dtype=torch.float32, device=dev) * 2 - 1
    lat = torch.empty((1, 128, 1, HEIGHT // 32, WIDTH // 32), dtype=torch.float32, device=dev)
    ctx.vae_encode_dev(px, lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ctx.vae_encode_dev(px, lat)
    torch.cuda.synchronize()
    res["vae_encode_ms"] = round(1e3 * (time.perf_counter() - t0) / iters, 3)
    ctx.vae_encoder_unload()
    return res


def bench_vae(ctx, ltx, torch, dev, F, H, W, iters=3):
    """VAE decode of one [1,128,4,16,24] latent -> (25,512,768,3), device-resident, ms + algorithmic GB/s."""
    ctx.vae_init_synthetic(seed=77)
    lat = torch.empty((1, 128, F, H, W), dtype=torch.float32, device=dev)
    ctx.op_fill_normal_f32(lat, seed=45)
    nf = 8 * (F - 1) + 1
    frames = torch.empty((nf, H * 32, W * 32, 3), dtype=torch.float32, device=dev)
    ctx.vae_decode_dev(lat, F, H, W, frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ctx.vae_decode_dev(lat, F, H, W, frames)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters  # un-instrumented
    ctx.prof_collect(2, reset=True)
    ctx.prof_enable(True)  # second pass with HIP events around the conv launches (kernel rate only)
    for _ in range(iters):
        ctx.vae_decode_dev(lat, F, H, W, frames)
    torch.cuda.synchronize()
    c = ctx.prof_collect(2)
    ctx.prof_enable(False)
    alg_bytes = 5.80e9  # BASELINE.md section 2 @ 4x16x24
    res = {"decode_ms": round(ms, 3), "algorithmic_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
           "tflops": round(12.96 / (ms * 1e-3), 1)}
    if c["ms"] > 0:
        res["conv_kernel_tflops"] = round(c["work"] / (c["ms"] * 1e-3) / 1e12, 1)
        res["conv_ms"] = round(c["ms"] / iters, 3)
    return res


if __name__ == "__main__":
    main()
