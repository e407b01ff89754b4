#!/usr/bin/env python3
"""bench.py - DiT denoise steps/sec (+ VAE decode ms) at 768x512x25 distilled on MI355X, one JSON line.

  python bench.py --gpus N --steps K --warmup W [--mode replica|cfg-pair|sp|vae-tiles]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher in the environment starts the N ranks itself (torch.distributed.run as a child
process, before this process touches the GPU) and relays rank 0's JSON line; a WORLD_SIZE that disagrees with --gpus is an error,
and `n_ranks_seen` in the line is an RCCL all-reduce of ones over the ranks that really ran.

A "step" is one denoise step of the hot path on one sample: patchify + bf16 cast -> 48-block DiT forward -> unpatchify -> Euler
update (reference loop body, LTXPipeline.swift:800-956), every input already resident in HBM.

Modes (what N GPUs do; the path shards at the granularity of forwards / samples / tiles):
  replica    (default; BASELINE configs[1]) distilled, 768x512x25 = 1536 tokens, 1024 text keys, CFG off. Every rank runs its own
             sample: no data-path collective, weak scaling, value = N*K steps / max-over-ranks time; `per_rank_ms_per_step`
             lists every rank's own time so that a slow box is visible. With N >= 2 and --extra-legs the line also carries two
             short legs measured after the timed region - `cfg_pair` and `sp` below; they are opt-in because they call
             collectives: under a watchdog, a hang or an error there still prints the line (`extra_legs_status`) and then ends
             every rank with exit status 3 - a run whose collectives did not return is never reported as a success.
  cfg-pair   (configs[2]) dev schedule, CFG 4.0: ranks (0,1), (2,3), ... each form a pair; rank 2p evaluates the negative branch,
             rank 2p+1 the positive one, ONE RCCL all-gather of the 786 KB velocities per step inside ltx_denoise_dev
             (LTX_SHARD_CFG). N = 1 runs the batched B = 2 forward on one GPU (the baseline a pair must beat). value = pairs*K
             steps / time; weak scaling over pairs.
  sp         (configs[4] shape) ONE sample of 768x512x201 = 9984 tokens split by tokens over all N ranks (LTX_SHARD_SEQUENCE):
             strong scaling, value = K steps / time.
  vae-tiles  (configs[4]) 26 latent frames, tile 8 / overlap 1: the four tiles decoded round-robin by the ranks, raw tiles
             broadcast, blended on every rank; value = decodes / s.

roofline: dominant kernel family = the bf16 MFMA GEMM. achieved = sum of the GEMM launches' algorithmic FLOPs (2*M*N*K) / sum of
their durations over K steps, measured with HIP events recorded on the launch stream around every GEMM launch (ltx_prof_*) in a
second pass of the same K steps right after the timed region (the event packets cost ~6 % of a step, so `value` comes from the
un-instrumented pass). peak = 2500 TFLOP/s dense bf16. traffic = fabric-side bytes per GEMM launch from the rocprofv3 PMC passes
committed under profiles/ (tools/pmc_traffic.py; `traffic_source` names the file - it is not re-measured inside this process).
cpu_baseline: the oracle (numpy restatement of the reference path, kind "port") timed on the host cores for ONE WHOLE denoise step
and ONE WHOLE VAE decode of the same workload, un-extrapolated (`extrapolated` is true only when a probe
predicts that the host would need more than the leg's budget, in which case whole blocks / a small latent are scaled and said so).
"""
import argparse
import copy
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, FRAMES, S_TEXT = 768, 512, 25, 1024
PEAK_BF16_TFLOPS = 2500.0
# What the part sustains in a bare v_mfma_f32_16x16x32_bf16 loop on random data (operands in registers, no memory traffic): the clock is held
# down under matrix load. Measured with tools/ubench/mfma_peak.hip (profiles/r01_ubench_mfma_sustained_peak.txt: 1.95-2.07 PFLOP/s).
# Reported beside `frac` for context only; `frac` stays priced against the 2.5 PFLOP/s headline.
SUSTAINED_BF16_TFLOPS = 2000.0
HBM_PEAK_GBPS = 8000.0
# (file, key): since round 4 one collection carries the dense GEMM launches ("gemm_all") and the VAE conv launches ("conv_all") side by side
TRAFFIC_FILES = (("r05_pmc_traffic.json", "gemm_all"), ("r04_pmc_traffic_v3.json", "gemm_all"), ("r04_pmc_traffic_v2.json", "gemm_all"), ("r04_pmc_traffic.json", "gemm_all"), ("r03_pmc_traffic.json", "gemm_all"), ("r02_pmc_traffic_v2.json", "gemm_all"),
                 ("r02_pmc_traffic.json", "gemm_all"), ("r01_pmc_traffic.json", "gemm_all"))
VAE_TRAFFIC_FILES = (("r05_pmc_traffic.json", "conv_all"), ("r04_pmc_traffic_v3.json", "conv_all"), ("r04_pmc_traffic_v2.json", "conv_all"), ("r04_pmc_traffic.json", "conv_all"), ("r03_pmc_traffic_vae_v2.json", "gemm_all"), ("r03_pmc_traffic_vae.json", "gemm_all"),
                     ("r02_pmc_traffic_vae.json", "gemm_all"))


def pmc_traffic(files=None):
    """HBM-side bytes per GEMM launch (mean over all GEMM launches of this workload) and the file they come from. PMC counters
    cannot be read from inside the process: they come from two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of this
    same command, corrected as MI355X_MICROARCH.md prescribes (KiB units, FETCH_SIZE x2 on gfx950) by tools/pmc_traffic.py."""
    for name, key in (files or TRAFFIC_FILES):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return round(json.load(f)[key]["hbm_bytes_per_launch"]), "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def dit_flops_per_step(T, S=1024, D=4096, L=48, B=1, executed=False):
    """BASELINE.md section 2. executed=True leaves out what the library computes once per prompt instead of once per step (the
    caption projection and the 48 layers' cross-attention K / V projections: output-identical caching)."""
    per = L * (8 * T * D * D + 4 * T * T * D + 4 * T * D * D + 4 * T * S * D + 16 * T * D * D)
    per += 2 * T * 128 * D + 2 * (256 * D + 7 * D * D) + 2 * T * D * 128
    if not executed:
        per += L * 4 * S * D * D + 2 * S * 3840 * D + 2 * S * D * D
    return B * per


def vae_flops(F, H, W):
    """2 * 27 * Cin * Cout * positions over the decoder's 42 convs: 12.96 TFLOP at the 4x16x24 latent."""
    ch = (1024, 512, 256, 128)
    fl = 2 * 27 * 128 * ch[0] * F * H * W
    f, h, w = F, H, W
    for g, c in enumerate(ch):
        fl += 10 * 2 * 27 * c * c * f * h * w
        if g < 3:
            fl += 2 * 27 * c * 4 * c * f * h * w
            f, h, w = 2 * f - 1, 2 * h, 2 * w
    return fl + 2 * 27 * 128 * 48 * f * h * w


def _host_threads():
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info

        nthreads = max([p.get("num_threads", 1) for p in threadpool_info()] or [ncores])
    except Exception:
        nthreads = ncores
    return int(ncores), int(nthreads)


def _fast_normal(rng, shape, scale):
    import numpy as np

    return rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)


def cpu_baseline_vae(F, H, W, budget_s=150.0):
    """The oracle's WHOLE VAE decode (oracle.decode_video: VideoDecoder.swift:358-449, numpy f32, one sgemm per conv tap) of the
    bench's own latent on the host cores: the CPU side of 'VAE decode ms' (BASELINE.md section 3), un-extrapolated. The time does
    not depend on the weight values, so one random tensor per distinct shape serves every layer of that shape. A reported baseline,
    not a target. If a small probe predicts more than `budget_s`, the probe is scaled by conv FLOPs instead and the entry says so."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ltx_oracle as o

    rng = np.random.default_rng(0)
    by_shape, w = {}, {}
    for k, shp in o.vae_param_shapes().items():
        if shp not in by_shape:
            by_shape[shp] = _fast_normal(rng, shp, 0.02) if len(shp) else np.float32(1.0)
        w[k] = by_shape[shp]
    w["mean_of_means"], w["std_of_means"] = np.zeros(128, np.float32), np.ones(128, np.float32)
    ncores, nthreads = _host_threads()
    sf, sh, sw = 2, 4, 6
    lat = rng.standard_normal((1, 128, sf, sh, sw), dtype=np.float32)
    t0 = time.perf_counter()
    o.decode_video(w, lat)
    probe = time.perf_counter() - t0
    ratio = vae_flops(F, H, W) / vae_flops(sf, sh, sw)
    if probe * ratio > 4 * budget_s:  # small convs run far below the large ones' BLAS rate: 4x is the margin before giving up
        return {"decode_ms": round(1e3 * probe * ratio, 1), "extrapolated": True, "cores": min(ncores, nthreads),
                "sample": f"oracle.decode_video of a {sf}x{sh}x{sw} latent in {probe:.2f} s scaled x{ratio:.1f} by conv FLOPs (the whole "
                          f"decode was predicted to exceed {4 * budget_s:.0f} s on this host)"}
    lat = rng.standard_normal((1, 128, F, H, W), dtype=np.float32)
    t0 = time.perf_counter()
    frames = o.decode_video(w, lat)
    el = time.perf_counter() - t0
    used = max(o._CONV_BLAS_THREADS, o._HOST_THREADS)
    return {"decode_ms": round(1e3 * el, 1), "extrapolated": False, "cores": min(ncores, used), "host_cores": ncores,
            "blas_threads": o._CONV_BLAS_THREADS, "pool_threads": o._HOST_THREADS, "tflops": round(vae_flops(F, H, W) / el / 1e12, 3),
            "unit": f"ms per whole decode of the {F}x{H}x{W} latent",
            "sample": f"ONE whole oracle.decode_video of the {F}x{H}x{W} latent -> {frames.shape[0]}x{frames.shape[1]}x{frames.shape[2]} "
                      f"frames ({vae_flops(F, H, W) / 1e12:.2f} TFLOP over 42 convs) in {el:.1f} s: numpy f32, one sgemm per conv tap on "
                      f"{o._CONV_BLAS_THREADS} BLAS threads (the fastest setting for that shape on a 256-core host, tools/host_blas_probe.py) + "
                      f"{o._HOST_THREADS} threads for the position-wise passes, {ncores} schedulable host cores"}


class _CycledBlocks(dict):
    """Weights of a 48-layer DiT in which block i aliases block i % n: the oracle's time does not depend on the values, and n
    distinct blocks (1.07 GB of f32 each) are more than the host's caches hold, so every block streams its weights from DRAM as a
    real model's would."""

    def __init__(self, base, n):
        super().__init__(base)
        self.n = n

    def _k(self, key):
        if key.startswith("transformer_blocks."):
            i, rest = key[len("transformer_blocks."):].split(".", 1)
            return f"transformer_blocks.{int(i) % self.n}.{rest}"
        return key

    def __getitem__(self, key):
        return dict.__getitem__(self, self._k(key))

    def __contains__(self, key):
        return dict.__contains__(self, self._k(key))


def cpu_baseline(T, S, F, H, W, budget_s=300.0):
    """Oracle (numpy restatement of the reference path: f32 activations x bf16-valued weights) on the host cores: ONE WHOLE denoise
    step of the bench workload - patchify projection, timestep path, caption projection, all 48 transformer blocks (text K/V
    recomputed, as the reference does every step), output head, Euler update - un-extrapolated. One block
    is timed first; if 48 of them would exceed `budget_s` the entry falls back to as many whole blocks as fit and says so."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ltx_oracle as o

    ncycle = 3
    rng = np.random.default_rng(0)
    shapes = o.dit_param_shapes(o.DiTConfig(num_layers=ncycle))
    w = {}
    for k, shp in shapes.items():
        if k.endswith("_norm.weight"):
            w[k] = np.ones(shp, np.float32)
        else:
            w[k] = _fast_normal(rng, shp, 0.01 if k.endswith(".bias") else 0.02)
    cfg = o.DiTConfig()
    wc = _CycledBlocks(w, ncycle)
    lat = o.bf16_round(rng.standard_normal((1, 128, F, H, W), dtype=np.float32))
    cx = o.bf16_round(rng.standard_normal((1, S, cfg.caption_channels), dtype=np.float32))
    ncores, nthreads = _host_threads()
    # probe: one block
    x = rng.standard_normal((1, T, cfg.dim), dtype=np.float32)
    ctxp = rng.standard_normal((1, S, cfg.dim), dtype=np.float32)
    temb = (0.02 * rng.standard_normal((1, 1, 6, cfg.dim))).astype(np.float32)
    rope = o.rope_tables(F, H, W)
    t0 = time.perf_counter()
    o.transformer_block(wc, 0, x, ctxp, temb, cfg, rope, None)
    per_block = time.perf_counter() - t0
    used = max(o._DIT_BLAS_THREADS, o._HOST_THREADS)
    nthreads = o._DIT_BLAS_THREADS
    common = {"unit": "steps/s", "cores": min(ncores, used), "host_cores": ncores, "blas_threads": o._DIT_BLAS_THREADS,
              "pool_threads": o._HOST_THREADS, "kind": "port"}
    if per_block * 48 > budget_s:
        nblk = max(1, int(budget_s / 4 / per_block))
        t0 = time.perf_counter()
        for i in range(nblk):
            x = o.transformer_block(wc, i, x, ctxp, temb, cfg, rope, None)
        el = time.perf_counter() - t0
        return dict(common, value=1.0 / (el / nblk * 48), extrapolated=True,
                    sample=f"{nblk} of the 48 transformer blocks of one 768x512x25 step in {el:.1f} s, scaled x48/{nblk} (a whole step was "
                           f"predicted to exceed {budget_s:.0f} s on this host: {per_block:.1f} s per block)")
    sig = o.sigmas(True, 8, T)
    t0 = time.perf_counter()
    o.denoise(wc, cfg, lat * np.float32(sig[0]), sig[:2], cx, None, F, H, W)
    el = time.perf_counter() - t0
    return dict(common, value=1.0 / el, extrapolated=False, step_s=round(el, 2),
                tflops=round(dit_flops_per_step(T, S) / el / 1e12, 3),
                sample=f"ONE whole denoise step of the 768x512x25 workload through oracle.denoise (T={T}, S={S}, D=4096, all 48 blocks, "
                       f"{dit_flops_per_step(T, S) / 1e12:.1f} TFLOP incl. the per-step text K/V the reference recomputes) in {el:.1f} s of "
                       f"numpy/BLAS f32 on {nthreads} BLAS threads (numpy's OpenBLAS is fastest there at these shapes: 3.4 TFLOP/s against 1.4 "
                       f"with its default 64, tools/host_blas_probe.py) + {o._HOST_THREADS} threads for the row-wise passes ({ncores} "
                       f"schedulable host cores); block weights cycle through "
                       f"{ncycle} distinct sets (time does not depend on the values)")


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a launcher in the environment
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    """Start N ranks as children (this process has not touched the GPU and never will), relay their output, exit with their code."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["LTX_BENCH_SPAWNED"] = "1"
    return subprocess.call(cmd, env=env)


class Watchdog:
    """The extra legs call collectives; if one of them hangs, rank 0 prints the line with what it has (from a snapshot taken under
    `lock`, which the main thread also holds whenever it changes the line) and every rank leaves with status 3. os._exit: a thread
    blocked inside a collective cannot be joined, and the GPU process is never restarted or re-executed."""

    def __init__(self, seconds, on_fire):
        self.t = threading.Timer(seconds, on_fire)
        self.t.daemon = True

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.t.cancel()
        return False


def guarded_extra_legs(out, rank, legs_fn, seconds):
    """Run the sharded legs after the headline region. The line is printed whatever happens here, but a hang or an error ends
    every rank with status 3 (`_exit_status`, applied by main() after the line is out): the cause must be found from the
    records, not hidden behind rc 0. `legs_fn(put)` calls put(name, result) per finished leg."""
    lock = threading.Lock()
    legs = {}
    out["extra_legs"] = legs
    out["extra_legs_status"] = "running"

    def fire():
        try:
            if rank == 0:
                with lock:
                    snap = copy.deepcopy(out)
                snap["extra_legs_status"] = "hang"
                snap.setdefault("extra_legs", {})["watchdog"] = "timed out: a collective of the extra legs did not return"
                snap.pop("_exit_status", None)
                print(json.dumps(snap, default=str), flush=True)
        finally:
            os._exit(3)

    def put(name, res):
        with lock:
            legs[name] = res

    failed = None
    with Watchdog(seconds, fire):
        try:
            legs_fn(put)
        except Exception as e:  # noqa: BLE001
            failed = repr(e)
    with lock:
        out["extra_legs_status"] = "error" if failed else "ok"
        if failed:
            legs["error"] = failed
            out["_exit_status"] = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("replica", "cfg-pair", "sp", "vae-tiles"), default="replica")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--vae-to", choices=("all", "root"), default="all",
                    help="--mode vae-tiles: raw tiles broadcast to every rank (all) or sent once to rank 0, which blends (root)")
    ap.add_argument("--no-aux", action="store_true", help="skip the once-per-prompt legs (text-embedding connector, VAE encoder)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-launch HIP events in the timed region")
    ap.add_argument("--extra-legs", action="store_true",
                    help="replica mode, N >= 2: add the cfg_pair / sp legs after the timed region (a hang or error in them ends the run with status 3)")
    ap.add_argument("--no-extra-legs", action="store_true", help="accepted for compatibility: the legs are off unless --extra-legs")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher self-test: ranks rendezvous over gloo, count themselves and print the line without touching a GPU")
    ap.add_argument("--sp-overlap", action="store_true",
                    help="--mode sp / the sp extra leg: run each block's V^T gather on the library's side stream under the q|k projection "
                         "(library option sp_overlap = 1; off by default until a multi-GPU run has confirmed it)")
    ap.add_argument("--reference-roundings", action="store_true",
                    help="store the q / k projections f32 and keep split-K partial tiles f32 (library options qk_f32 = 1, split_f32 = 1): the "
                         "reference's rounding points exactly; the line's extra_roundings is then []")
    ap.add_argument("--quant", type=int, choices=(0, 4, 8), default=0,
                    help="--mode sp: quantise the transformer to this many bits (MLX affine, group 64) after the bf16 run and report both - "
                         "BASELINE configs[4] names a qint8 transformer")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N ranks on ONE GPU over gloo (every rank on cuda:0, host-staged collectives): executes the multi-rank code of "
                         "this script where a single GPU is all there is; the line is labelled and is not a scaling measurement")
    ap.add_argument("--selftest-legs", choices=("hang", "error", "ok"), default=None,
                    help="with --launch-check: drive the extra-legs guard with a fake leg that hangs / raises / returns (no GPU)")
    args = ap.parse_args()
    if args.rehearse_one_gpu and (args.mode != "replica" or args.extra_legs):
        # every rank on cuda:0 over gloo: only the replica path (no RCCL communicator inside the library) can run like that - two ranks of
        # one RCCL group on one device end in a duplicate-GPU error or in the watchdog
        raise SystemExit("bench.py: --rehearse-one-gpu rehearses the replica path only (no --mode cfg-pair / sp / vae-tiles, no --extra-legs)")

    # ---- who launches the ranks? (before torch or the library are imported: no GPU call may precede a spawn) ----
    has_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not has_launcher:
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong rank count\n")
        raise SystemExit(2)
    if args.mode == "cfg-pair" and world > 1 and world % 2:
        raise SystemExit("bench.py: --mode cfg-pair needs an even number of ranks")

    import torch
    import torch.distributed as dist

    if args.launch_check:
        if world > 1:
            dist.init_process_group("gloo")
        ones = torch.ones(1)
        if world > 1:
            dist.all_reduce(ones)
        line = {"metric": "launch-check", "n_gpus": world, "n_ranks_seen": int(ones.item()), "mode": args.mode}
        if args.selftest_legs:
            def fake(put):
                put("first", {"ok": True})
                if args.selftest_legs == "hang":
                    threading.Event().wait()  # a collective that never returns
                if args.selftest_legs == "error":
                    raise RuntimeError("leg failed")

            guarded_extra_legs(line, rank, fake, 1.0)
        status = line.pop("_exit_status", 0)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if status:
            os._exit(status)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.rehearse_one_gpu:
        # every rank on cuda:0 over gloo: only the replica path (no RCCL communicator inside the library) can run like that - two
        # ranks of one RCCL group on one device end in a duplicate-GPU error or in the watchdog
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    side = None  # gloo side channel: carries the 128-byte RCCL ids of the library's own communicators
    if world > 1 and args.rehearse_one_gpu:
        dist.init_process_group("gloo")
        side = dist.group.WORLD
    elif world > 1:
        dist.init_process_group("nccl", device_id=dev)
        side = dist.new_group(backend="gloo")
    ones = torch.ones(1, device=dev)
    if world > 1:
        dist.all_reduce(ones)  # RCCL: counts the ranks that really take part
    n_ranks_seen = int(ones.item())

    ltx = importlib.import_module("ltx-video-swift-mlx_amd")
    dmod = importlib.import_module("ltx-video-swift-mlx_amd.dist")
    ctx = ltx.Context(local)
    if args.reference_roundings:  # the two roundings the reference does not have, off: +0.4-0.5 ms per step
        ctx.set_option("qk_f32", 1)
        ctx.set_option("split_f32", 1)
    if args.sp_overlap:
        ctx.set_option("sp_overlap", 1)  # spawned ranks get --sp-overlap in their own argv; the library never reads the environment
    cfg = ltx.default_transformer_config()
    runner = {"replica": run_replica, "cfg-pair": run_cfg_pair, "sp": run_sp, "vae-tiles": run_vae_tiles}[args.mode]
    out = runner(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side)
    out["n_ranks_seen"] = n_ranks_seen
    if args.rehearse_one_gpu:
        out["rehearsal"] = f"{world} ranks share ONE GPU over gloo: code-path rehearsal, not a scaling measurement"
    out["launcher"] = "bench.py" if os.environ.get("LTX_BENCH_SPAWNED") else ("torch.distributed.run" if world > 1 else "single process")
    status = out.pop("_exit_status", 0)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if status:  # an extra leg failed on this rank: the line is out, now fail the run (no barrier: the peers may be gone)
        sys.stdout.flush()
        os._exit(status)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def timed(torch, dist, world, dev, fn, warmup, steps):
    """W untimed + exactly K timed calls of fn(i), barrier + synchronize on both sides, MAX over ranks."""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0  # this rank's K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    timed.per_rank = [own]
    if world > 1:
        t = torch.tensor([el, own], dtype=torch.float64, device=dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)  # the line's time is the MAX of the barrier-to-barrier times; `own` shows a slow box
        timed.per_rank = [float(x[1].item()) for x in allt]
        el = max(float(x[0].item()) for x in allt)
    return el


def extra_roundings():
    """The roundings this path has beyond the reference's (precision contract), as the library is configured NOW:
    the headline number carries its own numerics (round-4 verdict, item 2). Both can be switched off (`--reference-roundings`)."""
    ltx = sys.modules.get("ltx-video-swift-mlx_amd")
    if ltx is None:  # --launch-check: the library is not loaded
        return None
    out = []
    if ltx.get_option("qk_f32") == 0:
        out.append("qk_bf16_store")
    if ltx.get_option("split_f32") == 0 and ltx.get_option("dtl_splitk") != 0:
        out.append("splitk_bf16_partials")
    return out


def base_line(args, world, el, value, scaling, workload, extra_cfg):
    return {
        "metric": "DiT denoise steps/sec + VAE decode ms, 768x512x25 distilled",
        "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic", "mode": args.mode, "extra_roundings": extra_roundings(),
        "config": dict({"workload": workload}, **extra_cfg),
    }


# ---------------------------------------------------------------------------------------------------------------
# replica (default): BASELINE configs[1]
# ---------------------------------------------------------------------------------------------------------------
def run_replica(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side):
    F, H, W = ltx.latent_shape(WIDTH, HEIGHT, FRAMES)
    T = F * H * W
    ctx.dit_init_synthetic(cfg, seed=1234)
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

    el = timed(torch, dist, world, dev, step, args.warmup, args.steps)
    per_rank = list(timed.per_rank)

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
        e = ctx.prof_collect(3)
        ctx.prof_enable(False)
        if g["ms"] > 0:
            ach = g["work"] / (g["ms"] * 1e-3) / 1e12
            traffic, src = pmc_traffic()
            roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel{_dtl,_v2,}", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
                        "sustained_peak_measured": SUSTAINED_BF16_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_BF16_TFLOPS, 4),
                        "launches": g["launches"], "avg_launch_us": round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                        "gemm_ms_per_step": round(g["ms"] / args.steps, 3),
                        "gemm_tflop_per_step": round(g["work"] / args.steps / 1e12, 2),
                        "ms_per_step_with_events": round(1e3 * el_prof / args.steps, 3)}
        if a["ms"] > 0:
            extra["attention"] = {"achieved_tflops": round(a["work"] / (a["ms"] * 1e-3) / 1e12, 1),
                                  "mfma_util": round(a["work"] / (a["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "ms_per_step": round(a["ms"] / args.steps, 3), "launches": a["launches"]}
        if e["ms"] > 0:
            extra["row_kernels"] = {"ms_per_step": round(e["ms"] / args.steps, 3), "launches": e["launches"],
                                    "algorithmic_GBps": round(e["work"] / (e["ms"] * 1e-3) / 1e9, 1),
                                    "frac_of_hbm_peak": round(e["work"] / (e["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}

    out = base_line(args, world, el, world * args.steps / el, "weak",
                    "distilled 8-step schedule, 768x512x25 -> latent 4x16x24 (1536 tokens), 1024 text keys, CFG off, batch 1 per GPU, "
                    "48-layer DiT bf16 weights (random init)",
                    {"tokens": T, "text_keys": S_TEXT, "parallelism": f"replica x{world} (one sample per GPU, no data-path collective)"})
    spp = el / args.steps
    out["per_rank_ms_per_step"] = [round(1e3 * x / args.steps, 3) for x in per_rank]
    out["dit_tflop_per_step"] = {"reference_algorithm": round(dit_flops_per_step(T) / 1e12, 2),
                                 "executed": round(dit_flops_per_step(T, executed=True) / 1e12, 2),
                                 "note": "executed = without the caption projection and cross-attention K/V projections, which the "
                                         "library computes once per prompt (output-identical to the reference's per-step recompute)"}
    out["dit_executed_tflops_per_s"] = round(dit_flops_per_step(T, executed=True) / 1e12 / spp, 1)
    out["dit_executed_frac_of_peak"] = round(dit_flops_per_step(T, executed=True) / 1e12 / spp / PEAK_BF16_TFLOPS, 4)
    if roofline:
        out["roofline"] = roofline
    out.update(extra)

    if rank == 0 and not args.no_vae:
        try:
            out["vae"] = bench_vae(ctx, ltx, torch, dev, F, H, W)
        except Exception as e:  # the DiT line must still be reported
            out["vae"] = {"error": str(e)}
    if rank == 0 and not args.no_aux:
        try:
            out["pre_loop"] = bench_pre_loop(ctx, ltx, torch, dev)
        except Exception as e:
            out["pre_loop"] = {"error": str(e)}
    if world > 1 and args.extra_legs and not args.no_extra_legs:
        def legs_fn(put):
            if world % 2 == 0:
                put("cfg_pair", leg_cfg_pair(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, min(args.steps, 4)))
            put("sp", leg_sp(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, 2))

        guarded_extra_legs(out, rank, legs_fn, 240.0)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T, S_TEXT, F, H, W)
        try:
            out["cpu_baseline"]["vae"] = cpu_baseline_vae(F, H, W)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"]["vae"] = {"error": repr(e)}
    return out


# ---------------------------------------------------------------------------------------------------------------
# cfg-pair: BASELINE configs[2]
# ---------------------------------------------------------------------------------------------------------------
def leg_cfg_pair(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, steps, warmup=1):
    """dev schedule, CFG 4.0 on 768x512x25. world == 1: the batched B = 2 forward; else one pair per two ranks."""
    F, H, W = ltx.latent_shape(WIDTH, HEIGHT, FRAMES)
    T = F * H * W
    context = torch.empty((2, S_TEXT, cfg.caption_channels), dtype=torch.bfloat16, device=dev)  # [negative, positive]
    ctx.op_fill_normal_bf16(context, seed=43)  # same seed on every rank = the broadcast context of one prompt
    mask = torch.ones((2, S_TEXT), dtype=torch.int32, device=dev)
    pair = rank // 2
    latent = torch.empty((1, 128, F, H, W), dtype=torch.float32, device=dev)
    ctx.op_fill_normal_f32(latent, seed=4200 + pair)  # both ranks of a pair hold the same sample
    sig = ltx.sigmas(False, 40, T)
    shard = ltx.SHARD_NONE
    if world > 1:
        group, _ = dmod.pair_groups()
        dmod.bootstrap(ctx, group)  # RCCL communicator of THIS pair, created by the library from an id carried over gloo
        shard = ltx.SHARD_CFG
    n0 = ctx.dist_info()["collectives"]

    def step(i):
        j = i % 40
        ctx.denoise_dev(latent, sig[j:j + 2], context, mask, F, H, W, ctx_version=11, mask_all_ones=True, cfg_scale=4.0, shard=shard)
        if j == 39:
            ctx.op_fill_normal_f32(latent, seed=5000 + i + pair)

    el = timed(torch, dist, world, dev, step, warmup, steps)
    res = {"steps_per_s": round(max(1, world // 2) * steps / el, 4), "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
           "pairs": max(1, world // 2), "cfg_scale": 4.0,
           "how": "one B=2 forward per step on one GPU" if world == 1 else
                  "rank 2p negative / rank 2p+1 positive branch, one RCCL all-gather of 2 x 786 KB per step (LTX_SHARD_CFG)",
           "collectives_per_step": (ctx.dist_info()["collectives"] - n0) / (warmup + steps) if world > 1 else 0}
    if world > 1:
        # both ranks of a pair must hold the same latent bit for bit (they apply CFG + Euler redundantly)
        chk = torch.stack([latent.double().sum(), latent.double().abs().sum()]).to(dev)
        allv = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allv, chk)
        res["pair_latents_identical"] = bool(all(torch.equal(allv[2 * p], allv[2 * p + 1]) for p in range(world // 2)))
        ctx.dist_shutdown()
    return res


def run_cfg_pair(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side):
    ctx.dit_init_synthetic(cfg, seed=1234)
    r = leg_cfg_pair(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, args.steps, args.warmup)
    el = r["ms_per_step"] * 1e-3 * args.steps
    out = base_line(args, world, el, r["steps_per_s"], "weak",
                    "dev 40-step schedule, CFG 4.0 (negative + positive forward per step), 768x512x25 -> 1536 tokens, 1024 text keys, "
                    "48-layer DiT bf16 weights (random init)",
                    {"tokens": 1536, "text_keys": S_TEXT, "parallelism": f"{r['pairs']} CFG pair(s): " + r["how"]})
    out["metric"] = "DiT denoise steps/sec, 768x512x25 dev CFG 4.0 (BASELINE configs[2])"
    out["cfg_pair"] = r
    return out


# ---------------------------------------------------------------------------------------------------------------
# sp: one 9984-token sample over all ranks (BASELINE configs[4] shape)
# ---------------------------------------------------------------------------------------------------------------
def leg_sp(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, steps, warmup=1):
    F, H, W = ltx.latent_shape(768, 512, 201)
    T = F * H * W
    assert (F, H, W) == (26, 16, 24)
    if T % world or (T // world) % 8:
        return {"skipped": f"{T} tokens do not split over {world} ranks into multiples of 8"}
    context = torch.empty((1, S_TEXT, cfg.caption_channels), dtype=torch.bfloat16, device=dev)
    ctx.op_fill_normal_bf16(context, seed=43)
    mask = torch.ones((1, S_TEXT), dtype=torch.int32, device=dev)
    latent = torch.empty((1, 128, F, H, W), dtype=torch.float32, device=dev)
    ctx.op_fill_normal_f32(latent, seed=77)  # the same sample on every rank
    sig = ltx.sigmas(True, 8, T)
    shard = ltx.SHARD_NONE
    if world > 1:
        dmod.bootstrap(ctx, side)
        shard = ltx.SHARD_SEQUENCE
    n0 = ctx.dist_info()["collectives"]

    def step(i):
        j = i % 8
        ctx.denoise_dev(latent, sig[j:j + 2], context, mask, F, H, W, ctx_version=13, mask_all_ones=True, shard=shard)
        if j == 7:
            ctx.op_fill_normal_f32(latent, seed=7000 + i)

    el = timed(torch, dist, world, dev, step, warmup, steps)
    res = {"steps_per_s": round(steps / el, 4), "ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "tokens": T,
           "tokens_per_rank": T // world,
           "how": "whole sample on one GPU" if world == 1 else
                  f"token slices of {T // world}; per block one RCCL all-gather of K rows and one of V^T "
                  f"({2 * (T // world) * 4096 * 2 // 1024} KB per rank each), one of the velocity slices per step (LTX_SHARD_SEQUENCE)",
           "collectives_per_step": (ctx.dist_info()["collectives"] - n0) / (warmup + steps) if world > 1 else 0}
    if world > 1:
        chk = torch.stack([latent.double().sum(), latent.double().abs().sum()]).to(dev)
        allv = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allv, chk)
        res["rank_latents_identical"] = bool(all(torch.equal(allv[0], v) for v in allv))
        ctx.dist_shutdown()
    return res


def run_sp(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side):
    ctx.dit_init_synthetic(cfg, seed=1234)
    r = leg_sp(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, args.steps, args.warmup)
    if "skipped" in r:
        raise SystemExit("bench.py --mode sp: " + r["skipped"])
    if args.quant and world == 1:
        # config 5's named setting: the same steps on the quantised model (codes + bf16 group scale / bias resident, bf16 weights released;
        # at this token count every quantised GEMM is preceded by a de-quantise pass into one scratch matrix: the GEMMs are MFMA-bound
        # here, so the price of the smaller model is that pass - LTXQuantizationConfig.swift:19-62)
        mem0 = ctx.dit_memory_info()
        ctx.dit_quantize(args.quant)
        mem1 = ctx.dit_memory_info()
        q = leg_sp(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side, args.steps, args.warmup)
        r["quant"] = {"bits": args.quant, "group": 64, "steps_per_s": q["steps_per_s"], "ms_per_step": q["ms_per_step"],
                      "vs_bf16": round(q["ms_per_step"] / r["ms_per_step"], 4),
                      "weights_before": mem0, "weights_after": mem1,
                      "note": "capacity feature at this size: the quantised step pays the de-quantise pre-pass of every Linear"}
    el = r["ms_per_step"] * 1e-3 * args.steps
    out = base_line(args, world, el, r["steps_per_s"], "strong",
                    "distilled schedule, 768x512x201 -> latent 26x16x24 (9984 tokens), 1024 text keys, CFG off, ONE sample over all "
                    "ranks, 48-layer DiT bf16 weights (random init)",
                    {"tokens": r["tokens"], "text_keys": S_TEXT, "parallelism": f"sequence parallel x{world}: " + r["how"]})
    out["metric"] = "DiT denoise steps/sec, 768x512x201 distilled, one sample (BASELINE configs[4] shape)"
    out["sp"] = r
    return out


# ---------------------------------------------------------------------------------------------------------------
# vae-tiles: BASELINE configs[4] decode
# ---------------------------------------------------------------------------------------------------------------
def run_vae_tiles(args, ctx, ltx, dmod, torch, dist, dev, cfg, rank, world, side):
    F, H, W, tile, ov = 26, 16, 24, 8, 1
    ctx.vae_init_synthetic(seed=77)
    plan, nf = ltx.vae_tile_plan(F, tile, ov)
    lat = torch.empty((1, 128, F, H, W), dtype=torch.float32, device=dev)
    ctx.op_fill_normal_f32(lat, seed=45)
    frames = torch.empty((nf, H * 32, W * 32, 3), dtype=torch.float32, device=dev)
    if world > 1:
        dmod.bootstrap(ctx, side)

    def step(i):
        if world > 1 and args.vae_to == "root":
            ctx.vae_decode_gathered_dev(lat, F, H, W, frames if rank == 0 else None, root=0, tile=tile, overlap=ov)
        elif world > 1:
            ctx.vae_decode_sharded_dev(lat, F, H, W, frames, tile=tile, overlap=ov)
        else:
            ctx.vae_decode_dev(lat, F, H, W, frames, tile=tile, overlap=ov)

    el = timed(torch, dist, world, dev, step, args.warmup, args.steps)
    out = base_line(args, world, el, args.steps / el, "strong",
                    f"VAE decode of 26 latent frames at 768x512, temporal tiles {plan} (tile 8, overlap 1) -> {nf} frames, tiles "
                    f"round-robin over {world} rank(s), raw tiles " + ("broadcast, blend + clip on every rank" if args.vae_to == "all" else
                                                                         "sent once to rank 0, which blends and clips") +
                    f"; the plan has {len(plan)} tiles (VideoDecoder.swift:534-548), so at most {len(plan)} ranks decode",
                    {"latent": [F, H, W], "tiles": len(plan), "frames": nf})
    out["metric"] = "tiled VAE decodes/sec, 768x512x201 (BASELINE configs[4])"
    out["unit"] = "decodes/s"
    out["decode_ms"] = out.pop("ms_per_step")
    if world > 1:
        if args.vae_to == "all":
            chk = torch.stack([frames.double().sum()]).to(dev)
            allv = [torch.empty_like(chk) for _ in range(world)]
            dist.all_gather(allv, chk)
            out["rank_frames_identical"] = bool(all(torch.equal(allv[0], v) for v in allv))
        out["busy_ranks"] = min(world, len(plan))
        ctx.dist_shutdown()
    return out


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
    """VAE decode of one [1,128,4,16,24] latent -> (25,512,768,3), device-resident: ms, and its own roofline object (the decoder is
    a dense contraction: 12.96 TFLOP against 5.80 GB of algorithmic bytes, BASELINE.md section 2 - MFMA-bound; GB/s reported too)."""
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
        ach = c["work"] / (c["ms"] * 1e-3) / 1e12
        vtraffic, vsrc = pmc_traffic(VAE_TRAFFIC_FILES)  # fabric-side bytes per conv launch, two --pmc passes of tools/bench_vae.py
        res["conv_kernel_tflops"] = round(ach, 1)
        res["conv_ms"] = round(c["ms"] / iters, 3)
        res["roofline"] = {"bound": "mfma", "kernel": "conv3d_halo2_kernel + conv3d_halo_kernel + gemm_bf16_kernel_v2<conv>", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": vtraffic, "traffic_source": vsrc,
                           "launches": c["launches"], "avg_launch_us": round(1e3 * c["ms"] / max(1, c["launches"]), 2),
                           "whole_decode_frac": round(12.96 / (ms * 1e-3) / PEAK_BF16_TFLOPS, 4),
                           "hbm_frac_at_algorithmic_bytes": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    return res


if __name__ == "__main__":
    main()
