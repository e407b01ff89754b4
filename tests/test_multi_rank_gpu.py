"""The library's multi-GPU paths (include/ltxhip.h "Multi-GPU"), driven through the C ABI by TWO PROCESSES.

The test box has one GPU and RCCL refuses two ranks on one device, so the two processes share the GPU and the group's all-gather
runs over gloo through the library's host-transport hook (ltx_dist_set_transport); on a multi-GPU node the same entry points run
over RCCL (ltx_dist_init; `test_native_rccl_group_of_one` below covers that code path with a group of one). What is checked is
the sharding itself, which is transport-independent:

* LTX_SHARD_CFG       rank 0 = negative branch, rank 1 = positive branch, one all-gather per step; both ranks must end with
                      bit-identical latents that match the single-process batched-CFG loop of the same library (B=1 vs B=2
                      launches pick different tiles: rel-L2 <= 5e-3) and the oracle (<= 5e-2 as test_denoise_gpu).
* LTX_SHARD_SEQUENCE  token slices, K / V^T all-gathered per block, velocity slices per forward; same checks.
* sequence-parallel forward through the context's transport (gather = NULL).
* tile-sharded VAE decode: raw tiles broadcast, blended on every rank == the single-process tiled decode, bit for bit.
* a failing transport aborts the call with an error instead of computing on unfilled buffers.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(oracle, ocfg, F, H, W, S, seed, nb):
    rng = np.random.default_rng(seed)
    lat = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    cx = oracle.bf16_round(rng.standard_normal((nb, S, ocfg.caption_channels)).astype(np.float32))
    mask = (rng.random((nb, S)) > 0.25).astype(np.int32)
    mask[:, 0] = 1
    return lat, cx, mask


def _worker(rank, world, port, scenario, wpath, dims, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ltx = importlib.import_module("ltx-video-swift-mlx_amd")
        d = importlib.import_module("ltx-video-swift-mlx_amd.dist")
        import ltx_oracle as oracle
        from test_dit_gpu import small_cfg

        F, H, W, S = dims
        ctx = ltx.Context(0)
        d.attach_gloo_transport(ctx, torch.device("cuda", 0))
        info0 = ctx.dist_info()
        assert info0 == {"rank": rank, "world": world, "native": False, "collectives": 0}, info0
        out = {}
        if scenario in ("cfg", "seq", "seq_cfg", "spfwd", "fail"):
            cfg, ocfg = small_cfg(ltx, oracle, heads=4, layers=3, caption=256)
            ctx.dit_load(wpath, cfg)
            nb = 2 if scenario in ("cfg", "seq_cfg") else 1
            lat, cx, mask = _inputs(oracle, ocfg, F, H, W, S, 11, nb)
            cdev = torch.from_numpy(ltx.f32_to_bf16_bits(cx).astype(np.int16)).cuda().view(torch.bfloat16)
            mdev = torch.from_numpy(mask).cuda()
        if scenario == "cfg":
            sig = ltx.sigmas(False, 4, F * H * W)
            latd = torch.from_numpy(lat * sig[0]).cuda()
            kw = dict(cfg_scale=3.0, guidance_rescale=0.5, stg_scale=0.8, stg_blocks=(1,), ge_gamma=0.4)
            ctx.denoise_dev(latd, sig, cdev, mdev, F, H, W, ctx_version=5, shard=ltx.SHARD_CFG, **kw)
            out["sharded"] = latd.cpu().numpy()
            out["collectives"] = ctx.dist_info()["collectives"]
            lat1 = torch.from_numpy(lat * sig[0]).cuda()
            ctx.denoise_dev(lat1, sig, cdev, mdev, F, H, W, ctx_version=5, **kw)  # same version: sub-keys must not collide
            out["single"] = lat1.cpu().numpy()
        elif scenario in ("seq", "seq_cfg"):
            sig = ltx.sigmas(True, 8, F * H * W)[:4] if scenario == "seq" else ltx.sigmas(False, 3, F * H * W)
            kw = dict(cfg_scale=2.5) if scenario == "seq_cfg" else {}
            latd = torch.from_numpy(lat * sig[0]).cuda()
            ctx.denoise_dev(latd, sig, cdev, mdev, F, H, W, ctx_version=9, shard=ltx.SHARD_SEQUENCE, **kw)
            out["sharded"] = latd.cpu().numpy()
            out["collectives"] = ctx.dist_info()["collectives"]
            lat1 = torch.from_numpy(lat * sig[0]).cuda()
            ctx.denoise_dev(lat1, sig, cdev, mdev, F, H, W, ctx_version=9, **kw)
            out["single"] = lat1.cpu().numpy()
        elif scenario == "spfwd":
            T = F * H * W
            t0, t1 = d.sp_token_slice(T, rank, world)
            tokens = torch.from_numpy(lat.reshape(128, T).T.copy()).cuda().to(torch.bfloat16).reshape(1, T, 128)
            ts = torch.full((1,), 0.7, dtype=torch.float32, device="cuda")
            full = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
            ctx.dit_forward_dev(tokens, cdev, ts, mdev, F, H, W, full, ctx_version=3)
            part = torch.empty((1, t1 - t0, 128), dtype=torch.float32, device="cuda")
            ctx.dit_forward_sp_dev(tokens[:, t0:t1].contiguous(), cdev, ts, mdev, F, H, W, part, rank, world, None, ctx_version=3)
            both = torch.empty((world, t1 - t0, 128), dtype=torch.float32, device="cuda")
            ctx.dist_allgather_dev(part, both)
            out["sharded"] = both.reshape(T, 128).cpu().numpy()
            out["single"] = full[0].cpu().numpy()
        elif scenario == "vae":
            ctx.vae_init_synthetic(seed=77)
            rng = np.random.default_rng(3)
            lat = torch.from_numpy(rng.standard_normal((1, 128, F, H, W)).astype(np.float32)).cuda()
            tile, ov = 3, 1
            plan, nf = ltx.vae_tile_plan(F, tile, ov)
            fr = torch.empty((nf, H * 32, W * 32, 3), dtype=torch.float32, device="cuda")
            n = ctx.vae_decode_sharded_dev(lat, F, H, W, fr, tile=tile, overlap=ov)
            out["sharded"] = fr[:n].cpu().numpy()
            fr1 = torch.empty_like(fr)
            n1 = ctx.vae_decode_dev(lat, F, H, W, fr1, tile=tile, overlap=ov)
            out["single"] = fr1[:n1].cpu().numpy()
            out["collectives"] = ctx.dist_info()["collectives"]
            out["tiles"] = len(plan)
            # gather form: raw tiles travel to rank 1 only, which blends; rank 0 passes no output buffer
            fg = torch.full_like(fr, 7.0) if rank == 1 else None
            ng = ctx.vae_decode_gathered_dev(lat, F, H, W, fg, root=1, tile=tile, overlap=ov)
            out["gathered_n"] = ng
            out["gathered"] = fg[:ng].cpu().numpy() if rank == 1 else None
        elif scenario == "fail":
            # a transport that raises on one rank: that rank's call must fail with the original error (not return OK on unfilled
            # buffers); the healthy rank is released by the gloo timeout of its peer's abort, so it only attempts a local call
            if rank == 1:
                def boom(send, recv, n):
                    raise RuntimeError("transport down")
                ctx.dist_set_transport(rank, world, boom)
                sig = ltx.sigmas(False, 2, F * H * W)
                latd = torch.from_numpy(lat * sig[0]).cuda()
                try:
                    ctx.denoise_dev(latd, sig, torch.cat([cdev, cdev]), None, F, H, W, cfg_scale=2.0, shard=ltx.SHARD_CFG)
                    out["raised"] = "nothing"
                except RuntimeError as e:
                    out["raised"] = str(e)
            else:
                out["raised"] = "n/a"
        torch.cuda.synchronize()
        q.put((rank, "ok", out))
        dist.barrier()
        dist.destroy_process_group()
        ctx.close()
    except Exception as e:  # surface the failure in the parent instead of a queue timeout
        import traceback

        q.put((rank, "error: " + repr(e) + "\n" + traceback.format_exc(), {}))


def _run(scenario, wpath, dims):
    import torch.multiprocessing as mp

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, scenario, wpath, dims, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in range(2)]
    finally:
        # never leave a worker holding the GPU: a rank whose peer died would sit in its collective forever
        for p in procs:
            p.join(timeout=60)
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=30)
    res.sort(key=lambda r: r[0])
    for r in res:
        assert r[1] == "ok", r[1]
    return [r[2] for r in res]


@pytest.fixture(scope="module")
def weights(ltx, oracle, tmp_path_factory):
    from test_dit_gpu import small_cfg, write_dit_file

    cfg, ocfg = small_cfg(ltx, oracle, heads=4, layers=3, caption=256)
    w = oracle.synth_dit_weights(ocfg, seed=33)
    path = str(tmp_path_factory.mktemp("mr") / "dit.safetensors")
    write_dit_file(oracle, w, path)
    return path, w, ocfg


def test_cfg_pair_sharded_denoise_two_ranks(ltx, oracle, weights):
    from test_dit_gpu import rel_l2

    path, w, ocfg = weights
    F, H, W, S = 2, 4, 4, 24
    r0, r1 = _run("cfg", path, (F, H, W, S))
    assert np.array_equal(r0["sharded"], r1["sharded"]), "ranks ended with different latents"
    assert r0["collectives"] == 4 and r1["collectives"] == 4  # exactly one all-gather per step
    assert rel_l2(r0["sharded"], r0["single"]) <= 5e-3, rel_l2(r0["sharded"], r0["single"])
    lat, cx, mask = _inputs(oracle, ocfg, F, H, W, S, 11, 2)
    sig = ltx.sigmas(False, 4, F * H * W)
    ref = oracle.denoise(w, ocfg, lat * sig[0], sig, cx[1:2], mask[1:2], F, H, W, cfg_scale=3.0, rescale=0.5, stg_scale=0.8,
                         stg_blocks=(1,), ge_gamma=0.4, neg_context=cx[0:1], neg_mask=mask[0:1])
    assert rel_l2(r0["sharded"], ref) <= 5e-2, rel_l2(r0["sharded"], ref)


@pytest.mark.parametrize("scenario", ["seq", "seq_cfg"])
def test_sequence_sharded_denoise_two_ranks(ltx, oracle, weights, scenario):
    from test_dit_gpu import rel_l2

    path, w, ocfg = weights
    F, H, W, S = 2, 8, 8, 40
    r0, r1 = _run(scenario, path, (F, H, W, S))
    assert np.array_equal(r0["sharded"], r1["sharded"]), "ranks ended with different latents"
    steps, fwd_per_step = (3, 1) if scenario == "seq" else (3, 2)
    assert r0["collectives"] == steps * fwd_per_step * (2 * 3 + 1)  # per forward: K and V^T per block + the velocity slices
    assert rel_l2(r0["sharded"], r0["single"]) <= 5e-3, rel_l2(r0["sharded"], r0["single"])


@pytest.mark.parametrize("F,H,W,S", [(2, 8, 8, 40), (4, 4, 6, 24)])
def test_sequence_parallel_forward_two_ranks(ltx, oracle, weights, F, H, W, S):
    from test_dit_gpu import rel_l2

    path, w, ocfg = weights
    r0, r1 = _run("spfwd", path, (F, H, W, S))
    assert np.array_equal(r0["sharded"], r1["sharded"])
    assert rel_l2(r0["sharded"], r0["single"]) <= 2e-3, rel_l2(r0["sharded"], r0["single"])
    lat, cx, mask = _inputs(oracle, ocfg, F, H, W, S, 11, 1)
    T = F * H * W
    tokens = oracle.bf16_round(lat.reshape(128, T).T.reshape(1, T, 128))
    ref = oracle.dit_forward(w, ocfg, tokens, cx, np.array([0.7], np.float32), mask, F, H, W)[0]
    assert rel_l2(r0["sharded"], ref) <= 2e-2, rel_l2(r0["sharded"], ref)


def test_tile_sharded_vae_decode_two_ranks(ltx, weights):
    r0, r1 = _run("vae", weights[0], (7, 2, 2, 0))
    assert r0["tiles"] == 3
    assert np.array_equal(r0["sharded"], r1["sharded"])
    assert np.array_equal(r0["sharded"], r0["single"]), "sharded tiles + blend differ from the single-process tiled decode"
    assert r0["collectives"] == 3  # one broadcast (as an all-gather on the host transport) per tile
    # ltx_vae_decode_gathered_dev: only the root holds frames, bit-identical to the single-process decode; every rank gets the count
    assert r0["gathered"] is None and r0["gathered_n"] == r1["gathered_n"] == r0["single"].shape[0]
    assert np.array_equal(r1["gathered"], r0["single"])


def test_failing_transport_aborts_the_call(ltx, weights):
    r0, r1 = _run("fail", weights[0], (1, 4, 4, 16))
    assert r1["raised"] == "transport down", r1["raised"]


def test_native_rccl_group_of_one(ltx, oracle, gpu_ctx, weights, tmp_path):
    """ltx_dist_unique_id / ltx_dist_init / ncclAllGather / ncclBroadcast on the context's stream with a one-rank communicator: the
    RCCL code path of dist.cpp (library load, communicator, stream ordering). A sequence-sharded loop over one rank must equal the
    unsharded loop bit for bit, and so must the tile-sharded decode."""
    import torch

    from test_dit_gpu import small_cfg

    path, w, ocfg = weights
    cfg, _ = small_cfg(ltx, oracle, heads=4, layers=3, caption=256)
    ctx = ltx.Context(0)
    try:
        ctx.dist_init(0, 1, ltx.dist_unique_id())
        assert ctx.dist_info()["native"] and ctx.dist_info()["world"] == 1
        a = torch.arange(4096, dtype=torch.float32, device="cuda")
        b = torch.zeros_like(a)
        ctx.dist_allgather_dev(a, b)
        ctx.dist_broadcast_dev(a, 0)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        ctx.dit_load(path, cfg)
        F, H, W, S = 2, 4, 4, 24
        lat, cx, mask = _inputs(oracle, ocfg, F, H, W, S, 4, 1)
        sig = ltx.sigmas(True, 8, F * H * W)[:3]
        cdev = torch.from_numpy(ltx.f32_to_bf16_bits(cx).astype(np.int16)).cuda().view(torch.bfloat16)
        l0 = torch.from_numpy(lat * sig[0]).cuda()
        l1 = l0.clone()
        ctx.denoise_dev(l0, sig, cdev, None, F, H, W, ctx_version=2, shard=ltx.SHARD_SEQUENCE)
        ctx.denoise_dev(l1, sig, cdev, None, F, H, W, ctx_version=2)
        assert torch.equal(l0, l1)
        ctx.vae_init_synthetic(seed=5)
        latv = torch.randn((1, 128, 5, 2, 2), device="cuda")
        plan, nf = ltx.vae_tile_plan(5, 3, 1)
        f0 = torch.empty((nf, 64, 64, 3), device="cuda")
        f1 = torch.empty_like(f0)
        assert ctx.vae_decode_sharded_dev(latv, 5, 2, 2, f0, tile=3, overlap=1) == nf
        assert ctx.vae_decode_dev(latv, 5, 2, 2, f1, tile=3, overlap=1) == nf
        assert torch.equal(f0, f1)
        f2 = torch.empty_like(f0)
        assert ctx.vae_decode_gathered_dev(latv, 5, 2, 2, f2, root=0, tile=3, overlap=1) == nf
        assert torch.equal(f2, f1)
        # the sequence-parallel branch of the forward on the native transport (V^T gather + interleave on the side stream under the
        # q|k projection, K gather on the context's stream, fork / join events), driven through a one-rank group by the
        # self-test hook: the bits must be those of the plain forward
        T = F * H * W
        tokens = torch.from_numpy(lat.reshape(128, T).T.copy()).cuda().to(torch.bfloat16).reshape(1, T, 128)
        ts = torch.full((1,), 0.7, dtype=torch.float32, device="cuda")
        va = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
        vb = torch.empty_like(va)
        n0 = ctx.dist_info()["collectives"]
        ctx.dit_forward_dev(tokens, cdev, ts, None, F, H, W, va, ctx_version=0)
        assert ctx.dist_info()["collectives"] == n0
        ctx.set_option("sp_selftest", 1)
        try:
            for _ in range(3):
                ctx.dit_forward_dev(tokens, cdev, ts, None, F, H, W, vb, ctx_version=0)
                torch.cuda.synchronize()
                assert torch.equal(va, vb)
        finally:
            ctx.set_option("sp_selftest", 0)
        assert ctx.dist_info()["collectives"] == n0 + 3 * 2 * 3  # K and V^T gathers of three blocks, three forwards
        with pytest.raises(ltx.LTXError):  # CFG sharding needs two ranks
            ctx.denoise_dev(l0, sig, torch.cat([cdev, cdev]), None, F, H, W, cfg_scale=2.0, shard=ltx.SHARD_CFG)
        ctx.dist_shutdown()
        assert ctx.dist_info()["world"] == 1 and not ctx.dist_info()["native"]
    finally:
        ctx.close()


def test_tile_building_blocks_match_the_tiled_decode(ltx, gpu_ctx):
    """ltx_vae_decode_tile_dev + ltx_vae_blend_tiles_dev composed by the host == ltx_vae_decode_dev with tiling, bit for bit; a
    tile's raw frames are NOT the clipped frames (blending clipped tiles is not the reference, VideoDecoder.swift:561-592,501-505)."""
    import torch

    gpu_ctx.vae_init_synthetic(seed=77)
    F, H, W, tile, ov = 7, 2, 3, 3, 1
    lat = torch.randn((1, 128, F, H, W), device="cuda") * 2.0
    plan, nf = ltx.vae_tile_plan(F, tile, ov)
    ref = torch.empty((nf, H * 32, W * 32, 3), device="cuda")
    assert gpu_ctx.vae_decode_dev(lat, F, H, W, ref, tile=tile, overlap=ov) == nf
    tiles, counts = [], []
    for i, (s, e) in enumerate(plan):
        n_i = 8 * (e - s - 1) + 1
        t = torch.empty((n_i, H * 32, W * 32, 3), device="cuda")
        assert gpu_ctx.vae_decode_tile_dev(lat, F, H, W, tile, ov, i, t) == n_i
        tiles.append(t)
        counts.append(n_i)
    out = torch.empty_like(ref)
    assert gpu_ctx.vae_blend_tiles_dev(tiles, counts, ov, H, W, out) == nf
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert float(tiles[0].min()) < 0.0 or float(tiles[0].max()) > 1.0, "raw tiles are pre-clip values"
    # an untiled decode of the first tile's latent frames == that tile's raw frames mapped through the clip
    s, e = plan[0]
    sub = lat[:, :, s:e].contiguous()
    whole = torch.empty((counts[0], H * 32, W * 32, 3), device="cuda")
    gpu_ctx.vae_decode_dev(sub, e - s, H, W, whole)
    assert torch.equal(whole, torch.clamp((tiles[0] + 1.0) * 0.5, 0.0, 1.0))
    with pytest.raises(ltx.LTXError):
        gpu_ctx.vae_decode_tile_dev(lat, F, H, W, tile, ov, len(plan), tiles[0])
