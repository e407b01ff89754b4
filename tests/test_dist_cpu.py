"""N > 1 path on CPU: world-size-2 gloo processes (no GPU). Checks that the CFG-pair sharding reproduces the
single-process loop bit for bit on every rank, that the context broadcast delivers rank 0's tensor, and that VAE tile
sharding + gather + blend equals the serial blend."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(ctx_pos, ctx_neg):
    """Deterministic stand-in for the DiT forward: a fixed nonlinear map of (latent, sigma, branch context)."""
    def fwd(latent, sigma, branch):
        c = ctx_pos if branch == 1 else ctx_neg
        return torch.tanh(latent * (0.5 + sigma)) * c.mean() + 0.1 * torch.roll(latent, 1, dims=-1) * sigma + c.std()
    return fwd


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("ltx-video-swift-mlx_amd.dist")
    g = torch.Generator().manual_seed(0)
    latent = torch.randn(1, 8, 2, 3, 4, generator=g)
    ctx = torch.randn(2, 5, 16, generator=g) if rank == 0 else torch.zeros(2, 5, 16)
    ctx, _ = d.broadcast_context(ctx)
    sig = [1.0, 0.8, 0.5, 0.1, 0.0]
    fwd = _fake_forward(ctx[1], ctx[0])
    out = d.denoise_cfg_sharded(latent.clone(), sig, fwd, cfg_scale=4.0, rescale=0.7)
    ref = d.denoise_cfg_single(latent.clone(), sig, fwd, cfg_scale=4.0, rescale=0.7)
    # VAE tiles: 5 raw tiles of different lengths, round-robin over ranks
    shapes = [(9, 4, 4, 3), (9, 4, 4, 3), (17, 4, 4, 3), (9, 4, 4, 3), (25, 4, 4, 3)]
    tiles = [torch.randn(s, generator=torch.Generator().manual_seed(10 + i)) for i, s in enumerate(shapes)]
    mine = {i: tiles[i] for i in d.shard_tiles(len(tiles))}
    gathered = d.gather_tiles_to_rank0(mine, len(tiles), shapes, "cpu")
    ok_tiles = True
    if rank == 0:
        ok_tiles = torch.equal(d.blend_tiles(gathered, 1), d.blend_tiles(tiles, 1))
    # sequence parallelism: token slices of one sample, velocity slices gathered back in global token order
    T, C = 48, 8
    t0, t1 = d.sp_token_slice(T)
    tok = torch.randn(T, C, generator=torch.Generator().manual_seed(5))
    per_token = lambda x: torch.tanh(x) * 3.0 + x.flip(-1)  # any row-wise map
    ok_sp = torch.equal(d.sp_gather_velocity(per_token(tok[t0:t1])), per_token(tok)) and (t0, t1) == (rank * 24, rank * 24 + 24)
    q.put((rank, torch.equal(out, ref), float(ctx.sum()), bool(ok_tiles) and bool(ok_sp), out.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_pair_sharding_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res), "sharded CFG loop differs from the single-process loop"
    assert res[0][2] == res[1][2] != 0.0, "context broadcast failed"
    assert all(r[3] for r in res), "tile gather/blend mismatch"
    assert res[0][4] == res[1][4], "ranks diverged"


def test_tile_sharding_plan():
    sys.path.insert(0, ROOT)
    d = importlib.import_module("ltx-video-swift-mlx_amd.dist")
    assert d.shard_tiles(4, 0, 8) == [0] and d.shard_tiles(4, 5, 8) == []
    assert sorted(sum((d.shard_tiles(5, r, 2) for r in range(2)), [])) == [0, 1, 2, 3, 4]
    assert [d.cfg_branch_for_rank(r) for r in range(4)] == [0, 1, 0, 1]
    # sequence-parallel token slices: equal, contiguous, multiples of 8 (1536 tokens over 8 ranks = 192 each)
    assert [d.sp_token_slice(1536, r, 8) for r in (0, 7)] == [(0, 192), (1344, 1536)]
    with pytest.raises(AssertionError):
        d.sp_token_slice(1536, 0, 5)
