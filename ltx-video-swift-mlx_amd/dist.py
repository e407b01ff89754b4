"""Process-group bootstrap for the library's multi-GPU layer (include/ltxhip.h "Multi-GPU", csrc/dist.cpp).

The sharded paths themselves - CFG-pair denoise, sequence-parallel forward / denoise, tile-sharded VAE decode - live in
libltxhip.so and call RCCL directly; nothing in this file is on the data path. What a host has to do, and what this file does for
the Python hosts (tests, bench.py), is membership: decide which ranks form a group, carry the 128-byte RCCL id from the group's
first rank to the others, and call ``ltx_dist_init`` on every rank. torch.distributed is used only as that side channel (and as
the barrier / max-over-ranks of bench.py); a Swift host would use whatever channel it has.

* ``bootstrap(ctx, group)``        native RCCL group = the ranks of a torch.distributed group (default: all ranks).
* ``attach_gloo_transport(ctx)``   the same membership with the all-gather done over gloo through host memory, for world-size-2
                                   tests where RCCL cannot run (two processes sharing the test box's one GPU).
* ``pair_groups()``                ranks (0,1), (2,3), ... as separate groups: one CFG pair per two GPUs (BASELINE config 3 on 2,
                                   4 pairs on an 8-GPU node).

The functions under "protocol model" restate the exchange protocol of the sharded loops with torch ops on CPU tensors. They are
test infrastructure for the world-size-2 gloo tests that run without a GPU (tests/test_dist_cpu.py); the product never calls them.
"""
import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------------------
# membership
# ---------------------------------------------------------------------------------------------------------------
def bootstrap(ctx, group=None):
    """RCCL communicator of `ctx` over the ranks of `group`: the id is created by the library on the group's first rank
    (ltx_dist_unique_id), carried to the others over torch.distributed, and every rank calls ltx_dist_init."""
    import importlib

    ltx = importlib.import_module(__package__)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [ltx.dist_unique_id() if rank == 0 else None]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    ctx.dist_init(rank, world, box[0])
    return rank, world


def pair_groups(backend="gloo"):
    """One torch.distributed group per consecutive rank pair (a side channel for the id exchange: gloo, so that creating it makes
    no RCCL communicator of torch's); returns (my_group, pair_index). Every rank must call it."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world % 2 == 0, "CFG pairs need an even number of ranks"
    mine = None
    for p in range(world // 2):
        g = dist.new_group(ranks=[2 * p, 2 * p + 1], backend=backend)
        if rank // 2 == p:
            mine = g
    return mine, rank // 2


class _DevMem:
    """Zero-copy view of raw device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def gloo_allgather_fn(device, group=None):
    """`gather(send_ptr, recv_ptr, nbytes)` over a gloo group: device -> host, all_gather on the CPU, host -> device, with the
    stream synchronised on both sides (the library's contract for a transport that does not enqueue on the context's stream)."""
    world = dist.get_world_size(group)

    def gather(send_ptr, recv_ptr, nbytes):
        send = torch.as_tensor(_DevMem(send_ptr, nbytes), device=device)
        recv = torch.as_tensor(_DevMem(recv_ptr, nbytes * world), device=device)
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, send.cpu(), group=group)  # .cpu() synchronises the current stream
        recv.copy_(torch.cat(parts).to(device))
        torch.cuda.current_stream(device).synchronize()

    return gather


def attach_gloo_transport(ctx, device, group=None):
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx.dist_set_transport(rank, world, gloo_allgather_fn(device, group))
    return rank, world


# ---------------------------------------------------------------------------------------------------------------
# partition rules (pure host logic; the library applies the same rules internally)
# ---------------------------------------------------------------------------------------------------------------
def cfg_branch_for_rank(rank=None):
    rank = dist.get_rank() if rank is None else rank
    return rank % 2  # 0 = negative (uncond), 1 = positive (cond): batch order [neg, pos] (LTXPipeline.swift:715-716)


def shard_tiles(n_tiles, rank=None, world=None):
    """Round-robin assignment of VAE temporal tiles to ranks (tile i on rank i % world)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(range(rank, n_tiles, world))


def sp_token_slice(T, rank=None, world=None):
    """Token range [t0, t1) of a rank: contiguous, equal, in global token order (token t = (f*H + h)*W + w)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    assert T % world == 0 and (T // world) % 8 == 0, f"{T} tokens do not split into {world} equal multiples of 8"
    n = T // world
    return rank * n, (rank + 1) * n


# ---------------------------------------------------------------------------------------------------------------
# protocol model (CPU tests only)
# ---------------------------------------------------------------------------------------------------------------
def broadcast_context(context, mask=None, src=0):
    """One-time broadcast of the text context ([nb,S,3840] bf16 = 7.9 MB per prompt) and its mask."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(context, src=src)
        if mask is not None:
            dist.broadcast(mask, src=src)
    return context, mask


def exchange_velocities(mine, group=None):
    """all-gather of this rank's branch velocity -> (uncond, cond). `mine` [C,F,H,W] f32."""
    world = dist.get_world_size(group)
    assert world == 2, "the CFG pair is sharded over exactly two ranks"
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine.contiguous(), group=group)
    return out[0], out[1]


def apply_cfg(uncond, cond, scale):
    """LatentUtils.swift:131-141 in f32."""
    return cond + (scale - 1.0) * (cond - uncond)


def guidance_rescale(v, cond, phi, eps=1e-8):
    """LatentUtils.swift:164-183 (population variance over all non-batch axes)."""
    if phi <= 0:
        return v
    cfg_std = torch.sqrt(v.var(unbiased=False) + eps)
    cond_std = torch.sqrt(cond.var(unbiased=False) + eps)
    return phi * (v * (cond_std / cfg_std)) + (1.0 - phi) * v


def euler_step(latent, velocity, sigma, sigma_next):
    """LTXScheduler.swift:305-327."""
    den = latent - sigma * velocity
    if sigma_next > 0:
        return den + sigma_next * (latent - den) / sigma
    return den


def denoise_cfg_sharded(latent, sigmas, forward_fn, cfg_scale, rescale=0.0, group=None):
    """The protocol of LTX_SHARD_CFG: rank r evaluates branch r, ONE all-gather per step, the update applied redundantly.
    forward_fn(latent_f32 [1,C,F,H,W], sigma, branch) -> velocity f32 [1,C,F,H,W] for that branch's context."""
    branch = cfg_branch_for_rank(dist.get_rank(group))
    for i in range(len(sigmas) - 1):
        s, sn = float(sigmas[i]), float(sigmas[i + 1])
        mine = forward_fn(latent, s, branch)
        uncond, cond = exchange_velocities(mine, group)
        v = apply_cfg(uncond, cond, cfg_scale)
        v = guidance_rescale(v, cond, rescale)
        latent = euler_step(latent, v, s, sn)
    return latent


def denoise_cfg_single(latent, sigmas, forward_fn, cfg_scale, rescale=0.0):
    """Same loop on one process (both branches locally) - what the sharded form must reproduce."""
    for i in range(len(sigmas) - 1):
        s, sn = float(sigmas[i]), float(sigmas[i + 1])
        uncond, cond = forward_fn(latent, s, 0), forward_fn(latent, s, 1)
        v = apply_cfg(uncond, cond, cfg_scale)
        v = guidance_rescale(v, cond, rescale)
        latent = euler_step(latent, v, s, sn)
    return latent


def gather_tiles_to_rank0(local_tiles, n_tiles, tile_shapes, device, group=None):
    """local_tiles: {tile_index: tensor (F_i,H,W,3)}; returns the ordered list on rank 0 (None elsewhere)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    out = [None] * n_tiles if rank == 0 else None
    for i in range(n_tiles):
        owner = i % world
        if owner == 0:
            if rank == 0:
                out[i] = local_tiles[i]
            continue
        if rank == owner:
            dist.send(local_tiles[i].contiguous(), dst=0, group=group)
        elif rank == 0:
            buf = torch.empty(tile_shapes[i], dtype=torch.float32, device=device)
            dist.recv(buf, src=owner, group=group)
            out[i] = buf
    return out


def blend_tiles(chunks, overlap):
    """decodeWithTemporalTiling's blend on (F,H,W,3) raw tiles (VideoDecoder.swift:561-592) + final clip."""
    po = 8 * overlap
    result = chunks[0]
    for nxt in chunks[1:]:
        rf, nf = result.shape[0], nxt.shape[0]
        if 0 < po < rf and po < nf:
            w = (torch.arange(po, dtype=torch.float32, device=result.device) / po).reshape(po, 1, 1, 1)
            blended = result[rf - po:] * (1 - w) + nxt[:po] * w
            result = torch.cat([result[:rf - po], blended, nxt[po:]], 0)
        else:
            result = torch.cat([result, nxt], 0)
    return torch.clamp((result + 1.0) / 2.0, 0.0, 1.0)


def sp_gather_velocity(vel_local, group=None):
    """[Tn, C] f32 slice of every rank -> [T, C] in global token order on every rank (one all-gather per forward)."""
    world = dist.get_world_size(group)
    parts = [torch.empty(vel_local.shape, dtype=vel_local.dtype) for _ in range(world)]
    dist.all_gather(parts, vel_local.cpu().contiguous(), group=group)
    return torch.cat(parts).to(vel_local.device)
