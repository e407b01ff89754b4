"""Multi-GPU layer of the path (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in CPU tests). The reference is single-device; this is new design:

* replicas (independent samples): no data-path collective; only `broadcast_context` once per prompt.
* CFG pair sharding (config 3): rank r evaluates branch r (0 = negative, 1 = positive), the two 786 KB velocities are
  exchanged with ONE all-gather per step and every rank applies CFG + Euler redundantly, so ranks stay bit-identical
  without a second collective. Messages are < 1 MB: latency-bound, so a direct exchange (all_gather over 2 ranks =
  one xGMI hop) rather than a ring.
* VAE temporal tiles (config 5): tiles are independent decoder calls -> round-robin over ranks, gathered to rank 0
  and blended there in tile order (the blend is order-dependent, VideoDecoder.swift:561-592).
* one sample on several GPUs (sequence parallelism): tokens are split into equal contiguous slices; all
  per-token work (projections, norms, FFN, cross-attention against the replicated text keys) is local, each block's self-attention
  all-gathers its K rows and V^T block (2 collectives per block, (T/N)*4096*2 B each per rank), and one all-gather of the velocity
  slices per step lets every rank run the scheduler redundantly. `hip_forward_fn_sp`.

`forward_fn(tokens_bf16_f32, branch, step)` abstracts the DiT forward so the same loop runs on the HIP path
(Context.dit_forward_dev) and, in CPU tests, on a stand-in; the arithmetic around it (CFG, rescale, Euler) is done with
torch ops on whatever device the tensors live on.
"""
import torch
import torch.distributed as dist


def broadcast_context(context, mask=None, src=0):
    """One-time broadcast of the text context ([nb,S,3840] bf16 = 7.9 MB per prompt) and its mask."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(context, src=src)
        if mask is not None:
            dist.broadcast(mask, src=src)
    return context, mask


def cfg_branch_for_rank(rank=None):
    rank = dist.get_rank() if rank is None else rank
    return rank % 2  # 0 = negative (uncond), 1 = positive (cond): batch order [neg, pos] (LTXPipeline.swift:715-716)


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
    """CFG denoise loop with the pair sharded over 2 ranks. Every rank returns the same final latent.

    forward_fn(latent_f32 [1,C,F,H,W], sigma, branch) -> velocity f32 [1,C,F,H,W] for that branch's context.
    """
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
    """Same loop on one process (both branches locally) - the reference the sharded form must reproduce."""
    for i in range(len(sigmas) - 1):
        s, sn = float(sigmas[i]), float(sigmas[i + 1])
        uncond, cond = forward_fn(latent, s, 0), forward_fn(latent, s, 1)
        v = apply_cfg(uncond, cond, cfg_scale)
        v = guidance_rescale(v, cond, rescale)
        latent = euler_step(latent, v, s, sn)
    return latent


def shard_tiles(n_tiles, rank=None, world=None):
    """Round-robin assignment of VAE temporal tiles to ranks."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(range(rank, n_tiles, world))


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


class _DevMem:
    """Zero-copy view of raw device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def sp_allgather_fn(device, group=None):
    """The `gather(send_ptr, recv_ptr, nbytes)` callback of Context.dit_forward_sp_dev over torch.distributed.

    backend nccl (= RCCL): the all-gather is enqueued behind the library's kernels - the context runs on torch's current stream and
    ProcessGroupNCCL orders its own stream after / before it with events, no host synchronisation. backend gloo (tests): device ->
    host, all_gather on the CPU, host -> device, synchronising the stream on both sides."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)

    def gather(send_ptr, recv_ptr, nbytes):
        send = torch.as_tensor(_DevMem(send_ptr, nbytes), device=device)
        recv = torch.as_tensor(_DevMem(recv_ptr, nbytes * world), device=device)
        if backend == "gloo":
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, send.cpu(), group=group)  # .cpu() synchronises the current stream
            recv.copy_(torch.cat(parts).to(device))
            torch.cuda.current_stream(device).synchronize()
        else:
            dist.all_gather_into_tensor(recv, send, group=group)

    return gather


def sp_token_slice(T, rank=None, world=None):
    """Token range [t0, t1) of a rank: contiguous, equal, in global token order (token t = (f*H + h)*W + w)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    assert T % world == 0 and (T // world) % 8 == 0, f"{T} tokens do not split into {world} equal multiples of 8"
    n = T // world
    return rank * n, (rank + 1) * n


def sp_gather_velocity(vel_local, group=None):
    """[Tn, C] f32 slice of every rank -> [T, C] in global token order on every rank (one all-gather per denoise step)."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and vel_local.is_cuda:
        parts = [torch.empty(vel_local.shape, dtype=vel_local.dtype) for _ in range(world)]
        dist.all_gather(parts, vel_local.cpu(), group=group)
        return torch.cat(parts).to(vel_local.device)
    full = torch.empty((world * vel_local.shape[0],) + tuple(vel_local.shape[1:]), dtype=vel_local.dtype, device=vel_local.device)
    dist.all_gather_into_tensor(full, vel_local.contiguous(), group=group)
    return full


def hip_forward_fn_sp(ctx, context, mask, F, H, W, mask_all_ones=False, group=None):
    """`forward_fn` for ONE sample sharded by tokens over all ranks of `group`: every rank keeps the full latent (786 KB at
    768x512x25), evaluates the DiT on its token slice (K / V^T all-gathered inside each block's self-attention), and one final
    all-gather of the [Tn,128] f32 velocity slices gives every rank the full velocity, so the scheduler arithmetic around it runs
    redundantly and ranks stay bit-identical without another collective. `context` [1,S,Cc] bf16, `mask` [1,S] int32 or None."""
    T = F * H * W
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t0, t1 = sp_token_slice(T, rank, world)
    gather = sp_allgather_fn(context.device, group)

    def fwd(latent, sigma, branch=0):
        c = latent.shape[1]
        tokens = latent.reshape(c, T).t()[t0:t1].contiguous().to(torch.bfloat16).reshape(1, t1 - t0, c)
        ts = torch.full((1,), float(sigma), dtype=torch.float32, device=latent.device)
        vel = torch.empty((1, t1 - t0, c), dtype=torch.float32, device=latent.device)
        ctx.dit_forward_sp_dev(tokens, context, ts, mask, F, H, W, vel, rank, world, gather, ctx_version=201 + branch,
                               mask_all_ones=mask_all_ones)
        return sp_gather_velocity(vel[0], group).t().reshape(1, c, F, H, W).contiguous()

    return fwd


def hip_forward_fn(ctx, context, mask, F, H, W, mask_all_ones=False):
    """`forward_fn` for the loops above on the HIP path: patchify -> bf16 -> Context.dit_forward_dev with this branch's slice of
    the [neg, pos] context -> unpatchify. `ctx` is an ltx Context on this rank's GPU, `context` [2,S,Cc] bf16 / `mask` [2,S] int32
    device tensors (broadcast once with `broadcast_context`). The projected context and the cross-attention K/V of each branch
    are cached inside the library under a per-branch version key, so only the first step pays for them."""
    T = F * H * W

    def fwd(latent, sigma, branch):
        c = latent.shape[1]
        tokens = latent.reshape(c, T).t().contiguous().to(torch.bfloat16).reshape(1, T, c)  # patchify (LatentUtils.swift:40-59)
        ts = torch.full((1,), float(sigma), dtype=torch.float32, device=latent.device)
        vel = torch.empty((1, T, c), dtype=torch.float32, device=latent.device)
        m = None if mask is None else mask[branch:branch + 1].contiguous()
        ctx.dit_forward_dev(tokens, context[branch:branch + 1].contiguous(), ts, m, F, H, W, vel, ctx_version=101 + branch,
                            mask_all_ones=mask_all_ones)
        return vel.reshape(T, c).t().reshape(1, c, F, H, W).contiguous()  # unpatchify

    return fwd
