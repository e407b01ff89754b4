"""ltx-video-swift-mlx_amd - MI355X-native LTX-2 denoise + VAE-decode path.

The product is ``csrc/build/libltxhip.so`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/ltxhip.h``). This Python package is the thin host-side mirror used by the test-suite and ``bench.py``:
``Context`` wraps an ``ltx_ctx``; device buffers are torch tensors used purely as HBM allocations (their
``data_ptr()`` is what crosses the ABI).

The directory name contains hyphens (it is the repo's mandated package name), so import it with
``importlib.import_module("ltx-video-swift-mlx_amd")`` - ``tests/conftest.py`` and ``bench.py`` do that and alias it
as ``ltx_amd``.
"""
import contextlib
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (ALLGATHER_FN, DIST_ID_BYTES, SHARD_CFG, SHARD_NONE, SHARD_SEQUENCE, ConnectorConfig, PROGRESS_CB, DenoiseOptions,
                   LTXError, TransformerConfig, lib)

__version__ = lib.ltx_version().decode()


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib.ltx_last_error(ctx).decode(errors="replace") if True else ""
        raise LTXError(rc, msg)


def default_transformer_config(**overrides):
    cfg = TransformerConfig()
    lib.ltx_transformer_config_default(C.byref(cfg))
    for k, v in overrides.items():
        if k == "max_pos":
            for i in range(3):
                cfg.max_pos[i] = v[i]
        else:
            setattr(cfg, k, v)
    return cfg


# ---------------------------------------------------------------------------------------------------------------
# pure-host helpers (no GPU)
# ---------------------------------------------------------------------------------------------------------------
def validate_generation_config(width, height, num_frames, num_steps, cfg_scale, two_stage=False):
    """``LTXVideoGenerationConfig.validate()``: raises LTXError.invalidConfiguration with the reference's text."""
    buf = C.create_string_buffer(256)
    rc = lib.ltx_validate_generation_config(width, height, num_frames, num_steps, cfg_scale, int(two_stage), buf, 256)
    if rc != 0:
        raise LTXError(rc, buf.value.decode())


def latent_shape(width, height, num_frames):
    f, h, w = C.c_int(), C.c_int(), C.c_int()
    _check(lib.ltx_latent_shape(width, height, num_frames, C.byref(f), C.byref(h), C.byref(w)))
    return f.value, h.value, w.value


def sigmas(distilled, num_steps, token_count=0):
    out = (C.c_float * 128)()
    n = lib.ltx_sigmas(int(distilled), num_steps, token_count or 0, out, 128)
    if n < 0:
        raise LTXError(-n, "invalid sigma request")
    return np.array(out[:n], dtype=np.float32)


def stage2_sigmas():
    out = (C.c_float * 4)()
    lib.ltx_stage2_sigmas(out, 4)
    return np.array(out[:], dtype=np.float32)


def rope_tables(cfg, F, H, W):
    D = cfg.num_attention_heads * cfg.attention_head_dim
    T = F * H * W
    cos = np.empty((T, D // 2), dtype=np.float32)
    sin = np.empty((T, D // 2), dtype=np.float32)
    _check(lib.ltx_rope_tables(C.byref(cfg), F, H, W, cos.ctypes.data, sin.ctypes.data))
    return cos, sin


def vae_tile_plan(latent_frames, tile, overlap):
    starts, ends = (C.c_int * 256)(), (C.c_int * 256)()
    out = C.c_int()
    n = lib.ltx_vae_tile_plan(latent_frames, tile, overlap, starts, ends, 256, C.byref(out))
    if n < 0:
        raise LTXError(-n, "invalid tiling request")
    return [(starts[i], ends[i]) for i in range(n)], out.value


def _map_key(fn, key):
    buf = C.create_string_buffer(1024)
    rc = fn(key.encode(), buf, 1024)
    if rc < 0:
        raise LTXError(2, "key too long")
    return buf.value.decode() if rc == 1 else None


def map_transformer_key(key):
    return _map_key(lib.ltx_map_transformer_key, key)


def map_vae_key(key):
    return _map_key(lib.ltx_map_vae_key, key)


def map_lora_key(key):
    return _map_key(lib.ltx_map_lora_key, key)


def mlx_random_normal(seed, shape, draw_index=0):
    """MLXRandom.seed(seed); MLXRandom.normal(shape) - the reference's generateNoise (LatentUtils.swift:69-83), restated."""
    out = np.empty(shape, np.float32)
    _check(lib.ltx_mlx_random_normal(int(seed), int(draw_index), out.ctypes.data, out.size))
    return out


def threefry2x32(key, ctr):
    k = np.asarray(key, np.uint32)
    c = np.asarray(ctr, np.uint32)
    o = np.empty(2, np.uint32)
    lib.ltx_threefry2x32(k.ctypes.data, c.ctypes.data, o.ctypes.data)
    return o




def set_option(key, value):
    """``ltx_ctx_set_option`` without a context: the launchers' switch table is process-wide (csrc/options.h)."""
    _check(lib.ltx_ctx_set_option(None, key.encode(), int(value)))


def get_option(key):
    v = C.c_int()
    _check(lib.ltx_ctx_get_option(None, key.encode(), C.byref(v)))
    return v.value


def option_table():
    """[(name, default, min, max, numerics, doc)] of every switch (``ltx_option_info``)."""
    n = lib.ltx_option_info(-1, None, None, None, None, None, None)
    out = []
    for i in range(n):
        name, doc = C.c_char_p(), C.c_char_p()
        d, lo, hi, num = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.ltx_option_info(i, C.byref(name), C.byref(d), C.byref(lo), C.byref(hi), C.byref(num), C.byref(doc))
        out.append((name.value.decode(), d.value, lo.value, hi.value, bool(num.value), doc.value.decode()))
    return out


@contextlib.contextmanager
def options(**kv):
    """Set switches for the duration of a ``with`` block and restore the previous values (tests, A/B runs)."""
    old = {k: get_option(k) for k in kv}
    try:
        for k, v in kv.items():
            set_option(k, v)
        yield
    finally:
        for k, v in old.items():
            set_option(k, v)


def dist_unique_id():
    """``ltx_dist_unique_id``: the 128 bytes one rank creates and the host carries to the others (RCCL's ncclUniqueId)."""
    buf = C.create_string_buffer(DIST_ID_BYTES)
    _check(lib.ltx_dist_unique_id(buf))
    return buf.raw


class _Transport:
    """Python all-gather behind ``ltx_allgather_fn``. ctypes would print and swallow an exception raised inside a callback and the
    C side would carry on with unfilled buffers: here the exception is stashed, the callback returns 1 (the library aborts the call
    with LTX_ERR_GENERATION_FAILED) and ``reraise`` surfaces the original error once the C call has returned."""

    def __init__(self, gather):
        self.error = None

        def _cb(_user, send, recv, nbytes):
            try:
                gather(send, recv, nbytes)
                return 0
            except BaseException as e:  # noqa: BLE001 - must not propagate through the C frames
                self.error = e
                return 1

        self.cfn = ALLGATHER_FN(_cb)

    def reraise(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e


def frames_to_u8(frames):
    """``VideoExporter.tensorToImages`` pixel conversion: uint8(clip(x,0,1)*255), truncating."""
    f = np.ascontiguousarray(frames, dtype=np.float32)
    out = np.empty(f.shape, dtype=np.uint8)
    _check(lib.ltx_frames_to_u8(f.ctypes.data, f.size, out.ctypes.data))
    return out


def write_png(path, rgb):
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = a.shape
    assert c == 3
    _check(lib.ltx_write_png(str(path).encode(), a.ctypes.data, w, h))


def map_vae_encoder_key(key):
    """``mapVAEEncoderWeights`` (ModelDownloader.swift:1222-1283); None for non-encoder tensors."""
    return _map_key(lib.ltx_map_vae_encoder_key, key)


def vae_encoder_latent_frames(T):
    return lib.ltx_vae_encoder_latent_frames(T)


def map_text_encoder_key(key):
    """``mapTextEncoderWeights`` (ModelDownloader.swift:911-968); None = dropped."""
    return _map_key(lib.ltx_map_text_encoder_key, key)


def connector_config(**kw):
    cfg = ConnectorConfig()
    lib.ltx_connector_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def rope_tables_1d(T, dim=3840, theta=10000.0, max_pos=4096):
    cos = np.empty((T, dim // 2), dtype=np.float32)
    sin = np.empty((T, dim // 2), dtype=np.float32)
    _check(lib.ltx_rope_tables_1d(T, dim, theta, max_pos, cos.ctypes.data, sin.ctypes.data))
    return cos, sin


# ---------------------------------------------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------------------------------------------
def _ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return t.ctypes.data
    assert t.is_contiguous()
    return t.data_ptr()


class Context:
    """Owns an ``ltx_ctx``. All ``*_dev`` methods take torch tensors resident on this context's GPU."""

    def __init__(self, device=0, use_torch_stream=True):
        h = C.c_void_p()
        rc = lib.ltx_ctx_create(device, C.byref(h))
        if rc != 0:
            raise LTXError(rc, lib.ltx_last_error(None).decode(errors="replace"))
        self._h = h
        self.device = device
        if use_torch_stream:
            import torch

            self.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def close(self):
        if self._h:
            lib.ltx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        tr = getattr(self, "_transport", None)
        if tr is not None:
            tr.reraise()
        if rc != 0:
            raise LTXError(rc, lib.ltx_last_error(self._h).decode(errors="replace"))

    # ---- multi-GPU group of this context (include/ltxhip.h, "Multi-GPU") ----
    def dist_init(self, rank, world, unique_id):
        """RCCL communicator over the group's ranks (collective call); ``unique_id`` from ``dist_unique_id()`` on one rank."""
        assert len(unique_id) == DIST_ID_BYTES
        self._transport = None
        self._ck(lib.ltx_dist_init(self._h, rank, world, C.c_char_p(bytes(unique_id))))

    def dist_set_transport(self, rank, world, gather):
        """Same membership over a Python all-gather ``gather(send_ptr, recv_ptr, nbytes)`` (tests over gloo)."""
        self._transport = None
        tr = _Transport(gather)
        self._ck(lib.ltx_dist_set_transport(self._h, rank, world, C.cast(tr.cfn, C.c_void_p), None))
        self._transport = tr

    def dist_shutdown(self):
        self._transport = None
        self._ck(lib.ltx_dist_shutdown(self._h))

    def dist_info(self):
        r, w, nat, n = C.c_int(), C.c_int(), C.c_int(), C.c_long()
        self._ck(lib.ltx_dist_info(self._h, C.byref(r), C.byref(w), C.byref(nat), C.byref(n)))
        return {"rank": r.value, "world": w.value, "native": bool(nat.value), "collectives": n.value}

    def dist_allgather_dev(self, send, recv):
        self._ck(lib.ltx_dist_allgather_dev(self._h, _ptr(send), _ptr(recv), send.numel() * send.element_size()))

    def dist_broadcast_dev(self, buf, root=0):
        self._ck(lib.ltx_dist_broadcast_dev(self._h, _ptr(buf), buf.numel() * buf.element_size(), root))

    def dit_export_param(self, key):
        """One resident parameter as an f32 numpy array (flat), e.g. to hand the synthetic weights to the oracle."""
        n = lib.ltx_dit_export_param(self._h, key.encode(), None, 0)
        if n < 0:
            raise LTXError(-n, lib.ltx_last_error(self._h).decode(errors="replace"))
        out = np.empty(n, dtype=np.float32)
        n2 = lib.ltx_dit_export_param(self._h, key.encode(), out.ctypes.data, n)
        if n2 < 0:
            raise LTXError(-n2, lib.ltx_last_error(self._h).decode(errors="replace"))
        return out

    def set_option(self, key, value):
        """``ltx_ctx_set_option``: move one of the launchers' switches (process-wide; the library never reads the environment)."""
        self._ck(lib.ltx_ctx_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int()
        self._ck(lib.ltx_ctx_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def options(self, **kv):
        """``with ctx.options(qk_f32=1): ...`` - set for the block, restore afterwards (the table is process-wide)."""
        return options(**kv)

    def set_stream(self, stream_handle):
        self._ck(lib.ltx_ctx_set_stream(self._h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._ck(lib.ltx_ctx_synchronize(self._h))

    def load_report(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._ck(lib.ltx_load_report(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"loaded": a.value, "missing": b.value, "unmatched": c.value}

    # ---- DiT ----
    def dit_load(self, path, cfg=None, quant_bits=16, group_size=64):
        self._ck(lib.ltx_dit_load(self._h, str(path).encode(), C.byref(cfg) if cfg is not None else None, quant_bits, group_size))

    def dit_init_synthetic(self, cfg=None, seed=1234):
        self._ck(lib.ltx_dit_init_synthetic(self._h, C.byref(cfg) if cfg is not None else None, seed))

    def dit_quantize(self, bits, group_size=64):
        self._ck(lib.ltx_dit_quantize(self._h, bits, group_size))

    def dit_memory_info(self):
        v = [C.c_long() for _ in range(4)]
        self._ck(lib.ltx_dit_memory_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("bf16_weights", "quantised_weights", "scratch", "other"), (x.value for x in v)))

    def fuse_lora(self, path, scale=1.0):
        """``LTXPipeline.fuseLoRA(from:scale:)`` -> number of fused layers."""
        n = C.c_int()
        self._ck(lib.ltx_dit_fuse_lora(self._h, str(path).encode(), scale, C.byref(n)))
        return n.value

    def dit_unload(self):
        self._ck(lib.ltx_dit_unload(self._h))

    def dit_forward(self, latent_bf16, context_bf16, timesteps, mask, F, H, W):
        """Host-pointer path. latent [B,T,C] / context [B,S,Cc] as uint16 bf16 bit arrays, returns f32 [B,T,Cout]."""
        B, T, _ = latent_bf16.shape
        S = context_bf16.shape[1]
        assert T == F * H * W
        out_c = self._out_channels if hasattr(self, "_out_channels") else latent_bf16.shape[2]
        vel = np.empty((B, T, out_c), dtype=np.float32)
        ts = np.ascontiguousarray(timesteps, dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        self._ck(lib.ltx_dit_forward(self._h, _ptr(np.ascontiguousarray(latent_bf16)), _ptr(np.ascontiguousarray(context_bf16)),
                                     _ptr(ts), _ptr(m), B, F, H, W, S, _ptr(vel)))
        return vel

    def dit_forward_tokens(self, latent_bf16, context_bf16, token_timesteps, mask, F, H, W):
        """Per-token timesteps [B,T] (image-to-video, ``prepareTimestep``); otherwise as ``dit_forward``."""
        B, T, _ = latent_bf16.shape
        S = context_bf16.shape[1]
        assert T == F * H * W
        vel = np.empty((B, T, latent_bf16.shape[2]), dtype=np.float32)
        ts = np.ascontiguousarray(token_timesteps, dtype=np.float32)
        assert ts.shape == (B, T)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        self._ck(lib.ltx_dit_forward_tokens(self._h, _ptr(np.ascontiguousarray(latent_bf16)), _ptr(np.ascontiguousarray(context_bf16)),
                                            _ptr(ts), _ptr(m), B, F, H, W, S, _ptr(vel)))
        return vel

    def dit_forward_dev(self, latent, context, timesteps, mask, F, H, W, velocity, ctx_version=0, mask_all_ones=False):
        B, T = latent.shape[0], latent.shape[1]
        S = context.shape[1]
        self._ck(lib.ltx_dit_forward_dev(self._h, _ptr(latent), _ptr(context), _ptr(timesteps), _ptr(mask),
                                         int(mask_all_ones), B, F, H, W, S, ctx_version, _ptr(velocity)))

    def dit_forward_sp_dev(self, latent, context, timesteps, mask, F, H, W, velocity, sp_rank, sp_world, gather, ctx_version=0,
                           mask_all_ones=False):
        """Sequence-parallel forward: `latent` [1,Tn,C] / `velocity` [1,Tn,C] hold this rank's token slice; `gather(send_ptr,
        recv_ptr, nbytes)` all-gathers device memory over the ranks (see dist.sp_allgather_fn). Device tensors."""
        S = context.shape[1]
        tr = _Transport(gather) if gather is not None else None  # None: the context's own transport (dist_init / dist_set_transport)
        rc = lib.ltx_dit_forward_sp_dev(self._h, _ptr(latent), _ptr(context), _ptr(timesteps), _ptr(mask), int(mask_all_ones),
                                        F, H, W, S, ctx_version, sp_rank, sp_world, C.cast(tr.cfn, C.c_void_p) if tr else None, None,
                                        _ptr(velocity))
        if tr:
            tr.reraise()
        self._ck(rc)

    def dit_set_cross_attn_scale(self, scale, first=0, last=-1):
        self._ck(lib.ltx_dit_set_cross_attn_scale(self._h, scale, first, last))

    def dit_set_stg(self, blocks, skip_self_attention=True, skip_feed_forward=False):
        arr = (C.c_int * len(blocks))(*blocks)
        self._ck(lib.ltx_dit_set_stg(self._h, arr, len(blocks), int(skip_self_attention), int(skip_feed_forward)))

    def dit_clear_stg(self):
        self._ck(lib.ltx_dit_clear_stg(self._h))

    # ---- VAE ----
    def vae_load(self, path, config_json=None):
        self._ck(lib.ltx_vae_load(self._h, str(path).encode(), str(config_json).encode() if config_json else None))

    def vae_init_synthetic(self, seed=77, timestep_conditioning=False):
        self._ck(lib.ltx_vae_init_synthetic(self._h, seed, int(timestep_conditioning)))

    def vae_unload(self):
        self._ck(lib.ltx_vae_unload(self._h))

    @property
    def vae_timestep_conditioning(self):
        r = lib.ltx_vae_timestep_conditioning(self._h)
        if r < 0:
            raise LTXError(-r, "Model component not loaded: vaeDecoder")
        return bool(r)

    def vae_decode(self, latent, timestep=None, noise=None, tile=0, overlap=1):
        """Host path: latent [1,128,F,H,W] f32 -> frames (n,32H,32W,3) f32 in [0,1] (decodeVideo)."""
        lat = np.ascontiguousarray(latent, dtype=np.float32)
        _, _, F, H, W = lat.shape
        _, nf = vae_tile_plan(F, tile, overlap)
        out = np.empty((nf, H * 32, W * 32, 3), dtype=np.float32)
        n = C.c_int()
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32)
        self._ck(lib.ltx_vae_decode(self._h, _ptr(lat), F, H, W, int(timestep is not None), float(timestep or 0.0), _ptr(nz),
                                    tile, overlap, _ptr(out), out.size, C.byref(n)))
        return out[:n.value]

    def vae_decode_dev(self, latent, F, H, W, frames, timestep=None, noise=None, tile=0, overlap=1):
        n = C.c_int()
        self._ck(lib.ltx_vae_decode_dev(self._h, _ptr(latent), F, H, W, int(timestep is not None), float(timestep or 0.0),
                                        _ptr(noise), tile, overlap, _ptr(frames), frames.numel(), C.byref(n)))
        return n.value

    def vae_decode_sharded_dev(self, latent, F, H, W, frames, timestep=None, noise=None, tile=0, overlap=1):
        """Tiles round-robin over the context's group, raw tiles broadcast, blend + clip on every rank."""
        n = C.c_int()
        self._ck(lib.ltx_vae_decode_sharded_dev(self._h, _ptr(latent), F, H, W, int(timestep is not None), float(timestep or 0.0),
                                                _ptr(noise), tile, overlap, _ptr(frames), frames.numel(), C.byref(n)))
        return n.value

    def vae_decode_gathered_dev(self, latent, F, H, W, frames, root=0, timestep=None, noise=None, tile=0, overlap=1):
        """Tiles decoded round-robin by the group's ranks, raw tiles sent to `root` only, which blends; frames=None elsewhere."""
        n = C.c_int()
        self._ck(lib.ltx_vae_decode_gathered_dev(self._h, _ptr(latent), F, H, W, int(timestep is not None), float(timestep or 0.0),
                                                 _ptr(noise), tile, overlap, root, _ptr(frames), 0 if frames is None else frames.numel(), C.byref(n)))
        return n.value

    def vae_decode_tile_dev(self, latent, F, H, W, tile, overlap, tile_index, out, timestep=None, noise=None):
        """RAW frames (pre-blend, pre-clip) of one tile of the plan -> out; returns its frame count."""
        n = C.c_int()
        self._ck(lib.ltx_vae_decode_tile_dev(self._h, _ptr(latent), F, H, W, int(timestep is not None), float(timestep or 0.0),
                                             _ptr(noise), tile, overlap, tile_index, _ptr(out), out.numel(), C.byref(n)))
        return n.value

    def vae_res_block_dev(self, group, block, x, F, H, W):
        """One res-block of the loaded decoder on the channels-last f32 stream x [F][H][W][C], in place."""
        self._ck(lib.ltx_vae_res_block_dev(self._h, group, block, _ptr(x), F, H, W))

    def vae_upsample_dev(self, group, x, F, H, W, out):
        """One depth-to-space upsampler of the loaded decoder: x [F][H][W][C] f32 -> out [2F-1][2H][2W][C/2] f32 (device tensors)."""
        self._ck(lib.ltx_vae_upsample_dev(self._h, group, _ptr(x), F, H, W, _ptr(out)))

    def vae_blend_tiles_dev(self, tiles, tile_frames, overlap, H, W, frames):
        n = C.c_int()
        ptrs = (C.c_void_p * len(tiles))(*[t.data_ptr() for t in tiles])
        nfs = (C.c_int * len(tiles))(*tile_frames)
        self._ck(lib.ltx_vae_blend_tiles_dev(self._h, ptrs, nfs, len(tiles), overlap, H, W, _ptr(frames), frames.numel(), C.byref(n)))
        return n.value

    def op_conv3d(self, x, w, bias, out, causal=False):
        F, H, W, Cin = x.shape
        Cout = w.shape[0]
        self._ck(lib.ltx_op_conv3d(self._h, _ptr(x), F, H, W, Cin, _ptr(w), _ptr(bias), Cout, int(causal), _ptr(out)))

    # ---- two-stage glue ----
    # ---- VAE encoder ----
    def vae_encoder_load(self, path, channel_base=0):
        self._ck(lib.ltx_vae_encoder_load(self._h, str(path).encode(), channel_base))

    def vae_encoder_init_synthetic(self, channel_base=0, seed=66):
        self._ck(lib.ltx_vae_encoder_init_synthetic(self._h, channel_base, seed))

    def vae_encoder_unload(self):
        self._ck(lib.ltx_vae_encoder_unload(self._h))

    def vae_encode(self, pixels, normalize=False):
        """``encodeImage`` without the image I/O: pixels [1,3,T,H,W] f32 -> latent [1,128,T',H/32,W/32] f32 (host arrays)."""
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        _, _, T, H, W = px.shape
        out = np.empty((1, 128, lib.ltx_vae_encoder_latent_frames(T), H // 32, W // 32), dtype=np.float32)
        self._ck(lib.ltx_vae_encode(self._h, _ptr(px), T, H, W, int(normalize), _ptr(out)))
        return out

    def vae_encode_dev(self, pixels, latent, normalize=False):
        _, _, T, H, W = pixels.shape
        self._ck(lib.ltx_vae_encode_dev(self._h, _ptr(pixels), T, H, W, int(normalize), _ptr(latent)))

    # ---- text-embedding connector ----
    def connector_load(self, path, cfg=None):
        self._ck(lib.ltx_connector_load(self._h, str(path).encode(), C.byref(cfg) if cfg is not None else None))

    def connector_init_synthetic(self, cfg=None, seed=91):
        self._ck(lib.ltx_connector_init_synthetic(self._h, C.byref(cfg) if cfg is not None else None, seed))

    def connector_unload(self):
        self._ck(lib.ltx_connector_unload(self._h))

    def connector_encode(self, hidden_bits, attention_mask, padding_right=False):
        """``encodeFromHiddenStates``: hidden [states,B,T,dim] uint16 (bf16 bits), mask [B,T] int32 -> (context bits
        [B,T,dim] uint16, mask [B,T] int32). HOST arrays."""
        h = np.ascontiguousarray(hidden_bits, dtype=np.uint16)
        m = np.ascontiguousarray(attention_mask, dtype=np.int32)
        _, B, T, D = h.shape
        out = np.empty((B, T, D), dtype=np.uint16)
        om = np.empty((B, T), dtype=np.int32)
        self._ck(lib.ltx_connector_encode(self._h, _ptr(h), _ptr(m), B, T, int(padding_right), _ptr(out), _ptr(om)))
        return out, om

    def connector_encode_dev(self, hidden, attention_mask, context, out_mask=None, padding_right=False, taps=None):
        """Device tensors: hidden [states,B,T,dim] bf16, mask [B,T] int32, context [B,T,dim] bf16. taps = optional
        (norm_concat, fe_out, after_registers) device tensors for parity tests."""
        _, B, T, _ = hidden.shape
        if taps is not None:
            self._ck(lib.ltx_connector_encode_taps_dev(self._h, _ptr(hidden), _ptr(attention_mask), B, T, int(padding_right),
                                                       _ptr(context), _ptr(taps[0]), _ptr(taps[1]), _ptr(taps[2])))
        else:
            self._ck(lib.ltx_connector_encode_dev(self._h, _ptr(hidden), _ptr(attention_mask), B, T, int(padding_right),
                                                  _ptr(context), _ptr(out_mask)))

    def upscaler_load(self, path):
        self._ck(lib.ltx_upscaler_load(self._h, str(path).encode()))

    def upscaler_unload(self):
        self._ck(lib.ltx_upscaler_unload(self._h))

    def upscale_latent(self, latent):
        """``upsampleLatents``: [1,128,F,H,W] f32 -> [1,128,F,2H,2W] f32 (needs the VAE's statistics loaded)."""
        lat = np.ascontiguousarray(latent, dtype=np.float32)
        _, Cc, F, H, W = lat.shape
        out = np.empty((1, Cc, F, 2 * H, 2 * W), dtype=np.float32)
        self._ck(lib.ltx_upscale_latent(self._h, _ptr(lat), F, H, W, _ptr(out)))
        return out

    def adain_filter_latent(self, latent, reference, factor=1.0):
        lat = np.ascontiguousarray(latent, dtype=np.float32).copy()
        ref = np.ascontiguousarray(reference, dtype=np.float32)
        Cc = lat.shape[1]
        self._ck(lib.ltx_adain_filter_latent(self._h, _ptr(lat), lat[0, 0].size, _ptr(ref), ref[0, 0].size, Cc, factor))
        return lat

    def generate_two_stage(self, noise1, noise2, context_bf16, mask, width, height, num_frames, num_steps=8, vae_tile=0,
                           decode=True, on_progress=None):
        """Host-level mirror of ``generateVideoTwoStage`` (LTXPipeline.swift:2420-2741) for the distilled T2V case:
        stage 1 at W/2 x H/2, latent upscale x2, AdaIN against the stage-1 latent, re-noise with the explicit
        ``noise2`` at sigma 0.909375, 3-step stage-2 refinement (never CFG), decode. Noise tensors are explicit
        inputs (the reference draws them from MLX's global RNG)."""
        validate_generation_config(width, height, num_frames, num_steps, 1.0, two_stage=True)
        F1, H1, W1 = latent_shape(width // 2, height // 2, num_frames)
        F2, H2, W2 = latent_shape(width, height, num_frames)
        sig1 = sigmas(True, num_steps, F1 * H1 * W1)
        lat = np.ascontiguousarray(noise1, dtype=np.float32) * sig1[0]
        lat = self.denoise(lat, sig1, context_bf16, mask, F1, H1, W1, on_progress=on_progress)
        stage1 = lat
        lat = self.upscale_latent(lat)
        lat = self.adain_filter_latent(lat, stage1)
        s2 = stage2_sigmas()
        lat = (np.float32(s2[0]) * np.asarray(noise2, np.float32) + np.float32(1.0 - s2[0]) * lat).astype(np.float32)
        lat = self.denoise(lat, s2, context_bf16, mask, F2, H2, W2, on_progress=on_progress)
        if not decode:
            return lat
        return self.vae_decode(lat, tile=vae_tile)

    # ---- denoise loop ----
    @staticmethod
    def _options(cfg_scale=1.0, guidance_rescale=0.0, stg_scale=0.0, stg_blocks=(29,), ge_gamma=0.0, cond_latent=None,
                 image_cond_noise_scale=0.0, cond_noise=None, shard=SHARD_NONE, step_stats=None):
        """cond_latent / cond_noise (image-to-video): numpy arrays for the host entry point, torch tensors for the device one.
        step_stats: a C-contiguous float32 numpy array [n_steps, 4] that receives (velocity mean, velocity std, latent mean, latent std)
        per step - the reference's --profile diagnostics (LTXPipeline.swift:945-951)."""
        arr = (C.c_int * max(1, len(stg_blocks)))(*stg_blocks)
        if isinstance(cond_latent, np.ndarray):
            cond_latent = np.ascontiguousarray(cond_latent, dtype=np.float32)
        if isinstance(cond_noise, np.ndarray):
            cond_noise = np.ascontiguousarray(cond_noise, dtype=np.float32)
        if step_stats is not None:
            assert isinstance(step_stats, np.ndarray) and step_stats.dtype == np.float32 and step_stats.flags["C_CONTIGUOUS"] and step_stats.shape[-1] == 4
        o = DenoiseOptions(C.sizeof(DenoiseOptions), cfg_scale, guidance_rescale, stg_scale, C.cast(arr, C.POINTER(C.c_int)), len(stg_blocks), ge_gamma,
                           _ptr(cond_latent), image_cond_noise_scale, _ptr(cond_noise), int(shard), _ptr(step_stats))
        o._keep = (arr, cond_latent, cond_noise, step_stats)
        return o

    def denoise(self, latent, sigmas_, context_bf16, mask, F, H, W, on_progress=None, **opts):
        """Host-pointer denoise loop. latent [1,C,F,H,W] f32 (scaled by sigmas[0]); returns the final latent."""
        lat = np.ascontiguousarray(latent, dtype=np.float32).copy()
        sg = np.ascontiguousarray(sigmas_, dtype=np.float32)
        cb = PROGRESS_CB(on_progress if on_progress else (lambda s, t, sig, u: None))
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        c = np.ascontiguousarray(context_bf16)
        o = self._options(**opts)
        self._ck(lib.ltx_denoise(self._h, _ptr(lat), F, H, W, sg.ctypes.data_as(C.POINTER(C.c_float)), len(sg), _ptr(c), _ptr(m),
                                 c.shape[1], C.byref(o), cb, None))
        return lat

    def denoise_dev(self, latent, sigmas_, context, mask, F, H, W, ctx_version=1, mask_all_ones=False, on_progress=None, **opts):
        sg = np.ascontiguousarray(sigmas_, dtype=np.float32)
        cb = PROGRESS_CB(on_progress if on_progress else (lambda s, t, sig, u: None))
        o = self._options(**opts)
        self._ck(lib.ltx_denoise_dev(self._h, _ptr(latent), F, H, W, sg.ctypes.data_as(C.POINTER(C.c_float)), len(sg), _ptr(context),
                                     _ptr(mask), int(mask_all_ones), context.shape[1], ctx_version, C.byref(o), cb, None))

    # ---- live kernel timing ----
    def prof_enable(self, on=True):
        self._ck(lib.ltx_prof_enable(self._h, int(on)))

    def prof_collect(self, kind, reset=False):
        ms, n, w = C.c_double(), C.c_long(), C.c_double()
        self._ck(lib.ltx_prof_collect(self._h, kind, C.byref(ms), C.byref(n), C.byref(w), int(reset)))
        return {"ms": ms.value, "launches": n.value, "work": w.value}

    # ---- kernel-level hooks (device tensors) ----
    def op_gemm(self, A, B, bias=None, act=0, tile_cfg=-1, out_f32=None, out_bf16=None):
        M, K = A.shape
        N = B.shape[0]
        self._ck(lib.ltx_op_gemm_bf16(self._h, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(bias), M, N, K, act, tile_cfg,
                                      _ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
                                      _ptr(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0))

    def op_gemm_q8(self, A, codes, scales, biases, bias, out, split_k=1, via_scratch=False, tile_cfg=30):
        """out[M][N] f32 = A[M][K] bf16 . dequant(codes [N][K] u8, scales / biases [N][K/64] bf16)^T + bias"""
        M, K = A.shape
        N = codes.shape[0]
        self._ck(lib.ltx_op_gemm_q8(self._h, _ptr(A), A.stride(0), _ptr(codes), _ptr(scales), _ptr(biases), _ptr(bias), M, N, K, split_k,
                                    int(via_scratch), tile_cfg, _ptr(out), out.stride(0)))

    def op_value_projection_t(self, X, W, bias, vt):
        tokens, K = X.shape
        self._ck(lib.ltx_op_value_projection_t(self._h, _ptr(X), X.stride(0), tokens, _ptr(W), _ptr(bias), W.shape[0], K, _ptr(vt), vt.stride(0)))

    def op_gemm_gated_residual(self, A, B, bias, gate, gate_scalar, x, mirror=None):
        M, K = A.shape
        N = B.shape[0]
        self._ck(lib.ltx_op_gemm_bf16_gated_residual(self._h, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(bias), _ptr(gate),
                                                     gate_scalar, M, N, K, _ptr(x), x.stride(0), _ptr(mirror),
                                                     mirror.stride(0) if mirror is not None else 0))

    def op_gemm_gated_residual_norm(self, A, B, bias, gate, gate_scalar, x, scale, shift, xn, eps=1e-6, fused=True):
        M, K = A.shape
        N = B.shape[0]
        self._ck(lib.ltx_op_gemm_bf16_gated_residual_norm(self._h, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(bias), _ptr(gate),
                                                          gate_scalar, M, N, K, _ptr(x), x.stride(0), _ptr(scale), _ptr(shift), eps,
                                                          _ptr(xn), xn.stride(0), int(fused)))

    def op_gemv(self, a, W, bias, out, in_act=0):
        M, K = a.shape
        N = W.shape[0]
        self._ck(lib.ltx_op_gemv_f32(self._h, _ptr(a), a.stride(0), _ptr(W), W.stride(0), _ptr(bias), _ptr(out), out.stride(0), M, N, K, in_act))

    def op_attention(self, Q, K, Vt, bias, H, O, scale=None):
        B, Tq, _ = Q.shape
        Tk = K.shape[1]
        ldvt = Vt.shape[2]
        if scale is None:
            scale = 1.0 / (128.0 ** 0.5)
        self._ck(lib.ltx_op_attention(self._h, _ptr(Q), _ptr(K), _ptr(Vt), ldvt, _ptr(bias), B, H, Tq, Tk, scale, _ptr(O)))

    @staticmethod
    def attention_key_splits(B, H, Tq, Tk):
        """Key ranges the attention launcher divides a launch of this shape into (1 = no split)."""
        return lib.ltx_attention_key_splits(B, H, Tq, Tk)

    def op_norm_mod(self, x, scale, shift, out, norm_kind=0, eps=1e-6, round_norm_bf16=False):
        rows, D = x.shape
        self._ck(lib.ltx_op_norm_mod(self._h, _ptr(x), _ptr(scale), _ptr(shift), rows, D, norm_kind, eps, int(round_norm_bf16), _ptr(out)))

    def op_qknorm_rope(self, x, w, cos, sin, T, out, eps=1e-6):
        rows, D = out.shape
        self._ck(lib.ltx_op_qknorm_rope(self._h, _ptr(x), x.stride(0), _ptr(w), _ptr(cos), _ptr(sin), T, rows, D, eps, _ptr(out)))

    def op_fill_normal_bf16(self, t, seed, mean=0.0, std=1.0):
        self._ck(lib.ltx_op_fill_normal_bf16(self._h, _ptr(t), t.numel(), seed, mean, std))

    def op_fill_normal_f32(self, t, seed, mean=0.0, std=1.0):
        self._ck(lib.ltx_op_fill_normal_f32(self._h, _ptr(t), t.numel(), seed, mean, std))


# ---------------------------------------------------------------------------------------------------------------
# bf16 helpers for host arrays (bit patterns as uint16)
# ---------------------------------------------------------------------------------------------------------------
def f32_to_bf16_bits(x):
    """Round-to-nearest-even f32 -> bf16 bit patterns (uint16), numpy."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    return r.reshape(x.shape)


def bf16_bits_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32).reshape(b.shape)
