"""ctypes binding of libltxhip.so (include/ltxhip.h).

The shared library is the product; this module only declares its C ABI to Python so that tests, bench.py and the
host-side mirror of the reference's `LTXPipeline` can call it. There is deliberately no fallback: if the library
has not been built (``__graft_entry__.build()`` / ``make -C csrc``) importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTX_LIB selects another build of the same ABI (tools/ and the `experiments`-marked tests use csrc/build_exp/libltxhip_exp.so)
SO_PATH = os.environ.get("LTX_LIB") or os.path.join(_HERE, "csrc", "build", "libltxhip.so")


class LTXError(RuntimeError):
    """Mirror of the reference's ``LTXError`` cases (LTXVideo.swift:66-141) keyed by C status code."""

    NAMES = {
        1: "modelNotLoaded", 2: "invalidConfiguration", 3: "insufficientMemory", 4: "weightLoadingFailed",
        5: "generationFailed", 6: "generationCancelled", 7: "invalidFrameCount", 8: "invalidDimensions",
        9: "fileNotFound", 10: "invalidLoRA", 11: "hipError",
    }

    def __init__(self, code, message):
        self.code = int(code)
        self.case = self.NAMES.get(self.code, "unknown")
        super().__init__(f"LTXError.{self.case}: {message}")


class DenoiseOptions(C.Structure):
    """``ltx_denoise_options``: the sampling knobs of ``LTXVideoGenerationConfig`` (LTXConfig.swift:216-300)."""

    _fields_ = [("struct_size", C.c_uint32), ("cfg_scale", C.c_float), ("guidance_rescale", C.c_float), ("stg_scale", C.c_float),
                ("stg_blocks", C.POINTER(C.c_int)), ("n_stg_blocks", C.c_int), ("ge_gamma", C.c_float),
                ("cond_latent", C.c_void_p), ("image_cond_noise_scale", C.c_float), ("cond_noise", C.c_void_p),
                ("shard", C.c_int), ("step_stats", C.c_void_p)]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_float, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long)  # ltx_allgather_fn
SHARD_NONE, SHARD_CFG, SHARD_SEQUENCE = 0, 1, 2
DIST_ID_BYTES = 128


class ConnectorConfig(C.Structure):
    """``ltx_connector_config`` / reference ``VideoGemmaTextEncoderModel`` defaults (LTXTextEncoder.swift:18-45)."""

    _fields_ = [("dim", C.c_int), ("heads", C.c_int), ("layers", C.c_int), ("registers", C.c_int), ("states", C.c_int),
                ("theta", C.c_float), ("max_pos", C.c_int)]


class TransformerConfig(C.Structure):
    """``ltx_transformer_config`` / reference ``LTXTransformerConfig`` (LTXConfig.swift:83-177)."""

    _fields_ = [
        ("num_layers", C.c_int), ("num_attention_heads", C.c_int), ("attention_head_dim", C.c_int),
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("cross_attention_dim", C.c_int),
        ("caption_channels", C.c_int), ("rope_theta", C.c_float), ("max_pos", C.c_int * 3),
        ("timestep_scale_multiplier", C.c_float), ("norm_eps", C.c_float),
    ]


def _load():
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"libltxhip.so not found at {SO_PATH}; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)")
    return C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _i, _f, _l, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_uint64
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol declared in include/ltxhip.h must appear here (tests check both ways)
SIGNATURES = {
    "ltx_version": (C.c_char_p, []),
    "ltx_abi_version": (_i, []),
    "ltx_ctx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "ltx_ctx_get_option": (_i, [_vp, C.c_char_p, C.POINTER(C.c_int)]),
    "ltx_option_info": (_i, [_i, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                             C.POINTER(C.c_char_p)]),
    "ltx_build_info": (C.c_char_p, []),
    "ltx_transformer_config_default": (None, [C.POINTER(TransformerConfig)]),
    "ltx_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "ltx_ctx_destroy": (None, [_vp]),
    "ltx_last_error": (C.c_char_p, [_vp]),
    "ltx_ctx_set_stream": (_i, [_vp, _vp]),
    "ltx_ctx_synchronize": (_i, [_vp]),
    "ltx_load_report": (_i, [_vp, _ip, _ip, _ip]),
    "ltx_validate_generation_config": (_i, [_i, _i, _i, _i, _f, _i, C.c_char_p, _i]),
    "ltx_latent_shape": (_i, [_i, _i, _i, _ip, _ip, _ip]),
    "ltx_sigmas": (_i, [_i, _i, _i, C.POINTER(C.c_float), _i]),
    "ltx_stage2_sigmas": (_i, [C.POINTER(C.c_float), _i]),
    "ltx_rope_tables": (_i, [C.POINTER(TransformerConfig), _i, _i, _i, _vp, _vp]),
    "ltx_vae_tile_plan": (_i, [_i, _i, _i, _ip, _ip, _i, _ip]),
    "ltx_map_transformer_key": (_i, [C.c_char_p, C.c_char_p, _i]),
    "ltx_map_vae_key": (_i, [C.c_char_p, C.c_char_p, _i]),
    "ltx_map_lora_key": (_i, [C.c_char_p, C.c_char_p, _i]),
    "ltx_st_info": (_i, [C.c_char_p, C.c_char_p, C.POINTER(C.c_long)]),
    "ltx_st_read": (_l, [C.c_char_p, C.c_char_p, _i, _vp, _l]),
    "ltx_dit_load": (_i, [_vp, C.c_char_p, C.POINTER(TransformerConfig), _i, _i]),
    "ltx_dit_init_synthetic": (_i, [_vp, C.POINTER(TransformerConfig), _u64]),
    "ltx_dit_unload": (_i, [_vp]),
    "ltx_dit_quantize": (_i, [_vp, _i, _i]),
    "ltx_dit_fuse_lora": (_i, [_vp, C.c_char_p, _f, _ip]),
    "ltx_dit_export_param": (_l, [_vp, C.c_char_p, _vp, _l]),
    "ltx_dist_unique_id": (_i, [_vp]),
    "ltx_dist_init": (_i, [_vp, _i, _i, _vp]),
    "ltx_dist_set_transport": (_i, [_vp, _i, _i, _vp, _vp]),
    "ltx_dist_shutdown": (_i, [_vp]),
    "ltx_dist_info": (_i, [_vp, _ip, _ip, _ip, C.POINTER(C.c_long)]),
    "ltx_dist_allgather_dev": (_i, [_vp, _vp, _vp, _l]),
    "ltx_dist_broadcast_dev": (_i, [_vp, _vp, _l, _i]),
    "ltx_vae_decode_tile_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _i, _vp, _l, _ip]),
    "ltx_dit_memory_info": (_i, [_vp, C.POINTER(_l), C.POINTER(_l), C.POINTER(_l), C.POINTER(_l)]),
    "ltx_op_gemm_q8": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _l]),
    "ltx_vae_res_block_dev": (_i, [_vp, _i, _i, _vp, _i, _i, _i]),
    "ltx_vae_upsample_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "ltx_vae_blend_tiles_dev": (_i, [_vp, C.POINTER(_vp), _ip, _i, _i, _i, _i, _vp, _l, _ip]),
    "ltx_vae_decode_sharded_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _vp, _l, _ip]),
    "ltx_vae_decode_gathered_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _i, _vp, _l, _ip]),
    "ltx_dit_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ltx_dit_forward_tokens": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ltx_dit_forward_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _u64, _vp]),
    "ltx_dit_forward_sp_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _u64, _i, _i, _vp, _vp, _vp]),
    "ltx_dit_set_cross_attn_scale": (_i, [_vp, _f, _i, _i]),
    "ltx_dit_set_stg": (_i, [_vp, _ip, _i, _i, _i]),
    "ltx_dit_clear_stg": (_i, [_vp]),
    "ltx_vae_load": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "ltx_vae_init_synthetic": (_i, [_vp, _u64, _i]),
    "ltx_vae_unload": (_i, [_vp]),
    "ltx_vae_timestep_conditioning": (_i, [_vp]),
    "ltx_vae_decode": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _vp, _l, _ip]),
    "ltx_vae_decode_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _vp, _l, _ip]),
    "ltx_op_conv3d": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ltx_connector_config_default": (None, [C.POINTER(ConnectorConfig)]),
    "ltx_connector_load": (_i, [_vp, C.c_char_p, C.POINTER(ConnectorConfig)]),
    "ltx_connector_init_synthetic": (_i, [_vp, C.POINTER(ConnectorConfig), C.c_ulong]),
    "ltx_connector_unload": (_i, [_vp]),
    "ltx_connector_encode_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ltx_connector_encode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ltx_connector_encode_taps_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ltx_map_text_encoder_key": (_i, [C.c_char_p, C.c_char_p, _i]),
    "ltx_rope_tables_1d": (_i, [_i, _i, _f, _i, _vp, _vp]),
    "ltx_vae_encoder_load": (_i, [_vp, C.c_char_p, _i]),
    "ltx_vae_encoder_init_synthetic": (_i, [_vp, _i, C.c_ulong]),
    "ltx_vae_encoder_unload": (_i, [_vp]),
    "ltx_vae_encode": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ltx_vae_encode_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ltx_vae_encoder_latent_frames": (_i, [_i]),
    "ltx_map_vae_encoder_key": (_i, [C.c_char_p, C.c_char_p, _i]),
    "ltx_frames_to_u8": (_i, [_vp, _l, _vp]),
    "ltx_mlx_random_normal": (_i, [_u64, _i, _vp, _l]),
    "ltx_threefry2x32": (None, [_vp, _vp, _vp]),
    "ltx_write_png": (_i, [C.c_char_p, _vp, _i, _i]),
    "ltx_upscaler_load": (_i, [_vp, C.c_char_p]),
    "ltx_upscaler_unload": (_i, [_vp]),
    "ltx_upscale_latent": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ltx_upscale_latent_dev": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ltx_adain_filter_latent": (_i, [_vp, _vp, _l, _vp, _l, _i, _f]),
    "ltx_adain_filter_latent_dev": (_i, [_vp, _vp, _l, _vp, _l, _i, _f]),
    "ltx_renoise_dev": (_i, [_vp, _vp, _vp, _f, _l]),
    "ltx_denoise": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.c_float), _i, _vp, _vp, _i, C.POINTER(DenoiseOptions), PROGRESS_CB, _vp]),
    "ltx_denoise_dev": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.c_float), _i, _vp, _vp, _i, _i, _u64, C.POINTER(DenoiseOptions), PROGRESS_CB, _vp]),
    "ltx_prof_enable": (_i, [_vp, _i]),
    "ltx_prof_collect": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double), _i]),
    "ltx_op_gemm_bf16": (_i, [_vp, _vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _l]),
    "ltx_op_value_projection_t": (_i, [_vp, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _l]),
    "ltx_op_gemm_bf16_gated_residual": (_i, [_vp, _vp, _l, _vp, _l, _vp, _vp, _f, _i, _i, _i, _vp, _l, _vp, _l]),
    "ltx_op_gemm_bf16_gated_residual_norm": (_i, [_vp, _vp, _l, _vp, _l, _vp, _vp, _f, _i, _i, _i, _vp, _l, _vp, _vp, _f, _vp, _l, _i]),
    "ltx_op_gemv_f32": (_i, [_vp, _vp, _l, _vp, _l, _vp, _vp, _l, _i, _i, _i, _i]),
    "ltx_op_attention": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _f, _vp]),
    "ltx_attention_key_splits": (_i, [_i, _i, _i, _i]),
    "ltx_op_norm_mod": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "ltx_op_qknorm_rope": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "ltx_op_fill_normal_bf16": (_i, [_vp, _vp, _l, _u64, _f, _f]),
    "ltx_op_fill_normal_f32": (_i, [_vp, _vp, _l, _u64, _f, _f]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

HAS_EXPERIMENTS = b"experiments=1" in lib.ltx_build_info()
