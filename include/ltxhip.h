/* ltxhip.h - C ABI of libltxhip.so: the MI355X-native LTX-2 denoise + VAE-decode path.
 *
 * This is the drop-in boundary for the hot path of VincentGourbin/ltx-video-swift-mlx. The reference has no FFI of
 * its own (it is a Swift library calling MLX directly); each entry point below replaces the Swift seam named in
 * its comment (file:line in the reference tree), so that a Swift `LTXPipeline` can keep its public surface and call
 * HIP through this header (see INTEGRATION.md for the module map and the Swift wrapper).
 *
 * Conventions
 *   - plain C types only; bf16 tensors are `uint16_t` bit patterns; all tensors are dense row-major.
 *   - every function returns an `ltx_status` (0 = ok). Codes mirror the reference's `LTXError` cases
 *     (LTXVideo.swift:66-141); `ltx_last_error(ctx)` returns the message the Swift wrapper re-throws.
 *   - ownership: the caller owns every buffer it passes; `ltx_ctx` owns all device memory. Entry points without a
 *     `_dev` suffix take HOST pointers and stage through HBM; `_dev` variants take DEVICE pointers valid on the
 *     context's GPU and run asynchronously on the context's stream (`ltx_ctx_set_stream`).
 *   - threading: one in-flight call per `ltx_ctx` (the reference's `LTXPipeline` is an actor; its models are not
 *     re-entrant: LTXTransformer.swift:30-31, LTXTransformerBlock.swift:117-123, LTXScheduler.swift:49-55).
 *   - the library fails loudly: there is no CPU fallback behind any of these calls.
 */
#ifndef LTXHIP_H
#define LTXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ltx_ctx ltx_ctx;

/* LTXError (LTXVideo.swift:66-141) -> status code */
typedef enum ltx_status {
    LTX_OK = 0,
    LTX_ERR_MODEL_NOT_LOADED = 1,      /* .modelNotLoaded */
    LTX_ERR_INVALID_CONFIGURATION = 2, /* .invalidConfiguration */
    LTX_ERR_INSUFFICIENT_MEMORY = 3,   /* .insufficientMemory */
    LTX_ERR_WEIGHT_LOADING_FAILED = 4, /* .weightLoadingFailed */
    LTX_ERR_GENERATION_FAILED = 5,     /* .generationFailed */
    LTX_ERR_GENERATION_CANCELLED = 6,  /* .generationCancelled */
    LTX_ERR_INVALID_FRAME_COUNT = 7,   /* .invalidFrameCount */
    LTX_ERR_INVALID_DIMENSIONS = 8,    /* .invalidDimensions */
    LTX_ERR_FILE_NOT_FOUND = 9,        /* .fileNotFound */
    LTX_ERR_INVALID_LORA = 10,         /* .invalidLoRA */
    LTX_ERR_HIP = 11                   /* device/runtime failure (no reference counterpart) */
} ltx_status;

/* LTXTransformerConfig (LTXConfig.swift:83-177) */
typedef struct ltx_transformer_config {
    int num_layers;                  /* 48 */
    int num_attention_heads;         /* 32 */
    int attention_head_dim;          /* 128 */
    int in_channels;                 /* 128 */
    int out_channels;                /* 128 */
    int cross_attention_dim;         /* 4096 */
    int caption_channels;            /* 3840 */
    float rope_theta;                /* 10000 */
    int max_pos[3];                  /* 20, 2048, 2048 */
    float timestep_scale_multiplier; /* 1000 */
    float norm_eps;                  /* 1e-6 */
} ltx_transformer_config;

/* ------------------------------------------------------------------------------------------------------------
 * Library / context
 * ---------------------------------------------------------------------------------------------------------- */
/* LTXVideo.version (LTXVideo.swift, asserted by Tests/LTXVideoTests/LTXVideoTests.swift:9-11) */
const char* ltx_version(void);
/* Revision of THIS header's binary interface (struct layouts, entry-point signatures); also the library's soname suffix
 * (libltxhip.so.2). ltx_version() mirrors the reference's framework version and does not move with it.
 *   1  rounds 1-4      2  round 5: ltx_denoise_options.struct_size first, ltx_ctx_set_option / ltx_ctx_get_option, no environment hooks */
#define LTX_ABI_VERSION 2
int ltx_abi_version(void);
/* "gfx950;experiments=0|1": whether this build carries the measured-but-not-selected kernels (-DLTX_EXPERIMENTS; the product does not). */
const char* ltx_build_info(void);
/* Filled with the reference defaults (LTXConfig.swift:83-177). */
void ltx_transformer_config_default(ltx_transformer_config* cfg);
/* Replaces `LTXPipeline.init` device/bookkeeping (LTXPipeline.swift:189-200). Fails with LTX_ERR_HIP when no
 * gfx950 device is usable. */
int ltx_ctx_create(int device, ltx_ctx** out);
void ltx_ctx_destroy(ltx_ctx* ctx);
const char* ltx_last_error(const ltx_ctx* ctx);
/* Run all work of this context on an existing HIP stream (hipStream_t passed as void*). The handle is used as given:
 * NULL selects the default (null) stream. A new context starts on a private non-blocking stream. */
int ltx_ctx_set_stream(ltx_ctx* ctx, void* hip_stream);
int ltx_ctx_synchronize(ltx_ctx* ctx);
/* Tuning / A-B switches of the kernel launchers (csrc/options.h holds the table; `ltx_option_info` enumerates it). The library NEVER
 * reads the environment (the -DLTX_EXPERIMENTS build of tools/ seeds the same table from LTX_<NAME>): this call is the only way to move
 * a switch, so a host's numerics do not depend on who launched it. Switches flagged `numerics` move results by rounding or accumulation
 * order only - e.g. "qk_f32" = 1 and "split_f32" = 1 remove the two roundings this path has beyond the reference's;
 * the others are bit-neutral. The table is process-wide (launchers are shared by all contexts); `ctx` receives the error message.
 * Unknown key or out-of-range value: LTX_ERR_INVALID_CONFIGURATION. No reference counterpart (MLX has no such switches). */
int ltx_ctx_set_option(ltx_ctx* ctx, const char* key, int value);
int ltx_ctx_get_option(const ltx_ctx* ctx, const char* key, int* value);
/* index 0 .. n-1 -> name / default / range / numerics flag / one-line description; returns the number of options (any pointer may be
 * NULL; index out of range: only the count is returned). */
int ltx_option_info(int index, const char** name, int* default_value, int* min_value, int* max_value, int* numerics, const char** doc);
/* Counts of the last load call: tensors applied / model parameters absent from the file (left at the reference's
 * initial values) / mapped file keys with no parameter (dropped), as logged at ModelDownloader.swift:992-1017. */
int ltx_load_report(const ltx_ctx* ctx, int* n_loaded, int* n_missing, int* n_unmatched);

/* ------------------------------------------------------------------------------------------------------------
 * Pure host logic (no GPU needed): shapes, schedules, tables, key mapping
 * ---------------------------------------------------------------------------------------------------------- */
/* LTXVideoGenerationConfig.validate (LTXConfig.swift:310-353) + two-stage %64 rule (LTXPipeline.swift:2443).
 * On failure writes the reference's message into msg (if non-NULL). */
int ltx_validate_generation_config(int width, int height, int num_frames, int num_steps, float cfg_scale,
                                   int two_stage, char* msg, int msg_cap);
/* latentFrames/Height/Width (LTXConfig.swift:356-361; VideoLatentShape.fromPixelDimensions :95-111) */
int ltx_latent_shape(int width, int height, int num_frames, int* latent_frames, int* latent_height,
                     int* latent_width);
/* LTXScheduler.setTimesteps (LTXScheduler.swift:74-182). token_count <= 0 means "not provided".
 * Writes up to cap values, returns the number of sigmas (9 for distilled, num_steps+1 for dev) or <0 on error. */
int ltx_sigmas(int distilled, int num_steps, int token_count, float* out, int cap);
/* STAGE_2_DISTILLED_SIGMA_VALUES (LTXScheduler.swift:31-36); returns 4. */
int ltx_stage2_sigmas(float* out, int cap);
/* createPositionGrid + precomputeFreqsCis(doublePrecision:true) (LTXRoPE.swift:552-610,375-527).
 * cos_out/sin_out: [F*H*W][inner_dim/2] f32; head h owns columns h*64..h*64+63 (the reference's [B,H,T,64]). */
int ltx_rope_tables(const ltx_transformer_config* cfg, int F, int H, int W, float* cos_out, float* sin_out);
/* decodeWithTemporalTiling's tile walk and blended frame count (VideoDecoder.swift:517-592).
 * Returns the number of tiles (<= cap written) or <0 on error. */
int ltx_vae_tile_plan(int latent_frames, int tile, int overlap, int* starts, int* ends, int cap, int* out_frames);
/* mapTransformerKey + loadTransformerWeights filters (ModelDownloader.swift:605-639,756-803). Returns 1 and writes
 * the module key, 0 when the file key is skipped, <0 when out is too small. */
int ltx_map_transformer_key(const char* file_key, char* out, int cap);
/* mapVAEWeights (ModelDownloader.swift:808-899). */
int ltx_map_vae_key(const char* file_key, char* out, int cap);
/* LoRAKeyMapper.loraKeyToModelKey (LoRALoader.swift:209-243). */
int ltx_map_lora_key(const char* lora_key, char* out, int cap);

/* safetensors file access for host tools (the CLI reads PrecomputedEmbeddings / noise tensors through these).
 * ltx_st_info: shape of `key` (up to 8 dims), returns ndim or <0. ltx_st_read: converts to `dtype` (0 f32, 1 bf16 bits,
 * 2 int32) into out (capacity in elements), returns the element count or <0. */
int ltx_st_info(const char* path, const char* key, long* shape8);
long ltx_st_read(const char* path, const char* key, int dtype, void* out, long cap);

/* ------------------------------------------------------------------------------------------------------------
 * DiT (LTXTransformer)
 * ---------------------------------------------------------------------------------------------------------- */
/* Replaces LTXTransformer(config:) + LTXWeightLoader.loadTransformerWeights/applyTransformerWeights
 * (LTXPipeline.swift:293-314, ModelDownloader.swift:605-639,972-1019). cfg NULL = reference defaults.
 * quant_bits: 16 (bf16), 8 (qint8) or 4 (int4) = LTXQuantizationConfig.transformer (LTXQuantizationConfig.swift:19-62):
 * after loading, every Linear weight is affine-quantised per 64-wide group along `in` (LTXPipeline.swift:323-333) and
 * kept as its de-quantised bf16 value (HBM capacity is not the constraint here; the MFMA path stays bf16). */
int ltx_dit_load(ltx_ctx* ctx, const char* safetensors_path, const ltx_transformer_config* cfg, int quant_bits,
                 int group_size);
/* Random-init weights of the given architecture, generated on device (bench / property tests). */
int ltx_dit_init_synthetic(ltx_ctx* ctx, const ltx_transformer_config* cfg, uint64_t seed);
/* quantize(model:groupSize:bits:) applied to an already loaded / synthetic model (LTXPipeline.swift:329). */
int ltx_dit_quantize(ltx_ctx* ctx, int bits, int group_size);
/* Device memory the resident transformer holds, by arena (what MLX's memory statistics would show the host, LTXPipeline.swift:3176-3187
 * estimateMemory): the bf16 Linear weights (0 once quantised: ltx_dit_quantize RELEASES them, as MLXNN.quantize replaces the
 * modules), the codes + group scales / biases, the one de-quantisation scratch matrix, and everything else (biases, norm weights,
 * tables). Any pointer may be NULL. */
int ltx_dit_memory_info(ltx_ctx* ctx, long* bf16_weight_bytes, long* quantised_weight_bytes, long* scratch_bytes, long* other_bytes);
/* Reads one parameter of the resident model back to the HOST as f32 (bf16 values widened; of a quantised model the de-quantised
 * value the GEMMs use). module_key: the reference's module path, e.g. "transformer_blocks.3.attn1.to_q.weight".
 * Returns the element count (written only if <= cap; out NULL = query) or <0. Parity tests use it to hand the on-device
 * synthetic weights of ltx_dit_init_synthetic to the CPU oracle at the reference's full width. */
long ltx_dit_export_param(ltx_ctx* ctx, const char* module_key, float* out, long cap);
/* Replaces LTXPipeline.fuseLoRA(from:scale:) -> Module.fuseLoRA (LTXPipeline.swift:3134-3153, LoRAAdapter.swift:64-166):
 * W' = W + cast(scale * (alpha/rank | 1) * (up @ down)) for every LoRA layer whose mapped key (ltx_map_lora_key) names a
 * Linear weight; on a quantised model dequant -> merge -> requant. *n_fused = number of fused layers. */
int ltx_dit_fuse_lora(ltx_ctx* ctx, const char* lora_path, float scale, int* n_fused);
/* `transformer = nil` (LTXPipeline.swift:989-999) */
int ltx_dit_unload(ltx_ctx* ctx);
/* Replaces transformer(latent:context:timesteps:contextMask:latentShape:) (LTXTransformer.swift:235).
 *   latent   [B][T][in_channels] bf16, T = F*H*W, token t = (f*H+h)*W+w
 *   context  [B][S][caption_channels] bf16
 *   timesteps[B] f32 (sigma; scaled by timestep_scale_multiplier inside)
 *   mask     [B][S] int32 (1 attend, 0 pad) or NULL
 *   velocity [B][T][out_channels] f32 (out)
 * HOST pointers. */
int ltx_dit_forward(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* timesteps,
                    const int32_t* mask, int B, int F, int H, int W, int S, float* velocity);
/* Same with DEVICE pointers, asynchronous on the context stream. ctx_version: non-zero value that changes whenever
 * the context/mask contents change (lets the library keep the projected caption and cross-attention K/V resident
 * across denoise steps - output-identical to recomputing them); 0 = recompute every call.
 * mask_all_ones: caller asserts the mask is all ones (connector output, LTXTextEncoder.swift:622-626). */
int ltx_dit_forward_dev(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* timesteps,
                        const int32_t* mask, int mask_all_ones, int B, int F, int H, int W, int S,
                        uint64_t ctx_version, float* velocity);
/* Sequence-parallel forward of ONE sample over sp_world ranks, one process per GPU (no counterpart in the single-device reference;
 * SURVEY 8(e) "single video on 8 GPUs", 8(f) item 4). Rank sp_rank owns tokens [sp_rank*Tn, (sp_rank+1)*Tn) of the F*H*W grid,
 * Tn = F*H*W / sp_world (must divide, Tn % 8 == 0): `latent` is [1][Tn][in_channels] and `velocity` [1][Tn][out_channels], that
 * rank's rows only; context, timestep and mask are the full ones on every rank. Everything per-token is local; each block's
 * self-attention all-gathers its K rows and V^T block through `gather`, called 2 x num_layers times per forward:
 *   gather(user, send, recv, bytes): all-gather `bytes` from every rank into recv = [sp_world][bytes] in rank order. DEVICE pointers.
 *   It must either enqueue the collective on the context's stream (RCCL on the stream given to ltx_ctx_set_stream) or finish it
 *   before returning (then synchronise that stream first), and return 0; any other value aborts the forward with
 *   LTX_ERR_GENERATION_FAILED (the kernels behind a failed gather would read unfilled buffers). Every rank must call the forward
 *   with the same arguments. gather NULL = the context's own transport (ltx_dist_init: RCCL; ltx_dist_set_transport: host
 *   callback), whose rank/world must equal sp_rank/sp_world.
 * DEVICE pointers, asynchronous on the context stream when the transport is. */
typedef int (*ltx_allgather_fn)(void* user, const void* send, void* recv, long bytes);
int ltx_dit_forward_sp_dev(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* timesteps,
                           const int32_t* mask, int mask_all_ones, int F, int H, int W, int S, uint64_t ctx_version,
                           int sp_rank, int sp_world, ltx_allgather_fn gather, void* user, float* velocity);
/* setCrossAttentionScale (LTXTransformer.swift:497); block range inclusive, (0,-1) = all blocks. */
int ltx_dit_set_cross_attn_scale(ltx_ctx* ctx, float scale, int first_block, int last_block);
/* setSTGSkipFlags / clearSTGSkipFlags (LTXTransformer.swift:512-526) */
int ltx_dit_set_stg(ltx_ctx* ctx, const int* blocks, int n_blocks, int skip_self_attention, int skip_feed_forward);
int ltx_dit_clear_stg(ltx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------------------
 * Video VAE decoder
 * ---------------------------------------------------------------------------------------------------------- */
/* Replaces VideoDecoder() + LTXWeightLoader.loadVAEWeights/applyVAEWeights + parseVAEConfig
 * (LTXPipeline.swift:338-346, ModelDownloader.swift:583-594,648-658,808-899,1022-1064). config_json NULL = look for
 * `config.json` next to the weights; its "timestep_conditioning" flag selects the conditioned decode. */
int ltx_vae_load(ltx_ctx* ctx, const char* safetensors_path, const char* config_json);
/* Random-init decoder of the reference architecture, generated on device (bench / property tests). */
int ltx_vae_init_synthetic(ltx_ctx* ctx, uint64_t seed, int timestep_conditioning);
int ltx_vae_unload(ltx_ctx* ctx);
/* VideoDecoder.timestepConditioning (VideoDecoder.swift:299) of the loaded model: 1/0, <0 on error. */
int ltx_vae_timestep_conditioning(const ltx_ctx* ctx);
/* Replaces decodeVideo(latent:decoder:timestep:temporalTileSize:temporalTileOverlap:) (VideoDecoder.swift:466).
 *   latent  [1][128][F][H][W] f32 (normalised latent, as the denoise loop leaves it)
 *   has_timestep/timestep: pass (1, 0.05) iff the model is timestep-conditioned (LTXPipeline.swift:1004); then
 *           `noise` (same shape as latent, N(0,1)) is REQUIRED - the reference draws it from the global MLX RNG
 *           (VideoDecoder.swift:369), here every noise tensor is an explicit input.
 *   tile/overlap: MemoryOptimizationConfig.vaeTemporalTileSize/Overlap (0 = untiled)
 *   frames_out (n_frames, 32H, 32W, 3) f32 in [0,1]; frames_cap = capacity in floats; *n_frames_out = frames written
 *           (8(F-1)+1 untiled; see ltx_vae_tile_plan for tiled counts, e.g. 26 latent frames, tile 8 -> 180).
 * HOST pointers. */
int ltx_vae_decode(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                   const float* noise, int tile, int overlap, float* frames_out, long frames_cap, int* n_frames_out);
/* Same with DEVICE pointers (n_frames_out stays a host pointer); asynchronous on the context stream. */
int ltx_vae_decode_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                       const float* noise, int tile, int overlap, float* frames_out, long frames_cap,
                       int* n_frames_out);
/* One Conv3dFull (VideoConvolution.swift:202-348) on a channels-last tensor, exposed for parity tests:
 * x [F][H][W][Cin] bf16 (device), w [Cout][27][Cin] bf16 (tap = (kt*3+kh)*3+kw), bias [Cout] f32 -> out [F][H][W][Cout] f32.
 * causal: 0 replicate-pad T on both sides, 1 repeat the first frame twice. */
int ltx_op_conv3d(ltx_ctx* ctx, const uint16_t* x, int F, int H, int W, int Cin, const uint16_t* w, const float* bias,
                  int Cout, int causal, float* out);

/* ------------------------------------------------------------------------------------------------------------
 * Two-stage generation glue (generateVideoTwoStage, LTXPipeline.swift:2420-2741)
 * ---------------------------------------------------------------------------------------------------------- */
/* Replaces loadSpatialUpscaler(from:) (SpatialUpscaler.swift:271-349); mid_channels is detected from the file. */
int ltx_upscaler_load(ltx_ctx* ctx, const char* safetensors_path);
int ltx_upscaler_unload(ltx_ctx* ctx);
/* Replaces upsampleLatents(_:upscaler:latentMean:latentStd:) (SpatialUpscaler.swift:352-379; call site
 * LTXPipeline.swift:2595-2618): denormalise with the VAE's mean_of_means/std_of_means (the VAE must be loaded),
 * SpatialUpscaler forward, renormalise. latent [1][128][F][H][W] f32 -> out [1][128][F][2H][2W] f32. HOST pointers. */
int ltx_upscale_latent(ltx_ctx* ctx, const float* latent, int F, int H, int W, float* out);
int ltx_upscale_latent_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, float* out);
/* adainFilterLatent (LatentUtils.swift:201-227): per-channel mean/std of `latent` [1][C][n] matched to those of
 * `reference` [1][C][n_ref] (population variance, +1e-8 on the std), blended by factor. In place. HOST pointers. */
int ltx_adain_filter_latent(ltx_ctx* ctx, float* latent, long n_per_channel, const float* reference,
                            long n_ref_per_channel, int channels, float factor);
int ltx_adain_filter_latent_dev(ltx_ctx* ctx, float* latent, long n_per_channel, const float* reference,
                                long n_ref_per_channel, int channels, float factor);
/* Stage-2 re-noise (LTXPipeline.swift:2644-2647): latent = sigma*noise + (1-sigma)*latent. DEVICE pointers. */
int ltx_renoise_dev(ltx_ctx* ctx, float* latent, const float* noise, float sigma, long n);

/* One forward with PER-TOKEN timesteps (image-to-video; LTXTransformer.prepareTimestep, LTXTransformer.swift:105-124):
 * token_timesteps [B][T] f32 sigmas. The adaLN path is evaluated once per distinct value of a batch row (at most 8 distinct
 * (batch, value) pairs in total - I2V has two per row: 0 for frame 0 and sigma elsewhere). Other arguments as
 * ltx_dit_forward. HOST pointers. */
int ltx_dit_forward_tokens(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* token_timesteps,
                           const int32_t* mask, int B, int F, int H, int W, int S, float* velocity);

/* ------------------------------------------------------------------------------------------------------------
 * Frame export helpers (the part that is verifiable here; VideoExporter.swift:563-580). Pure host code.
 * ---------------------------------------------------------------------------------------------------------- */
/* MLX-compatible initial noise (replaces generateNoise, LatentUtils.swift:69-83 = MLXRandom.seed(seed) then
 * MLXRandom.normal(shape, float32)): writes the `draw_index`-th keyless normal draw after seeding (0 = the first) as n f32
 * values. mlx-swift is not part of the reference tree: the generator is restated from MLX's published algorithm (threefry2x32-20
 * counter hash, key split per draw, uniform -> sqrt(2)*erfinv). The hash is pinned by the Random123 known-answer vectors
 * (ltx_threefry2x32); the bits -> normal pipeline could not be checked against an MLX run. Pure host code. */
int ltx_mlx_random_normal(uint64_t seed, int draw_index, float* out, long n);
void ltx_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);

/* tensorToImages' pixel conversion: out = uint8(clip(x, 0, 1) * 255), f32 multiply then truncation toward zero. */
int ltx_frames_to_u8(const float* frames, long n, uint8_t* out);
/* Writes one (H, W, 3) uint8 frame as a PNG (8-bit RGB, stored/uncompressed deflate blocks - no external codec). The
 * reference hands CGImages to AVAssetWriter (Apple-only); an MP4 muxer is outside this library. */
int ltx_write_png(const char* path, const uint8_t* rgb, int width, int height);

/* ------------------------------------------------------------------------------------------------------------
 * VAE encoder (VideoEncoder.swift:211-312, call site encodeImage LTXPipeline.swift:1902-1932): turns the
 * image-to-video conditioning image into the latent that ltx_denoise_options.cond_latent takes.
 * ---------------------------------------------------------------------------------------------------------- */
/* Replaces loadVAEEncoder (LTXPipeline.swift:1871-1885): reads the `encoder.*` tensors of the VAE file (key rules
 * ModelDownloader.swift:1222-1283). channel_base 0 = reference (128 -> 128..2048 channels). */
int ltx_vae_encoder_load(ltx_ctx* ctx, const char* safetensors_path, int channel_base);
int ltx_vae_encoder_init_synthetic(ltx_ctx* ctx, int channel_base, unsigned long seed);
int ltx_vae_encoder_unload(ltx_ctx* ctx);
/* pixels [1][3][T][H][W] f32 (the reference feeds images scaled to [-1,1]; image loading/resizing stays with the caller) ->
 * latent [1][128][T'][H/32][W/32] f32, T' = ltx_vae_encoder_latent_frames(T). normalize != 0 applies
 * (latent - mean_of_means) / std_of_means with the loaded decoder's statistics (LTXPipeline.swift:1920-1927; the decoder must
 * be loaded). HOST pointers. */
int ltx_vae_encode(ltx_ctx* ctx, const float* pixels, int T, int H, int W, int normalize, float* latent);
/* Same with DEVICE pointers, asynchronous on the context stream. */
int ltx_vae_encode_dev(ltx_ctx* ctx, const float* pixels, int T, int H, int W, int normalize, float* latent);
int ltx_vae_encoder_latent_frames(int T);
/* mapVAEEncoderWeights (ModelDownloader.swift:1222-1283): length of the module key, 0 if the key is not an encoder tensor. */
int ltx_map_vae_encoder_key(const char* file_key, char* out, int cap);

/* ------------------------------------------------------------------------------------------------------------
 * Text-embedding connector (VideoGemmaTextEncoderModel.encodeFromHiddenStates,
 * LTXTextEncoder.swift:574-643 - call site LTXPipeline.swift:640-700): from the 49 Gemma-3 hidden states to the
 * [B,T,3840] bf16 context + all-ones mask that ltx_denoise / ltx_dit_forward take. The language model itself is not
 * part of this library: its hidden states are the input.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct ltx_connector_config {
    int dim;        /* 3840 = heads x 128 */
    int heads;      /* 30 */
    int layers;     /* 2 */
    int registers;  /* 128 learnable registers */
    int states;     /* 49 hidden states (embedding + 48 layers) */
    float theta;    /* 10000 */
    int max_pos;    /* 4096 */
} ltx_connector_config;
void ltx_connector_config_default(ltx_connector_config* cfg);
/* Replaces the connector part of loadModels (LTXPipeline.swift:420-540, key rules ModelDownloader.swift:911-968): reads
 * `text_embedding_projection.*` / `video_embeddings_connector.*` from a unified checkpoint, or `text_proj_in.*` /
 * `video_connector.*` from a standalone connector file. cfg NULL = reference architecture. */
int ltx_connector_load(ltx_ctx* ctx, const char* safetensors_path, const ltx_connector_config* cfg);
int ltx_connector_init_synthetic(ltx_ctx* ctx, const ltx_connector_config* cfg, unsigned long seed);
int ltx_connector_unload(ltx_ctx* ctx);
/* hidden [states][B][T][dim] bf16, attention_mask [B][T] int32 0/1 (padding_right: 0 = left padding, the reference's
 * default) -> context [B][T][dim] bf16 and out_mask [B][T] int32 (all ones; may be NULL). T must be a multiple of
 * `registers` (the reference aborts otherwise). DEVICE pointers. */
int ltx_connector_encode_dev(ltx_ctx* ctx, const uint16_t* hidden, const int32_t* attention_mask, int B, int T,
                             int padding_right, uint16_t* context, int32_t* out_mask);
/* HOST-pointer variant of the same call. */
int ltx_connector_encode(ltx_ctx* ctx, const uint16_t* hidden, const int32_t* attention_mask, int B, int T,
                         int padding_right, uint16_t* context, int32_t* out_mask);
/* Parity taps (DEVICE pointers, any may be NULL): the normalised concat [B][T][dim*states] bf16, the feature-extractor
 * output [B][T][dim] bf16 and the stream after register replacement [B][T][dim] f32. */
int ltx_connector_encode_taps_dev(ltx_ctx* ctx, const uint16_t* hidden, const int32_t* attention_mask, int B, int T,
                                  int padding_right, uint16_t* context, uint16_t* norm_concat, uint16_t* fe_out,
                                  float* after_registers);
/* mapTextEncoderWeights (ModelDownloader.swift:911-968): returns the length of the module key, 0 if the key is dropped. */
int ltx_map_text_encoder_key(const char* file_key, char* out, int cap);
/* Connector RoPE tables (positions 0..T-1, one axis, f64 math): cos/sin [T][dim/2] f32, before the cast to bf16. */
int ltx_rope_tables_1d(int T, int dim, float theta, int max_pos, float* cos_out, float* sin_out);

/* ------------------------------------------------------------------------------------------------------------
 * Denoising loop
 * ---------------------------------------------------------------------------------------------------------- */
/* GenerationProgressCallback (LTXPipeline.swift:50-72): invoked synchronously on the calling thread once per step,
 * before that step's forward (LTXPipeline.swift:805-810). */
typedef void (*ltx_progress_cb)(int current_step, int total_steps, float sigma, void* user);

/* LTXVideoGenerationConfig's sampling knobs (LTXConfig.swift:216-300) for one denoise run.
 * ABI revision 2: the struct starts with its own size as the CALLER compiled it. The library reads a field only when struct_size covers
 * it, so a host built against an older header (a shorter struct) stays safe when fields are appended; struct_size == 0 or smaller than
 * the revision-2 core (everything up to and including `shard`) is rejected with LTX_ERR_INVALID_CONFIGURATION. Initialise with
 * LTX_DENOISE_OPTIONS_INIT (C / C++) or MemoryLayout<ltx_denoise_options>.size (Swift). */
typedef struct ltx_denoise_options {
    uint32_t struct_size;   /* sizeof(ltx_denoise_options) in the caller's translation unit */
    float cfg_scale;        /* > 1 enables CFG; context batch is then [negative, positive] (LatentUtils.swift:104-117) */
    float guidance_rescale; /* phi (LatentUtils.swift:164-183) */
    float stg_scale;        /* STG (LTXPipeline.swift:897-921) */
    const int* stg_blocks;  /* default [29] */
    int n_stg_blocks;
    float ge_gamma;         /* GE momentum (LTXPipeline.swift:924-927) */
    /* Image-to-video (denoise(...) with conditioningMask / conditionedLatent, LTXPipeline.swift:2191-2401).
     * cond_latent: the encoded conditioning image [1][C][1][H][W] f32, in the same memory space as `latent` (NULL = text-to-video).
     * The VAE encoder is outside this library, so the image latent is an input. Frame 0 of `latent` is set to it (:2092-2094),
     * its tokens run at timestep 0 (:2237-2252) and the Euler step leaves it untouched (:2344-2357).
     * cond_noise: optional N(0,1) draws [n_sigmas-1][C][1][H][W] for the per-step re-noising of frame 0,
     * frame0 = cond + image_cond_noise_scale * cond_noise[step] * sigma^2 (:2225-2229); NULL = no re-noising. */
    const float* cond_latent;
    float image_cond_noise_scale;
    const float* cond_noise;
    /* Multi-GPU sharding of the loop over the ranks of this context's group (ltx_dist_init / ltx_dist_set_transport; the reference is
     * single-device - SURVEY 8(e)). Every rank calls ltx_denoise(_dev) with the same arguments and ends with the same latent, bit
     * for bit: CFG / rescale / STG / GE / Euler run redundantly on every rank with the library's own kernels.
     *   LTX_SHARD_NONE     (0) this context evaluates everything.
     *   LTX_SHARD_CFG      (1) the CFG pair is split: rank 0 = negative branch, rank 1 = positive branch (the two B=1 forwards of
     *                          LTXPipeline.swift:829-848 / the B=2 forward of :2236-2264), rank 2 of a three-rank group = the STG
     *                          pass; ONE all-gather of the [C*T] f32 velocities per step. Needs cfg_scale > 1 and 2 (or 3) ranks.
     *   LTX_SHARD_SEQUENCE (2) one sample's tokens are split into equal contiguous slices (F*H*W divisible by the group size into
     *                          multiples of 8): per transformer block one all-gather of K rows and one of V^T, per forward one
     *                          all-gather of the velocity slices. */
    int shard;
    /* Per-step diagnostics, optional: a HOST array of 4 * (n_sigmas - 1) floats that receives, for every step, the mean and the
     * (population) standard deviation of the guided velocity and of the latent after the Euler update - the numbers the reference logs
     * under --profile ("Step i: ... vel mean=, std=, latent mean=, std=", LTXPipeline.swift:945-951). With it the call ends with a
     * stream synchronisation (the device entry point is otherwise asynchronous). NULL = none. A maintainer with the reference on a Mac
     * can compare these lines for the same weights, embeddings and noise without touching the reference's code. */
    float* step_stats;
} ltx_denoise_options;
/* size set, CFG / rescale / STG / GE off, text-to-video, no sharding, no diagnostics */
#define LTX_DENOISE_OPTIONS_INIT {(uint32_t)sizeof(ltx_denoise_options), 1.0f, 0.0f, 0.0f, NULL, 0, 0.0f, NULL, 0.0f, NULL, 0, NULL}
enum { LTX_SHARD_NONE = 0, LTX_SHARD_CFG = 1, LTX_SHARD_SEQUENCE = 2 };

/* Replaces the denoise loop of generateVideo (LTXPipeline.swift:800-956) / denoise(...) (:2191-2401), T2V.
 *   latent  [1][C][F][H][W] f32, in/out; must already be scaled by sigmas[0] (LTXPipeline.swift:793)
 *   sigmas  HOST array of n_sigmas values (n_sigmas-1 Euler steps), e.g. from ltx_sigmas
 *   context [nb][S][caption_channels] bf16, mask [nb][S] int32 or NULL; nb = 2 ([neg,pos]) iff cfg_scale > 1
 * HOST pointers for latent/context/mask. */
int ltx_denoise(ltx_ctx* ctx, float* latent, int F, int H, int W, const float* sigmas, int n_sigmas,
                const uint16_t* context, const int32_t* mask, int S, const ltx_denoise_options* opt,
                ltx_progress_cb cb, void* user);
/* Same with DEVICE pointers for latent/context/mask (sigmas stays a host array); asynchronous on the context
 * stream. ctx_version / mask_all_ones as in ltx_dit_forward_dev. */
int ltx_denoise_dev(ltx_ctx* ctx, float* latent, int F, int H, int W, const float* sigmas, int n_sigmas,
                    const uint16_t* context, const int32_t* mask, int mask_all_ones, int S, uint64_t ctx_version,
                    const ltx_denoise_options* opt, ltx_progress_cb cb, void* user);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, one ltx_ctx per process, RCCL over xGMI called by the library itself (librccl is loaded
 * on the first ltx_dist_* call; no PyTorch on this path). The reference is a single-device program: this section has no Swift
 * counterpart, it shards what SURVEY 8(e) says shards - the CFG pair (LTXPipeline.swift:820-865, :2235-2283), one sample's
 * tokens, the VAE's temporal tiles (VideoDecoder.swift:517-592). Independent samples need nothing from here (replicas).
 * ---------------------------------------------------------------------------------------------------------- */
#define LTX_DIST_ID_BYTES 128
/* ncclGetUniqueId: called on ONE rank; the host carries the 128 bytes to the other ranks of the group (any channel: MPI, a TCP
 * store, a file) and every rank passes them to ltx_dist_init. */
int ltx_dist_unique_id(void* id_out);
/* ncclCommInitRank on the context's device; collective over the group's ranks. A group is whatever set of contexts shares one
 * id: the CFG pair of one sample (2 ranks), the ranks decoding one video's tiles, all 8 GPUs of one sequence-parallel sample. */
int ltx_dist_init(ltx_ctx* ctx, int rank, int world, const void* id);
/* The same group membership with a caller-supplied all-gather instead of RCCL (contract as ltx_allgather_fn above). For hosts
 * that already own a transport, and for world-size-2 tests where RCCL cannot run (two processes on one GPU). */
int ltx_dist_set_transport(ltx_ctx* ctx, int rank, int world, ltx_allgather_fn gather, void* user);
int ltx_dist_shutdown(ltx_ctx* ctx);
/* rank / world of the context's group (0 / 1 without one); *native = 1 when the transport is RCCL; *n_collectives = collectives
 * enqueued so far. Any pointer may be NULL. */
int ltx_dist_info(const ltx_ctx* ctx, int* rank, int* world, int* native_transport, long* n_collectives);
/* Collectives on the context's stream for the host's own exchanges (the one-time broadcast of the text context).
 * DEVICE pointers. recv = [world][bytes] in rank order. */
int ltx_dist_allgather_dev(ltx_ctx* ctx, const void* send, void* recv, long bytes);
int ltx_dist_broadcast_dev(ltx_ctx* ctx, void* buf, long bytes, int root);

/* Tiled VAE decode as building blocks (decodeWithTemporalTiling, VideoDecoder.swift:517-602), for hosts that place tiles on GPUs
 * themselves. ltx_vae_decode_tile_dev: the RAW frames - before blending and before the (x+1)/2 clip of :501-505 - of tile
 * `tile_index` of the plan (tile, overlap) over `latent` [1][128][F][H][W] -> tile_out (n, 32H, 32W, 3) f32, n = 8(len-1)+1.
 * Blending clipped tiles is not the reference: raw frames are what travels between GPUs. DEVICE pointers. */
int ltx_vae_decode_tile_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                            const float* noise, int tile, int overlap, int tile_index, float* tile_out, long tile_cap,
                            int* n_frames_out);
/* One VAEResBlock3d of the loaded decoder (VideoDecoder.swift:75-131: PixelNorm -> scale/shift -> SiLU -> conv1 -> PixelNorm ->
 * scale/shift -> SiLU -> conv2 -> + x; no timestep conditioning) applied in place to a channels-last f32 stream x [F][H][W][C],
 * C = channels of up-block group `group` (0..3: 1024, 512, 256, 128), block 0..4. Runs the very kernels and fused epilogues of the
 * decode, so a host (or a parity test) can check one stage at full resolution without a full-size reference decode. DEVICE pointer. */
int ltx_vae_res_block_dev(ltx_ctx* ctx, int group, int block, float* x, int F, int H, int W);
/* One VAEDepthToSpaceUpsample3d of the loaded decoder (VideoDecoder.swift:201-251: conv C -> 4C, depth-to-space (2,2,2), drop the first
 * frame, + the channel-tiled depth-to-space of the input) - the stage behind up-block group `group` (0..2: 1024, 512, 256 channels in,
 * half of that out) - on a caller-supplied channels-last f32 stream x [F][H][W][C] -> out [2F-1][2H][2W][C/2]. Runs the decode's own
 * conv kernel and fused depth-to-space epilogue, so one upsampler can be checked at full resolution. DEVICE pointers. */
int ltx_vae_upsample_dev(ltx_ctx* ctx, int group, const float* x, int F, int H, int W, float* out);
/* Blend raw tiles in tile order over 8*overlap frames (VideoDecoder.swift:561-592), then clip((x+1)/2, 0, 1) (:501-505).
 * tiles: HOST array of n_tiles DEVICE pointers, tile_frames their frame counts; tiles[0] may be frames_out. */
int ltx_vae_blend_tiles_dev(ltx_ctx* ctx, const float* const* tiles, const int* tile_frames, int n_tiles, int overlap, int H,
                            int W, float* frames_out, long frames_cap, int* n_frames_out);
/* ltx_vae_decode_dev with the tiles decoded round-robin by the ranks of the context's group (tile i on rank i % world), each raw
 * tile broadcast from its owner, and the blend + clip done on every rank: every rank returns the full frames, bit-identical to
 * the single-GPU ltx_vae_decode_dev. Every rank must call it with the same arguments. */
int ltx_vae_decode_sharded_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                               const float* noise, int tile, int overlap, float* frames_out, long frames_cap,
                               int* n_frames_out);
/* The gather form of the same: every raw tile travels ONCE, from its owner to rank `root` (ncclSend / ncclRecv over xGMI), and only
 * `root` blends, clips and fills frames_out (bit-identical to ltx_vae_decode_dev); the other ranks pass frames_out = NULL and get the
 * frame count. This is what a host that exports one video wants (decodeVideo returns ONE tensor, VideoDecoder.swift:466-507): with
 * the 4 tiles of 768x512x201 (tile 8, overlap 1 - the plan is fixed by VideoDecoder.swift:534-548 and must not be changed to fill
 * more GPUs: it decides the blended frames) on 8 ranks, 0.8 GB reach one rank instead of 1 GB reaching all eight. Every rank of the
 * group must call it with the same arguments. */
int ltx_vae_decode_gathered_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                                const float* noise, int tile, int overlap, int root, float* frames_out, long frames_cap,
                                int* n_frames_out);

/* ------------------------------------------------------------------------------------------------------------
 * Live kernel timing (replaces the reference's wall-clock GenerationTimings, LTXVideo.swift:255-297, with HIP
 * events recorded on the launch stream around every launch of a kernel family). kind: 0 GEMM, 1 attention,
 * 2 conv3d, 3 elementwise. work = algorithmic FLOPs of the timed launches.
 * ---------------------------------------------------------------------------------------------------------- */
int ltx_prof_enable(ltx_ctx* ctx, int on);
/* Synchronises the stream, folds finished events into the totals and returns them; reset != 0 clears afterwards. */
int ltx_prof_collect(ltx_ctx* ctx, int kind, double* total_ms, long* launches, double* work, int reset);

/* ------------------------------------------------------------------------------------------------------------
 * Kernel-level entry points (DEVICE pointers). These expose the individual gfx950 kernels so that parity tests can
 * pin each one against the oracle; they are not needed by a pipeline caller.
 * ---------------------------------------------------------------------------------------------------------- */
/* C[M,N] = A[M,K] . B[N,K]^T (+bias[N]) ; act: 0 none, 1 gelu-tanh, 2 silu; tile_cfg: -1 auto, else a forced tile configuration
 * (0,1,3,4: two-stage 4-wave tiles; 21,23,25: 8-wave LDS-ring tiles; 75: 192x256 one-wave-per-SIMD tile; 41,42,71-74,90
 * only in the experiments build; S*100 + ring cfg: the
 * same with an S-way deterministic split of the K reduction, as the VAE's 1024-channel convolutions use).
 * Exactly one of out_f32/out_bf16 may be NULL. */
int ltx_op_gemm_bf16(ltx_ctx* ctx, const uint16_t* A, long lda, const uint16_t* B, long ldb, const float* bias, int M,
                     int N, int K, int act, int tile_cfg, float* out_f32, long ld_f32, uint16_t* out_bf16,
                     long ld_bf16);
/* out = A . dequant(codes)^T + bias for a few-row launch (M <= 256) on an affine-quantised Linear as MLX's QuantizedLinear holds it
 * (LTXQuantizationConfig.swift:19-62): 8-bit codes [N][K], bf16 scale / bias per 64-wide group [N][K/64], w' = bf16(q * scale + bias).
 * via_scratch = 0: the codes are de-quantised in the GEMM's B stage (what a quantised DiT does at few tokens, e.g. 256x256x9);
 * via_scratch = 1: codes -> bf16 scratch matrix -> the same kernel (what every other launch does). Same bits either way.
 * split_k >= 1: deterministic split of the K reduction. tile_cfg: 30 = the few-row kernel (M <= 128; operand rings of their own, codes
 * staged 15 K-tiles ahead), 29 = the 128x64 ring kernel. */
int ltx_op_gemm_q8(ltx_ctx* ctx, const uint16_t* A, long lda, const uint8_t* codes, const uint16_t* scales, const uint16_t* biases,
                   const float* bias, int M, int N, int K, int split_k, int via_scratch, int tile_cfg, float* out_f32, long ld_f32);
/* vt[n][t] = bf16(sum_k X[t][k] W[n][k] + bias[n]): the value projection in the transposed layout the attention op takes
 * (to_v of LTXAttention.swift:100-104 followed by the head split; columns t >= tokens of vt are left untouched), ldvt % 4 == 0 */
int ltx_op_value_projection_t(ltx_ctx* ctx, const uint16_t* X, long ldx, int tokens, const uint16_t* W, const float* bias, int out_features,
                              int in_features, uint16_t* vt, long ldvt);
/* x[m][n] += gate[n] * (A.B^T + bias) in place on an f32 stream, optional bf16 mirror of the result */
int ltx_op_gemm_bf16_gated_residual(ltx_ctx* ctx, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                    const float* bias, const float* gate, float gate_scalar, int M, int N, int K,
                                    float* x, long ldx, uint16_t* mirror_bf16, long ld_mirror);
/* the same update followed by the adaLN pass over the updated rows: xn[m][:] = bf16(rms_norm(x[m][:]) * (1 + scale) + shift)
 * (LTXTransformerBlock.swift:72-92: a block's last residual update and the next block's first norm-modulate). fused != 0: the way the
 * DiT graph launches the pair (the norm rides on the GEMM's split-K finish pass where the launch has one); fused == 0: two launches.
 * Both give the same bits. */
int ltx_op_gemm_bf16_gated_residual_norm(ltx_ctx* ctx, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                         const float* bias, const float* gate, float gate_scalar, int M, int N, int K,
                                         float* x, long ldx, const float* scale, const float* shift, float eps,
                                         uint16_t* xn, long ldxn, int fused);
/* out[m][n] = sum_k in_act(a[m][k]) W[n][k] + bias[n], f32 x bf16 -> f32, M <= 8 */
int ltx_op_gemv_f32(ltx_ctx* ctx, const float* a, long lda, const uint16_t* W, long ldw, const float* bias, float* out,
                    long ldo, int M, int N, int K, int in_act);
/* O = softmax(Q K^T * scale + bias) V ; Q [B][Tq][H*128], K [B][Tk][H*128], Vt [B][H*128][ldvt] (keys contiguous,
 * ldvt >= roundup(Tk,64)), bias [B][Tk] f32 or NULL, O [B][Tq][H*128] ; all bf16. scale <= 0: Q already carries
 * (1/sqrt(128)) * log2(e) - what the DiT's q-norm + RoPE pass writes, one rounding to bf16 - and the scores are base-2 exponents:
 * O = softmax_2(Q K^T + bias * log2(e)) V. */
int ltx_op_attention(ltx_ctx* ctx, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, long ldvt,
                     const float* bias, int B, int H, int Tq, int Tk, float scale, uint16_t* O);
/* Number of key ranges the attention launcher divides a launch of this shape into when it has a workspace (1 = it does not): few
 * queries against many keys - the cross-attention of a 128-token clip is 32 workgroups walking 16 key tiles each - run one workgroup
 * per (query block, head, key range) and a combine pass (attention.h, AttnArgs::split_ws). ltx_op_attention and the DiT forward
 * both lend that workspace. No reference counterpart: MLXFast.scaledDotProductAttention (LTXAttention.swift:209) is one call. */
int ltx_attention_key_splits(int B, int H, int Tq, int Tk);
/* adaLN: out = norm(x) * (1+scale) + shift -> bf16 ; norm_kind 0 RMS, 1 LayerNorm; scale/shift [D] or NULL */
int ltx_op_norm_mod(ltx_ctx* ctx, const float* x, const float* scale, const float* shift, int rows, int D,
                    int norm_kind, float eps, int round_norm_bf16, uint16_t* out);
/* q/k RMSNorm (weight w[D]) + split RoPE (cos/sin [T][D/2] or NULL) -> bf16 */
int ltx_op_qknorm_rope(ltx_ctx* ctx, const float* x, long ldx, const float* w, const float* cos_t, const float* sin_t,
                       int T, int rows, int D, float eps, uint16_t* out);
/* deterministic device fills used by bench.py to create synthetic inputs */
int ltx_op_fill_normal_bf16(ltx_ctx* ctx, uint16_t* p, long n, uint64_t seed, float mean, float stddev);
int ltx_op_fill_normal_f32(ltx_ctx* ctx, float* p, long n, uint64_t seed, float mean, float stddev);

#ifdef __cplusplus
}
#endif
#endif /* LTXHIP_H */
