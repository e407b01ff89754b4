// elementwise.h - HBM-bound helper kernels around the GEMM/attention cores (elementwise.hip).
// Every kernel is vectorised (16 B per lane where the layout allows) and fuses the full chain of pointwise work
// between two matrix products, so each activation tensor is read once and written once.
#pragma once
#include "common.h"

enum { LTX_NORM_RMS = 0, LTX_NORM_LAYER = 1 };

// y = norm(x) * (1 + scale[b]) + shift[b]  -> bf16            (adaLN: LTXTransformerBlock.swift:72-83;
//                                                              final head: LTXTransformer.swift:208-224)
// x: f32 [rows][D]; scale/shift: f32 [B][mod_bstride] rows (b = row / rows_per_batch), nullable (plain norm).
// round_norm_bf16: round norm(x) to bf16 before the modulation (block 0 sees a bf16 residual stream in the
// reference, so MLXFast.rmsNorm returns bf16 there - SURVEY R10 dtype column).
void launch_norm_mod(const float* x, long ldx, const float* scale, const float* shift, long mod_bstride,
                     int rows_per_batch, bf16_t* out, long ldo, int rows, int D, int norm_kind, float eps,
                     int round_norm_bf16, hipStream_t stream, const int32_t* row_map = nullptr);

// q/k RMSNorm across ALL heads with learnable weight, then split-RoPE per head (LTXAttention.swift:179-189,
// LTXRoPE.swift:84-149). x: f32 [rows][ldx] (D valid columns), w: f32 [D], cos/sin: f32 [T][D/2] indexed by
// (row % T) with per-head slices of 64 (nullable -> no RoPE, cross-attention). out: bf16 [rows][ldo].
// q and k of a fused projection in ONE launch (job 1 optional: x1 == nullptr)
void launch_qknorm_rope2(const float* x0, const float* w0, bf16_t* out0, const float* x1, const float* w1, bf16_t* out1,
                         long ldx, long ldo, const float* cosT, const float* sinT, int T, int rows, int D, float eps,
                         hipStream_t stream, float out_scale0 = 1.0f, bool x_bf16 = false);
// out_scale0 / out_scale: multiplier applied to job 0's (the query's) normed + rotated row before its single rounding to bf16. The
// DiT folds the softmax scale and log2(e) into it (kAttnQueryPrescale), so that the attention kernel's scores arrive as base-2
// exponents: one rounding of q as before, and 48 multiplies per key tile less in the kernel (AttnArgs::q_prescaled).
void launch_qknorm_rope(const float* x, long ldx, const float* w, const float* cosT, const float* sinT, int T,
                        bf16_t* out, long ldo, int rows, int D, float eps, hipStream_t stream, float out_scale = 1.0f, bool x_bf16 = false);

// sequence-parallel self-attention: gathered V^T blocks [N][D][Tn] -> one V^T [D][ld] with rank r's keys at columns r*Tn ..
void launch_sp_vt_interleave(const bf16_t* gathered, bf16_t* vt, int N, int D, int Tn, long ld, hipStream_t stream);

void launch_cast_f32_bf16(const float* x, bf16_t* out, long n, hipStream_t stream);
void launch_cast_bf16_f32(const bf16_t* x, float* out, long n, hipStream_t stream);

// Sinusoidal timestep embedding [cos(t f_i), sin(t f_i)], f_i = exp(-ln(10000) i/128), t = sigma*mult
// (LTXTimestepEmbedding.swift:17-54). ts: f32 [n], out: f32 [n][256].
void launch_timestep_embedding(const float* ts, float mult, float* out, int n, int dim, hipStream_t stream);

// mod[b][l][j][d] = table[l][j][d] + ada[b][j][d]   (getAdaValues, LTXTransformerBlock.swift:163-185)
void launch_make_mod(const float* tables, const float* ada, float* mod, int B, int L, int J, int D, hipStream_t stream);

// int mask -> additive f32 bias (1-m)*-10000 (LTXTransformer.swift:141-156)
void launch_mask_to_bias(const int32_t* mask, float* bias, long n, hipStream_t stream);

// latent f32 [B][C][F*H*W] -> tokens bf16 [B][T][C] (patchify + .asType(.bfloat16): LatentUtils.swift:20-34,
// LTXPipeline.swift:815) and the inverse in f32 (LatentUtils.swift:42-54).
void launch_patchify_bf16(const float* latent, bf16_t* tokens, int B, int C, int T, hipStream_t stream);
void launch_unpatchify_f32(const float* tokens, float* latent, int B, int C, int T, hipStream_t stream);

// Euler flow-matching update (LTXScheduler.swift:305-327): den = x - s*v; x' = s_next>0 ? den + s_next*(x-den)/s : den
void launch_euler_step(float* latent, const float* velocity, float sigma, float sigma_next, long n, hipStream_t stream,
                       int skip_hw = 0, int frames = 1);
// image-to-video helpers (LTXPipeline.swift:2092-2094, :2225-2252)
void launch_set_frame0(float* latent, const float* cond, const float* noise, float scale, float sigma2, int C, int F, int HW,
                       hipStream_t stream);
void launch_i2v_rowmap(int32_t* row_map, int B, int T, int first, int G, hipStream_t stream);
// CFG combine v = cond + (scale-1)*(cond-uncond) (LatentUtils.swift:131-141)
void launch_cfg_combine(const float* uncond, const float* cond, float scale, float* out, long n, hipStream_t stream);
// out = a + s*(a - b)   (STG: LTXPipeline.swift:920) ; out = g*(a-b)+b (GE: :924-927) share one kernel
void launch_axpby(const float* a, const float* b, float ca, float cb, float* out, long n, hipStream_t stream);
// GE momentum: out = g*(a - b) + b (LTXPipeline.swift:924-927)
void launch_ge(const float* a, const float* b, float g, float* out, long n, hipStream_t stream);
// per-batch population mean/variance over n elements -> stats[b] = {mean, var}
void launch_mean_var(const float* x, long n_per_batch, int B, float* stats, hipStream_t stream);
// guidance rescale (LatentUtils.swift:164-183): out = phi*cfg*(std_cond/std_cfg) + (1-phi)*cfg
void launch_guidance_rescale(float* cfg, const float* stats_cfg, const float* stats_cond, float phi, long n_per_batch,
                             int B, hipStream_t stream);
// AdaIN (LatentUtils.swift:201-227): per (b,c) over F*H*W; x <- (x-mu)/(sd+1e-8)*sd_ref + mu_ref, blended by factor
void launch_adain(float* x, const float* stats_x, const float* stats_ref, float factor, long n_per_chan, int BC,
                  hipStream_t stream);
void launch_scale_inplace(float* x, float s, long n, hipStream_t stream);
// out = ca*a + cb*b
void launch_lincomb(const float* a, const float* b, float ca, float cb, float* out, long n, hipStream_t stream);

// ---- VAE decoder helpers (channels-last f32 stream [P][C]) ----
// latent [C][P] f32 (+ optional noise blend) * std + mean -> channels-last bf16 [P][C] (VideoDecoder.swift:366-381)
// latent/noise element (c,p) lives at [c*chan_stride + p] (lets a temporal tile be a strided view of the full latent)
void launch_vae_prepare(const float* latent, long chan_stride, const float* noise, float noise_scale, const float* mean,
                        const float* std_, bf16_t* out, int C, long P, hipStream_t stream);
// VAE modulation tables, up to 24 in one launch: out[r][c] = table[r][c] + (te ? te[r][c] : 0) + (r odd ? 1 : 0), rows
// shift, scale+1[, shift2, scale2+1] (VAEResBlock3d modulation, VideoDecoder.swift:93-113; final norm :421-434)
struct VaeModsJob {
    const float* table;
    const float* te;
    float* out;
    int rows, C;
};
struct VaeModsBatch {
    static constexpr int MAX_JOBS = 24;
    VaeModsJob job[MAX_JOBS];
    int n = 0;
};
void launch_vae_make_mods_batch(const VaeModsBatch& b, hipStream_t stream);
// temporal-tile blend (VideoDecoder.swift:561-592): r[i] = r[i]*(1-i/n) + nx[i]*(i/n) for frame i < n
void launch_blend_frames(float* r, const float* nx, int n_frames, long frame_elems, hipStream_t stream);
void launch_clip01(float* x, long n, hipStream_t stream);
// pixel-norm over channels (eps 1e-8) * scale + shift -> SiLU -> bf16 (VideoDecoder.swift:29-32,118-127,419-436).
// scale already contains the +1. x: f32 [P][C].
void launch_pixelnorm_silu(const float* x, const float* scale, const float* shift, bf16_t* out, long P, int C,
                           hipStream_t stream);
// conv_out [F*H*W][ldx] f32 (48 valid channels) -> frames (F, 4H, 4W, 3) f32 = clip((x+1)/2, 0, 1)
// (unpatchify VideoDecoder.swift:257-275 + decodeVideo :501-505). When blend_w >= 0 nothing is blended here;
// temporal-tile blending is done by launch_blend_frames on the (F,H,W,3) tensors.

// ---- latent upscaler helpers (SpatialUpscaler.swift) ----
// GroupNorm statistics over (all positions, C/G channels) per group, population variance: stats[g] = {mean, rstd}
// with rstd = 1/sqrt(var + eps) (UpscalerGroupNorm3D, SpatialUpscaler.swift:14-58). x: f32 [P][C].
void launch_groupnorm_stats(const float* x, long P, int C, int G, float eps, float* stats, hipStream_t stream);
// y = (x - mean_g) * rstd_g * w[c] + b[c]; if resid: y += resid; y = SiLU(y) when act; writes f32 and/or bf16
void launch_groupnorm_apply(const float* x, const float* stats, const float* w, const float* b, const float* resid,
                            int act_silu, float* out_f32, bf16_t* out_bf16, long P, int C, int G, hipStream_t stream);
// x [P][C] f32 -> out [C][P] f32 with (x - mean[c]) / std[c]  (upsampleLatents renormalisation, SpatialUpscaler.swift:352-379)
void launch_upscaler_finish(const float* x, const float* mean, const float* std_, float* out, long P, int C, hipStream_t stream);

// ---- text-embedding connector (LTXTextEncoder.swift:62-122, :428-470) ----
// masked (sum, min, max) over the valid tokens of every (state, batch) plane of hidden [states][B][T][D]; partials per chunk,
// then mean and 8/(range+eps) per plane -> stats[(l*B+b)*2 + {0,1}]
void launch_fe_stats(const bf16_t* hidden, const int32_t* mask, int states, int B, int T, int D, int padding_right, float eps,
                     float* partials, float* stats, hipStream_t stream);
// out[b][t][d*states + l] = valid ? bf16(8*(x-mean)/(range+eps)) : 0
long fe_stats_partials_floats(int states, int B, int T, int D);
void launch_fe_norm_concat(const bf16_t* hidden, const int32_t* mask, const float* stats, int states, int B, int T, int D,
                           int padding_right, float eps, bf16_t* out, hipStream_t stream);
// src[b][p] = token index whose row lands at position p, or -1 for a learnable register (valid tokens first, in order;
// position p keeps a token where reverse(valid)[p] is set)
void launch_register_plan(const int32_t* mask, int B, int T, int32_t* src, hipStream_t stream);
// x[b][p][:] = src >= 0 ? enc[b][src][:] : registers[p % R][:]   (f32 stream)
void launch_register_gather(const bf16_t* enc, const float* registers, const int32_t* src, int B, int T, int D, int R, float* x,
                            hipStream_t stream);
void launch_fill_const_i32(int32_t* p, long n, int32_t v, hipStream_t stream);

// ---- VAE encoder (VideoEncoder.swift) ----
void launch_enc_patchify(const float* pixels, bf16_t* out, int T, int H, int W, hipStream_t stream);
void launch_enc_s2d_residual(const float* conv, int Cc, const float* x, int Cx, float* out, int Cout, int T, int H, int W, int ft,
                             int fh, int fw, hipStream_t stream);
void launch_enc_finish(const float* z, long ldz, const float* mean, const float* stdv, float* out, int C, long P, hipStream_t stream);
