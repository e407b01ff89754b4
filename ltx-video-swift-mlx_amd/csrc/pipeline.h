// pipeline.h - device-resident denoising loop (reference LTXPipeline.generateVideo loop :800-956 and
// denoise(...) :2191-2401, T2V path) sequenced over the DiT forward and the Euler/CFG/STG/GE kernels.
#pragma once
#include "runtime.h"

typedef void (*ltx_progress_fn)(int step, int total, float sigma, void* user);

struct DenoiseParams {
    float* latent = nullptr;  // device f32 [1][C][F][H][W], in/out (already scaled by sigmas[0])
    int F = 0, H = 0, W = 0;
    const float* sigmas = nullptr;  // host, n_sigmas values (n_sigmas-1 steps)
    int n_sigmas = 0;
    // text conditioning, device. With cfg_scale > 1 the batch is [negative, positive] (LTXPipeline.swift:715-716)
    const bf16_t* context = nullptr;  // [nb][S][caption_channels], nb = cfg>1 ? 2 : 1
    const int32_t* mask = nullptr;    // [nb][S] or null
    int mask_all_ones = 0;
    int S = 0;
    uint64_t ctx_version = 0;
    float cfg_scale = 1.0f;
    float guidance_rescale = 0.0f;
    float stg_scale = 0.0f;
    const int* stg_blocks = nullptr;
    int n_stg = 0;
    float ge_gamma = 0.0f;
    // image-to-video (denoise(...) with conditioningMask / conditionedLatent, LTXPipeline.swift:2191-2401): the encoded image
    // latent [1][C][1][H][W] (device f32; the VAE encoder is outside this path, so it is an input like the text embeddings).
    // Frame 0 is set to it, optionally re-noised per step with cond_noise[step] * image_cond_noise_scale * sigma^2, its tokens
    // run at timestep 0 and the Euler step leaves it untouched.
    const float* cond_latent = nullptr;
    float image_cond_noise_scale = 0.0f;
    const float* cond_noise = nullptr;  // device f32 [n_sigmas-1][C][1][H][W] N(0,1) draws, or null (no injection)
    ltx_progress_fn progress = nullptr;
    void* user = nullptr;
    // multi-GPU CFG sharding hook (config 3): when set, this rank evaluates only branch `cfg_branch`
    // (0 = negative, 1 = positive) and `exchange` must return both velocities (device f32 [2][C*T]) - bench/dist
    // layer provides it via RCCL all-gather. Null = single-GPU batched CFG.
    int cfg_branch = -1;
    void (*exchange)(float* both /*[2][n]*/, const float* mine /*[n]*/, long n, void* user) = nullptr;
    void* exchange_user = nullptr;
};

void denoise_run(ltx_ctx* ctx, const DenoiseParams& p);
