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
    // multi-GPU sharding of this loop over the context's ranks (dist.h; the reference is single-device):
    //   SHARD_CFG: the CFG pair (LTXPipeline.swift:820-865 two B=1 forwards, :2235-2283 one B=2 forward) is split over the ranks -
    //     rank 0 evaluates the negative branch, rank 1 the positive one (rank 2, when the group has three, the STG-perturbed pass);
    //     ONE all-gather of the [C*T] f32 velocities per step, then every rank applies CFG / rescale / STG / GE / Euler redundantly
    //     with the library's own kernels, so the ranks' latents stay bit-identical without a second collective.
    //   SHARD_SEQUENCE: one sample's tokens are split over the ranks (dit_forward's sequence parallelism); one all-gather of the
    //     [T/N][C] velocity slices per forward; scheduler arithmetic redundant on every rank. CFG runs as two sequential B=1
    //     forwards, as generateVideo does (LTXPipeline.swift:829-848).
    int shard = 0;
    // optional HOST array [n_sigmas-1][4]: mean / population std of the guided velocity and of the latent after each step
    // (the reference's --profile diagnostics, LTXPipeline.swift:945-951); the run then ends with a stream synchronisation
    float* step_stats = nullptr;
};

enum { SHARD_NONE = 0, SHARD_CFG = 1, SHARD_SEQUENCE = 2 };

void denoise_run(ltx_ctx* ctx, const DenoiseParams& p);
