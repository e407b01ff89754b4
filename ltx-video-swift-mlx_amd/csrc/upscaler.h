// upscaler.h - latent spatial upscaler of the two-stage pipeline (reference SpatialUpscaler.swift:181-258 and
// upsampleLatents :352-379): Conv3d 128->mid (zero padding) + GroupNorm(32) + SiLU, 4 ResBlocks, per-frame
// Conv2d mid->4*mid + PixelShuffle(2), 4 ResBlocks, Conv3d mid->128; channels-last throughout.
#pragma once
#include <map>
#include <string>

#include "runtime.h"
#include "vae.h"

struct UpNorm {
    float* w = nullptr;
    float* b = nullptr;
};
struct UpResBlock {
    ConvW conv1, conv2;
    UpNorm norm1, norm2;
};
struct UpscalerModel {
    int in_channels = 128, mid = 1024, blocks_per_stage = 4;
    DeviceArena arena;
    ConvW initial_conv, final_conv, up_conv;  // up_conv: conv2d (9 taps), rows stored (i,j)-major for the pixel shuffle
    UpNorm initial_norm;
    UpResBlock pre[4], post[4];
    struct Slot {
        void* dst = nullptr;
        int kind = 0;  // 0 conv weight (taps x cin relayout), 1 f32 vector
        long numel = 0;
        int cout = 0, cin = 0, taps = 27;
        bool perm4 = false;
        bool loaded = false;
    };
    std::map<std::string, Slot> slots;
    DevBuf h, t, hb, hb2, stats, out_cl;
    long ws_P = 0;
};

UpscalerModel* upscaler_create(int mid_channels);
void upscaler_destroy(UpscalerModel* m);
UpscalerModel* upscaler_load(ltx_ctx* ctx, const std::string& path);
// latent: device f32 [1][128][F][H][W] (normalised); mean/std: device f32 [128] (VAE per-channel statistics);
// out: device f32 [1][128][F][2H][2W] (renormalised)
void upscaler_forward(ltx_ctx* ctx, UpscalerModel* m, const float* latent, int F, int H, int W, const float* mean,
                      const float* std_, float* out);
