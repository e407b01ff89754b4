// vae_encoder.h - the LTX-2 video VAE *encoder* (reference Models/VAE/VideoEncoder.swift:211-312) that turns the
// image-to-video conditioning image into the latent of frame 0 (encodeImage, LTXPipeline.swift:1902-1932). Same building
// blocks as the decoder graph (vae.h): channels-last activations, f32 residual stream, bf16 conv inputs produced by the fused
// pixel-norm + SiLU pass, implicit-GEMM conv3d (pad mode 3: zeros in H/W, first frame replicated in T) - plus the
// space-to-depth downsampler with its group-mean residual.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "runtime.h"
#include "vae.h"

struct VaeEncoderModel {
    int base = 128;              // channel ladder base, base*2, ... base*16 (reference: 128 .. 2048)
    int ch[5] = {128, 256, 512, 1024, 2048};
    DeviceArena arena;
    ConvW conv_in, conv_out;     // conv_in input channels padded 48 -> 64; conv_out keeps the first 128 of 129 rows (the mean)
    struct Res { ConvW c1, c2; };
    std::vector<Res> down[4];
    ConvW ds[4];
    Res mid[2];
    struct Slot {
        void* dst = nullptr;
        int kind = 0;  // 0 conv weight (relayout), 1 f32 vector
        long file_numel = 0;
        int cout = 0, cin = 0, cin_pad = 0, file_cout = 0;
        bool loaded = false;
    };
    std::map<std::string, Slot> slots;
    size_t weight_bytes = 0;
    DevBuf xa, xb, t1, hb;  // f32 streams and the bf16 conv input (+ one zero row)
    long ws_elems = 0;
};

VaeEncoderModel* vae_encoder_create(int base);
void vae_encoder_destroy(VaeEncoderModel* m);
void vae_encoder_load_safetensors(ltx_ctx* ctx, VaeEncoderModel* m, const std::string& path);
void vae_encoder_init_synthetic(ltx_ctx* ctx, VaeEncoderModel* m, uint64_t seed);
// pixels: device f32 [3][T][H][W]; latent: device f32 [128][T'][H/32][W/32], T' = ceil-chain of the three temporal halvings
// ((T-1)/8+1 for T = 8k+1). mean/std (device [128], the decoder's mean_of_means / std_of_means) may be null = raw latent.
void vae_encoder_encode(ltx_ctx* ctx, VaeEncoderModel* m, const float* pixels, int T, int H, int W, const float* mean,
                        const float* stdv, float* latent, int* Tp_out);
int vae_encoder_latent_frames(int T);
