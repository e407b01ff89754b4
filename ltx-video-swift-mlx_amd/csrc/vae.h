// vae.h - host graph of the LTX-2 video VAE decoder (reference VideoDecoder.swift, VideoConvolution.swift) over the
// implicit-GEMM conv3d path of gemm.hip. Activations are channels-last ([F][H][W][C]); the residual stream stays
// f32 in HBM, conv inputs are the bf16 output of the fused pixel-norm/scale-shift/SiLU pass. Conv weights are
// re-laid at load time to [Cout][27 taps][Cin] bf16 (K-contiguous implicit-GEMM B operand); the upsamplers' output
// channels are additionally permuted to (dt,dh,dw)-major so the depth-to-space store of a tile is contiguous.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "runtime.h"

struct ConvW {
    bf16_t* w = nullptr;  // [cout][27*cin]
    float* b = nullptr;   // [cout]
    int cin = 0, cout = 0;
    bool d2s_perm = false;  // rows stored in sub-major order (upsampler convs)
};
struct VaeResBlock {
    ConvW conv1, conv2;
    float* sst = nullptr;  // [4][C]: shift1, scale1, shift2, scale2 (raw table)
};
struct VaeTimeEmbedder {  // timestep_embedder.linear_1 / linear_2 (VideoDecoder.swift:36-70)
    bf16_t* w1 = nullptr;
    float* b1 = nullptr;
    bf16_t* w2 = nullptr;
    float* b2 = nullptr;
    int hidden = 256, out = 0;
};
struct VaeGroup {
    int C = 0;
    VaeResBlock blocks[5];
    VaeTimeEmbedder te;
};

struct VaeModel {
    int latent_channels = 128;
    int channels[4] = {1024, 512, 256, 128};
    bool timestep_conditioning = false;
    DeviceArena arena;
    ConvW conv_in, conv_out;
    VaeGroup groups[4];
    ConvW up[3];
    VaeTimeEmbedder last_te;
    float* last_sst = nullptr;  // [2][128]
    float* mean = nullptr;      // [128]
    float* std_ = nullptr;      // [128]
    float* ts_mult = nullptr;   // scalar (device) + host copy
    float ts_mult_host = 1000.0f;
    size_t weight_bytes = 0;

    struct Slot {
        void* dst = nullptr;
        int kind = 0;  // 0 conv weight (relayout), 1 f32 vector (optionally d2s-permuted), 2 bf16 matrix [out][in], 3 scalar
        long numel = 0;
        int cout = 0, cin = 0;
        int perm = 0;  // output-channel order: 0 as in the file, 1 depth-to-space sub-major (upsampler convs), 2 un-patchify row-major (conv_out)
        int init = 0;
        bool loaded = false;
    };
    std::map<std::string, Slot> slots;

    // workspace
    DevBuf xa, xb, t1, hb, hb2, mods, tile_frames, temb, skws;  // hb / hb2: the convs' bf16 inputs, ping-pong where PixelNorm is fused
    long ws_elems = 0;
};

VaeModel* vae_create();
void vae_destroy(VaeModel* m);
void vae_load_safetensors(ltx_ctx* ctx, VaeModel* m, const std::string& path, const std::string& config_json);
void vae_init_synthetic(ltx_ctx* ctx, VaeModel* m, uint64_t seed, bool timestep_conditioning);

struct VaeDecodeArgs {
    const float* latent = nullptr;  // device [1][128][F][H][W] f32
    int F = 0, H = 0, W = 0;
    int has_timestep = 0;
    float timestep = 0.05f;
    const float* noise = nullptr;  // device, same shape as latent (required when has_timestep)
    int tile = 0, overlap = 1;
    float* frames = nullptr;  // device (n_frames, 32H, 32W, 3) f32
    long frames_cap = 0;      // capacity in floats
    int* n_frames_out = nullptr;
    int shard = 0;  // 1: temporal tiles are decoded round-robin by the context's ranks (dist.h) and broadcast raw before the blend;
                    // 2: ... and sent raw to rank `root` only, which blends (the other ranks' `frames` may be null)
    int root = 0;
};
void vae_decode(ltx_ctx* ctx, VaeModel* m, const VaeDecodeArgs& a);
// Building blocks of the tiled decode (decodeWithTemporalTiling, VideoDecoder.swift:517-602), exposed so that a host can place
// tiles on GPUs itself: the RAW frames (before blend and clip) of tile `tile_index` of the plan (tile, overlap) -> a.frames;
// returns the tile's frame count.
int vae_decode_tile(ltx_ctx* ctx, VaeModel* m, const VaeDecodeArgs& a, int tile_index);
// One VAEResBlock3d of the loaded decoder (VideoDecoder.swift:75-131; no timestep conditioning) on a caller-supplied channels-last
// f32 stream x [F][H][W][C] (device, in place), C = channels of `group`; the kernels and epilogues the decode itself runs.
void vae_res_block(ltx_ctx* ctx, VaeModel* m, int group, int block, float* x, int F, int H, int W);
// one VAEDepthToSpaceUpsample3d of the loaded decoder (the stage behind up-block group `group`, 0..2) on a caller-supplied stream
void vae_upsample(ltx_ctx* ctx, VaeModel* m, int group, const float* x, int F, int H, int W, float* out);
// Blend raw tiles in tile order over 8*overlap frames, then (x+1)/2 clipped to [0,1] (VideoDecoder.swift:561-592, :501-505).
// tiles[i] (n_i, 32H, 32W, 3) device f32; tiles[0] may alias `frames`. Returns the blended frame count.
int vae_blend_tiles(ltx_ctx* ctx, const float* const* tiles, const int* tile_frames, int n_tiles, int overlap, int H, int W, float* frames,
                    long frames_cap);
