// attention.h - launch interface of the flash attention forward kernel (attention.hip).
#pragma once
#include "common.h"

struct AttnArgs {
    const bf16_t* Q = nullptr;   // [B][Tq][ldq], head h occupies columns h*128 .. h*128+127
    long ldq = 0, q_bstride = 0;
    const bf16_t* K = nullptr;   // [B][Tk][ldk]
    long ldk = 0, k_bstride = 0;
    const bf16_t* Vt = nullptr;  // [B][H*128][ldvt], keys contiguous, ldvt >= roundup(Tk,64), pad columns finite
    long ldvt = 0, vt_bstride = 0;
    bf16_t* O = nullptr;         // [B][Tq][ldo]
    long ldo = 0, o_bstride = 0;
    const float* bias = nullptr;  // optional additive key mask [B][Tk] (reference: (1-m)*-10000, LTXTransformer.swift:141-156)
    long bias_bstride = 0;
    int B = 0, H = 0, Tq = 0, Tk = 0;
    float scale = 0.08838834764831845f;  // 1/sqrt(128)
    // 1: Q already carries scale * log2(e) (written so by qknorm_rope's out_scale, one rounding to bf16): the scores ARE the base-2
    // exponents and `scale` is ignored. The assembly kernel then runs a stream without its 48 v_mul per key tile; the other
    // kernels run with scale = ln 2. The bias stays in the reference's units (added to score * scale).
    int q_prescaled = 0;
    int plain_order = 0;  // 1: workgroups in (query block, head, batch) order instead of the XCD-aware order (A/B measurements only)
    // Key split (few queries against many keys: the cross-attention of a 128-token launch is 32 workgroups walking 16 key tiles each).
    // With a workspace the launcher may divide the keys over up to 8 workgroups per (query block, head): each writes its normalised
    // O to a slice of the workspace plus log2 of its softmax denominators, attn_combine_kernel weights the slices together.
    // attn_split_ws_bytes() says how much a launch can use; null = never split.
    void* split_ws = nullptr;
    long split_ws_bytes = 0;
    // set by the launcher for the kernels (callers leave them alone)
    int key_splits = 1, split_keys = 0;   // number of key ranges, keys per range (a multiple of 64)
    bf16_t* o_part = nullptr;             // [splits][B][Tq][H*128]
    float* lse = nullptr;                 // [splits][B][H][roundup(Tq, 192)]
};
// workspace a key-split launch of this shape would use (0: the launcher would not split it)
long attn_split_ws_bytes(int B, int H, int Tq, int Tk);
int attn_key_splits(int B, int H, int Tq, int Tk);  // the number of key ranges (1 = no split)
// what a producer multiplies q by for q_prescaled: (1 / sqrt(128)) * log2(e)
constexpr float kAttnQueryPrescale = 0.08838834764831845f * 1.4426950408889634f;

void launch_attention(const AttnArgs& args, hipStream_t stream);
