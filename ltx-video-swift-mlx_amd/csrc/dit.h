// dit.h - host graph of the video-only LTX-2 DiT (reference LTXTransformer.swift / LTXTransformerBlock.swift /
// LTXAttention.swift / LTXFeedForward.swift / LTXTimestepEmbedding.swift), sequenced over hand-written gfx950
// kernels. Weights are resident in one HBM arena in a kernel-friendly layout chosen at load time (q/k projection
// rows fused so one GEMM feeds the q/k-norm+RoPE pass; biases, norm weights and scale-shift tables as f32 copies
// of their bf16 values). The safetensors key layout is only the file contract.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "hostmath.h"
#include "runtime.h"

// One scratch matrix for the weights of the GEMM that is about to run on a quantised model (largest Linear: 16384 x 4096 bf16).
struct DequantScratch {
    DevBuf buf;
};

struct LinearW {
    bf16_t* w = nullptr;  // [out][in] bf16; null once the model is quantised (dit_quantize releases the bf16 weights)
    float* b = nullptr;   // [out] f32 (bf16-representable values)
    int out = 0, in = 0;
    // affine group quantisation (LTXQuantizationConfig.swift:19-62; MLX layout: codes + per-group scale and bias, group = 64 along
    // `in`): w'[o][i] = bf16(q[o][i] * scale[o][i/64] + bias[o][i/64]). qbits 8: one code per byte; 4: two per byte, low nibble first.
    const uint8_t* q = nullptr;
    const bf16_t* qs = nullptr;
    const bf16_t* qb = nullptr;
    int qbits = 0;
    DequantScratch* dq = nullptr;
};

struct DiTBlock {
    LinearW qk1, v1, o1;  // attn1 (self): to_q|to_k fused [2D][D], to_v, to_out
    float* qn1 = nullptr;
    float* kn1 = nullptr;
    LinearW q2, k2, v2, o2;  // attn2 (text cross-attention)
    float* qn2 = nullptr;
    float* kn2 = nullptr;
    LinearW ff1, ff2;
    float cross_scale = 1.0f;  // setCrossAttentionScale (LTXTransformer.swift:497)
    bool skip_attn = false;    // STG flags (LTXTransformer.swift:512-526)
    bool skip_ff = false;
};

enum SlotKind { SLOT_BF16 = 0, SLOT_F32 = 1 };
struct ParamSlot {
    // quantised model: the slot of a Linear weight points at its codes / scales / biases instead of dst (rows of a fused matrix
    // are row offsets into all three)
    uint8_t* q = nullptr;
    bf16_t* qs = nullptr;
    bf16_t* qb = nullptr;
    void* dst = nullptr;
    int kind = SLOT_BF16;
    long numel = 0;
    long rows = 0, cols = 0;  // expected shape (cols = 0 -> vector)
    int init = 0;             // 0: zeros, 1: ones (reference initialisers for parameters absent from the file)
    bool loaded = false;
};

struct DiTModel {
    TransformerConfig cfg;
    int D = 0, L = 0;
    DeviceArena arena;    // biases, norm weights, scale-shift tables (f32)
    DeviceArena warena;   // Linear weights, bf16 (26.07 GB at the reference architecture); released by dit_quantize
    DeviceArena qarena;   // after dit_quantize: codes + group scales / biases (13.9 GB at 8 bits, 7.3 GB at 4)
    DequantScratch dq;
    LinearW patchify, ada_l1, ada_l2, ada_lin, cap_l1, cap_l2, proj_out;
    float* sst_blocks = nullptr;  // [L][6][D]
    float* sst_out = nullptr;     // [2][D]
    std::vector<DiTBlock> blocks;
    std::map<std::string, ParamSlot> slots;  // module key -> destination
    size_t weight_bytes = 0;
    int quant_bits = 16;  // 16 (bf16), 8 or 4 after dit_quantize

    // ---- caches ----
    DevBuf rope_cos, rope_sin;  // [T][D/2] f32
    int rope_F = 0, rope_H = 0, rope_W = 0;
    // projected caption context and per-layer cross-attention K / V^T (constant across denoise steps:
    // recomputing them every step as the reference does is output-identical - SURVEY 9.2). A few entries are kept
    // so that the CFG pair [neg,pos] and the STG cond-only pass do not evict each other.
    struct CtxCache {
        DevBuf proj, k, vt, bias;
        uint64_t version = 0;
        int kind = 0;  // which pass of a denoise step owns the entry (DiTForwardArgs::ctx_kind): part of the key, never folded into `version`
        int B = 0, S = 0, Spad = 0;
        bool has_bias = false;
        uint64_t last_use = 0;
    };
    std::vector<CtxCache*> ctx_cache;
    uint64_t ctx_clock = 0;
    DevBuf ctx_tmp_h, ctx_tmp_kraw;
    ~DiTModel() {
        for (auto* c : ctx_cache) delete c;
    }

    // ---- activation workspace (grown on demand, never freed inside a step) ----
    DevBuf ws_x, ws_xn, ws_xb, ws_qk, ws_q, ws_k, ws_vt, ws_ao, ws_ffh, ws_qc;
    DevBuf ws_sp_k, ws_sp_vt, ws_sp_vtg;  // sequence-parallel: gathered K [T][D], V^T [D][Tpad], V^T gather staging
    DevBuf ws_ts, ws_emb256, ws_h1, ws_embts, ws_ada, ws_mod, ws_modout, ws_splitk, ws_attn_split;
    int ws_rows = 0, ws_B = 0, ws_Tpad = 0;
};

DiTModel* dit_create(const TransformerConfig& cfg);
void dit_destroy(DiTModel* m);
// Fill every parameter from a safetensors file using the reference's key mapping; returns counts via ctx.
void dit_load_safetensors(ltx_ctx* ctx, DiTModel* m, const std::string& path);
void dit_init_synthetic(ltx_ctx* ctx, DiTModel* m, uint64_t seed);
// one parameter back to the host as f32 (parity tests hand the resident weights to the CPU oracle)
void dit_export_slot(ltx_ctx* ctx, DiTModel* m, const ParamSlot& s, float* out);

struct DiTForwardArgs {
    const bf16_t* latent = nullptr;   // device [B][T][in_channels] bf16
    const bf16_t* context = nullptr;  // device [B][S][caption_channels] bf16
    const float* timesteps = nullptr; // device [B*n_groups] f32 (sigma, unscaled): group g of batch element b at b*n_groups+g
    // per-token timesteps (image-to-video, LTXTransformer.swift:105-124, LTXPipeline.swift:2237-2252): tokens are
    // partitioned into n_groups timestep groups; row_map[b*T+t] = b*n_groups + group(b,t). The adaLN tables are built
    // per (b, group) row and the kernels pick their row through the map. n_groups = 1, row_map = null: one timestep per b.
    int n_groups = 1;
    const int32_t* row_map = nullptr;  // device [B][T] int32
    const int32_t* mask = nullptr;    // device [B][S] int32 or null
    int mask_all_ones = 0;            // host hint: skip the additive bias entirely (bit-identical: +0.0)
    int B = 1, F = 0, H = 0, W = 0, S = 0;
    uint64_t ctx_version = 0;         // 0 = recompute context every call; else cache key (any non-zero 64-bit value, e.g. a hash)
    int ctx_kind = 0;                 // second key field: 0 = a raw forward; 1.. = the passes of denoise_run (batched pair, negative, positive)
    float* velocity = nullptr;        // device [B][T][out_channels] f32
    // Sequence parallelism over sp_world ranks (single sample on several GPUs): this rank owns tokens
    // [sp_rank*Tn, (sp_rank+1)*Tn), Tn = F*H*W / sp_world, and `latent`, `row_map`, `velocity` hold only those Tn rows. Every
    // per-token operation is local; self-attention all-gathers K and V^T of each layer through `sp_gather`, which must
    // enqueue on the context's stream (or synchronise) an all-gather of `bytes` per rank: recv = [sp_world][bytes], rank order,
    // and return 0 (anything else aborts the forward). Null = the context's own transport (dist.h: RCCL or the host transport).
    int sp_rank = 0, sp_world = 1;
    int (*sp_gather)(void* user, const void* send, void* recv, long bytes) = nullptr;
    void* sp_user = nullptr;
};
void dit_forward(ltx_ctx* ctx, DiTModel* m, const DiTForwardArgs& a);

// R22: fuse a LoRA file into the resident weights: W' = W + cast(scale * (alpha/rank | 1) * (up @ down))
// (LoRALoader.swift:63-111,162-178; LoRAAdapter.swift:64-166). Returns the number of fused layers.
int dit_fuse_lora(ltx_ctx* ctx, DiTModel* m, const std::string& path, float scale);
// R21: on-the-fly affine quantisation of every Linear weight, group size 64 along `in`, 8 or 4 bits
// (LTXQuantizationConfig.swift:19-62, LTXPipeline.swift:323-333). The bf16 weights are REPLACED by codes + bf16 group scale / bias
// (the layout MLX's QuantizedLinear holds) and their 26 GB are released. Consumer: dit_linear_weights() below - see the note there
// on which launches de-quantise inside the GEMM's B stage and which through the scratch matrix.
void dit_quantize(ltx_ctx* ctx, DiTModel* m, int bits, int group);
// bf16 weights of a Linear for a launch on `stream`: the resident matrix, or the scratch matrix filled from the codes
const bf16_t* dit_linear_weights(const LinearW& w, hipStream_t stream);
void launch_dequant(const uint8_t* q, const bf16_t* qs, const bf16_t* qb, long out, long in, int bits, bf16_t* dst, hipStream_t stream);
