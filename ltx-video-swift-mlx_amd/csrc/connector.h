// connector.h - text-embedding connector: the step immediately before the denoise loop.
// Reference: Models/TextEncoder/LTXTextEncoder.swift - normAndConcatPaddedBatch (:62-122), GemmaFeaturesExtractor
// (:126-187), ConnectorAttention (:197-269), BasicTransformerBlock1D (:316-371), Embeddings1DConnector (:375-522),
// VideoGemmaTextEncoderModel.encodeFromHiddenStates (:574-643). Input = the Gemma-3 hidden states (the language model
// itself stays out of scope), output = the [B,T,3840] bf16 context + all-ones mask that ltx_denoise / ltx_dit_forward take.
// It re-uses the DiT's GEMM / attention / row kernels (30 heads x 128); only the feature-extractor statistics, the
// concat-normalise pass and the register replacement are new kernels (elementwise.hip).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "dit.h"
#include "runtime.h"

struct ConnectorConfig {
    int dim = 3840, heads = 30, layers = 2, registers = 128, states = 49;
    float theta = 10000.0f;
    int max_pos = 4096;
    float eps = 1e-6f;
};

struct ConnBlock {
    LinearW qk, v, o, ff1, ff2;  // to_q|to_k fused [2D][D]
    float* qn = nullptr;
    float* kn = nullptr;
};

struct ConnectorModel {
    ConnectorConfig cfg;
    DeviceArena arena;
    LinearW fe;                  // aggregate_embed [D][D*states], no bias
    float* registers = nullptr;  // [R][D] f32 (bf16-representable)
    std::vector<ConnBlock> blocks;
    std::map<std::string, ParamSlot> slots;
    size_t weight_bytes = 0;
    DevBuf rope_cos, rope_sin;  // [T][D/2] f32, values rounded to bf16 (LTXTextEncoder.swift:498)
    int rope_T = 0;
    DevBuf ws_nc, ws_part, ws_stats, ws_enc, ws_src, ws_x, ws_xn, ws_qk, ws_q, ws_k, ws_vt, ws_ao, ws_ffh;
};

ConnectorModel* connector_create(const ConnectorConfig& cfg);
void connector_destroy(ConnectorModel* m);
void connector_load_safetensors(ltx_ctx* ctx, ConnectorModel* m, const std::string& path);
void connector_init_synthetic(ltx_ctx* ctx, ConnectorModel* m, uint64_t seed);

struct ConnectorArgs {
    const bf16_t* hidden = nullptr;   // device [states][B][T][D] bf16 (the 49 Gemma hidden states)
    const int32_t* mask = nullptr;    // device [B][T] int32 0/1
    int B = 1, T = 0;
    int padding_right = 0;            // 0 = left padding (the reference's default)
    bf16_t* out = nullptr;            // device [B][T][D] bf16
    int32_t* out_mask = nullptr;      // device [B][T] int32 (all ones) or null
    bf16_t* dbg_nc = nullptr;         // optional taps for parity tests: [B][T][D*states], [B][T][D], [B][T][D] f32
    bf16_t* dbg_fe = nullptr;
    float* dbg_reg = nullptr;
};
void connector_encode(ltx_ctx* ctx, ConnectorModel* m, const ConnectorArgs& a);
