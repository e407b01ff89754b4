// abi.cpp - extern "C" surface of libltxhip.so (declared in include/ltxhip.h). Every entry point converts C++
// exceptions into an ltx_status + message; nothing here computes on the CPU on behalf of the GPU path.
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/ltxhip.h"
#include "attention.h"
#include "connector.h"
#include "dist.h"
#include "dit.h"
#include "elementwise.h"
#include "gemm.h"
#include "hostmath.h"
#include "pipeline.h"
#include "upscaler.h"
#include "vae.h"
#include "vae_encoder.h"
#include "runtime.h"
#include "linear_ops.h"

namespace {

thread_local std::string g_noctx_error;

template <class Fn>
int guarded(ltx_ctx* ctx, Fn&& fn) {
    try {
        if (ctx) HIP_CHECK(hipSetDevice(ctx->device));
        prof_set_current(ctx ? &ctx->prof : nullptr);
        fn();
        return LTX_OK;
    } catch (const LtxError& e) {
        if (ctx) ctx->last_error = e.msg; else g_noctx_error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->last_error = e.what(); else g_noctx_error = e.what();
        return LTX_ERR_GENERATION_FAILED;
    }
}

TransformerConfig to_cfg(const ltx_transformer_config* c) {
    TransformerConfig t;
    if (!c) return t;
    t.num_layers = c->num_layers;
    t.num_heads = c->num_attention_heads;
    t.head_dim = c->attention_head_dim;
    t.in_channels = c->in_channels;
    t.out_channels = c->out_channels;
    t.cross_attention_dim = c->cross_attention_dim;
    t.caption_channels = c->caption_channels;
    t.rope_theta = c->rope_theta;
    for (int i = 0; i < 3; ++i) t.max_pos[i] = c->max_pos[i];
    t.timestep_scale_multiplier = c->timestep_scale_multiplier;
    t.norm_eps = c->norm_eps;
    return t;
}

int copy_str(const std::string& s, char* out, int cap) {
    if (!out || cap <= (int)s.size()) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return 1;
}

// FNV-1a over 8-byte words: cheap change detector for host-pointer context buffers
uint64_t hash_bytes(const void* p, size_t n, uint64_t h) {
    const uint8_t* b = (const uint8_t*)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x100000001B3ull;
        h ^= h >> 29;
    }
    for (; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h ? h : 1;
}

DiTModel* need_dit(ltx_ctx* ctx) {
    if (!ctx->dit) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: transformer");
    return ctx->dit;
}

}  // namespace

extern "C" {

const char* ltx_version(void) { return "0.1.0"; }
int ltx_abi_version(void) { return LTX_ABI_VERSION; }

int ltx_ctx_set_option(ltx_ctx* ctx, const char* key, int value) {
    return guarded(ctx, [&] {
        const int i = ltx_opt_find(key);
        LTX_REQUIRE(i >= 0, "ltx_ctx_set_option: unknown option '%s' (ltx_option_info enumerates them)", key ? key : "(null)");
        const LtxOptInfo& o = ltx_opt_info(i);
        LTX_REQUIRE(ltx_opt_set(i, value), "ltx_ctx_set_option: %s = %d outside [%d, %d]", o.name, value, o.lo, o.hi);
    });
}
int ltx_ctx_get_option(const ltx_ctx* ctx, const char* key, int* value) {
    return guarded(const_cast<ltx_ctx*>(ctx), [&] {
        const int i = ltx_opt_find(key);
        LTX_REQUIRE(i >= 0 && value, "ltx_ctx_get_option: unknown option '%s'", key ? key : "(null)");
        *value = ltx_opt((LtxOpt)i);
    });
}
int ltx_option_info(int index, const char** name, int* def, int* lo, int* hi, int* numerics, const char** doc) {
    if (index >= 0 && index < OPT_COUNT) {
        const LtxOptInfo& o = ltx_opt_info(index);
        if (name) *name = o.name;
        if (def) *def = o.def;
        if (lo) *lo = o.lo;
        if (hi) *hi = o.hi;
        if (numerics) *numerics = o.numerics;
        if (doc) *doc = o.doc;
    }
    return OPT_COUNT;
}

const char* ltx_build_info(void) {
#ifdef LTX_EXPERIMENTS
    return "gfx950;experiments=1";
#else
    return "gfx950;experiments=0";
#endif
}

void ltx_transformer_config_default(ltx_transformer_config* c) {
    if (!c) return;
    TransformerConfig t;
    c->num_layers = t.num_layers;
    c->num_attention_heads = t.num_heads;
    c->attention_head_dim = t.head_dim;
    c->in_channels = t.in_channels;
    c->out_channels = t.out_channels;
    c->cross_attention_dim = t.cross_attention_dim;
    c->caption_channels = t.caption_channels;
    c->rope_theta = t.rope_theta;
    for (int i = 0; i < 3; ++i) c->max_pos[i] = t.max_pos[i];
    c->timestep_scale_multiplier = t.timestep_scale_multiplier;
    c->norm_eps = t.norm_eps;
}

int ltx_ctx_create(int device, ltx_ctx** out) {
    if (!out) return LTX_ERR_INVALID_CONFIGURATION;
    *out = nullptr;
    ltx_ctx* ctx = nullptr;
    const int rc = guarded(nullptr, [&] {
        int n = 0;
        HIP_CHECK(hipGetDeviceCount(&n));
        if (device < 0 || device >= n) LTX_THROW(LTXS_HIP_ERROR, "no HIP device %d (found %d)", device, n);
        HIP_CHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            LTX_THROW(LTXS_HIP_ERROR, "libltxhip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
        ctx = new ltx_ctx();
        ctx->device = device;
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    });
    if (rc != LTX_OK) {
        delete ctx;
        return rc;
    }
    *out = ctx;
    return LTX_OK;
}

void ltx_ctx_destroy(ltx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    try { dist_shutdown(ctx); } catch (...) {}
    if (ctx->dit) dit_destroy(ctx->dit);
    if (ctx->vae) vae_destroy(ctx->vae);
    if (ctx->upscaler) upscaler_destroy(ctx->upscaler);
    if (ctx->connector) connector_destroy(ctx->connector);
    if (ctx->vae_encoder) vae_encoder_destroy(ctx->vae_encoder);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* ltx_last_error(const ltx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_noctx_error.c_str(); }

int ltx_ctx_set_stream(ltx_ctx* ctx, void* hip_stream) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (ctx->stream) HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->own_stream && ctx->stream) HIP_CHECK(hipStreamDestroy(ctx->stream));
        // the handle is used as given: NULL is the (legacy) default stream, which is what torch's default stream is
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    });
}

int ltx_ctx_synchronize(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { HIP_CHECK(hipStreamSynchronize(ctx->stream)); });
}

int ltx_load_report(const ltx_ctx* ctx, int* n_loaded, int* n_missing, int* n_unmatched) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    if (n_loaded) *n_loaded = ctx->n_loaded;
    if (n_missing) *n_missing = ctx->n_missing;
    if (n_unmatched) *n_unmatched = ctx->n_unmatched;
    return LTX_OK;
}

// ---- pure host ----
int ltx_validate_generation_config(int width, int height, int num_frames, int num_steps, float cfg_scale,
                                   int two_stage, char* msg, int msg_cap) {
    std::string m;
    const int rc = validate_generation_config(width, height, num_frames, num_steps, cfg_scale, two_stage, &m);
    if (msg && msg_cap > 0) {
        strncpy(msg, m.c_str(), msg_cap - 1);
        msg[msg_cap - 1] = 0;
    }
    return rc;
}

int ltx_latent_shape(int width, int height, int num_frames, int* F, int* H, int* W) {
    if (!F || !H || !W) return LTX_ERR_INVALID_CONFIGURATION;
    latent_shape(width, height, num_frames, F, H, W);
    return LTX_OK;
}

int ltx_sigmas(int distilled, int num_steps, int token_count, float* out, int cap) {
    if (!out || (!distilled && num_steps < 1)) return -LTX_ERR_INVALID_CONFIGURATION;
    const std::vector<float> s = compute_sigmas(distilled != 0, num_steps, token_count);
    for (int i = 0; i < (int)s.size() && i < cap; ++i) out[i] = s[i];
    return (int)s.size();
}

int ltx_stage2_sigmas(float* out, int cap) {
    for (int i = 0; i < 4 && i < cap; ++i) out[i] = kStage2DistilledSigmas[i];
    return 4;
}

int ltx_rope_tables(const ltx_transformer_config* cfg, int F, int H, int W, float* cos_out, float* sin_out) {
    if (!cos_out || !sin_out || F < 1 || H < 1 || W < 1) return LTX_ERR_INVALID_CONFIGURATION;
    const TransformerConfig t = to_cfg(cfg);
    std::vector<float> c, s;
    rope_tables(t, F, H, W, 24.0f, &c, &s);
    memcpy(cos_out, c.data(), c.size() * 4);
    memcpy(sin_out, s.data(), s.size() * 4);
    return LTX_OK;
}

int ltx_vae_tile_plan(int latent_frames, int tile, int overlap, int* starts, int* ends, int cap, int* out_frames) {
    if (latent_frames < 1) return -LTX_ERR_INVALID_CONFIGURATION;
    const TilePlan p = vae_tile_plan(latent_frames, tile, overlap);
    if (p.start.empty()) return -LTX_ERR_INVALID_CONFIGURATION;
    for (int i = 0; i < (int)p.start.size() && i < cap; ++i) {
        if (starts) starts[i] = p.start[i];
        if (ends) ends[i] = p.end[i];
    }
    if (out_frames) *out_frames = p.out_frames;
    return (int)p.start.size();
}

int ltx_map_transformer_key(const char* file_key, char* out, int cap) {
    std::string mk;
    if (!file_key || !map_transformer_file_key(file_key, &mk)) return 0;
    return copy_str(mk, out, cap);
}
int ltx_map_vae_key(const char* file_key, char* out, int cap) {
    std::string mk;
    if (!file_key || !map_vae_file_key(file_key, &mk)) return 0;
    return copy_str(mk, out, cap);
}
int ltx_map_lora_key(const char* lora_key, char* out, int cap) {
    std::string mk;
    if (!lora_key || !map_lora_key(lora_key, &mk)) return 0;
    return copy_str(mk, out, cap);
}

int ltx_st_info(const char* path, const char* key, long* shape8) {
    if (!path || !key) return -LTX_ERR_INVALID_CONFIGURATION;
    int nd = -1;
    const int rc = guarded(nullptr, [&] {
        SafeTensors st;
        st.open(path);
        auto it = st.tensors.find(key);
        if (it == st.tensors.end()) LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: no tensor '%s' in %s", key, path);
        nd = (int)it->second.shape.size();
        for (int i = 0; i < nd && i < 8; ++i)
            if (shape8) shape8[i] = it->second.shape[i];
    });
    return rc == 0 ? nd : -rc;
}

long ltx_st_read(const char* path, const char* key, int dtype, void* out, long cap) {
    if (!path || !key || !out) return -LTX_ERR_INVALID_CONFIGURATION;
    long n = -1;
    const int rc = guarded(nullptr, [&] {
        SafeTensors st;
        st.open(path);
        auto it = st.tensors.find(key);
        if (it == st.tensors.end()) LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: no tensor '%s' in %s", key, path);
        const StTensor& t = it->second;
        n = t.numel();
        LTX_REQUIRE(n <= cap, "ltx_st_read: buffer too small (%ld < %ld)", cap, n);
        if (dtype == 1) {
            st_to_bf16(st, t, (bf16_t*)out);
        } else if (dtype == 0) {
            st_to_f32(st, t, (float*)out);
        } else {
            if (t.dtype == "I32") {
                memcpy(out, st.ptr(t), (size_t)n * 4);
            } else if (t.dtype == "I64") {
                const int64_t* s64 = (const int64_t*)st.ptr(t);
                for (long i = 0; i < n; ++i) ((int32_t*)out)[i] = (int32_t)s64[i];
            } else {
                std::vector<float> f(n);
                st_to_f32(st, t, f.data());
                for (long i = 0; i < n; ++i) ((int32_t*)out)[i] = (int32_t)f[i];
            }
        }
    });
    return rc == 0 ? n : -rc;
}

// ---- DiT ----
int ltx_dit_load(ltx_ctx* ctx, const char* path, const ltx_transformer_config* cfg, int quant_bits, int group_size) {
    if (!ctx || !path) return LTX_ERR_INVALID_CONFIGURATION;
    (void)group_size;
    return guarded(ctx, [&] {
        LTX_REQUIRE(quant_bits == 16 || quant_bits == 0 || quant_bits == 8 || quant_bits == 4,
                    "transformer quantization must be bf16 (16), qint8 (8) or int4 (4); got %d bits", quant_bits);
        LTX_REQUIRE(quant_bits == 16 || quant_bits == 0 || group_size == 64, "quantization group size must be 64, got %d", group_size);
        if (ctx->dit) {
            dit_destroy(ctx->dit);
            ctx->dit = nullptr;
        }
        DiTModel* m = dit_create(to_cfg(cfg));
        try {
            dit_load_safetensors(ctx, m, path);
            if (quant_bits == 8 || quant_bits == 4) dit_quantize(ctx, m, quant_bits, 64);
        } catch (...) {
            dit_destroy(m);
            throw;
        }
        ctx->dit = m;
    });
}

int ltx_dit_init_synthetic(ltx_ctx* ctx, const ltx_transformer_config* cfg, uint64_t seed) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (ctx->dit) {
            dit_destroy(ctx->dit);
            ctx->dit = nullptr;
        }
        DiTModel* m = dit_create(to_cfg(cfg));
        try {
            dit_init_synthetic(ctx, m, seed);
        } catch (...) {
            dit_destroy(m);
            throw;
        }
        ctx->dit = m;
    });
}

int ltx_dit_quantize(ltx_ctx* ctx, int bits, int group_size) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { dit_quantize(ctx, need_dit(ctx), bits, group_size); });
}

int ltx_dit_memory_info(ltx_ctx* ctx, long* bf16_weight_bytes, long* quantised_weight_bytes, long* scratch_bytes, long* other_bytes) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = ctx->dit;
        if (!m) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: transformer");
        if (bf16_weight_bytes) *bf16_weight_bytes = m->warena.buf.p ? (long)m->warena.buf.bytes : 0;
        if (quantised_weight_bytes) *quantised_weight_bytes = m->qarena.buf.p ? (long)m->qarena.buf.bytes : 0;
        if (scratch_bytes) *scratch_bytes = m->dq.buf.p ? (long)m->dq.buf.bytes : 0;
        if (other_bytes) *other_bytes = m->arena.buf.p ? (long)m->arena.buf.bytes : 0;
    });
}

int ltx_dit_fuse_lora(ltx_ctx* ctx, const char* path, float scale, int* n_fused) {
    if (!ctx || !path) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (!ctx->dit) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: Transformer not loaded");
        const int n = dit_fuse_lora(ctx, ctx->dit, path, scale);
        if (n_fused) *n_fused = n;
    });
}

int ltx_dit_unload(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->dit) dit_destroy(ctx->dit);
        ctx->dit = nullptr;
    });
}

int ltx_dit_forward_dev(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* timesteps,
                        const int32_t* mask, int mask_all_ones, int B, int F, int H, int W, int S,
                        uint64_t ctx_version, float* velocity) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        DiTForwardArgs a;
        a.latent = latent;
        a.context = context;
        a.timesteps = timesteps;
        a.mask = mask;
        a.mask_all_ones = mask_all_ones;
        a.B = B; a.F = F; a.H = H; a.W = W; a.S = S;
        a.ctx_version = ctx_version;
        a.velocity = velocity;
        dit_forward(ctx, m, a);
    });
}

int ltx_dit_forward_sp_dev(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* timesteps,
                           const int32_t* mask, int mask_all_ones, int F, int H, int W, int S, uint64_t ctx_version,
                           int sp_rank, int sp_world, ltx_allgather_fn gather, void* user, float* velocity) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        LTX_REQUIRE(sp_world >= 1 && (sp_world == 1 || gather || ctx->dist), "ltx_dit_forward_sp_dev: %d ranks need a gather callback or ltx_dist_init", sp_world);
        DiTForwardArgs a;
        a.latent = latent;
        a.context = context;
        a.timesteps = timesteps;
        a.mask = mask;
        a.mask_all_ones = mask_all_ones;
        a.B = 1; a.F = F; a.H = H; a.W = W; a.S = S;
        a.ctx_version = ctx_version;
        a.velocity = velocity;
        a.sp_rank = sp_rank;
        a.sp_world = sp_world;
        a.sp_gather = gather;
        a.sp_user = user;
        dit_forward(ctx, m, a);
    });
}

int ltx_dit_forward(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* timesteps,
                    const int32_t* mask, int B, int F, int H, int W, int S, float* velocity) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        LTX_REQUIRE(latent && context && timesteps && velocity && B >= 1 && F >= 1 && H >= 1 && W >= 1 && S >= 1, "ltx_dit_forward: bad arguments");
        const long T = (long)F * H * W;
        const size_t n_lat = (size_t)B * T * m->cfg.in_channels * 2;
        const size_t n_ctx = (size_t)B * S * m->cfg.caption_channels * 2;
        const size_t n_vel = (size_t)B * T * m->cfg.out_channels * 4;
        ctx->h2d[0].ensure(n_lat);
        ctx->h2d[1].ensure(n_ctx);
        ctx->h2d[2].ensure((size_t)B * 4);
        ctx->h2d[3].ensure((size_t)B * S * 4);
        ctx->h2d[4].ensure(n_vel);
        hipStream_t st = ctx->stream;
        // the context is constant across denoise steps: upload + re-project it only when its bytes change
        uint64_t hv = hash_bytes(context, n_ctx, 0xCBF29CE484222325ull);
        int all_ones = 1;
        if (mask) {
            hv = hash_bytes(mask, (size_t)B * S * 4, hv);
            for (long i = 0; i < (long)B * S; ++i)
                if (mask[i] != 1) { all_ones = 0; break; }
        }
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[0].p, latent, n_lat, hipMemcpyHostToDevice, st));
        if (!hv) hv = 1;  // 0 means "do not cache"
        bool cached = false;
        for (auto* e : m->ctx_cache) cached = cached || (e->version == hv && e->kind == 0 && e->B == B && e->S == S);
        if (!cached) {
            HIP_CHECK(hipMemcpyAsync(ctx->h2d[1].p, context, n_ctx, hipMemcpyHostToDevice, st));
            if (mask) HIP_CHECK(hipMemcpyAsync(ctx->h2d[3].p, mask, (size_t)B * S * 4, hipMemcpyHostToDevice, st));
        }
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[2].p, timesteps, (size_t)B * 4, hipMemcpyHostToDevice, st));
        DiTForwardArgs a;
        a.latent = ctx->h2d[0].as<bf16_t>();
        a.context = ctx->h2d[1].as<bf16_t>();
        a.timesteps = ctx->h2d[2].as<float>();
        a.mask = mask ? ctx->h2d[3].as<int32_t>() : nullptr;
        a.mask_all_ones = all_ones;
        a.B = B; a.F = F; a.H = H; a.W = W; a.S = S;
        a.ctx_version = hv;
        a.velocity = ctx->h2d[4].as<float>();
        dit_forward(ctx, m, a);
        HIP_CHECK(hipMemcpyAsync(velocity, ctx->h2d[4].p, n_vel, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

int ltx_dit_forward_tokens(ltx_ctx* ctx, const uint16_t* latent, const uint16_t* context, const float* token_timesteps,
                           const int32_t* mask, int B, int F, int H, int W, int S, float* velocity) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        LTX_REQUIRE(latent && context && token_timesteps && velocity && B >= 1 && F >= 1 && H >= 1 && W >= 1 && S >= 1,
                    "ltx_dit_forward_tokens: bad arguments");
        const long T = (long)F * H * W;
        // distinct timestep values per batch row -> groups (order of first appearance); G = the largest count
        std::vector<std::vector<float>> uniq(B);
        std::vector<int32_t> rowmap((size_t)B * T);
        int G = 1;
        for (int b = 0; b < B; ++b) {
            for (long t = 0; t < T; ++t) {
                const float v = token_timesteps[(size_t)b * T + t];
                size_t g = 0;
                while (g < uniq[b].size() && uniq[b][g] != v) ++g;
                if (g == uniq[b].size()) uniq[b].push_back(v);
                rowmap[(size_t)b * T + t] = (int32_t)g;
                LTX_REQUIRE(uniq[b].size() <= 8, "ltx_dit_forward_tokens: more than 8 distinct timesteps in batch row %d", b);
            }
            G = std::max(G, (int)uniq[b].size());
        }
        LTX_REQUIRE(B * G <= 8, "ltx_dit_forward_tokens: batch x distinct timesteps = %d exceeds 8", B * G);
        std::vector<float> ts((size_t)B * G);
        for (int b = 0; b < B; ++b)
            for (int g = 0; g < G; ++g) ts[(size_t)b * G + g] = uniq[b][g < (int)uniq[b].size() ? g : 0];
        for (int b = 0; b < B; ++b)
            for (long t = 0; t < T; ++t) rowmap[(size_t)b * T + t] += b * G;
        const size_t n_lat = (size_t)B * T * m->cfg.in_channels * 2;
        const size_t n_ctx = (size_t)B * S * m->cfg.caption_channels * 2;
        const size_t n_vel = (size_t)B * T * m->cfg.out_channels * 4;
        ctx->h2d[0].ensure(n_lat);
        ctx->h2d[1].ensure(n_ctx);
        ctx->h2d[2].ensure(8 * 4);
        ctx->h2d[3].ensure((size_t)B * S * 4);
        ctx->h2d[4].ensure(n_vel);
        ctx->dn_rowmap.ensure(std::max((size_t)B * T * 4, (size_t)2 * T * 4));
        hipStream_t st = ctx->stream;
        int all_ones = 1;
        if (mask)
            for (long i = 0; i < (long)B * S; ++i)
                if (mask[i] != 1) { all_ones = 0; break; }
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[0].p, latent, n_lat, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[1].p, context, n_ctx, hipMemcpyHostToDevice, st));
        if (mask) HIP_CHECK(hipMemcpyAsync(ctx->h2d[3].p, mask, (size_t)B * S * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[2].p, ts.data(), ts.size() * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(ctx->dn_rowmap.p, rowmap.data(), rowmap.size() * 4, hipMemcpyHostToDevice, st));
        DiTForwardArgs a;
        a.latent = ctx->h2d[0].as<bf16_t>();
        a.context = ctx->h2d[1].as<bf16_t>();
        a.timesteps = ctx->h2d[2].as<float>();
        a.mask = mask ? ctx->h2d[3].as<int32_t>() : nullptr;
        a.mask_all_ones = all_ones;
        a.B = B; a.F = F; a.H = H; a.W = W; a.S = S;
        a.n_groups = G;
        a.row_map = ctx->dn_rowmap.as<int32_t>();
        a.velocity = ctx->h2d[4].as<float>();
        dit_forward(ctx, m, a);
        HIP_CHECK(hipMemcpyAsync(velocity, ctx->h2d[4].p, n_vel, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));  // also keeps ts / rowmap alive until the copies are done
    });
}

int ltx_dit_set_cross_attn_scale(ltx_ctx* ctx, float scale, int first_block, int last_block) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        if (last_block < 0) last_block = m->L - 1;
        for (int i = first_block; i <= last_block; ++i)
            if (i >= 0 && i < m->L) m->blocks[i].cross_scale = scale;
    });
}

int ltx_dit_set_stg(ltx_ctx* ctx, const int* blocks, int n_blocks, int skip_attn, int skip_ff) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        for (int j = 0; j < n_blocks; ++j) {
            const int i = blocks[j];
            if (i < 0 || i >= m->L) continue;
            m->blocks[i].skip_attn = skip_attn != 0;
            m->blocks[i].skip_ff = skip_ff != 0;
        }
    });
}

int ltx_dit_clear_stg(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        for (auto& b : m->blocks) b.skip_attn = b.skip_ff = false;
    });
}

// ---- VAE ----
static VaeModel* need_vae(ltx_ctx* ctx) {
    if (!ctx->vae) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: vaeDecoder");
    return ctx->vae;
}

int ltx_vae_load(ltx_ctx* ctx, const char* path, const char* config_json) {
    if (!ctx || !path) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (ctx->vae) {
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            vae_destroy(ctx->vae);
            ctx->vae = nullptr;
        }
        VaeModel* m = vae_create();
        try {
            vae_load_safetensors(ctx, m, path, config_json ? config_json : "");
        } catch (...) {
            vae_destroy(m);
            throw;
        }
        ctx->vae = m;
    });
}

int ltx_vae_init_synthetic(ltx_ctx* ctx, uint64_t seed, int timestep_conditioning) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (ctx->vae) {
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            vae_destroy(ctx->vae);
            ctx->vae = nullptr;
        }
        VaeModel* m = vae_create();
        try {
            vae_init_synthetic(ctx, m, seed, timestep_conditioning != 0);
        } catch (...) {
            vae_destroy(m);
            throw;
        }
        ctx->vae = m;
    });
}

int ltx_vae_unload(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->vae) vae_destroy(ctx->vae);
        ctx->vae = nullptr;
    });
}

int ltx_vae_timestep_conditioning(const ltx_ctx* ctx) {
    if (!ctx || !ctx->vae) return -LTX_ERR_MODEL_NOT_LOADED;
    return ctx->vae->timestep_conditioning ? 1 : 0;
}

int ltx_vae_decode_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                       const float* noise, int tile, int overlap, float* frames_out, long frames_cap,
                       int* n_frames_out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        VaeModel* m = need_vae(ctx);
        VaeDecodeArgs a;
        a.latent = latent; a.F = F; a.H = H; a.W = W;
        a.has_timestep = has_timestep; a.timestep = timestep; a.noise = noise;
        a.tile = tile; a.overlap = overlap;
        a.frames = frames_out; a.frames_cap = frames_cap; a.n_frames_out = n_frames_out;
        vae_decode(ctx, m, a);
    });
}

int ltx_vae_decode_sharded_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                               const float* noise, int tile, int overlap, float* frames_out, long frames_cap,
                               int* n_frames_out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        VaeModel* m = need_vae(ctx);
        LTX_REQUIRE(ctx->dist, "ltx_vae_decode_sharded_dev: no group on this context (ltx_dist_init / ltx_dist_set_transport)");
        VaeDecodeArgs a;
        a.latent = latent; a.F = F; a.H = H; a.W = W;
        a.has_timestep = has_timestep; a.timestep = timestep; a.noise = noise;
        a.tile = tile; a.overlap = overlap;
        a.frames = frames_out; a.frames_cap = frames_cap; a.n_frames_out = n_frames_out;
        a.shard = 1;
        vae_decode(ctx, m, a);
    });
}

int ltx_vae_decode_gathered_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                                const float* noise, int tile, int overlap, int root, float* frames_out, long frames_cap,
                                int* n_frames_out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        VaeModel* m = need_vae(ctx);
        LTX_REQUIRE(ctx->dist, "ltx_vae_decode_gathered_dev: no group on this context (ltx_dist_init / ltx_dist_set_transport)");
        VaeDecodeArgs a;
        a.latent = latent; a.F = F; a.H = H; a.W = W;
        a.has_timestep = has_timestep; a.timestep = timestep; a.noise = noise;
        a.tile = tile; a.overlap = overlap;
        a.frames = frames_out; a.frames_cap = frames_cap; a.n_frames_out = n_frames_out;
        a.shard = 2;
        a.root = root;
        vae_decode(ctx, m, a);
    });
}

int ltx_vae_decode_tile_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                            const float* noise, int tile, int overlap, int tile_index, float* tile_out, long tile_cap,
                            int* n_frames_out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        VaeModel* m = need_vae(ctx);
        VaeDecodeArgs a;
        a.latent = latent; a.F = F; a.H = H; a.W = W;
        a.has_timestep = has_timestep; a.timestep = timestep; a.noise = noise;
        a.tile = tile; a.overlap = overlap;
        a.frames = tile_out; a.frames_cap = tile_cap; a.n_frames_out = n_frames_out;
        vae_decode_tile(ctx, m, a, tile_index);
    });
}

int ltx_vae_res_block_dev(ltx_ctx* ctx, int group, int block, float* x, int F, int H, int W) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { vae_res_block(ctx, need_vae(ctx), group, block, x, F, H, W); });
}

int ltx_vae_upsample_dev(ltx_ctx* ctx, int group, const float* x, int F, int H, int W, float* out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { vae_upsample(ctx, need_vae(ctx), group, x, F, H, W, out); });
}

int ltx_vae_blend_tiles_dev(ltx_ctx* ctx, const float* const* tiles, const int* tile_frames, int n_tiles, int overlap, int H,
                            int W, float* frames_out, long frames_cap, int* n_frames_out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LTX_REQUIRE(H >= 1 && W >= 1, "ltx_vae_blend_tiles_dev: bad latent size");
        const int n = vae_blend_tiles(ctx, tiles, tile_frames, n_tiles, overlap, H, W, frames_out, frames_cap);
        if (n_frames_out) *n_frames_out = n;
    });
}

int ltx_vae_decode(ltx_ctx* ctx, const float* latent, int F, int H, int W, int has_timestep, float timestep,
                   const float* noise, int tile, int overlap, float* frames_out, long frames_cap, int* n_frames_out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        VaeModel* m = need_vae(ctx);
        LTX_REQUIRE(latent && frames_out && F >= 1 && H >= 1 && W >= 1, "ltx_vae_decode: bad arguments");
        const size_t n_lat = (size_t)128 * F * H * W * 4;
        hipStream_t st = ctx->stream;
        DevBuf dl, dn, df;
        dl.ensure(n_lat);
        HIP_CHECK(hipMemcpyAsync(dl.p, latent, n_lat, hipMemcpyHostToDevice, st));
        if (has_timestep && noise) {
            dn.ensure(n_lat);
            HIP_CHECK(hipMemcpyAsync(dn.p, noise, n_lat, hipMemcpyHostToDevice, st));
        }
        df.ensure((size_t)frames_cap * 4);
        int nf = 0;
        VaeDecodeArgs a;
        a.latent = dl.as<float>(); a.F = F; a.H = H; a.W = W;
        a.has_timestep = has_timestep; a.timestep = timestep; a.noise = (has_timestep && noise) ? dn.as<float>() : nullptr;
        a.tile = tile; a.overlap = overlap;
        a.frames = df.as<float>(); a.frames_cap = frames_cap; a.n_frames_out = &nf;
        vae_decode(ctx, m, a);
        HIP_CHECK(hipMemcpyAsync(frames_out, df.p, (size_t)nf * H * 32 * W * 32 * 3 * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (n_frames_out) *n_frames_out = nf;
    });
}

int ltx_op_conv3d(ltx_ctx* ctx, const uint16_t* x, int F, int H, int W, int Cin, const uint16_t* w, const float* bias,
                  int Cout, int causal, float* out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        GemmArgs g;
        g.A = x; g.B = w; g.ldb = 27L * Cin;
        g.M = F * H * W; g.N = Cout; g.K = 27 * Cin;
        g.conv = 1;
        g.geom.F = F; g.geom.H = H; g.geom.W = W; g.geom.C = Cin; g.geom.causal = causal; g.geom.pad_mode = 0;
        g.ep.bias_n = bias;
        g.ep.out_f32 = out; g.ep.ld_f32 = Cout;
        // a workspace, as the VAE graph passes one: the launcher may run a last partial round of tiles as a split-K launch
        if (ctx->op_ws.ensure((size_t)64 << 20)) HIP_CHECK(hipStreamSynchronize(ctx->stream));
        g.split_ws = ctx->op_ws.as<float>();
        g.split_ws_elems = (long)(ctx->op_ws.bytes / 4);
        launch_gemm_bf16(g, ctx->stream);
    });
}

// ---- two-stage glue ----
int ltx_upscaler_load(ltx_ctx* ctx, const char* path) {
    if (!ctx || !path) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (ctx->upscaler) {
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            upscaler_destroy(ctx->upscaler);
            ctx->upscaler = nullptr;
        }
        ctx->upscaler = upscaler_load(ctx, path);
    });
}

int ltx_upscaler_unload(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->upscaler) upscaler_destroy(ctx->upscaler);
        ctx->upscaler = nullptr;
    });
}

int ltx_upscale_latent_dev(ltx_ctx* ctx, const float* latent, int F, int H, int W, float* out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (!ctx->upscaler) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: spatial upscaler");
        VaeModel* v = need_vae(ctx);  // per-channel statistics come from the VAE (LTXPipeline.swift:2598-2599)
        upscaler_forward(ctx, ctx->upscaler, latent, F, H, W, v->mean, v->std_, out);
    });
}

int ltx_upscale_latent(ltx_ctx* ctx, const float* latent, int F, int H, int W, float* out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (!ctx->upscaler) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: spatial upscaler");
        VaeModel* v = need_vae(ctx);
        LTX_REQUIRE(latent && out && F >= 1 && H >= 1 && W >= 1, "ltx_upscale_latent: bad arguments");
        const size_t n_in = (size_t)128 * F * H * W * 4;
        DevBuf di, dout;
        di.ensure(n_in);
        dout.ensure(n_in * 4);
        HIP_CHECK(hipMemcpyAsync(di.p, latent, n_in, hipMemcpyHostToDevice, ctx->stream));
        upscaler_forward(ctx, ctx->upscaler, di.as<float>(), F, H, W, v->mean, v->std_, dout.as<float>());
        HIP_CHECK(hipMemcpyAsync(out, dout.p, n_in * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int ltx_adain_filter_latent_dev(ltx_ctx* ctx, float* latent, long n, const float* reference, long n_ref, int channels,
                                float factor) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LTX_REQUIRE(latent && reference && n > 0 && n_ref > 0 && channels > 0, "adain: bad arguments");
        if (factor <= 0.f) return;
        ctx->dn_stats.ensure((size_t)channels * 4 * 4 + 64);
        float* sx = ctx->dn_stats.as<float>();
        float* sr = sx + 2 * channels;
        launch_mean_var(latent, n, channels, sx, ctx->stream);
        launch_mean_var(reference, n_ref, channels, sr, ctx->stream);
        launch_adain(latent, sx, sr, factor, n, channels, ctx->stream);
    });
}

int ltx_adain_filter_latent(ltx_ctx* ctx, float* latent, long n, const float* reference, long n_ref, int channels,
                            float factor) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LTX_REQUIRE(latent && reference && n > 0 && n_ref > 0 && channels > 0, "adain: bad arguments");
        if (factor <= 0.f) return;
        DevBuf dl, dr;
        dl.ensure((size_t)n * channels * 4);
        dr.ensure((size_t)n_ref * channels * 4);
        HIP_CHECK(hipMemcpyAsync(dl.p, latent, dl.bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(dr.p, reference, dr.bytes, hipMemcpyHostToDevice, ctx->stream));
        ctx->dn_stats.ensure((size_t)channels * 4 * 4 + 64);
        float* sx = ctx->dn_stats.as<float>();
        float* sr = sx + 2 * channels;
        launch_mean_var(dl.as<float>(), n, channels, sx, ctx->stream);
        launch_mean_var(dr.as<float>(), n_ref, channels, sr, ctx->stream);
        launch_adain(dl.as<float>(), sx, sr, factor, n, channels, ctx->stream);
        HIP_CHECK(hipMemcpyAsync(latent, dl.p, dl.bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int ltx_renoise_dev(ltx_ctx* ctx, float* latent, const float* noise, float sigma, long n) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { launch_lincomb(noise, latent, sigma, 1.0f - sigma, latent, n, ctx->stream); });
}

// ---- denoise loop ----
static void fill_params(DenoiseParams& p, const ltx_denoise_options* o) {
    if (!o) return;
    // ABI revision 2: the caller declares how much of the struct it knows; nothing past that is read (round-4 advice: a host compiled
    // against the older, shorter struct must not have `step_stats` read past its end)
    const size_t sz = o->struct_size;
    LTX_REQUIRE(sz >= offsetof(ltx_denoise_options, shard) + sizeof(o->shard) && sz <= 4096,
                "ltx_denoise_options.struct_size = %zu: set it to sizeof(ltx_denoise_options) (LTX_DENOISE_OPTIONS_INIT); this library is ABI revision %d",
                sz, LTX_ABI_VERSION);
    p.cfg_scale = o->cfg_scale;
    p.guidance_rescale = o->guidance_rescale;
    p.stg_scale = o->stg_scale;
    p.stg_blocks = o->stg_blocks;
    p.n_stg = o->n_stg_blocks;
    p.ge_gamma = o->ge_gamma;
    p.cond_latent = o->cond_latent;  // device variant: used as given; host variant: replaced by staged copies below
    p.image_cond_noise_scale = o->image_cond_noise_scale;
    p.cond_noise = o->cond_noise;
    p.shard = o->shard;
    if (sz >= offsetof(ltx_denoise_options, step_stats) + sizeof(o->step_stats)) p.step_stats = o->step_stats;
}

int ltx_denoise_dev(ltx_ctx* ctx, float* latent, int F, int H, int W, const float* sigmas, int n_sigmas,
                    const uint16_t* context, const int32_t* mask, int mask_all_ones, int S, uint64_t ctx_version,
                    const ltx_denoise_options* opt, ltx_progress_cb cb, void* user) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DenoiseParams p;
        p.latent = latent; p.F = F; p.H = H; p.W = W;
        p.sigmas = sigmas; p.n_sigmas = n_sigmas;
        p.context = context; p.mask = mask; p.mask_all_ones = mask_all_ones; p.S = S;
        p.ctx_version = ctx_version;
        fill_params(p, opt);
        p.progress = cb; p.user = user;
        denoise_run(ctx, p);
    });
}

int ltx_denoise(ltx_ctx* ctx, float* latent, int F, int H, int W, const float* sigmas, int n_sigmas,
                const uint16_t* context, const int32_t* mask, int S, const ltx_denoise_options* opt,
                ltx_progress_cb cb, void* user) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        LTX_REQUIRE(latent && sigmas && context && F >= 1 && H >= 1 && W >= 1 && S >= 1 && n_sigmas >= 2, "ltx_denoise: bad arguments");
        const int nb = (opt && opt->cfg_scale > 1.0f) ? 2 : 1;
        const size_t n_lat = (size_t)m->cfg.in_channels * F * H * W * 4;
        const size_t n_ctx = (size_t)nb * S * m->cfg.caption_channels * 2;
        hipStream_t st = ctx->stream;
        ctx->h2d[5].ensure(n_lat);
        ctx->h2d[6].ensure(n_ctx);
        ctx->h2d[7].ensure((size_t)nb * S * 4);
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[5].p, latent, n_lat, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[6].p, context, n_ctx, hipMemcpyHostToDevice, st));
        int all_ones = 1;
        uint64_t hv = hash_bytes(context, n_ctx, 0xCBF29CE484222325ull);
        if (mask) {
            HIP_CHECK(hipMemcpyAsync(ctx->h2d[7].p, mask, (size_t)nb * S * 4, hipMemcpyHostToDevice, st));
            hv = hash_bytes(mask, (size_t)nb * S * 4, hv);
            for (long i = 0; i < (long)nb * S; ++i)
                if (mask[i] != 1) { all_ones = 0; break; }
        }
        DenoiseParams p;
        p.latent = ctx->h2d[5].as<float>(); p.F = F; p.H = H; p.W = W;
        p.sigmas = sigmas; p.n_sigmas = n_sigmas;
        p.context = ctx->h2d[6].as<bf16_t>();
        p.mask = mask ? ctx->h2d[7].as<int32_t>() : nullptr;
        p.mask_all_ones = all_ones; p.S = S;
        p.ctx_version = hv ? hv : 1;
        fill_params(p, opt);
        if (opt && opt->cond_latent) {  // image-to-video: stage the image latent (and the re-noising draws) on the device
            const size_t n_c = (size_t)m->cfg.in_channels * H * W * 4;
            ctx->i2v_cond.ensure(n_c);
            HIP_CHECK(hipMemcpyAsync(ctx->i2v_cond.p, opt->cond_latent, n_c, hipMemcpyHostToDevice, st));
            p.cond_latent = ctx->i2v_cond.as<float>();
            if (opt->cond_noise) {
                ctx->i2v_noise.ensure(n_c * (n_sigmas - 1));
                HIP_CHECK(hipMemcpyAsync(ctx->i2v_noise.p, opt->cond_noise, n_c * (n_sigmas - 1), hipMemcpyHostToDevice, st));
                p.cond_noise = ctx->i2v_noise.as<float>();
            }
        }
        p.progress = cb; p.user = user;
        denoise_run(ctx, p);
        HIP_CHECK(hipMemcpyAsync(latent, ctx->h2d[5].p, n_lat, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

// ---- multi-GPU ----
int ltx_dist_unique_id(void* id_out) {
    if (!id_out) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(nullptr, [&] { dist_unique_id(id_out); });
}

int ltx_dist_init(ltx_ctx* ctx, int rank, int world, const void* id) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { dist_init_native(ctx, rank, world, id); });
}

int ltx_dist_set_transport(ltx_ctx* ctx, int rank, int world, ltx_allgather_fn gather, void* user) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { dist_set_transport(ctx, rank, world, gather, user); });
}

int ltx_dist_shutdown(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { dist_shutdown(ctx); });
}

int ltx_dist_info(const ltx_ctx* ctx, int* rank, int* world, int* native_transport, long* n_collectives) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    if (rank) *rank = dist_rank(ctx);
    if (world) *world = dist_world(ctx);
    if (native_transport) *native_transport = (ctx->dist && ctx->dist->comm) ? 1 : 0;
    if (n_collectives) *n_collectives = ctx->dist ? ctx->dist->n_collectives : 0;
    return LTX_OK;
}

int ltx_dist_allgather_dev(ltx_ctx* ctx, const void* send, void* recv, long bytes) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { dist_allgather(ctx, send, recv, bytes); });
}

int ltx_dist_broadcast_dev(ltx_ctx* ctx, void* buf, long bytes, int root) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { dist_broadcast(ctx, buf, bytes, root); });
}

long ltx_dit_export_param(ltx_ctx* ctx, const char* module_key, float* out, long cap) {
    if (!ctx || !module_key) return -LTX_ERR_INVALID_CONFIGURATION;
    long n = 0;
    const int rc = guarded(ctx, [&] {
        DiTModel* m = need_dit(ctx);
        auto it = m->slots.find(module_key);
        LTX_REQUIRE(it != m->slots.end(), "ltx_dit_export_param: no parameter '%s'", module_key);
        const ParamSlot& sl = it->second;
        n = sl.numel;
        if (!out) return;
        LTX_REQUIRE(cap >= n, "ltx_dit_export_param: buffer too small (%ld < %ld)", cap, n);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        dit_export_slot(ctx, m, sl, out);
    });
    return rc == 0 ? n : -rc;
}

int ltx_prof_enable(ltx_ctx* ctx, int on) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->prof.collect();
        ctx->prof.on = on != 0;
    });
}

int ltx_prof_collect(ltx_ctx* ctx, int kind, double* total_ms, long* launches, double* work, int reset) {
    if (!ctx || kind < 0 || kind >= PROF_NKINDS) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->prof.collect();
        if (total_ms) *total_ms = ctx->prof.total_ms[kind];
        if (launches) *launches = ctx->prof.launches[kind];
        if (work) *work = ctx->prof.total_work[kind];
        if (reset) ctx->prof.reset();
    });
}

// ---- kernel-level hooks ----
int ltx_op_gemm_bf16(ltx_ctx* ctx, const uint16_t* A, long lda, const uint16_t* B, long ldb, const float* bias, int M,
                     int N, int K, int act, int tile_cfg, float* out_f32, long ld_f32, uint16_t* out_bf16,
                     long ld_bf16) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        GemmArgs g;
        g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
        g.ep.bias_n = bias;
        g.ep.act = act;
        g.ep.out_f32 = out_f32; g.ep.ld_f32 = ld_f32;
        g.ep.out_bf16 = out_bf16; g.ep.ld_bf16 = ld_bf16;
        if (tile_cfg >= 100) {  // S*100 + cfg: S-way split-K on a ring tile (workspace owned by the context)
            g.split_k = tile_cfg / 100;
            tile_cfg %= 100;
            if (ctx->op_ws.ensure((size_t)g.split_k * M * N * 4)) HIP_CHECK(hipStreamSynchronize(ctx->stream));
            g.split_ws = ctx->op_ws.as<float>();
            g.split_ws_elems = (long)(ctx->op_ws.bytes / 4);
        }
        if (tile_cfg < 0) launch_gemm_bf16(g, ctx->stream); else launch_gemm_bf16_cfg(g, tile_cfg, ctx->stream);
    });
}

int ltx_op_gemm_q8(ltx_ctx* ctx, const uint16_t* A, long lda, const uint8_t* codes, const uint16_t* scales, const uint16_t* biases,
                   const float* bias, int M, int N, int K, int split_k, int via_scratch, int tile_cfg, float* out_f32, long ld_f32) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LTX_REQUIRE(A && codes && scales && biases && out_f32 && split_k >= 1, "ltx_op_gemm_q8: null argument");
        LTX_REQUIRE(gemm_takes_codes(M, N, K), "ltx_op_gemm_q8: few-row launches only (M <= 256, K %% 64 == 0, K >= 256, N %% 4 == 0; M=%d N=%d K=%d)", M, N, K);
        GemmArgs g;
        g.A = A; g.lda = lda; g.ldb = K; g.M = M; g.N = N; g.K = K;
        g.ep.bias_n = bias;
        g.ep.out_f32 = out_f32; g.ep.ld_f32 = ld_f32;
        DevBuf scratch;
        if (via_scratch) {  // the path every other launch of a quantised Linear takes: codes -> bf16 scratch matrix -> the same kernel
            scratch.ensure((size_t)N * K * 2);
            launch_dequant(codes, scales, biases, N, K, 8, scratch.as<bf16_t>(), ctx->stream);
            g.B = scratch.as<bf16_t>();
        } else {
            g.Bq = codes; g.Bqs = scales; g.Bqb = biases;
        }
        if (split_k > 1) {
            g.split_k = split_k;
            if (ctx->op_ws.ensure((size_t)split_k * M * N * 4)) HIP_CHECK(hipStreamSynchronize(ctx->stream));
            g.split_ws = ctx->op_ws.as<float>();
        }
        LTX_REQUIRE(tile_cfg == 29 || (tile_cfg == 30 && gemm_fewrow_takes(M, N, K, split_k)), "ltx_op_gemm_q8: tile_cfg 29 (ring kernel) or 30 (few-row kernel: M <= 128, N %% 64 == 0, whole 256-wide macro-tiles of K per split)");
        launch_gemm_bf16_cfg(g, tile_cfg, ctx->stream);
        if (via_scratch) HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int ltx_op_value_projection_t(ltx_ctx* ctx, const uint16_t* X, long ldx, int tokens, const uint16_t* W, const float* bias, int out_features,
                              int in_features, uint16_t* vt, long ldvt) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LinearW w;
        w.w = (bf16_t*)W;
        w.b = (float*)bias;
        w.out = out_features;
        w.in = in_features;
        gemm_vt(X, ldx, tokens, w, vt, ldvt, ctx->stream);
    });
}

int ltx_op_gemm_bf16_gated_residual(ltx_ctx* ctx, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                    const float* bias, const float* gate, float gate_scalar, int M, int N, int K,
                                    float* x, long ldx, uint16_t* mirror, long ld_mirror) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        GemmArgs g;
        g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
        g.ep.bias_n = bias;
        g.ep.out_f32 = x; g.ep.ld_f32 = ldx;
        g.ep.resid = 1;
        g.ep.gate = gate; g.ep.gate_bstride = 0; g.ep.rows_per_batch = M; g.ep.gate_scalar = gate_scalar;
        g.ep.out_bf16 = mirror; g.ep.ld_bf16 = ld_mirror;
        launch_gemm_bf16(g, ctx->stream);
    });
}

int ltx_op_gemm_bf16_gated_residual_norm(ltx_ctx* ctx, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                         const float* bias, const float* gate, float gate_scalar, int M, int N, int K,
                                         float* x, long ldx, const float* scale, const float* shift, float eps,
                                         uint16_t* xn, long ldxn, int fused) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LTX_REQUIRE(A && B && x && scale && shift && xn, "ltx_op_gemm_bf16_gated_residual_norm: null argument");
        GemmArgs g;
        g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
        g.ep.bias_n = bias;
        g.ep.out_f32 = x; g.ep.ld_f32 = ldx;
        g.ep.resid = 1;
        g.ep.gate = gate; g.ep.gate_bstride = 0; g.ep.rows_per_batch = M; g.ep.gate_scalar = gate_scalar;
        // the DiT graph's launches bring a workspace and leave the split to the launcher
        if (ctx->op_ws.ensure((size_t)2 * M * N * 4)) HIP_CHECK(hipStreamSynchronize(ctx->stream));
        g.split_k = 0;
        g.split_ws = ctx->op_ws.as<float>();
        g.split_ws_elems = (long)(ctx->op_ws.bytes / 4);
        NormAfter na;
        na.scale = scale; na.shift = shift; na.mod_bstride = 0; na.rows_per_batch = M;
        na.out = xn; na.ldo = ldxn; na.eps = eps; na.norm_kind = LTX_NORM_RMS;
        if (fused) {
            launch_gemm_bf16(g, ctx->stream, &na);
        } else {
            launch_gemm_bf16(g, ctx->stream);
            launch_norm_mod(x, ldx, scale, shift, 0, M, xn, ldxn, M, N, LTX_NORM_RMS, eps, 0, ctx->stream);
        }
    });
}

int ltx_op_gemv_f32(ltx_ctx* ctx, const float* a, long lda, const uint16_t* W, long ldw, const float* bias, float* out,
                    long ldo, int M, int N, int K, int in_act) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { launch_gemv_f32(a, lda, W, ldw, bias, out, ldo, M, N, K, in_act, ctx->stream); });
}

int ltx_op_attention(ltx_ctx* ctx, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, long ldvt,
                     const float* bias, int B, int H, int Tq, int Tk, float scale, uint16_t* O) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        AttnArgs a;
        const long D = (long)H * 128;
        a.Q = Q; a.ldq = D; a.q_bstride = (long)Tq * D;
        a.K = K; a.ldk = D; a.k_bstride = (long)Tk * D;
        a.Vt = Vt; a.ldvt = ldvt; a.vt_bstride = D * ldvt;
        a.O = O; a.ldo = D; a.o_bstride = (long)Tq * D;
        a.bias = bias; a.bias_bstride = Tk;
        a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale;
        a.q_prescaled = scale <= 0.f ? 1 : 0;  // Q already carries scale * log2(e): scores are base-2 exponents
        // few queries against many keys: lend the launcher room to divide the keys over workgroups, as the DiT forward does
        if (const long need = attn_split_ws_bytes(B, H, Tq, Tk)) {
            if (ctx->op_ws.ensure((size_t)need)) HIP_CHECK(hipStreamSynchronize(ctx->stream));
            a.split_ws = ctx->op_ws.p;
            a.split_ws_bytes = (long)ctx->op_ws.bytes;
        }
        launch_attention(a, ctx->stream);
    });
}

int ltx_attention_key_splits(int B, int H, int Tq, int Tk) {
    if (B < 1 || H < 1 || Tq < 1 || Tk < 1) return 1;
    return attn_key_splits(B, H, Tq, Tk);
}

int ltx_op_norm_mod(ltx_ctx* ctx, const float* x, const float* scale, const float* shift, int rows, int D,
                    int norm_kind, float eps, int round_norm_bf16, uint16_t* out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        launch_norm_mod(x, D, scale, shift, 0, rows, out, D, rows, D, norm_kind, eps, round_norm_bf16, ctx->stream);
    });
}

int ltx_op_qknorm_rope(ltx_ctx* ctx, const float* x, long ldx, const float* w, const float* cos_t, const float* sin_t,
                       int T, int rows, int D, float eps, uint16_t* out) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { launch_qknorm_rope(x, ldx, w, cos_t, sin_t, T, out, D, rows, D, eps, ctx->stream); });
}

int ltx_op_fill_normal_bf16(ltx_ctx* ctx, uint16_t* p, long n, uint64_t seed, float mean, float stddev) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { launch_fill_normal_bf16(p, n, seed, mean, stddev, ctx->stream); });
}
int ltx_op_fill_normal_f32(ltx_ctx* ctx, float* p, long n, uint64_t seed, float mean, float stddev) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { launch_fill_normal_f32(p, n, seed, mean, stddev, 0, ctx->stream); });
}



/* ---- text-embedding connector ---- */
namespace {
extern "C++" ConnectorConfig to_conn_cfg(const ltx_connector_config* c) {
    ConnectorConfig k;
    if (c) {
        k.dim = c->dim; k.heads = c->heads; k.layers = c->layers; k.registers = c->registers; k.states = c->states;
        k.theta = c->theta; k.max_pos = c->max_pos;
    }
    return k;
}
void drop_connector(ltx_ctx* ctx) {
    if (ctx->connector) {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        connector_destroy(ctx->connector);
        ctx->connector = nullptr;
    }
}
}  // namespace

void ltx_connector_config_default(ltx_connector_config* cfg) {
    if (!cfg) return;
    const ConnectorConfig k;
    cfg->dim = k.dim; cfg->heads = k.heads; cfg->layers = k.layers; cfg->registers = k.registers; cfg->states = k.states;
    cfg->theta = k.theta; cfg->max_pos = k.max_pos;
}

int ltx_connector_load(ltx_ctx* ctx, const char* path, const ltx_connector_config* cfg) {
    if (!ctx || !path) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        drop_connector(ctx);
        ConnectorModel* m = connector_create(to_conn_cfg(cfg));
        try {
            connector_load_safetensors(ctx, m, path);
        } catch (...) {
            connector_destroy(m);
            throw;
        }
        ctx->connector = m;
    });
}

int ltx_connector_init_synthetic(ltx_ctx* ctx, const ltx_connector_config* cfg, unsigned long seed) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        drop_connector(ctx);
        ctx->connector = connector_create(to_conn_cfg(cfg));
        connector_init_synthetic(ctx, ctx->connector, seed);
    });
}

int ltx_connector_unload(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { drop_connector(ctx); });
}

int ltx_connector_encode_taps_dev(ltx_ctx* ctx, const uint16_t* hidden, const int32_t* attention_mask, int B, int T,
                                  int padding_right, uint16_t* context, uint16_t* norm_concat, uint16_t* fe_out,
                                  float* after_registers) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (!ctx->connector) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model not loaded: text-embedding connector");
        ConnectorArgs a;
        a.hidden = hidden; a.mask = attention_mask; a.B = B; a.T = T; a.padding_right = padding_right;
        a.out = context; a.dbg_nc = norm_concat; a.dbg_fe = fe_out; a.dbg_reg = after_registers;
        connector_encode(ctx, ctx->connector, a);
    });
}

int ltx_connector_encode_dev(ltx_ctx* ctx, const uint16_t* hidden, const int32_t* attention_mask, int B, int T,
                             int padding_right, uint16_t* context, int32_t* out_mask) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (!ctx->connector) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model not loaded: text-embedding connector");
        ConnectorArgs a;
        a.hidden = hidden; a.mask = attention_mask; a.B = B; a.T = T; a.padding_right = padding_right;
        a.out = context; a.out_mask = out_mask;
        connector_encode(ctx, ctx->connector, a);
    });
}

int ltx_connector_encode(ltx_ctx* ctx, const uint16_t* hidden, const int32_t* attention_mask, int B, int T,
                         int padding_right, uint16_t* context, int32_t* out_mask) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        if (!ctx->connector) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model not loaded: text-embedding connector");
        LTX_REQUIRE(hidden && attention_mask && context && B >= 1 && T >= 1, "connector_encode: bad arguments");
        const ConnectorConfig& c = ctx->connector->cfg;
        const size_t hb = (size_t)c.states * B * T * c.dim * 2, mb = (size_t)B * T * 4, ob = (size_t)B * T * c.dim * 2;
        ctx->h2d[0].ensure(hb);
        ctx->h2d[1].ensure(mb);
        ctx->h2d[2].ensure(ob);
        ctx->h2d[3].ensure(mb);
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[0].p, hidden, hb, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[1].p, attention_mask, mb, hipMemcpyHostToDevice, ctx->stream));
        ConnectorArgs a;
        a.hidden = ctx->h2d[0].as<bf16_t>(); a.mask = ctx->h2d[1].as<int32_t>(); a.B = B; a.T = T; a.padding_right = padding_right;
        a.out = ctx->h2d[2].as<bf16_t>(); a.out_mask = ctx->h2d[3].as<int32_t>();
        connector_encode(ctx, ctx->connector, a);
        HIP_CHECK(hipMemcpyAsync(context, ctx->h2d[2].p, ob, hipMemcpyDeviceToHost, ctx->stream));
        if (out_mask) HIP_CHECK(hipMemcpyAsync(out_mask, ctx->h2d[3].p, mb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int ltx_map_text_encoder_key(const char* file_key, char* out, int cap) {
    std::string mk;
    if (!file_key || !map_text_encoder_file_key(file_key, &mk)) return 0;
    return copy_str(mk, out, cap);
}

int ltx_rope_tables_1d(int T, int dim, float theta, int max_pos, float* cos_out, float* sin_out) {
    if (!cos_out || !sin_out || T < 1 || dim < 2 || max_pos < 1) return LTX_ERR_INVALID_CONFIGURATION;
    std::vector<float> c, s;
    rope_tables_1d(T, dim, (double)theta, max_pos, &c, &s);
    memcpy(cos_out, c.data(), c.size() * 4);
    memcpy(sin_out, s.data(), s.size() * 4);
    return LTX_OK;
}


/* ---- VAE encoder ---- */
namespace {
void drop_vae_encoder(ltx_ctx* ctx) {
    if (ctx->vae_encoder) {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        vae_encoder_destroy(ctx->vae_encoder);
        ctx->vae_encoder = nullptr;
    }
}
void encode_common(ltx_ctx* ctx, const float* px_dev, int T, int H, int W, int normalize, float* lat_dev) {
    if (!ctx->vae_encoder) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model not loaded: VAE encoder failed to load");
    const float* mean = nullptr;
    const float* stdv = nullptr;
    if (normalize) {
        if (!ctx->vae) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model not loaded: VAE decoder not loaded (needed for latent statistics)");
        mean = ctx->vae->mean;
        stdv = ctx->vae->std_;
    }
    vae_encoder_encode(ctx, ctx->vae_encoder, px_dev, T, H, W, mean, stdv, lat_dev, nullptr);
}
}  // namespace

int ltx_vae_encoder_load(ltx_ctx* ctx, const char* path, int channel_base) {
    if (!ctx || !path) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        drop_vae_encoder(ctx);
        VaeEncoderModel* m = vae_encoder_create(channel_base > 0 ? channel_base : 128);
        try {
            vae_encoder_load_safetensors(ctx, m, path);
        } catch (...) {
            vae_encoder_destroy(m);
            throw;
        }
        ctx->vae_encoder = m;
    });
}

int ltx_vae_encoder_init_synthetic(ltx_ctx* ctx, int channel_base, unsigned long seed) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        drop_vae_encoder(ctx);
        ctx->vae_encoder = vae_encoder_create(channel_base > 0 ? channel_base : 128);
        vae_encoder_init_synthetic(ctx, ctx->vae_encoder, seed);
    });
}

int ltx_vae_encoder_unload(ltx_ctx* ctx) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { drop_vae_encoder(ctx); });
}

int ltx_vae_encoder_latent_frames(int T) { return T < 1 ? 0 : vae_encoder_latent_frames(T); }

int ltx_vae_encode_dev(ltx_ctx* ctx, const float* pixels, int T, int H, int W, int normalize, float* latent) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] { encode_common(ctx, pixels, T, H, W, normalize, latent); });
}

int ltx_vae_encode(ltx_ctx* ctx, const float* pixels, int T, int H, int W, int normalize, float* latent) {
    if (!ctx) return LTX_ERR_INVALID_CONFIGURATION;
    return guarded(ctx, [&] {
        LTX_REQUIRE(pixels && latent && T >= 1 && H >= 1 && W >= 1, "ltx_vae_encode: bad arguments");
        const size_t nb_in = (size_t)3 * T * H * W * 4;
        const size_t nb_out = (size_t)128 * vae_encoder_latent_frames(T) * (H / 32) * (W / 32) * 4;
        ctx->h2d[0].ensure(nb_in);
        ctx->h2d[1].ensure(nb_out ? nb_out : 4);
        HIP_CHECK(hipMemcpyAsync(ctx->h2d[0].p, pixels, nb_in, hipMemcpyHostToDevice, ctx->stream));
        encode_common(ctx, ctx->h2d[0].as<float>(), T, H, W, normalize, ctx->h2d[1].as<float>());
        HIP_CHECK(hipMemcpyAsync(latent, ctx->h2d[1].p, nb_out, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int ltx_map_vae_encoder_key(const char* file_key, char* out, int cap) {
    std::string mk;
    if (!file_key || !map_vae_encoder_file_key(file_key, &mk)) return 0;
    return copy_str(mk, out, cap);
}


/* ---- MLX-compatible noise ---- */
int ltx_mlx_random_normal(uint64_t seed, int draw_index, float* out, long n) {
    if (!out || n < 0 || draw_index < 0) return LTX_ERR_INVALID_CONFIGURATION;
    mlx_random_normal(seed, draw_index, n, out);
    return LTX_OK;
}

void ltx_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) { threefry2x32(key, ctr, out); }

/* ---- frame export ---- */
int ltx_frames_to_u8(const float* frames, long n, uint8_t* out) {
    if (!frames || !out || n < 0) return LTX_ERR_INVALID_CONFIGURATION;
    frames_to_u8(frames, n, out);
    return LTX_OK;
}

int ltx_write_png(const char* path, const uint8_t* rgb, int width, int height) {
    return write_png_rgb8(path, rgb, width, height) ? LTX_OK : LTX_ERR_FILE_NOT_FOUND;
}

}  // extern "C"
