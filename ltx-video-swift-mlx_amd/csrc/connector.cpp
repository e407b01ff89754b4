// connector.cpp - see connector.h.
#include "connector.h"

#include <string.h>

#include "attention.h"
#include "elementwise.h"
#include "gemm.h"
#include "hostmath.h"
#include "linear_ops.h"

namespace {
struct Pending {
    std::string key;
    void** dst;
    int kind;
    long rows, cols;
    int init;
};
long slot_bytes(int kind, long numel) { return numel * (kind == SLOT_BF16 ? 2 : 4); }
}  // namespace

ConnectorModel* connector_create(const ConnectorConfig& cfg) {
    LTX_REQUIRE(cfg.dim == cfg.heads * 128, "connector: dim %d must be heads (%d) x 128", cfg.dim, cfg.heads);
    LTX_REQUIRE(cfg.dim % 64 == 0 && cfg.dim <= 8192, "connector: dim %d must be a multiple of 64 and <= 8192", cfg.dim);
    LTX_REQUIRE(cfg.layers >= 1 && cfg.registers >= 1 && cfg.states >= 1, "connector: empty model");
    ConnectorModel* m = new ConnectorModel();
    m->cfg = cfg;
    m->blocks.resize(cfg.layers);
    const int D = cfg.dim;
    std::vector<Pending> pend;
    auto lin = [&](const std::string& name, LinearW& l, int out, int in) {
        l.out = out;
        l.in = in;
        pend.push_back({name + ".weight", (void**)&l.w, SLOT_BF16, out, in, 0});
        pend.push_back({name + ".bias", (void**)&l.b, SLOT_F32, out, 0, 0});
    };
    m->fe.out = D;
    m->fe.in = D * cfg.states;
    pend.push_back({"feature_extractor.aggregate_embed.weight", (void**)&m->fe.w, SLOT_BF16, D, (long)D * cfg.states, 0});
    pend.push_back({"embeddings_connector.learnable_registers", (void**)&m->registers, SLOT_F32, cfg.registers, D, 0});
    for (int i = 0; i < cfg.layers; ++i) {
        ConnBlock& b = m->blocks[i];
        const std::string p = "embeddings_connector.transformer_1d_blocks." + std::to_string(i) + ".";
        b.qk.out = 2 * D;
        b.qk.in = D;
        pend.push_back({p + "attn1.__qk.weight", (void**)&b.qk.w, SLOT_BF16, 2 * D, D, 0});
        pend.push_back({p + "attn1.__qk.bias", (void**)&b.qk.b, SLOT_F32, 2 * D, 0, 0});
        lin(p + "attn1.to_v", b.v, D, D);
        lin(p + "attn1.to_out", b.o, D, D);
        pend.push_back({p + "attn1.q_norm.weight", (void**)&b.qn, SLOT_F32, D, 0, 1});
        pend.push_back({p + "attn1.k_norm.weight", (void**)&b.kn, SLOT_F32, D, 0, 1});
        lin(p + "ff.project_in.proj", b.ff1, 4 * D, D);
        lin(p + "ff.project_out", b.ff2, D, 4 * D);
    }
    size_t total = 0;
    for (auto& q : pend) total += DeviceArena::padded(slot_bytes(q.kind, q.rows * (q.cols ? q.cols : 1)));
    m->weight_bytes = total;
    m->arena.reserve(total + 256);
    HIP_CHECK(hipMemset(m->arena.buf.p, 0, m->arena.buf.bytes));
    for (auto& q : pend) {
        const long numel = q.rows * (q.cols ? q.cols : 1);
        *q.dst = m->arena.take(slot_bytes(q.kind, numel));
        ParamSlot s;
        s.dst = *q.dst;
        s.kind = q.kind;
        s.numel = numel;
        s.rows = q.rows;
        s.cols = q.cols;
        s.init = q.init;
        m->slots[q.key] = s;
    }
    for (int i = 0; i < cfg.layers; ++i) {
        ConnBlock& b = m->blocks[i];
        const std::string p = "embeddings_connector.transformer_1d_blocks." + std::to_string(i) + ".";
        auto view = [&](const std::string& key, void* dst, int kind, long rows, long cols) {
            ParamSlot s;
            s.dst = dst;
            s.kind = kind;
            s.rows = rows;
            s.cols = cols;
            s.numel = rows * (cols ? cols : 1);
            m->slots[key] = s;
        };
        view(p + "attn1.to_q.weight", b.qk.w, SLOT_BF16, D, D);
        view(p + "attn1.to_k.weight", b.qk.w + (long)D * D, SLOT_BF16, D, D);
        view(p + "attn1.to_q.bias", b.qk.b, SLOT_F32, D, 0);
        view(p + "attn1.to_k.bias", b.qk.b + D, SLOT_F32, D, 0);
        m->slots.erase(p + "attn1.__qk.weight");
        m->slots.erase(p + "attn1.__qk.bias");
    }
    for (auto& kv : m->slots)
        if (kv.second.init == 1) launch_fill_const_f32((float*)kv.second.dst, kv.second.numel, 1.0f, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return m;
}

void connector_destroy(ConnectorModel* m) { delete m; }

void connector_load_safetensors(ltx_ctx* ctx, ConnectorModel* m, const std::string& path) {
    SafeTensors st;
    st.open(path);
    ctx->n_loaded = ctx->n_missing = ctx->n_unmatched = 0;
    for (auto& kv : m->slots) kv.second.loaded = false;
    std::vector<uint8_t> staging;
    for (auto& kv : st.tensors) {
        std::string mk;
        if (!map_text_encoder_file_key(kv.first, &mk)) continue;  // DiT / VAE / audio tensors of a unified file
        if (mk.compare(0, 26, "audio_embeddings_connector") == 0) continue;
        auto it = m->slots.find(mk);
        if (it == m->slots.end()) {
            ctx->n_unmatched++;
            continue;
        }
        ParamSlot& s = it->second;
        const StTensor& t = kv.second;
        if (t.numel() != s.numel)
            LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: %s has %ld elements, expected %ld", kv.first.c_str(), t.numel(), s.numel);
        if (s.kind == SLOT_BF16) {
            staging.resize((size_t)s.numel * 2);
            st_to_bf16(st, t, (bf16_t*)staging.data());
        } else {
            staging.resize((size_t)s.numel * 4);
            float* f = (float*)staging.data();
            st_to_f32(st, t, f);
            for (long i = 0; i < s.numel; ++i) f[i] = host_bf16_to_f32(host_f32_to_bf16(f[i]));
        }
        HIP_CHECK(hipMemcpy(s.dst, staging.data(), staging.size(), hipMemcpyHostToDevice));
        s.loaded = true;
        ctx->n_loaded++;
    }
    for (auto& kv : m->slots)
        if (!kv.second.loaded) ctx->n_missing++;
}

void connector_init_synthetic(ltx_ctx* ctx, ConnectorModel* m, uint64_t seed) {
    uint64_t k = 0;
    for (auto& kv : m->slots) {
        ParamSlot& s = kv.second;
        const uint64_t sd = seed * 0x9E3779B97F4A7C15ull + (++k) * 0xD1B54A32D192ED03ull;
        const std::string& key = kv.first;
        const bool is_bias = key.size() > 5 && key.compare(key.size() - 5, 5, ".bias") == 0;
        const bool is_norm = key.find("_norm.weight") != std::string::npos;
        const bool is_reg = key.find("learnable_registers") != std::string::npos;
        const bool is_fe = key.find("aggregate_embed") != std::string::npos;
        if (s.kind == SLOT_BF16)
            launch_fill_normal_bf16((bf16_t*)s.dst, s.numel, sd, 0.f, is_fe ? 1.0f / sqrtf((float)m->fe.in) : 0.02f, ctx->stream);
        else if (is_norm)
            launch_fill_normal_f32((float*)s.dst, s.numel, sd, 1.0f, 0.02f, 1, ctx->stream);
        else
            launch_fill_normal_f32((float*)s.dst, s.numel, sd, 0.f, is_reg ? 0.5f : (is_bias ? 0.01f : 0.02f), 1, ctx->stream);
        s.loaded = true;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

void connector_encode(ltx_ctx* ctx, ConnectorModel* m, const ConnectorArgs& a) {
    const ConnectorConfig& c = m->cfg;
    const int D = c.dim, NS = c.states, B = a.B, T = a.T;
    LTX_REQUIRE(a.hidden && a.mask && a.out, "connector_encode: null argument");
    LTX_REQUIRE(B >= 1 && T >= 1, "connector_encode: empty batch");
    if (T % c.registers != 0)  // fatalError in the reference (LTXTextEncoder.swift:435-437)
        LTX_THROW(LTXS_INVALID_CONFIGURATION, "Sequence length %d must be divisible by numLearnableRegisters %d", T, c.registers);
    hipStream_t st = ctx->stream;
    const long rows = (long)B * T;
    const int Tpad = ((T + 63) / 64) * 64;
    bool grew = false;
    grew |= m->ws_nc.ensure((size_t)rows * D * NS * 2);
    grew |= m->ws_part.ensure((size_t)fe_stats_partials_floats(NS, B, T, D) * 4);
    grew |= m->ws_stats.ensure((size_t)NS * B * 2 * 4);
    grew |= m->ws_enc.ensure((size_t)rows * D * 2);
    grew |= m->ws_src.ensure((size_t)rows * 4);
    grew |= m->ws_x.ensure((size_t)rows * D * 4);
    grew |= m->ws_xn.ensure((size_t)rows * D * 2);
    grew |= m->ws_qk.ensure((size_t)rows * 2 * D * 4);
    grew |= m->ws_q.ensure((size_t)rows * D * 2);
    grew |= m->ws_k.ensure((size_t)rows * D * 2);
    grew |= m->ws_vt.ensure((size_t)B * D * Tpad * 2, true);
    grew |= m->ws_ao.ensure((size_t)rows * D * 2);
    grew |= m->ws_ffh.ensure((size_t)rows * 4 * D * 2);
    if (m->rope_T != T) {
        std::vector<float> hc, hs;
        rope_tables_1d(T, D, (double)c.theta, c.max_pos, &hc, &hs);
        for (auto& v : hc) v = host_bf16_to_f32(host_f32_to_bf16(v));  // cast to the activations' dtype (:498)
        for (auto& v : hs) v = host_bf16_to_f32(host_f32_to_bf16(v));
        HIP_CHECK(hipStreamSynchronize(st));
        m->rope_cos.ensure(hc.size() * 4);
        m->rope_sin.ensure(hs.size() * 4);
        HIP_CHECK(hipMemcpy(m->rope_cos.p, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->rope_sin.p, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        m->rope_T = T;
    }
    (void)grew;
    bf16_t* nc = m->ws_nc.as<bf16_t>();
    bf16_t* enc = m->ws_enc.as<bf16_t>();
    float* x = m->ws_x.as<float>();
    bf16_t* xn = m->ws_xn.as<bf16_t>();
    float* qk = m->ws_qk.as<float>();
    bf16_t* q = m->ws_q.as<bf16_t>();
    bf16_t* k = m->ws_k.as<bf16_t>();
    bf16_t* vt = m->ws_vt.as<bf16_t>();
    bf16_t* ao = m->ws_ao.as<bf16_t>();
    bf16_t* ffh = m->ws_ffh.as<bf16_t>();

    // feature extractor: per-(state, batch) statistics over the valid tokens, 8*(x-mean)/(range+eps), concat, projection
    launch_fe_stats(a.hidden, a.mask, NS, B, T, D, a.padding_right, c.eps, m->ws_part.as<float>(), m->ws_stats.as<float>(), st);
    launch_fe_norm_concat(a.hidden, a.mask, m->ws_stats.as<float>(), NS, B, T, D, a.padding_right, c.eps, nc, st);
    {
        GemmEpilogue e;  // f32 accumulate over 188160 products, one rounding to bf16 (LTXTextEncoder.swift:180-186)
        e.out_bf16 = enc;
        e.ld_bf16 = D;
        gemm_linear(nc, (long)D * NS, m->fe, (int)rows, e, st);
    }
    if (a.dbg_nc) HIP_CHECK(hipMemcpyAsync(a.dbg_nc, nc, (size_t)rows * D * NS * 2, hipMemcpyDeviceToDevice, st));
    if (a.dbg_fe) HIP_CHECK(hipMemcpyAsync(a.dbg_fe, enc, (size_t)rows * D * 2, hipMemcpyDeviceToDevice, st));
    // learnable registers replace the padded positions (valid tokens first)
    launch_register_plan(a.mask, B, T, m->ws_src.as<int32_t>(), st);
    launch_register_gather(enc, m->registers, m->ws_src.as<int32_t>(), B, T, D, c.registers, x, st);
    if (a.dbg_reg) HIP_CHECK(hipMemcpyAsync(a.dbg_reg, x, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, st));

    for (int l = 0; l < c.layers; ++l) {
        const ConnBlock& blk = m->blocks[l];
        launch_norm_mod(x, D, nullptr, nullptr, 0, T, xn, D, (int)rows, D, LTX_NORM_RMS, c.eps, 0, st);
        GemmEpilogue eqk;
        eqk.out_f32 = qk;
        eqk.ld_f32 = 2 * D;
        gemm_linear(xn, D, blk.qk, (int)rows, eqk, st);
        launch_qknorm_rope2(qk, blk.qn, q, qk + D, blk.kn, k, 2 * D, D, m->rope_cos.as<float>(), m->rope_sin.as<float>(), T, (int)rows, D,
                            c.eps, st);
        for (int b = 0; b < B; ++b) gemm_vt(xn + (size_t)b * T * D, D, T, blk.v, vt + (size_t)b * D * Tpad, Tpad, st);
        AttnArgs at;
        at.Q = q; at.ldq = D; at.q_bstride = (long)T * D;
        at.K = k; at.ldk = D; at.k_bstride = (long)T * D;
        at.Vt = vt; at.ldvt = Tpad; at.vt_bstride = (long)D * Tpad;
        at.O = ao; at.ldo = D; at.o_bstride = (long)T * D;
        at.B = B; at.H = c.heads; at.Tq = T; at.Tk = T;
        launch_attention(at, st);
        GemmEpilogue eo;  // x += to_out(attn)
        eo.out_f32 = x;
        eo.ld_f32 = D;
        eo.resid = 1;
        eo.gate_scalar = 1.0f;
        gemm_linear(ao, D, blk.o, (int)rows, eo, st);
        launch_norm_mod(x, D, nullptr, nullptr, 0, T, xn, D, (int)rows, D, LTX_NORM_RMS, c.eps, 0, st);
        GemmEpilogue e1;
        e1.out_bf16 = ffh;
        e1.ld_bf16 = 4 * D;
        e1.act = LTX_ACT_GELU_TANH;
        gemm_linear(xn, D, blk.ff1, (int)rows, e1, st);
        GemmEpilogue e2;  // x += project_out(h)
        e2.out_f32 = x;
        e2.ld_f32 = D;
        e2.resid = 1;
        e2.gate_scalar = 1.0f;
        gemm_linear(ffh, 4 * D, blk.ff2, (int)rows, e2, st);
    }
    launch_norm_mod(x, D, nullptr, nullptr, 0, T, a.out, D, (int)rows, D, LTX_NORM_RMS, c.eps, 0, st);
    // after the register replacement every position is valid: the mask the loop receives is all ones (:466-468, :622-626)
    if (a.out_mask) launch_fill_const_i32(a.out_mask, rows, 1, st);
}
