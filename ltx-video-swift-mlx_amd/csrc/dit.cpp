// dit.cpp - see dit.h. One forward = reference LTXTransformer.callAsFunction (LTXTransformer.swift:235-486).
#include "dit.h"

#include <stdlib.h>
#include <string.h>

#include "attention.h"
#include "dist.h"
#include "elementwise.h"
#include "gemm.h"
#include "linear_ops.h"
#include "options.h"

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

DiTModel* dit_create(const TransformerConfig& cfg) {
    LTX_REQUIRE(cfg.head_dim == 128, "DiT: attention_head_dim must be 128 (got %d)", cfg.head_dim);
    LTX_REQUIRE(cfg.num_layers >= 1 && cfg.num_heads >= 1, "DiT: empty model");
    LTX_REQUIRE(cfg.cross_attention_dim == cfg.inner_dim(), "DiT: cross_attention_dim (%d) must equal inner dim (%d)",
                cfg.cross_attention_dim, cfg.inner_dim());
    LTX_REQUIRE(cfg.in_channels % 64 == 0 && cfg.caption_channels % 64 == 0, "DiT: in_channels/caption_channels must be multiples of 64");
    LTX_REQUIRE(cfg.out_channels % 4 == 0, "DiT: out_channels must be a multiple of 4");
    DiTModel* m = new DiTModel();
    m->cfg = cfg;
    m->D = cfg.inner_dim();
    m->L = cfg.num_layers;
    m->blocks.resize(m->L);
    const int D = m->D;

    std::vector<Pending> pend;
    auto lin = [&](const std::string& name, LinearW& l, int out, int in, bool fused_rows = false) {
        l.out = out;
        l.in = in;
        if (!fused_rows) {
            pend.push_back({name + ".weight", (void**)&l.w, SLOT_BF16, out, in, 0});
            pend.push_back({name + ".bias", (void**)&l.b, SLOT_F32, out, 0, 0});
        }
    };
    auto vec = [&](const std::string& name, float*& p, long rows, long cols, int init) {
        pend.push_back({name, (void**)&p, SLOT_F32, rows, cols, init});
    };
    lin("patchify_proj", m->patchify, D, cfg.in_channels);
    lin("adaln_single.emb.linear_1", m->ada_l1, D, 256);
    lin("adaln_single.emb.linear_2", m->ada_l2, D, D);
    lin("adaln_single.linear", m->ada_lin, 6 * D, D);
    lin("caption_projection.linear_1", m->cap_l1, D, cfg.caption_channels);
    lin("caption_projection.linear_2", m->cap_l2, D, D);
    lin("proj_out", m->proj_out, cfg.out_channels, D);
    vec("scale_shift_table", m->sst_out, 2, D, 0);
    // all per-block scale-shift tables live in one [L][6][D] tensor so one kernel builds every layer's modulation
    pend.push_back({"__sst_blocks", (void**)&m->sst_blocks, SLOT_F32, (long)m->L * 6, D, 0});
    for (int i = 0; i < m->L; ++i) {
        DiTBlock& b = m->blocks[i];
        const std::string p = "transformer_blocks." + std::to_string(i) + ".";
        // attn1: to_q and to_k rows fused into one [2D][D] matrix (+ [2D] bias)
        b.qk1.out = 2 * D;
        b.qk1.in = D;
        pend.push_back({p + "attn1.__qk.weight", (void**)&b.qk1.w, SLOT_BF16, 2 * D, D, 0});
        pend.push_back({p + "attn1.__qk.bias", (void**)&b.qk1.b, SLOT_F32, 2 * D, 0, 0});
        lin(p + "attn1.to_v", b.v1, D, D);
        lin(p + "attn1.to_out", b.o1, D, D);
        vec(p + "attn1.q_norm.weight", b.qn1, D, 0, 1);
        vec(p + "attn1.k_norm.weight", b.kn1, D, 0, 1);
        lin(p + "attn2.to_q", b.q2, D, D);
        lin(p + "attn2.to_k", b.k2, D, cfg.cross_attention_dim);
        lin(p + "attn2.to_v", b.v2, D, cfg.cross_attention_dim);
        lin(p + "attn2.to_out", b.o2, D, D);
        vec(p + "attn2.q_norm.weight", b.qn2, D, 0, 1);
        vec(p + "attn2.k_norm.weight", b.kn2, D, 0, 1);
        lin(p + "ff.project_in.proj", b.ff1, 4 * D, D);
        lin(p + "ff.project_out", b.ff2, D, 4 * D);
    }
    size_t total = 0, total_w = 0;
    for (auto& q : pend) (q.kind == SLOT_BF16 ? total_w : total) += DeviceArena::padded(slot_bytes(q.kind, q.rows * (q.cols ? q.cols : 1)));
    m->weight_bytes = total + total_w;
    m->arena.reserve(total + 256);
    m->warena.reserve(total_w + 256);   // the Linear weights on their own: dit_quantize releases exactly these
    HIP_CHECK(hipMemset(m->arena.buf.p, 0, m->arena.buf.bytes));
    HIP_CHECK(hipMemset(m->warena.buf.p, 0, m->warena.buf.bytes));
    for (auto& q : pend) {
        const long numel = q.rows * (q.cols ? q.cols : 1);
        *q.dst = (q.kind == SLOT_BF16 ? m->warena : m->arena).take(slot_bytes(q.kind, numel));
        ParamSlot s;
        s.dst = *q.dst;
        s.kind = q.kind;
        s.numel = numel;
        s.rows = q.rows;
        s.cols = q.cols;
        s.init = q.init;
        m->slots[q.key] = s;
    }
    // module-key views into the fused tensors (these are the keys files actually carry)
    for (int i = 0; i < m->L; ++i) {
        DiTBlock& b = m->blocks[i];
        const std::string p = "transformer_blocks." + std::to_string(i) + ".";
        auto view = [&](const std::string& key, void* dst, int kind, long rows, long cols, int init) {
            ParamSlot s;
            s.dst = dst;
            s.kind = kind;
            s.rows = rows;
            s.cols = cols;
            s.numel = rows * (cols ? cols : 1);
            s.init = init;
            m->slots[key] = s;
        };
        view(p + "attn1.to_q.weight", b.qk1.w, SLOT_BF16, D, D, 0);
        view(p + "attn1.to_k.weight", b.qk1.w + (long)D * D, SLOT_BF16, D, D, 0);
        view(p + "attn1.to_q.bias", b.qk1.b, SLOT_F32, D, 0, 0);
        view(p + "attn1.to_k.bias", b.qk1.b + D, SLOT_F32, D, 0, 0);
        view(p + "scale_shift_table", m->sst_blocks + (long)i * 6 * D, SLOT_F32, 6, D, 0);
        m->slots.erase(p + "attn1.__qk.weight");
        m->slots.erase(p + "attn1.__qk.bias");
    }
    m->slots.erase("__sst_blocks");
    // reference initialisers for parameters a file may omit: RMSNorm weights = 1 (LTXAttention.swift:18)
    for (auto& kv : m->slots)
        if (kv.second.init == 1) launch_fill_const_f32((float*)kv.second.dst, kv.second.numel, 1.0f, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return m;
}

void dit_destroy(DiTModel* m) { delete m; }

void dit_load_safetensors(ltx_ctx* ctx, DiTModel* m, const std::string& path) {
    LTX_REQUIRE(m->quant_bits == 16, "weights cannot be loaded into a quantised model");
    SafeTensors st;
    st.open(path);
    ctx->n_loaded = ctx->n_missing = ctx->n_unmatched = 0;
    for (auto& kv : m->slots) kv.second.loaded = false;
    std::vector<uint8_t> staging;
    for (auto& kv : st.tensors) {
        std::string mk;
        if (!map_transformer_file_key(kv.first, &mk)) continue;
        auto it = m->slots.find(mk);
        if (it == m->slots.end()) {
            ctx->n_unmatched++;  // silently dropped by the reference (ModelDownloader.swift:992-1003)
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
            // every f32 parameter is cast to bf16 when applied (ModelDownloader.swift:1005-1012); keep the
            // bf16-rounded value in an f32 container
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
    for (auto* c : m->ctx_cache) c->version = 0;
}

void dit_init_synthetic(ltx_ctx* ctx, DiTModel* m, uint64_t seed) {
    // Linear weights N(0,0.02^2), biases N(0,0.01^2), q/k norm weights 1+N(0,0.02^2),
    // scale-shift tables N(0,0.02^2); everything rounded to bf16 once.
    uint64_t k = 0;
    for (auto& kv : m->slots) {
        ParamSlot& s = kv.second;
        const uint64_t sd = seed * 0x9E3779B97F4A7C15ull + (++k) * 0xD1B54A32D192ED03ull;
        const std::string& key = kv.first;
        const bool is_bias = key.size() > 5 && key.compare(key.size() - 5, 5, ".bias") == 0;
        const bool is_norm = key.find("_norm.weight") != std::string::npos;
        if (s.kind == SLOT_BF16) {
            launch_fill_normal_bf16((bf16_t*)s.dst, s.numel, sd, 0.f, 0.02f, ctx->stream);
        } else if (is_norm) {
            launch_fill_normal_f32((float*)s.dst, s.numel, sd, 1.0f, 0.02f, 1, ctx->stream);
        } else {
            launch_fill_normal_f32((float*)s.dst, s.numel, sd, 0.f, is_bias ? 0.01f : 0.02f, 1, ctx->stream);
        }
        s.loaded = true;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto* c : m->ctx_cache) c->version = 0;
}

void dit_export_slot(ltx_ctx* ctx, DiTModel* m, const ParamSlot& s, float* out) {
    if (s.q) {  // quantised Linear weight: the value the GEMMs use, bf16(q * scale + bias)
        DevBuf tmp;
        tmp.ensure((size_t)s.numel * 2);
        launch_dequant(s.q, s.qs, s.qb, s.rows, s.cols, m->quant_bits, tmp.as<bf16_t>(), ctx->stream);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::vector<bf16_t> h((size_t)s.numel);
        HIP_CHECK(hipMemcpy(h.data(), tmp.p, (size_t)s.numel * 2, hipMemcpyDeviceToHost));
        for (long i = 0; i < s.numel; ++i) out[i] = host_bf16_to_f32(h[i]);
        return;
    }
    if (s.kind == SLOT_F32) {
        HIP_CHECK(hipMemcpy(out, s.dst, (size_t)s.numel * 4, hipMemcpyDeviceToHost));
        return;
    }
    std::vector<bf16_t> h((size_t)s.numel);
    HIP_CHECK(hipMemcpy(h.data(), s.dst, (size_t)s.numel * 2, hipMemcpyDeviceToHost));
    for (long i = 0; i < s.numel; ++i) out[i] = host_bf16_to_f32(h[i]);
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
namespace {

void ensure_workspace(DiTModel* m, int B, int T, hipStream_t st) {
    const int D = m->D;
    const long rows = (long)B * T;
    const int Tpad = ((T + 63) / 64) * 64;
    if (rows <= m->ws_rows && B <= m->ws_B && Tpad == m->ws_Tpad) return;
    m->ws_x.ensure(rows * D * 4);
    m->ws_xn.ensure(rows * D * 2);
    m->ws_xb.ensure(rows * D * 2);
    m->ws_qk.ensure(rows * 2 * D * 4);
    m->ws_q.ensure(rows * D * 2);
    m->ws_k.ensure(rows * D * 2);
    m->ws_qc.ensure(rows * D * 4);
    if (m->ws_vt.ensure((size_t)B * D * Tpad * 2) || Tpad != m->ws_Tpad) HIP_CHECK(hipMemsetAsync(m->ws_vt.p, 0, m->ws_vt.bytes, st));
    m->ws_ao.ensure(rows * D * 2);
    m->ws_ffh.ensure(rows * 4 * D * 2);
    m->ws_ts.ensure(8 * 4);
    m->ws_emb256.ensure(8 * 256 * 4);
    m->ws_h1.ensure((size_t)8 * D * 4);
    m->ws_embts.ensure((size_t)8 * D * 4);
    m->ws_ada.ensure((size_t)8 * 6 * D * 4);
    m->ws_mod.ensure((size_t)8 * m->L * 6 * D * 4);
    m->ws_modout.ensure((size_t)8 * 2 * D * 4);
    // 8 M floats: split-K partials of the few-tile GEMMs at small token counts; two slices of [rows][D] where a one-round launch may run
    // as two K halves of the 192x256 kernel
    m->ws_splitk.ensure(std::max((size_t)8 << 22, rows <= 1536 ? (size_t)2 * rows * D * 4 : (size_t)0));
    m->ws_rows = (int)rows;
    m->ws_B = B;
    m->ws_Tpad = Tpad;
}

void ensure_rope(ltx_ctx* ctx, DiTModel* m, int F, int H, int W) {
    if (m->rope_F == F && m->rope_H == H && m->rope_W == W && m->rope_cos.p) return;
    std::vector<float> c, s;
    rope_tables(m->cfg, F, H, W, 24.0f, &c, &s);
    m->rope_cos.ensure(c.size() * 4);
    m->rope_sin.ensure(s.size() * 4);
    HIP_CHECK(hipMemcpyAsync(m->rope_cos.p, c.data(), c.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(m->rope_sin.p, s.data(), s.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    m->rope_F = F;
    m->rope_H = H;
    m->rope_W = W;
}

// caption projection + per-layer cross-attention K / V^T (LTXTransformer.swift:127-134; LTXAttention.swift:173-180)
DiTModel::CtxCache* prepare_context(ltx_ctx* ctx, DiTModel* m, const DiTForwardArgs& a) {
    const int D = m->D, L = m->L, B = a.B, S = a.S;
    const int Spad = ((S + 63) / 64) * 64;
    const long rows = (long)B * S;
    hipStream_t st = ctx->stream;
    m->ws_splitk.ensure((size_t)8 << 22);
    const SplitWs sk{m->ws_splitk.as<float>(), (long)(m->ws_splitk.bytes / 4)};
    m->ctx_clock++;
    DiTModel::CtxCache* c = nullptr;
    if (a.ctx_version != 0)
        for (auto* e : m->ctx_cache)
            if (e->version == a.ctx_version && e->kind == a.ctx_kind && e->B == B && e->S == S) {
                e->last_use = m->ctx_clock;
                return e;
            }
    if (m->ctx_cache.size() < 3) {
        c = new DiTModel::CtxCache();
        m->ctx_cache.push_back(c);
    } else {
        c = m->ctx_cache[0];
        for (auto* e : m->ctx_cache)
            if (e->last_use < c->last_use) c = e;
    }
    c->last_use = m->ctx_clock;
    m->ctx_tmp_h.ensure(rows * D * 2);
    m->ctx_tmp_kraw.ensure(rows * D * 4);
    c->proj.ensure(rows * D * 2);
    c->k.ensure((size_t)L * rows * D * 2);
    if (c->vt.ensure((size_t)L * B * D * Spad * 2) || c->Spad != Spad || c->B != B) HIP_CHECK(hipMemsetAsync(c->vt.p, 0, c->vt.bytes, st));
    c->bias.ensure(rows * 4);
    {
        GemmEpilogue e1;
        e1.out_bf16 = m->ctx_tmp_h.as<bf16_t>();
        e1.ld_bf16 = D;
        e1.act = LTX_ACT_GELU_TANH;
        gemm_linear(a.context, m->cfg.caption_channels, m->cap_l1, (int)rows, e1, st, sk);
        GemmEpilogue e2;
        e2.out_bf16 = c->proj.as<bf16_t>();
        e2.ld_bf16 = D;
        gemm_linear(m->ctx_tmp_h.as<bf16_t>(), D, m->cap_l2, (int)rows, e2, st, sk);
    }
    for (int l = 0; l < L; ++l) {
        const DiTBlock& blk = m->blocks[l];
        GemmEpilogue ek;
        ek.out_f32 = m->ctx_tmp_kraw.as<float>();
        ek.ld_f32 = D;
        ek.round_bf16 = 1;  // the reference's K projection runs bf16 x bf16 -> bf16 (SURVEY R5/R11 dtype notes)
        gemm_linear(c->proj.as<bf16_t>(), D, blk.k2, (int)rows, ek, st, sk);
        launch_qknorm_rope(m->ctx_tmp_kraw.as<float>(), D, blk.kn2, nullptr, nullptr, S,
                           c->k.as<bf16_t>() + (size_t)l * rows * D, D, (int)rows, D, m->cfg.norm_eps, st);
        for (int b = 0; b < B; ++b)
            gemm_vt(c->proj.as<bf16_t>() + (size_t)b * S * D, D, S, blk.v2,
                    c->vt.as<bf16_t>() + ((size_t)l * B + b) * D * Spad, Spad, st, sk);
    }
    c->has_bias = (a.mask != nullptr) && !a.mask_all_ones;
    if (c->has_bias) launch_mask_to_bias(a.mask, c->bias.as<float>(), rows, st);
    c->version = a.ctx_version;
    c->kind = a.ctx_kind;
    c->B = B;
    c->S = S;
    c->Spad = Spad;
    return c;
}

}  // namespace

// The q|k projection and the cross-attention q projection are stored as bf16 (round 4): half the bytes of the round trip to the norm + RoPE
// pass (34.82 -> 34.60 ms per step, rows 2.02 -> 1.77 ms). This is a rounding the REFERENCE DOES NOT HAVE: its q / k reach RMSNorm as f32
// (f32 activations x bf16-rounded weights; LTXAttention.swift:173-189) and the oracle keeps them f32. It was accepted on the
// measured cost - the 48-layer headline forward sits 2.419e-3 from the oracle against 2.416e-3 with the f32 store, the 8-step loops move in
// the fourth digit ("extra roundings") - and option "qk_f32" = 1 (ltx_ctx_set_option) restores the f32 store. The
// choice depends on nothing but that option, so sequence-parallel ranks and single-GPU runs round alike. The f32 workspaces (ws_qk, ws_qc)
// are reused at half their size.
static bool qk_store_bf16() {
    return ltx_opt(OPT_QK_F32) == 0;
}

void dit_forward(ltx_ctx* ctx, DiTModel* m, const DiTForwardArgs& a) {
    const int D = m->D, L = m->L, B = a.B;
    const int Tfull = a.F * a.H * a.W;
    const int NW = a.sp_world < 1 ? 1 : a.sp_world;
    LTX_REQUIRE(NW == 1 || ((a.sp_gather || (ctx->dist && ctx->dist->world == NW && ctx->dist->rank == a.sp_rank)) && B == 1 &&
                            a.sp_rank >= 0 && a.sp_rank < NW && Tfull % NW == 0 && (Tfull / NW) % 8 == 0),
                "dit_forward: sequence parallelism needs a transport (callback or ltx_dist_init with the same rank/world), batch 1 and "
                "F*H*W = %d divisible by %d ranks into multiples of 8", Tfull, NW);
    // all-gather of this forward: the caller's callback, else the context's transport. A failing transport stops the forward - the
    // kernels behind it would read unfilled K / V^T while the peer rank blocks in its collective.
    auto sp_allgather = [&](const void* send, void* recv, long bytes) {
        if (a.sp_gather) {
            const int rc = a.sp_gather(a.sp_user, send, recv, bytes);
            if (rc != 0) LTX_THROW(LTXS_GENERATION_FAILED, "sequence-parallel all-gather failed on rank %d of %d (status %d)", a.sp_rank, NW, rc);
        } else {
            dist_allgather(ctx, send, recv, bytes);
        }
    };
    // Self-test option ("sp_selftest" = 1 through ltx_ctx_set_option) sends a ONE-rank group of the native transport through the sequence-parallel branch (gathers of
    // one part, side stream, events): the only way to execute that code where a single GPU is all there is. Output bits == NW = 1 path.
    // (the option is consulted LAST, i.e. only on a context that holds a one-rank native group - a test artefact.)
    const bool sp_selftest = NW == 1 && B == 1 && !a.sp_gather && dist_can_overlap(ctx) && dist_world(ctx) == 1 && ltx_opt(OPT_SP_SELFTEST) != 0;
    const bool sp = NW > 1 || sp_selftest;
    const int T = Tfull / NW;            // rows this rank evaluates
    const int tok0 = a.sp_rank * T;      // first global token of this rank (NW == 1: 0)
    LTX_REQUIRE(B >= 1 && B <= 8 && T >= 1 && a.S >= 1, "dit_forward: bad shapes B=%d T=%d S=%d", B, T, a.S);
    const int G = a.n_groups < 1 ? 1 : a.n_groups;
    const int BG = B * G;  // rows of the timestep path / modulation tables
    LTX_REQUIRE(BG <= 8, "dit_forward: batch x timestep groups = %d exceeds 8", BG);
    LTX_REQUIRE(G == 1 || a.row_map, "dit_forward: timestep groups need a row map");
    const int32_t* rmap = G > 1 ? a.row_map : nullptr;
    LTX_REQUIRE(a.latent && a.context && a.timesteps && a.velocity, "dit_forward: null argument");
    hipStream_t st = ctx->stream;
    const long rows = (long)B * T;
    const int Tpad = ((T + 63) / 64) * 64;
    ensure_workspace(m, B, T, st);
    ensure_rope(ctx, m, a.F, a.H, a.W);
    DiTModel::CtxCache* cc = prepare_context(ctx, m, a);
    const int S = a.S, Spad = cc->Spad;

    const SplitWs sk{m->ws_splitk.as<float>(), (long)(m->ws_splitk.bytes / 4)};
    // FFN-down only: when its launch runs as K ranges of the 192x256 kernel (1536 tokens) the partial tiles may cross the workspace as bf16 -
    // an extra rounding the reference does not have (MLX accumulates the whole product in f32), accepted on its measured cost (headline
    // forward 2.419e-3 -> 2.457e-3 from the oracle for 0.7 % of the step); option "split_f32" = 1 turns it off
    SplitWs sk_ff2 = sk;
    sk_ff2.bf16_partials = true;
    // few tokens: the attention launcher may divide the keys of a launch over workgroups (attention.h, key split) - lend it the room
    {
        // (a sequence-parallel rank's self-attention sees all Tfull keys)
        const long need_x = attn_split_ws_bytes(B, m->cfg.num_heads, T, S), need_s = attn_split_ws_bytes(B, m->cfg.num_heads, T, Tfull);
        const long need = need_x > need_s ? need_x : need_s;
        if (need > 0) m->ws_attn_split.ensure((size_t)need);
    }
    float* x = m->ws_x.as<float>();
    bf16_t* xn = m->ws_xn.as<bf16_t>();
    bf16_t* xb = m->ws_xb.as<bf16_t>();
    float* qk = m->ws_qk.as<float>();
    bf16_t* q = m->ws_q.as<bf16_t>();
    bf16_t* k = m->ws_k.as<bf16_t>();
    bf16_t* vt = m->ws_vt.as<bf16_t>();
    bf16_t* ao = m->ws_ao.as<bf16_t>();
    bf16_t* ffh = m->ws_ffh.as<bf16_t>();
    float* qc = m->ws_qc.as<float>();
    float* mod = m->ws_mod.as<float>();
    const float eps = m->cfg.norm_eps;
    // RoPE rows of this rank's tokens (the table is [F*H*W][D/2] in global token order)
    const float* rope_c = m->rope_cos.as<float>() + (long)tok0 * (D >> 1);
    const float* rope_s = m->rope_sin.as<float>() + (long)tok0 * (D >> 1);
    const int TfullPad = ((Tfull + 63) / 64) * 64;
    bf16_t* k_full = k;
    bf16_t* vt_full = vt;
    if (sp) {
        m->ws_sp_k.ensure((size_t)Tfull * D * 2);
        if (m->ws_sp_vt.ensure((size_t)D * TfullPad * 2)) HIP_CHECK(hipMemsetAsync(m->ws_sp_vt.p, 0, m->ws_sp_vt.bytes, st));
        m->ws_sp_vtg.ensure((size_t)NW * D * T * 2);
        k_full = m->ws_sp_k.as<bf16_t>();
        vt_full = m->ws_sp_vt.as<bf16_t>();
    }

    // 1. patchify_proj: bf16 x bf16 -> bf16 in the reference; the residual stream starts as that bf16 value
    {
        GemmEpilogue e;
        e.out_f32 = x;
        e.ld_f32 = D;
        e.round_bf16 = 1;
        e.out_bf16 = xb;  // bf16 mirror (used by cross-attention when block 0 skips self-attention)
        e.ld_bf16 = D;
        gemm_linear(a.latent, m->cfg.in_channels, m->patchify, (int)rows, e, st, sk);
    }
    // 2. timestep path in f32 activations x bf16 weights (LTXTimestepEmbedding.swift:62-124)
    launch_timestep_embedding(a.timesteps, m->cfg.timestep_scale_multiplier, m->ws_emb256.as<float>(), BG, 256, st);
    launch_gemv_f32(m->ws_emb256.as<float>(), 256, dit_linear_weights(m->ada_l1, st), 256, m->ada_l1.b, m->ws_h1.as<float>(), D, BG, D, 256, LTX_ACT_NONE, st);
    launch_gemv_f32(m->ws_h1.as<float>(), D, dit_linear_weights(m->ada_l2, st), D, m->ada_l2.b, m->ws_embts.as<float>(), D, BG, D, D, LTX_ACT_SILU, st);
    launch_gemv_f32(m->ws_embts.as<float>(), D, dit_linear_weights(m->ada_lin, st), D, m->ada_lin.b, m->ws_ada.as<float>(), 6 * D, BG, 6 * D, D, LTX_ACT_SILU, st);
    launch_make_mod(m->sst_blocks, m->ws_ada.as<float>(), mod, BG, L, 6, D, st);
    // output modulation: shift = SST_out[0] + emb_ts, scale = SST_out[1] + emb_ts (LTXTransformer.swift:208-224)
    for (int j = 0; j < 2; ++j) {
        // mod_out[b][j][:] = sst_out[j] + emb_ts[b]
        for (int b = 0; b < BG; ++b)
            launch_lincomb(m->sst_out + (long)j * D, m->ws_embts.as<float>() + (long)b * D, 1.f, 1.f,
                           m->ws_modout.as<float>() + ((long)b * 2 + j) * D, D, st);
    }
    const long mod_bs = (long)L * 6 * D;  // batch stride of mod

    bool xn_ready = false;  // the previous block's last GEMM launch already wrote this block's first adaLN rows (NormAfter)
    for (int l = 0; l < L; ++l) {
        const DiTBlock& blk = m->blocks[l];
        const float* ml = mod + (long)l * 6 * D;  // rows: 0 shift_msa 1 scale_msa 2 gate_msa 3 shift_mlp 4 scale_mlp 5 gate_mlp
        if (!blk.skip_attn) {
            if (!xn_ready) launch_norm_mod(x, D, ml + 1 * D, ml + 0 * D, mod_bs, T, xn, D, (int)rows, D, LTX_NORM_RMS, eps, l == 0 ? 1 : 0, st, rmap);
            xn_ready = false;
            // Sequence parallelism on the native transport: the V^T gather (and its interleave) runs on the side stream under
            // the q|k projection and its norm + RoPE pass; only the K gather stays on the critical path. Same kernels, same
            // operands, same order of every reduction: the bits do not depend on which stream carried a collective.
            // Opt-in (option "sp_overlap" = 1) until a run on two or more GPUs has confirmed it: the side-stream gather shares ONE RCCL
            // communicator with the K gather on the context's stream, and no box reachable so far could execute that with two ranks
            // (round-3 advice). The self-test hook always takes the branch - that is what it is for.
            const bool sp_overlap_on = ltx_opt(OPT_SP_OVERLAP) != 0;
            const bool overlap = sp && !a.sp_gather && dist_can_overlap(ctx) && (sp_overlap_on || sp_selftest);
            if (sp) {
                gemm_vt(xn, D, T, blk.v1, vt, T, st, sk);  // V^T of the local tokens, dense [D][T]
                if (overlap) {
                    hipStream_t side = dist_side_stream(ctx);
                    dist_fork(ctx, 0);
                    dist_allgather_on(ctx, vt, m->ws_sp_vtg.p, (long)D * T * 2, side);
                    launch_sp_vt_interleave(m->ws_sp_vtg.as<bf16_t>(), vt_full, NW, D, T, TfullPad, side);
                }
            }
            const bool qkb = qk_store_bf16();
            GemmEpilogue eqk;
            if (qkb) {
                eqk.out_bf16 = (bf16_t*)qk;
                eqk.ld_bf16 = 2 * D;
            } else {
                eqk.out_f32 = qk;
                eqk.ld_f32 = 2 * D;
            }
            gemm_linear(xn, D, blk.qk1, (int)rows, eqk, st, sk);
            if (qkb) launch_qknorm_rope2(qk, blk.qn1, q, (const float*)((const bf16_t*)qk + D), blk.kn1, k, 2 * D, D, rope_c, rope_s, T, (int)rows, D, eps, st, kAttnQueryPrescale, true);
            else
            launch_qknorm_rope2(qk, blk.qn1, q, qk + D, blk.kn1, k, 2 * D, D, rope_c, rope_s, T, (int)rows, D, eps, st, kAttnQueryPrescale);
            AttnArgs at;
            if (!sp) {
                for (int b = 0; b < B; ++b) gemm_vt(xn + (size_t)b * T * D, D, T, blk.v1, vt + (size_t)b * D * Tpad, Tpad, st, sk);
                at.Vt = vt; at.ldvt = Tpad; at.vt_bstride = (long)D * Tpad;
                at.K = k; at.k_bstride = (long)T * D;
                at.Tk = T;
            } else {
                // keys / values of every rank's tokens: K rows gather straight into global token order; V^T blocks [D][T] of
                // the ranks are interleaved into [D][Tfull] after the gather
                sp_allgather(k, k_full, (long)T * D * 2);
                if (overlap) {
                    dist_join(ctx, 1);
                } else {
                    sp_allgather(vt, m->ws_sp_vtg.p, (long)D * T * 2);
                    launch_sp_vt_interleave(m->ws_sp_vtg.as<bf16_t>(), vt_full, NW, D, T, TfullPad, st);
                }
                at.Vt = vt_full; at.ldvt = TfullPad; at.vt_bstride = (long)D * TfullPad;
                at.K = k_full; at.k_bstride = (long)Tfull * D;
                at.Tk = Tfull;
            }
            at.Q = q; at.ldq = D; at.q_bstride = (long)T * D;
            at.ldk = D;
            at.O = ao; at.ldo = D; at.o_bstride = (long)T * D;
            at.B = B; at.H = m->cfg.num_heads; at.Tq = T;
            at.q_prescaled = 1;
            at.split_ws = m->ws_attn_split.p; at.split_ws_bytes = (long)m->ws_attn_split.bytes;
            launch_attention(at, st);
            GemmEpilogue eo;
            eo.out_f32 = x;
            eo.ld_f32 = D;
            eo.resid = 1;
            eo.gate = ml + 2 * D;
            eo.gate_bstride = mod_bs;
            eo.rows_per_batch = T;
            eo.gate_rowmap = rmap;
            eo.out_bf16 = xb;
            eo.ld_bf16 = D;
            gemm_linear(ao, D, blk.o1, (int)rows, eo, st, sk);
        } else if (l > 0) {
            launch_cast_f32_bf16(x, xb, rows * D, st);
        }
        // cross-attention on the un-normalised stream (LTXTransformerBlock.swift:205-214)
        {
            GemmEpilogue eq;
            eq.out_f32 = qc;
            eq.ld_f32 = D;
            const bool q2b = qk_store_bf16();
            if (q2b) {
                eq.out_f32 = nullptr;
                eq.out_bf16 = (bf16_t*)qc;
                eq.ld_bf16 = D;
            }
            gemm_linear(xb, D, blk.q2, (int)rows, eq, st, sk);
            launch_qknorm_rope(qc, D, blk.qn2, nullptr, nullptr, T, q, D, (int)rows, D, eps, st, kAttnQueryPrescale, q2b);
            AttnArgs at;
            at.Q = q; at.ldq = D; at.q_bstride = (long)T * D;
            at.K = cc->k.as<bf16_t>() + (size_t)l * B * S * D; at.ldk = D; at.k_bstride = (long)S * D;
            at.Vt = cc->vt.as<bf16_t>() + (size_t)l * B * D * Spad; at.ldvt = Spad; at.vt_bstride = (long)D * Spad;
            at.O = ao; at.ldo = D; at.o_bstride = (long)T * D;
            at.bias = cc->has_bias ? cc->bias.as<float>() : nullptr;
            at.bias_bstride = S;
            at.B = B; at.H = m->cfg.num_heads; at.Tq = T; at.Tk = S;
            at.q_prescaled = 1;
            at.split_ws = m->ws_attn_split.p; at.split_ws_bytes = (long)m->ws_attn_split.bytes;
            launch_attention(at, st);
            GemmEpilogue eo;
            eo.out_f32 = x;
            eo.ld_f32 = D;
            eo.resid = 1;
            eo.gate = nullptr;
            eo.gate_scalar = blk.cross_scale;
            gemm_linear(ao, D, blk.o2, (int)rows, eo, st, sk);
        }
        if (!blk.skip_ff) {
            launch_norm_mod(x, D, ml + 4 * D, ml + 3 * D, mod_bs, T, xn, D, (int)rows, D, LTX_NORM_RMS, eps, 0, st, rmap);
            GemmEpilogue e1;
            e1.out_bf16 = ffh;
            e1.ld_bf16 = 4 * D;
            e1.act = LTX_ACT_GELU_TANH;
            gemm_linear(xn, D, blk.ff1, (int)rows, e1, st, sk);
            GemmEpilogue e2;
            e2.out_f32 = x;
            e2.ld_f32 = D;
            e2.resid = 1;
            e2.gate = ml + 5 * D;
            e2.gate_bstride = mod_bs;
            e2.rows_per_batch = T;
            e2.gate_rowmap = rmap;
            // no bf16 mirror here: the next reader of xb is a cross-attention q projection, and by then either this block's successor has
            // rewritten it (attention-out epilogue) or, when that block skips self-attention, the cast above has
            // the next block opens with the adaLN pass over the stream this launch finishes: it rides on the launch (gemm.h NormAfter)
            if (l + 1 < L && !m->blocks[l + 1].skip_attn) {
                const float* mn = mod + (long)(l + 1) * 6 * D;
                NormAfter na;
                na.scale = mn + 1 * D;
                na.shift = mn + 0 * D;
                na.mod_bstride = mod_bs;
                na.rows_per_batch = T;
                na.out = xn;
                na.ldo = D;
                na.eps = eps;
                na.norm_kind = LTX_NORM_RMS;
                na.row_map = rmap;
                gemm_linear(ffh, 4 * D, blk.ff2, (int)rows, e2, st, sk_ff2, &na);
                xn_ready = true;
            } else {
                gemm_linear(ffh, 4 * D, blk.ff2, (int)rows, e2, st, sk_ff2);
            }
        }
    }
    // 6. output head: LayerNorm (no affine) * (1+scale) + shift -> proj_out (LTXTransformer.swift:208-224)
    {
        const float* mo = m->ws_modout.as<float>();
        launch_norm_mod(x, D, mo + D, mo, 2L * D, T, xn, D, (int)rows, D, LTX_NORM_LAYER, eps, 0, st, rmap);
        GemmEpilogue e;
        e.out_f32 = a.velocity;
        e.ld_f32 = m->cfg.out_channels;
        gemm_linear(xn, D, m->proj_out, (int)rows, e, st, sk);
    }
}
