// upscaler.cpp - see upscaler.h.
#include "upscaler.h"

#include <string.h>

#include "elementwise.h"
#include "gemm.h"

namespace {
struct Pending {
    std::string key;
    void** dst;
    int kind;
    long numel;
    int cout, cin, taps;
    bool perm4;
};
}  // namespace

UpscalerModel* upscaler_create(int mid) {
    LTX_REQUIRE(mid % 64 == 0 && mid % 32 == 0, "upscaler: mid_channels=%d must be a multiple of 64", mid);
    UpscalerModel* m = new UpscalerModel();
    m->mid = mid;
    std::vector<Pending> pend;
    auto conv = [&](const std::string& name, ConvW& c, int cin, int cout, int taps, bool perm4) {
        c.cin = cin;
        c.cout = cout;
        pend.push_back({name + ".weight", (void**)&c.w, 0, (long)cout * cin * taps, cout, cin, taps, perm4});
        pend.push_back({name + ".bias", (void**)&c.b, 1, cout, cout, 0, 0, perm4});
    };
    auto norm = [&](const std::string& name, UpNorm& n) {
        pend.push_back({name + ".weight", (void**)&n.w, 1, mid, 0, 0, 0, false});
        pend.push_back({name + ".bias", (void**)&n.b, 1, mid, 0, 0, 0, false});
    };
    conv("initial_conv", m->initial_conv, m->in_channels, mid, 27, false);
    norm("initial_norm", m->initial_norm);
    for (int i = 0; i < 4; ++i) {
        const std::string p = "res_blocks." + std::to_string(i) + ".";
        conv(p + "conv1", m->pre[i].conv1, mid, mid, 27, false);
        norm(p + "norm1", m->pre[i].norm1);
        conv(p + "conv2", m->pre[i].conv2, mid, mid, 27, false);
        norm(p + "norm2", m->pre[i].norm2);
        const std::string q = "post_upsample_res_blocks." + std::to_string(i) + ".";
        conv(q + "conv1", m->post[i].conv1, mid, mid, 27, false);
        norm(q + "norm1", m->post[i].norm1);
        conv(q + "conv2", m->post[i].conv2, mid, mid, 27, false);
        norm(q + "norm2", m->post[i].norm2);
    }
    conv("upsampler.conv", m->up_conv, mid, 4 * mid, 9, true);
    conv("final_conv", m->final_conv, mid, m->in_channels, 27, false);
    size_t total = 0;
    for (auto& p : pend) total += DeviceArena::padded((size_t)p.numel * (p.kind == 0 ? 2 : 4));
    m->arena.reserve(total + 256);
    HIP_CHECK(hipMemset(m->arena.buf.p, 0, m->arena.buf.bytes));
    for (auto& p : pend) {
        *p.dst = m->arena.take((size_t)p.numel * (p.kind == 0 ? 2 : 4));
        UpscalerModel::Slot s;
        s.dst = *p.dst;
        s.kind = p.kind;
        s.numel = p.numel;
        s.cout = p.cout;
        s.cin = p.cin;
        s.taps = p.taps;
        s.perm4 = p.perm4;
        m->slots[p.key] = s;
    }
    // GroupNorm affine defaults: weight 1, bias 0 (SpatialUpscaler.swift:25-26)
    for (auto& kv : m->slots)
        if (kv.second.kind == 1 && kv.first.find("norm") != std::string::npos && kv.first.size() > 7 &&
            kv.first.compare(kv.first.size() - 7, 7, ".weight") == 0)
            launch_fill_const_f32((float*)kv.second.dst, kv.second.numel, 1.0f, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return m;
}

void upscaler_destroy(UpscalerModel* m) { delete m; }

UpscalerModel* upscaler_load(ltx_ctx* ctx, const std::string& path) {
    SafeTensors st;
    st.open(path);
    // "Detect mid_channels from weight shape" (SpatialUpscaler.swift:276-281)
    int mid = 1024;
    auto it = st.tensors.find("res_blocks.0.conv1.weight");
    if (it != st.tensors.end() && !it->second.shape.empty()) mid = (int)it->second.shape[0];
    UpscalerModel* m = upscaler_create(mid);
    try {
        ctx->n_loaded = ctx->n_missing = ctx->n_unmatched = 0;
        std::vector<uint8_t> tmp, staging;
        for (auto& kv : st.tensors) {
            if (kv.first.find("blur_down") != std::string::npos) continue;  // fixed constant (SpatialUpscaler.swift:300-303)
            auto sit = m->slots.find(kv.first);
            if (sit == m->slots.end()) {
                ctx->n_unmatched++;
                continue;
            }
            UpscalerModel::Slot& s = sit->second;
            const StTensor& t = kv.second;
            if (t.numel() != s.numel)
                LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: %s has %ld elements, expected %ld", kv.first.c_str(), t.numel(), s.numel);
            if (s.kind == 0) {
                tmp.resize((size_t)s.numel * 2);
                staging.resize((size_t)s.numel * 2);
                st_to_bf16(st, t, (bf16_t*)tmp.data());
                const bf16_t* src = (const bf16_t*)tmp.data();
                bf16_t* dst = (bf16_t*)staging.data();
                const int O = s.cout, I = s.cin, T = s.taps, co = O / 4;
                for (int op = 0; op < O; ++op) {
                    const int o = s.perm4 ? ((op % co) * 4 + op / co) : op;
                    for (int tap = 0; tap < T; ++tap)
                        for (int i = 0; i < I; ++i) dst[((size_t)op * T + tap) * I + i] = src[((size_t)o * I + i) * T + tap];
                }
            } else {
                staging.resize((size_t)s.numel * 4);
                float* f = (float*)staging.data();
                st_to_f32(st, t, f);
                if (s.perm4) {
                    std::vector<float> q(f, f + s.numel);
                    const int O = (int)s.numel, co = O / 4;
                    for (int op = 0; op < O; ++op) f[op] = q[(op % co) * 4 + op / co];
                }
            }
            HIP_CHECK(hipMemcpy(s.dst, staging.data(), staging.size(), hipMemcpyHostToDevice));
            s.loaded = true;
            ctx->n_loaded++;
        }
        for (auto& kv : m->slots)
            if (!kv.second.loaded) ctx->n_missing++;
    } catch (...) {
        upscaler_destroy(m);
        throw;
    }
    return m;
}

namespace {
void conv_zero(const bf16_t* x, int F, int H, int W, const ConvW& cw, int kt, GemmEpilogue ep, hipStream_t st) {
    GemmArgs g;
    g.A = x;
    g.B = cw.w;
    g.ldb = 9L * kt * cw.cin;
    g.M = F * H * W;
    g.N = cw.cout;
    g.K = 9 * kt * cw.cin;
    g.conv = 1;
    g.geom.F = F;
    g.geom.H = H;
    g.geom.W = W;
    g.geom.C = cw.cin;
    g.geom.pad_mode = 1;
    g.geom.kt = kt;
    ep.bias_n = cw.b;
    g.ep = ep;
    launch_gemm_bf16(g, st);
}
}  // namespace

void upscaler_forward(ltx_ctx* ctx, UpscalerModel* m, const float* latent, int F, int H, int W, const float* mean,
                      const float* std_, float* out) {
    LTX_REQUIRE(latent && out && mean && std_ && F >= 1 && H >= 1 && W >= 1, "upscale: bad arguments");
    hipStream_t st = ctx->stream;
    const int C = m->mid, Cin = m->in_channels, G = 32;
    const long P1 = (long)F * H * W, P2 = P1 * 4;
    if (P2 > m->ws_P) {
        HIP_CHECK(hipStreamSynchronize(st));
        // bf16 conv inputs carry one extra all-zero position row at index P (zero padding by gather)
        m->h.ensure((size_t)P2 * C * 4);
        m->t.ensure((size_t)P2 * C * 4);
        m->hb.ensure((size_t)(P2 + 1) * C * 2);
        m->hb2.ensure((size_t)(P2 + 1) * C * 2);
        m->stats.ensure(G * 2 * 4);
        m->out_cl.ensure((size_t)P2 * Cin * 4);
        m->ws_P = P2;
    }
    float* h = m->h.as<float>();
    float* t = m->t.as<float>();
    bf16_t* hb = m->hb.as<bf16_t>();
    bf16_t* hb2 = m->hb2.as<bf16_t>();
    float* stats = m->stats.as<float>();
    auto zero_row = [&](bf16_t* buf, long P, int ch) { HIP_CHECK(hipMemsetAsync(buf + P * ch, 0, (size_t)ch * 2, st)); };

    // denormalise + channels-last bf16 (upsampleLatents: latent*std + mean, SpatialUpscaler.swift:366-367)
    launch_vae_prepare(latent, P1, nullptr, 0.f, mean, std_, hb, Cin, P1, st);
    zero_row(hb, P1, Cin);
    {
        GemmEpilogue e;
        e.out_f32 = t;
        e.ld_f32 = C;
        conv_zero(hb, F, H, W, m->initial_conv, 3, e, st);
    }
    launch_groupnorm_stats(t, P1, C, G, 1e-5f, stats, st);
    launch_groupnorm_apply(t, stats, m->initial_norm.w, m->initial_norm.b, nullptr, 1, h, hb, P1, C, G, st);

    auto res_block = [&](const UpResBlock& rb, int f, int hh, int ww) {
        const long P = (long)f * hh * ww;
        zero_row(hb, P, C);
        GemmEpilogue e1;
        e1.out_f32 = t;
        e1.ld_f32 = C;
        conv_zero(hb, f, hh, ww, rb.conv1, 3, e1, st);
        launch_groupnorm_stats(t, P, C, G, 1e-5f, stats, st);
        launch_groupnorm_apply(t, stats, rb.norm1.w, rb.norm1.b, nullptr, 1, nullptr, hb2, P, C, G, st);
        zero_row(hb2, P, C);
        GemmEpilogue e2;
        e2.out_f32 = t;
        e2.ld_f32 = C;
        conv_zero(hb2, f, hh, ww, rb.conv2, 3, e2, st);
        launch_groupnorm_stats(t, P, C, G, 1e-5f, stats, st);
        // SiLU(norm2(h2) + residual) (SpatialUpscaler.swift:103-105); new stream value in f32 + bf16
        launch_groupnorm_apply(t, stats, rb.norm2.w, rb.norm2.b, h, 1, h, hb, P, C, G, st);
    };
    for (int i = 0; i < 4; ++i) res_block(m->pre[i], F, H, W);
    // per-frame Conv2d mid -> 4*mid + PixelShuffle(2) fused into the store (SpatialUpscaler.swift:136-162)
    zero_row(hb, P1, C);
    {
        GemmEpilogue e;
        e.out_f32 = h;
        e.ld_f32 = C;
        e.out_bf16 = hb2;
        e.ld_bf16 = C;
        e.d2s = 2;
        conv_zero(hb, F, H, W, m->up_conv, 1, e, st);
    }
    // hb2 now holds the bf16 mirror of the up-sampled stream: make it the conv input buffer
    HIP_CHECK(hipMemcpyAsync(hb, hb2, (size_t)P2 * C * 2, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < 4; ++i) res_block(m->post[i], F, 2 * H, 2 * W);
    zero_row(hb, P2, C);
    {
        GemmEpilogue e;
        e.out_f32 = m->out_cl.as<float>();
        e.ld_f32 = Cin;
        conv_zero(hb, F, 2 * H, 2 * W, m->final_conv, 3, e, st);
    }
    // renormalise (x - mean)/std and return to [C][F][2H][2W] (SpatialUpscaler.swift:372-376)
    launch_upscaler_finish(m->out_cl.as<float>(), mean, std_, out, P2, Cin, st);
}
