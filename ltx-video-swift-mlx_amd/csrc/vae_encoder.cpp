// vae_encoder.cpp - see vae_encoder.h.
#include "vae_encoder.h"

#include <string.h>

#include "elementwise.h"
#include "gemm.h"
#include "hostmath.h"

namespace {
const int kResnets[4] = {4, 6, 6, 2};
const int kFactor[4][3] = {{1, 2, 2}, {2, 1, 1}, {2, 2, 2}, {2, 2, 2}};

struct Dims {
    int F, H, W;
    long P() const { return (long)F * H * W; }
};

void conv3d_enc(const bf16_t* x, const Dims& d, const ConvW& cw, GemmEpilogue ep, hipStream_t st) {
    GemmArgs g;
    g.A = x;
    g.B = cw.w;
    g.ldb = 27L * cw.cin;
    g.M = (int)d.P();
    g.N = cw.cout;
    g.K = 27 * cw.cin;
    g.conv = 1;
    g.geom.F = d.F;
    g.geom.H = d.H;
    g.geom.W = d.W;
    g.geom.C = cw.cin;
    g.geom.causal = 1;    // VideoEncoder(causal: true) (VideoEncoder.swift:221)
    g.geom.pad_mode = 3;  // spatialPaddingMode .zeros (:225), first frame replicated in T
    ep.bias_n = cw.b;
    g.ep = ep;
    launch_gemm_bf16(g, st);
}
}  // namespace

int vae_encoder_latent_frames(int T) {
    int t = T;
    for (int i = 0; i < 4; ++i) t = (t + kFactor[i][0] - 1) / kFactor[i][0];
    return t;
}

VaeEncoderModel* vae_encoder_create(int base) {
    LTX_REQUIRE(base >= 64 && base % 64 == 0 && base * 16 <= 2048, "vae encoder: channel base %d must be a multiple of 64, <= 128", base);
    VaeEncoderModel* m = new VaeEncoderModel();
    m->base = base;
    for (int i = 0; i < 5; ++i) m->ch[i] = base << i;
    struct Pending { std::string key; void** dst; int kind; long bytes; VaeEncoderModel::Slot s; };
    std::vector<Pending> pend;
    auto conv = [&](const std::string& name, ConvW& c, int cin, int cout, int cin_pad, int file_cout) {
        c.cin = cin_pad;
        c.cout = cout;
        VaeEncoderModel::Slot sw;
        sw.kind = 0; sw.cout = cout; sw.cin = cin; sw.cin_pad = cin_pad; sw.file_cout = file_cout;
        sw.file_numel = (long)file_cout * cin * 27;
        pend.push_back({name + ".conv.weight", (void**)&c.w, 0, (long)cout * 27 * cin_pad * 2, sw});
        VaeEncoderModel::Slot sb;
        sb.kind = 1; sb.cout = cout; sb.file_cout = file_cout; sb.file_numel = file_cout;
        pend.push_back({name + ".conv.bias", (void**)&c.b, 1, (long)cout * 4, sb});
    };
    conv("conv_in", m->conv_in, 48, m->ch[0], 64, m->ch[0]);
    conv("conv_out", m->conv_out, m->ch[4], 128, m->ch[4], 129);  // row 128 = logvar, dropped (:307)
    for (int i = 0; i < 4; ++i) {
        m->down[i].resize(kResnets[i]);
        for (int j = 0; j < kResnets[i]; ++j) {
            const std::string p = "down_blocks_" + std::to_string(i) + ".resnets.resnets." + std::to_string(j) + ".";
            conv(p + "conv1", m->down[i][j].c1, m->ch[i], m->ch[i], m->ch[i], m->ch[i]);
            conv(p + "conv2", m->down[i][j].c2, m->ch[i], m->ch[i], m->ch[i], m->ch[i]);
        }
        const int S = kFactor[i][0] * kFactor[i][1] * kFactor[i][2];
        conv("down_blocks_" + std::to_string(i) + ".downsamplers.conv", m->ds[i], m->ch[i], m->ch[i + 1] / S, m->ch[i], m->ch[i + 1] / S);
    }
    for (int j = 0; j < 2; ++j) {
        const std::string p = "mid_block.resnets." + std::to_string(j) + ".";
        conv(p + "conv1", m->mid[j].c1, m->ch[4], m->ch[4], m->ch[4], m->ch[4]);
        conv(p + "conv2", m->mid[j].c2, m->ch[4], m->ch[4], m->ch[4], m->ch[4]);
    }
    size_t total = 0;
    for (auto& q : pend) total += DeviceArena::padded(q.bytes);
    m->weight_bytes = total;
    m->arena.reserve(total + 256);
    HIP_CHECK(hipMemset(m->arena.buf.p, 0, m->arena.buf.bytes));
    for (auto& q : pend) {
        *q.dst = m->arena.take(q.bytes);
        q.s.dst = *q.dst;
        m->slots[q.key] = q.s;
    }
    return m;
}

void vae_encoder_destroy(VaeEncoderModel* m) { delete m; }

void vae_encoder_load_safetensors(ltx_ctx* ctx, VaeEncoderModel* m, const std::string& path) {
    SafeTensors st;
    st.open(path);
    ctx->n_loaded = ctx->n_missing = ctx->n_unmatched = 0;
    for (auto& kv : m->slots) kv.second.loaded = false;
    std::vector<uint8_t> tmp, staging;
    for (auto& kv : st.tensors) {
        std::string mk;
        if (!map_vae_encoder_file_key(kv.first, &mk)) continue;  // decoder.* / statistics of the same file
        auto it = m->slots.find(mk);
        if (it == m->slots.end()) {
            ctx->n_unmatched++;
            continue;
        }
        VaeEncoderModel::Slot& s = it->second;
        const StTensor& t = kv.second;
        if (t.numel() != s.file_numel)
            LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: %s has %ld elements, expected %ld", kv.first.c_str(), t.numel(), s.file_numel);
        if (s.kind == 0) {
            // (O,I,kT,kH,kW) -> [O][tap][Ipad] bf16, rows beyond cout (conv_out's logvar row) dropped, padded input channels zero
            tmp.resize((size_t)s.file_numel * 2);
            st_to_bf16(st, t, (bf16_t*)tmp.data());
            staging.assign((size_t)s.cout * 27 * s.cin_pad * 2, 0);
            const bf16_t* src = (const bf16_t*)tmp.data();
            bf16_t* dst = (bf16_t*)staging.data();
            for (int o = 0; o < s.cout; ++o)
                for (int tap = 0; tap < 27; ++tap)
                    for (int i = 0; i < s.cin; ++i) dst[((size_t)o * 27 + tap) * s.cin_pad + i] = src[((size_t)o * s.cin + i) * 27 + tap];
        } else {
            tmp.resize((size_t)s.file_numel * 4);
            st_to_f32(st, t, (float*)tmp.data());
            staging.assign(tmp.begin(), tmp.begin() + (size_t)s.cout * 4);
        }
        HIP_CHECK(hipMemcpy(s.dst, staging.data(), staging.size(), hipMemcpyHostToDevice));
        s.loaded = true;
        ctx->n_loaded++;
    }
    for (auto& kv : m->slots)
        if (!kv.second.loaded) ctx->n_missing++;
}

void vae_encoder_init_synthetic(ltx_ctx* ctx, VaeEncoderModel* m, uint64_t seed) {
    uint64_t k = 0;
    for (auto& kv : m->slots) {
        VaeEncoderModel::Slot& s = kv.second;
        const uint64_t sd = seed * 0x9E3779B97F4A7C15ull + (++k) * 0xD1B54A32D192ED03ull;
        if (s.kind == 0)  // padded input channels of conv_in multiply zero activations: their values do not matter
            launch_fill_normal_bf16((bf16_t*)s.dst, (long)s.cout * 27 * s.cin_pad, sd, 0.f, 1.0f / sqrtf(27.0f * s.cin), ctx->stream);
        else
            launch_fill_normal_f32((float*)s.dst, s.cout, sd, 0.f, 0.01f, 1, ctx->stream);
        s.loaded = true;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

void vae_encoder_encode(ltx_ctx* ctx, VaeEncoderModel* m, const float* pixels, int T, int H, int W, const float* mean,
                        const float* stdv, float* latent, int* Tp_out) {
    LTX_REQUIRE(pixels && latent && T >= 1, "vae_encode: null argument");
    // three spatial halvings after the 4x4 patchify: H, W multiples of 32 (LTXVideoGenerationConfig.validate already demands it)
    LTX_REQUIRE(H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, "vae_encode: height %d and width %d must be multiples of 32", H, W);
    hipStream_t st = ctx->stream;
    Dims d{T, H / 4, W / 4};
    // workspace: the widest stream is conv_in's output (P0 x base) - every later stage has at most the same element count
    const long elems = d.P() * (long)std::max(m->ch[0], 64) * 2;
    if (elems > m->ws_elems) {
        HIP_CHECK(hipStreamSynchronize(st));
        m->xa.ensure((size_t)elems * 4);
        m->xb.ensure((size_t)elems * 4);
        m->t1.ensure((size_t)elems * 4);
        m->hb.ensure((size_t)(elems + 4096) * 2);
        m->ws_elems = elems;
    }
    float* x = m->xa.as<float>();
    float* xo = m->xb.as<float>();
    float* t1 = m->t1.as<float>();
    bf16_t* hb = m->hb.as<bf16_t>();
    // pad mode 3 reads a zero row at index P of the conv input: keep one behind every bf16 activation tensor
    auto zero_row = [&](long P, int C) { HIP_CHECK(hipMemsetAsync(hb + P * C, 0, (size_t)C * 2, st)); };

    launch_enc_patchify(pixels, hb, T, H, W, st);
    zero_row(d.P(), 64);
    {
        GemmEpilogue e;
        e.out_f32 = x;
        e.ld_f32 = m->ch[0];
        conv3d_enc(hb, d, m->conv_in, e, st);
    }
    auto res_block = [&](const VaeEncoderModel::Res& rb, int C) {
        launch_pixelnorm_silu(x, nullptr, nullptr, hb, d.P(), C, st);
        zero_row(d.P(), C);
        GemmEpilogue e1;
        e1.out_f32 = t1;
        e1.ld_f32 = C;
        conv3d_enc(hb, d, rb.c1, e1, st);
        launch_pixelnorm_silu(t1, nullptr, nullptr, hb, d.P(), C, st);
        zero_row(d.P(), C);
        GemmEpilogue e2;  // x = conv2(h) + x, in place
        e2.out_f32 = x;
        e2.ld_f32 = C;
        e2.resid = 1;
        e2.gate_scalar = 1.0f;
        conv3d_enc(hb, d, rb.c2, e2, st);
    };
    for (int i = 0; i < 4; ++i) {
        const int C = m->ch[i];
        for (auto& rb : m->down[i]) res_block(rb, C);
        // downsampler: conv on the raw stream, space-to-depth, + group mean of space-to-depth(x) (VideoEncoder.swift:146-167)
        launch_cast_f32_bf16(x, hb, d.P() * C, st);
        zero_row(d.P(), C);
        GemmEpilogue e;
        e.out_f32 = t1;
        e.ld_f32 = m->ds[i].cout;
        conv3d_enc(hb, d, m->ds[i], e, st);
        const int* f = kFactor[i];
        launch_enc_s2d_residual(t1, m->ds[i].cout, x, C, xo, m->ch[i + 1], d.F, d.H, d.W, f[0], f[1], f[2], st);
        d.F = (d.F + f[0] - 1) / f[0];
        d.H /= f[1];
        d.W /= f[2];
        std::swap(x, xo);
    }
    for (int j = 0; j < 2; ++j) res_block(m->mid[j], m->ch[4]);
    launch_pixelnorm_silu(x, nullptr, nullptr, hb, d.P(), m->ch[4], st);
    zero_row(d.P(), m->ch[4]);
    {
        GemmEpilogue e;
        e.out_f32 = t1;
        e.ld_f32 = 128;
        conv3d_enc(hb, d, m->conv_out, e, st);
    }
    launch_enc_finish(t1, 128, mean, stdv, latent, 128, d.P(), st);
    if (Tp_out) *Tp_out = d.F;
}
