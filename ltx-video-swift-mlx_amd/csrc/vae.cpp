// vae.cpp - see vae.h. One decode = reference VideoDecoder.callAsFunction (VideoDecoder.swift:358-449) per temporal
// tile + decodeVideo / decodeWithTemporalTiling (VideoDecoder.swift:466-602).
#include "vae.h"

#include "dist.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "elementwise.h"
#include "gemm.h"
#include "hostmath.h"
#include "options.h"

namespace {

struct Pending {
    std::string key;
    void** dst;
    int kind;
    long numel;
    int cout, cin;
    int perm;
    int init;
};

}  // namespace

// stored output channel `op` of a conv <- channel of the file. perm 1: depth-to-space sub-major (op = sub*Cout/8 + c <- c*8 + sub);
// perm 2: conv_out in un-patchify order (op = b*12 + a*3 + c <- (c*4 + a)*4 + b, VideoDecoder.swift:257-275)
static int vae_file_channel(int perm, int op, int O) {
    if (perm == 1) {
        const int co = O / 8;
        return (op % co) * 8 + op / co;
    }
    if (perm == 2) {
        const int b = op / 12, a = (op % 12) / 3, c = op % 3;
        return (c * 4 + a) * 4 + b;
    }
    return op;
}

VaeModel* vae_create() {
    VaeModel* m = new VaeModel();
    std::vector<Pending> pend;
    auto conv = [&](const std::string& name, ConvW& c, int cin, int cout, int perm) {
        c.cin = cin;
        c.cout = cout;
        c.d2s_perm = perm == 1;
        pend.push_back({name + ".conv.weight", (void**)&c.w, 0, (long)cout * cin * 27, cout, cin, perm, 0});
        pend.push_back({name + ".conv.bias", (void**)&c.b, 1, cout, cout, 0, perm, 0});
    };
    auto te = [&](const std::string& name, VaeTimeEmbedder& t, int out) {
        t.hidden = 256;
        t.out = out;
        pend.push_back({name + ".timestep_embedder.linear_1.weight", (void**)&t.w1, 2, 256L * 256, 256, 256, 0, 0});
        pend.push_back({name + ".timestep_embedder.linear_1.bias", (void**)&t.b1, 1, 256, 256, 0, 0, 0});
        pend.push_back({name + ".timestep_embedder.linear_2.weight", (void**)&t.w2, 2, (long)out * 256, out, 256, 0, 0});
        pend.push_back({name + ".timestep_embedder.linear_2.bias", (void**)&t.b2, 1, out, out, 0, 0, 0});
    };
    conv("conv_in", m->conv_in, m->latent_channels, m->channels[0], 0);
    conv("conv_out", m->conv_out, m->channels[3], 48, 2);  // stored in the un-patchify order: its epilogue writes the frames
    for (int g = 0; g < 4; ++g) {
        const int C = m->channels[g];
        m->groups[g].C = C;
        const std::string gp = "up_blocks_" + std::to_string(2 * g) + ".";
        for (int r = 0; r < 5; ++r) {
            const std::string rp = gp + "res_blocks." + std::to_string(r) + ".";
            conv(rp + "conv1", m->groups[g].blocks[r].conv1, C, C, 0);
            conv(rp + "conv2", m->groups[g].blocks[r].conv2, C, C, 0);
            pend.push_back({rp + "scale_shift_table", (void**)&m->groups[g].blocks[r].sst, 1, 4L * C, 4 * C, 0, 0, 0});
        }
        te(gp + "time_embedder", m->groups[g].te, 4 * C);
        if (g < 3) conv("up_blocks_" + std::to_string(2 * g + 1) + ".conv", m->up[g], C, 4 * C, 1);
    }
    te("last_time_embedder", m->last_te, 2 * m->channels[3]);
    pend.push_back({"last_scale_shift_table", (void**)&m->last_sst, 1, 2L * m->channels[3], 0, 0, 0, 0});
    pend.push_back({"mean_of_means", (void**)&m->mean, 1, m->latent_channels, 0, 0, 0, 0});
    pend.push_back({"std_of_means", (void**)&m->std_, 1, m->latent_channels, 0, 0, 0, 1});
    pend.push_back({"timestep_scale_multiplier", (void**)&m->ts_mult, 3, 1, 0, 0, 0, 0});

    size_t total = 0;
    for (auto& p : pend) total += DeviceArena::padded((size_t)p.numel * ((p.kind == 0 || p.kind == 2) ? 2 : 4));
    m->weight_bytes = total;
    m->arena.reserve(total + 256);
    HIP_CHECK(hipMemset(m->arena.buf.p, 0, m->arena.buf.bytes));
    for (auto& p : pend) {
        *p.dst = m->arena.take((size_t)p.numel * ((p.kind == 0 || p.kind == 2) ? 2 : 4));
        VaeModel::Slot s;
        s.dst = *p.dst;
        s.kind = p.kind;
        s.numel = p.numel;
        s.cout = p.cout;
        s.cin = p.cin;
        s.perm = p.perm;
        s.init = p.init;
        m->slots[p.key] = s;
    }
    // reference initialisers (VideoDecoder.swift:324-327): std = 1, timestep_scale_multiplier = 1000
    launch_fill_const_f32(m->std_, m->latent_channels, 1.0f, nullptr);
    launch_fill_const_f32(m->ts_mult, 1, 1000.0f, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return m;
}

void vae_destroy(VaeModel* m) { delete m; }

static bool parse_timestep_conditioning(const std::string& weights_path, const std::string& config_json) {
    // parseVAEConfig (ModelDownloader.swift:583-594): config.json next to the weights file
    std::string path = config_json;
    if (path.empty()) {
        const size_t slash = weights_path.rfind('/');
        path = (slash == std::string::npos ? std::string(".") : weights_path.substr(0, slash)) + "/config.json";
    }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string txt;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, n);
    fclose(f);
    const size_t k = txt.find("\"timestep_conditioning\"");
    if (k == std::string::npos) return false;
    size_t p = txt.find(':', k);
    if (p == std::string::npos) return false;
    ++p;
    while (p < txt.size() && (txt[p] == ' ' || txt[p] == '\n' || txt[p] == '\t' || txt[p] == '\r')) ++p;
    return txt.compare(p, 4, "true") == 0;
}

void vae_load_safetensors(ltx_ctx* ctx, VaeModel* m, const std::string& path, const std::string& config_json) {
    SafeTensors st;
    st.open(path);
    ctx->n_loaded = ctx->n_missing = ctx->n_unmatched = 0;
    for (auto& kv : m->slots) kv.second.loaded = false;
    std::vector<uint8_t> tmp, staging;
    for (auto& kv : st.tensors) {
        std::string mk;
        if (!map_vae_file_key(kv.first, &mk)) continue;
        auto it = m->slots.find(mk);
        if (it == m->slots.end()) {
            ctx->n_unmatched++;  // dropped without error (ModelDownloader.swift:1040-1050)
            continue;
        }
        VaeModel::Slot& s = it->second;
        const StTensor& t = kv.second;
        if (t.numel() != s.numel)
            LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: %s has %ld elements, expected %ld", kv.first.c_str(), t.numel(), s.numel);
        if (s.kind == 0) {
            // (O,I,kT,kH,kW) -> [O'][tap][I] bf16, O' = sub*Cout/8... (d2s permutation for upsamplers)
            tmp.resize((size_t)s.numel * 2);
            staging.resize((size_t)s.numel * 2);
            st_to_bf16(st, t, (bf16_t*)tmp.data());
            const bf16_t* src = (const bf16_t*)tmp.data();
            bf16_t* dst = (bf16_t*)staging.data();
            const int O = s.cout, I = s.cin;
            const int co = O / 8;
            for (int op = 0; op < O; ++op) {
                const int o = vae_file_channel(s.perm, op, O);
                for (int tap = 0; tap < 27; ++tap)
                    for (int i = 0; i < I; ++i) dst[((size_t)op * 27 + tap) * I + i] = src[((size_t)o * I + i) * 27 + tap];
            }
        } else if (s.kind == 2) {
            staging.resize((size_t)s.numel * 2);
            st_to_bf16(st, t, (bf16_t*)staging.data());
        } else {
            staging.resize((size_t)s.numel * 4);
            float* f = (float*)staging.data();
            st_to_f32(st, t, f);
            if (s.perm) {
                std::vector<float> q(f, f + s.numel);
                const int O = (int)s.numel;
                for (int op = 0; op < O; ++op) f[op] = q[vae_file_channel(s.perm, op, O)];
            }
            if (s.kind == 3) m->ts_mult_host = f[0];
        }
        HIP_CHECK(hipMemcpy(s.dst, staging.data(), staging.size(), hipMemcpyHostToDevice));
        s.loaded = true;
        ctx->n_loaded++;
    }
    for (auto& kv : m->slots)
        if (!kv.second.loaded) ctx->n_missing++;
    m->timestep_conditioning = parse_timestep_conditioning(path, config_json);
}

void vae_init_synthetic(ltx_ctx* ctx, VaeModel* m, uint64_t seed, bool timestep_conditioning) {
    uint64_t k = 0;
    for (auto& kv : m->slots) {
        VaeModel::Slot& s = kv.second;
        const uint64_t sd = seed * 0x9E3779B97F4A7C15ull + (++k) * 0xD1B54A32D192ED03ull;
        const std::string& key = kv.first;
        if (s.kind == 0) {
            launch_fill_normal_bf16((bf16_t*)s.dst, s.numel, sd, 0.f, 1.0f / sqrtf(27.0f * s.cin), ctx->stream);
        } else if (s.kind == 2) {
            launch_fill_normal_bf16((bf16_t*)s.dst, s.numel, sd, 0.f, 1.0f / 16.0f, ctx->stream);
        } else if (key == "std_of_means") {
            launch_fill_const_f32((float*)s.dst, s.numel, 1.0f, ctx->stream);
        } else if (key == "mean_of_means") {
            launch_fill_const_f32((float*)s.dst, s.numel, 0.0f, ctx->stream);
        } else if (s.kind == 3) {
            launch_fill_const_f32((float*)s.dst, 1, 1000.0f, ctx->stream);
        } else {
            launch_fill_normal_f32((float*)s.dst, s.numel, sd, 0.f, 0.02f, 1, ctx->stream);
        }
        s.loaded = true;
    }
    m->ts_mult_host = 1000.0f;
    m->timestep_conditioning = timestep_conditioning;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

// ---------------------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Dims {
    int F, H, W;
    long P() const { return (long)F * H * W; }
};

// `skws`/`skws_elems`: split-K workspace (floats) - convs with too few output tiles to fill the chip (the 1024-channel
// stage: 1536 positions -> 64 tiles of 192x128) split their 27*Cin reduction over several workgroups per tile
void conv3d(const bf16_t* x, const Dims& d, const ConvW& cw, GemmEpilogue ep, hipStream_t st, float* skws = nullptr,
            long skws_elems = 0) {
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
    g.geom.causal = 0;    // the pipeline builds the decoder with causal:false (LTXPipeline.swift:338)
    g.geom.pad_mode = 0;  // reflect
    const bool blk_on = ltx_opt(OPT_CONV_BLOCK) != 0;  // A/B option "conv_block"
    if (blk_on && (d.H * d.W) % 192 == 0 && ((d.H * d.W) / 192 < 8 || (d.H * d.W) / 192 % 8 == 0)) g.geom.blk_rg = d.H * d.W / 192;
    ep.bias_n = cw.b;
    g.ep = ep;
    if (skws && !ep.d2s && cw.cout % 4 == 0) {
        int sk = gemm_suggest_split_k(g.M, g.N, g.K);
        while (sk > 1 && (long)sk * g.M * g.N > skws_elems) --sk;
        if (sk > 1) g.split_k = sk;
        g.split_ws = skws;  // also lets the launcher run a last partial round of tiles as a split-K launch
        g.split_ws_elems = skws_elems;
    }
    launch_gemm_bf16(g, st);
}

// One VAEResBlock3d (VideoDecoder.swift:75-131) on the f32 stream x [P][C], in place. `md` = the block's modulation rows (shift1,
// 1 + scale1, shift2, 1 + scale2). With C == 128 the convs' epilogues carry the PixelNorm passes: conv1 emits conv2's input, conv2
// emits the next consumer's (`next_scale` / `next_shift`: PixelNorm + modulation + SiLU; `copy`: a plain bf16 cast; neither: nothing)
// into the bf16 buffer it is not reading; hin / hout swap accordingly. `first`: x has no bf16 form yet (the group's first block).
void res_block(const VaeResBlock& rb, const float* md, int C, const Dims& d, float* x, float* t1, bf16_t*& hin, bf16_t*& hout, bool first,
               const float* next_scale, const float* next_shift, bool copy, hipStream_t st, float* skws, long skn) {
    const bool fuse = C == 128;
    if (!fuse || first) launch_pixelnorm_silu(x, md + 1 * C, md + 0 * C, hin, d.P(), C, st);
    GemmEpilogue e1;
    if (fuse) {
        e1.pn_out = hout;  // h = silu(pixelnorm(conv1(.)) * (1 + scale2) + shift2), the f32 value is not kept
        e1.ld_pn = C;
        e1.pn_scale = md + 3 * C;
        e1.pn_shift = md + 2 * C;
    } else {
        e1.out_f32 = t1;
        e1.ld_f32 = C;
    }
    conv3d(hin, d, rb.conv1, e1, st, skws, skn);
    if (fuse) {
        std::swap(hin, hout);
    } else {
        launch_pixelnorm_silu(t1, md + 3 * C, md + 2 * C, hin, d.P(), C, st);
    }
    GemmEpilogue e2;  // x = conv2(h) + x, in place
    e2.out_f32 = x;
    e2.ld_f32 = C;
    e2.resid = 1;
    e2.gate_scalar = 1.0f;
    if (fuse && next_scale) {
        e2.pn_out = hout;
        e2.ld_pn = C;
        e2.pn_scale = next_scale;
        e2.pn_shift = next_shift;
    } else if (copy) {
        // into the other buffer: neighbouring tiles are still reading this conv's input
        e2.out_bf16 = hout;
        e2.ld_bf16 = C;
    }
    conv3d(hin, d, rb.conv2, e2, st, skws, skn);
    if (fuse) std::swap(hin, hout);
}

// decode one temporal tile into `frames` (raw, layout (F_out, 32H, 32W, 3)); returns F_out
int decode_tile(ltx_ctx* ctx, VaeModel* m, const float* latent, long chan_stride, const float* noise, int has_ts,
                float timestep, Dims d0, float* frames, int apply_clip) {
    hipStream_t st = ctx->stream;
    float* xa = m->xa.as<float>();
    float* xb = m->xb.as<float>();
    float* t1 = m->t1.as<float>();
    bf16_t* hb = m->hb.as<bf16_t>();
    float* mods = m->mods.as<float>();
    float* skws = m->skws.as<float>();
    const long skn = (long)(m->skws.bytes / 4);

    // per-block modulation vectors: rows shift1, scale1+1, shift2, scale2+1 (VideoDecoder.swift:93-113)
    float* emb = m->temb.as<float>();        // [256] sinusoid, [256] hidden, then per-group outputs
    if (has_ts) {
        float* tsd = emb + 16384;
        launch_fill_const_f32(tsd, 1, timestep, st);
        launch_timestep_embedding(tsd, m->ts_mult_host, emb, 1, 256, st);
    }
    long mod_off = 0;
    long mod_ofs_group[4];
    VaeModsBatch mb;  // every table of the decode in one launch; the time embeddings of the groups sit side by side in `emb`
    float* te_next = emb + 512;
    auto time_embed = [&](const VaeTimeEmbedder& t) -> const float* {
        if (!has_ts) return nullptr;
        float* hid = emb + 256;
        float* out = te_next;
        te_next += t.out;
        launch_gemv_f32(emb, 256, t.w1, 256, t.b1, hid, 256, 1, 256, 256, LTX_ACT_NONE, st);
        launch_gemv_f32(hid, 256, t.w2, 256, t.b2, out, t.out, 1, t.out, 256, LTX_ACT_SILU, st);
        return out;
    };
    for (int g = 0; g < 4; ++g) {
        const int C = m->groups[g].C;
        const float* te = time_embed(m->groups[g].te);
        mod_ofs_group[g] = mod_off;
        for (int r = 0; r < 5; ++r) {
            mb.job[mb.n++] = VaeModsJob{m->groups[g].blocks[r].sst, te, mods + mod_off, 4, C};
            mod_off += 4L * C;
        }
    }
    const long last_mod = mod_off;
    mb.job[mb.n++] = VaeModsJob{m->last_sst, time_embed(m->last_te), mods + last_mod, 2, m->channels[3]};
    launch_vae_make_mods_batch(mb, st);

    // noise blend + denormalise -> channels-last bf16 (VideoDecoder.swift:366-381)
    Dims d = d0;
    launch_vae_prepare(latent, chan_stride, has_ts ? noise : nullptr, 0.025f, m->mean, m->std_, hb, m->latent_channels, d.P(), st);
    {
        GemmEpilogue e;
        e.out_f32 = xa;
        e.ld_f32 = m->channels[0];
        conv3d(hb, d, m->conv_in, e, st, skws, skn);
    }
    float* x = xa;
    float* xo = xb;
    // Stages whose channels fit one 128-column GEMM tile (the last, most expensive one) never run a PixelNorm pass: each conv's
    // epilogue also emits the NEXT conv's input, PixelNorm + modulation + SiLU of the value it just produced, into the other of
    // two bf16 buffers (the conv reads one while it writes the other). Wider stages keep the separate row pass.
    bf16_t* hb2 = m->hb2.as<bf16_t>();
    const int C3 = m->channels[3];
    bool hin_ready = false;  // the upsampler in front of this group already wrote the first res-block's input
    for (int g = 0; g < 4; ++g) {
        const int C = m->groups[g].C;
        const bool fuse = C == 128;
        bf16_t* hin = hb;   // the next conv's input
        bf16_t* hout = hb2;
        for (int r = 0; r < 5; ++r) {
            const float* md = mods + mod_ofs_group[g] + (long)r * 4 * C;
            // what the block's last conv emits beside the f32 stream: the next consumer's bf16 input
            const float* nsc = nullptr;
            const float* nsh = nullptr;
            bool copy = false;
            if (fuse) {
                // the next block's conv1 (PixelNorm + its modulation + SiLU), conv_out (the last table), or the upsampler's conv (plain cast)
                const float* nmd = r + 1 < 5 ? md + 4L * C : (g == 3 ? mods + last_mod : nullptr);
                if (nmd || g == 3) {
                    nsc = g == 3 && r == 4 ? mods + last_mod + C3 : nmd + 1 * C;
                    nsh = g == 3 && r == 4 ? mods + last_mod : nmd + 0 * C;
                } else {
                    copy = true;
                }
            } else if (r == 4 && g < 3) {
                copy = true;  // the group's last conv also emits the bf16 copy of the stream that the upsampler's conv reads
            }
            res_block(m->groups[g].blocks[r], md, C, d, x, t1, hin, hout, r == 0 && !hin_ready, nsc, nsh, copy, st, skws, skn);
        }
        hin_ready = false;
        if (g < 3) {
            // depth-to-space upsampler (VideoDecoder.swift:215-251): conv on the raw stream, D2S, drop frame 0, + D2S(x)
            if (!fuse) {
                bf16_t* t = hin; hin = hout; hout = t;  // written by the last conv2 above
            }
            GemmEpilogue e;
            e.out_f32 = xo;
            e.ld_f32 = C / 2;
            e.d2s = 1;
            e.resid_src = x;
            e.ld_resid = C;
            // The upsampler in front of the 128-channel stage (round 5): a 128-column tile of this launch is one sub-position with ALL
            // channels of its voxel, so the epilogue also emits the first res-block's input - PixelNorm + its modulation + SiLU - where a
            // row pass re-read the whole upsampled f32 stream (315 MB at 768x512). Only where a halo-staged kernel takes the launch (they
            // carry that epilogue) and the conv's input is not the buffer the next stage starts from.
            const bool halo_w = (d.W <= 192 && d.W >= 48 && 192 % d.W == 0) || d.W % 192 == 0;
            if (C / 2 == 128 && m->groups[g + 1].C == 128 && halo_w && d.H >= 2 && hin != hb && ltx_opt(OPT_CONV_D2S_PN) != 0) {
                const float* nmd = mods + mod_ofs_group[g + 1];
                e.pn_out = hb;
                e.ld_pn = 128;
                e.pn_scale = nmd + 1 * 128;
                e.pn_shift = nmd + 0 * 128;
                hin_ready = true;
            }
            conv3d(hin, d, m->up[g], e, st);
            d.F = 2 * d.F - 1;
            d.H *= 2;
            d.W *= 2;
            float* tmp = x;
            x = xo;
            xo = tmp;
        } else if (!fuse) {
            launch_pixelnorm_silu(x, mods + last_mod + C3, mods + last_mod, hin, d.P(), C3, st);
        }
        if (g == 3) {
            GemmEpilogue e;  // conv_out's channels are stored in un-patchify order: the epilogue writes (F, 4H, 4W, 3) itself
            e.out_f32 = frames;
            e.ld_f32 = 1;
            e.d2s = 3;
            e.clip01 = apply_clip;
            conv3d(hin, d, m->conv_out, e, st);
        }
    }
    return d.F;
}

}  // namespace

namespace {

void ensure_decode_workspace(ltx_ctx* ctx, VaeModel* m, const TilePlan& plan, int H, int W) {
    int maxf = 0;
    for (size_t i = 0; i < plan.start.size(); ++i) maxf = std::max(maxf, plan.end[i] - plan.start[i]);
    const long P3 = (long)(8 * (maxf - 1) + 1) * (H * 8) * (W * 8);
    const long elems = P3 * 128;
    if (elems > m->ws_elems) {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        m->xa.ensure((size_t)elems * 4);
        m->xb.ensure((size_t)elems * 4);
        m->t1.ensure((size_t)elems * 4);
        m->hb.ensure((size_t)elems * 2);
        m->hb2.ensure((size_t)elems * 2);
        m->skws.ensure((size_t)elems * 4);
        m->ws_elems = elems;
    }
    m->mods.ensure((size_t)(20 * 4 * 1024 + 256) * 4);
    m->temb.ensure((size_t)(16384 + 64) * 4);  // sinusoid [256], hidden [256], the groups' time embeddings (4C each), the timestep at 16384
}

TilePlan checked_plan(const VaeDecodeArgs& a) {
    LTX_REQUIRE(a.latent && a.F >= 1 && a.H >= 2 && a.W >= 2, "vae_decode: bad arguments (F=%d H=%d W=%d)", a.F, a.H, a.W);
    LTX_REQUIRE(!a.has_timestep || a.noise, "vae_decode: timestep conditioning needs an explicit noise tensor");
    const TilePlan plan = vae_tile_plan(a.F, a.tile, a.overlap);
    LTX_REQUIRE(!plan.start.empty(), "vae_decode: temporal tile size %d must exceed overlap %d", a.tile, a.overlap);
    return plan;
}

// raw frames of tile i of the plan -> dst; returns its frame count
int decode_plan_tile(ltx_ctx* ctx, VaeModel* m, const VaeDecodeArgs& a, const TilePlan& plan, int i, float* dst, int apply_clip) {
    const long chan_stride = (long)a.F * a.H * a.W;
    const long hw = (long)a.H * a.W;
    const int s = plan.start[i], e = plan.end[i];
    return decode_tile(ctx, m, a.latent + s * hw, chan_stride, a.noise ? a.noise + s * hw : nullptr, a.has_timestep, a.timestep,
                       Dims{e - s, a.H, a.W}, dst, apply_clip);
}

}  // namespace

int vae_decode_tile(ltx_ctx* ctx, VaeModel* m, const VaeDecodeArgs& a, int tile_index) {
    const TilePlan plan = checked_plan(a);
    LTX_REQUIRE(tile_index >= 0 && tile_index < (int)plan.start.size(), "vae_decode_tile: tile %d of %d", tile_index, (int)plan.start.size());
    LTX_REQUIRE(a.frames, "vae_decode_tile: null output");
    const long HWpix = (long)a.H * 32 * a.W * 32 * 3;
    const int nf = 8 * (plan.end[tile_index] - plan.start[tile_index] - 1) + 1;
    LTX_REQUIRE(a.frames_cap >= (long)nf * HWpix, "vae_decode_tile: output buffer too small (%ld < %ld floats)", a.frames_cap, (long)nf * HWpix);
    ensure_decode_workspace(ctx, m, plan, a.H, a.W);
    const int got = decode_plan_tile(ctx, m, a, plan, tile_index, a.frames, 0);
    if (a.n_frames_out) *a.n_frames_out = got;
    return got;
}

void vae_res_block(ltx_ctx* ctx, VaeModel* m, int group, int block, float* x, int F, int H, int W) {
    LTX_REQUIRE(group >= 0 && group < 4 && block >= 0 && block < 5, "vae_res_block: group %d block %d", group, block);
    LTX_REQUIRE(x && F >= 1 && H >= 2 && W >= 2, "vae_res_block: bad arguments (F=%d H=%d W=%d)", F, H, W);
    hipStream_t st = ctx->stream;
    const int C = m->groups[group].C;
    const Dims d{F, H, W};
    const size_t elems = (size_t)d.P() * C;
    // buffers of its own size: the decode workspace is sized by latent shape, this call by the stream it is given
    DevBuf t1, hb, hb2, mods, skws;
    t1.ensure(elems * 4);
    hb.ensure(elems * 2);
    hb2.ensure(elems * 2);
    skws.ensure(elems * 4);
    mods.ensure((size_t)4 * C * 4);
    VaeModsBatch mb;
    mb.job[mb.n++] = VaeModsJob{m->groups[group].blocks[block].sst, nullptr, mods.as<float>(), 4, C};
    launch_vae_make_mods_batch(mb, st);
    bf16_t* hin = hb.as<bf16_t>();
    bf16_t* hout = hb2.as<bf16_t>();
    res_block(m->groups[group].blocks[block], mods.as<float>(), C, d, x, t1.as<float>(), hin, hout, true, nullptr, nullptr, false, st,
              skws.as<float>(), (long)elems);
    HIP_CHECK(hipStreamSynchronize(st));  // the scratch buffers above go out of scope
}

void vae_upsample(ltx_ctx* ctx, VaeModel* m, int group, const float* x, int F, int H, int W, float* out) {
    LTX_REQUIRE(group >= 0 && group < 3, "vae_upsample: group %d (0..2)", group);
    LTX_REQUIRE(x && out && F >= 1 && H >= 2 && W >= 2, "vae_upsample: bad arguments (F=%d H=%d W=%d)", F, H, W);
    hipStream_t st = ctx->stream;
    const int C = m->groups[group].C;
    const Dims d{F, H, W};
    DevBuf hb;
    hb.ensure((size_t)d.P() * C * 2);
    launch_cast_f32_bf16(x, hb.as<bf16_t>(), d.P() * C, st);  // the conv reads the bf16 copy of the stream, as in the decode
    GemmEpilogue e;
    e.out_f32 = out;
    e.ld_f32 = C / 2;
    e.d2s = 1;
    e.resid_src = x;
    e.ld_resid = C;
    conv3d(hb.as<bf16_t>(), d, m->up[group], e, st);
    HIP_CHECK(hipStreamSynchronize(st));  // the scratch buffer above goes out of scope
}

int vae_blend_tiles(ltx_ctx* ctx, const float* const* tiles, const int* tile_frames, int n_tiles, int overlap, int H, int W, float* frames,
                    long frames_cap) {
    LTX_REQUIRE(tiles && tile_frames && n_tiles >= 1 && frames && overlap >= 0, "vae_blend_tiles: bad arguments");
    hipStream_t st = ctx->stream;
    const long HWpix = (long)H * 32 * W * 32 * 3;
    const int po = 8 * overlap;
    // frame count first (the walk of VideoDecoder.swift:561-592), so that a short buffer is refused before anything is written
    long total = tile_frames[0];
    for (int i = 1; i < n_tiles; ++i) total += (po > 0 && po < total && po < tile_frames[i]) ? tile_frames[i] - po : tile_frames[i];
    LTX_REQUIRE(frames_cap >= total * HWpix, "vae_blend_tiles: output buffer too small (%ld < %ld floats)", frames_cap, total * HWpix);
    long cur = 0;
    for (int i = 0; i < n_tiles; ++i) {
        const float* src = tiles[i];
        const int nf = tile_frames[i];
        LTX_REQUIRE(src && nf >= 1, "vae_blend_tiles: tile %d is empty", i);
        if (i == 0) {
            if (src != frames) HIP_CHECK(hipMemcpyAsync(frames, src, (size_t)nf * HWpix * 4, hipMemcpyDeviceToDevice, st));
            cur = nf;
            continue;
        }
        if (po > 0 && po < cur && po < nf) {
            launch_blend_frames(frames + (cur - po) * HWpix, src, po, HWpix, st);
            HIP_CHECK(hipMemcpyAsync(frames + cur * HWpix, src + (long)po * HWpix, (size_t)(nf - po) * HWpix * 4, hipMemcpyDeviceToDevice, st));
            cur += nf - po;
        } else {
            HIP_CHECK(hipMemcpyAsync(frames + cur * HWpix, src, (size_t)nf * HWpix * 4, hipMemcpyDeviceToDevice, st));
            cur += nf;
        }
    }
    launch_clip01(frames, cur * HWpix, st);  // ((x + 1) / 2 clipped to [0,1]) AFTER the blend (VideoDecoder.swift:501-505)
    return (int)cur;
}

void vae_decode(ltx_ctx* ctx, VaeModel* m, const VaeDecodeArgs& a) {
    const TilePlan plan = checked_plan(a);
    // gather form (shard == 2): only the root rank produces frames
    LTX_REQUIRE(a.shard != 2 || (a.root >= 0 && a.root < dist_world(ctx)), "vae_decode: root %d of %d ranks", a.root, dist_world(ctx));
    const bool makes_frames = a.shard != 2 || dist_rank(ctx) == a.root;
    LTX_REQUIRE(a.frames || !makes_frames, "vae_decode: null output");
    hipStream_t st = ctx->stream;
    const long HWpix = (long)a.H * 32 * a.W * 32 * 3;
    LTX_REQUIRE(!makes_frames || a.frames_cap >= (long)plan.out_frames * HWpix, "vae_decode: output buffer too small (%ld < %ld floats)",
                a.frames_cap, (long)plan.out_frames * HWpix);
    ensure_decode_workspace(ctx, m, plan, a.H, a.W);
    const int n_tiles = (int)plan.start.size();
    if (n_tiles == 1) {
        int nf = plan.out_frames;
        if (makes_frames) nf = decode_plan_tile(ctx, m, a, plan, 0, a.frames, 1);
        if (a.n_frames_out) *a.n_frames_out = nf;
        return;
    }
    // temporal tiling with linear blending of 8*overlap pixel frames (VideoDecoder.swift:517-602)
    const int world = a.shard ? dist_world(ctx) : 1, rank = a.shard ? dist_rank(ctx) : 0;
    if (world == 1) {
        // one tile buffer, blended into the output as it is produced
        int maxf = 0;
        for (int i = 0; i < n_tiles; ++i) maxf = std::max(maxf, 8 * (plan.end[i] - plan.start[i] - 1) + 1);
        m->tile_frames.ensure((size_t)maxf * HWpix * 4);
        const int po = 8 * a.overlap;
        long cur = 0;
        for (int i = 0; i < n_tiles; ++i) {
            float* dst = (i == 0) ? a.frames : m->tile_frames.as<float>();
            const int nf = decode_plan_tile(ctx, m, a, plan, i, dst, 0);
            if (i == 0) {
                cur = nf;
                continue;
            }
            if (po > 0 && po < cur && po < nf) {
                launch_blend_frames(a.frames + (cur - po) * HWpix, m->tile_frames.as<float>(), po, HWpix, st);
                HIP_CHECK(hipMemcpyAsync(a.frames + cur * HWpix, m->tile_frames.as<float>() + (long)po * HWpix,
                                         (size_t)(nf - po) * HWpix * 4, hipMemcpyDeviceToDevice, st));
                cur += nf - po;
            } else {
                HIP_CHECK(hipMemcpyAsync(a.frames + cur * HWpix, m->tile_frames.as<float>(), (size_t)nf * HWpix * 4, hipMemcpyDeviceToDevice, st));
                cur += nf;
            }
        }
        launch_clip01(a.frames, cur * HWpix, st);
        if (a.n_frames_out) *a.n_frames_out = (int)cur;
        return;
    }
    // tiles sharded over the ranks: tile i is decoded by rank i % world into its slot of one raw-tile
    // buffer, each slot is broadcast from its owner over xGMI, and every rank blends the raw tiles in tile order - the blend is
    // order-dependent and comes BEFORE the clip (VideoDecoder.swift:561-592, :501-505), so raw frames travel, not clipped ones.
    std::vector<int> nfs(n_tiles);
    std::vector<long> off(n_tiles + 1, 0);
    for (int i = 0; i < n_tiles; ++i) {
        nfs[i] = 8 * (plan.end[i] - plan.start[i] - 1) + 1;
        off[i + 1] = off[i] + (long)nfs[i] * HWpix;
    }
    m->tile_frames.ensure((size_t)off[n_tiles] * 4);
    float* raw = m->tile_frames.as<float>();
    for (int i = rank; i < n_tiles; i += world) decode_plan_tile(ctx, m, a, plan, i, raw + off[i], 0);
    std::vector<const float*> ptrs(n_tiles);
    if (a.shard == 2) {
        // gather form: every raw tile travels ONCE, from its owner to the rank that blends (a.root); the other ranks return the frame
        // count only. With 4 tiles on 8 ranks the broadcast form puts 1 GB on every rank, this one 0.8 GB on one.
        LTX_REQUIRE(a.root >= 0 && a.root < world, "vae_decode: root %d of %d ranks", a.root, world);
        for (int i = 0; i < n_tiles; ++i) {
            dist_send_to_root(ctx, raw + off[i], (long)nfs[i] * HWpix * 4, i % world, a.root);
            ptrs[i] = raw + off[i];
        }
        if (rank == a.root) {
            const int total = vae_blend_tiles(ctx, ptrs.data(), nfs.data(), n_tiles, a.overlap, a.H, a.W, a.frames, a.frames_cap);
            LTX_REQUIRE(total == plan.out_frames, "vae_decode: blended %d frames, the plan says %d", total, plan.out_frames);
        }
        if (a.n_frames_out) *a.n_frames_out = plan.out_frames;
        return;
    }
    for (int i = 0; i < n_tiles; ++i) {
        dist_broadcast(ctx, raw + off[i], (long)nfs[i] * HWpix * 4, i % world);
        ptrs[i] = raw + off[i];
    }
    const int total = vae_blend_tiles(ctx, ptrs.data(), nfs.data(), n_tiles, a.overlap, a.H, a.W, a.frames, a.frames_cap);
    if (a.n_frames_out) *a.n_frames_out = total;
}
