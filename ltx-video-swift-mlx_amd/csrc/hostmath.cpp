#include <cmath>
// hostmath.cpp - see hostmath.h. Scalar arithmetic follows the reference operation by operation (Swift `Float`
// = IEEE f32, no fused multiply-add), so floating-point contraction is disabled for this file.
#pragma clang fp contract(off)
#include "hostmath.h"

#include <math.h>
#include <string.h>
#include <stdio.h>

#include <vector>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// R1: validation + latent shape
// ---------------------------------------------------------------------------------------------------------------
int validate_generation_config(int width, int height, int num_frames, int num_steps, float cfg_scale, int two_stage,
                               std::string* msg) {
    char buf[256];
    auto fail = [&](const char* fmt, auto... a) {
        snprintf(buf, sizeof(buf), fmt, a...);
        if (msg) *msg = buf;
        return LTXS_INVALID_CONFIGURATION;
    };
    // order of checks = LTXConfig.swift:310-353
    if (width % 32 != 0) return fail("Width must be divisible by 32, got %d", width);
    if (height % 32 != 0) return fail("Height must be divisible by 32, got %d", height);
    if ((num_frames - 1) % 8 != 0)
        return fail("Number of frames must be 8n + 1 (e.g., 9, 17, 25, ..., 121), got %d", num_frames);
    if (!(width >= 64 && width <= 2048)) return fail("Width must be between 64 and 2048, got %d", width);
    if (!(height >= 64 && height <= 2048)) return fail("Height must be between 64 and 2048, got %d", height);
    if (!(num_frames >= 9 && num_frames <= 257))
        return fail("Number of frames must be between 9 and 257, got %d", num_frames);
    if (!(num_steps >= 1 && num_steps <= 100))
        return fail("Number of steps must be between 1 and 100, got %d", num_steps);
    if (!(cfg_scale >= 1.0f && cfg_scale <= 20.0f))
        return fail("CFG scale must be between 1.0 and 20.0, got %g", (double)cfg_scale);
    // generateVideoTwoStage (LTXPipeline.swift:2443): final dims must be divisible by 64
    if (two_stage && (width % 64 != 0 || height % 64 != 0))
        return fail("Two-stage generation requires width and height divisible by 64, got %dx%d", width, height);
    if (msg) msg->clear();
    return LTXS_OK;
}

void latent_shape(int width, int height, int num_frames, int* F, int* H, int* W) {
    *F = (num_frames - 1) / 8 + 1;
    *H = height / 32;
    *W = width / 32;
}

// ---------------------------------------------------------------------------------------------------------------
// R3: sigma schedules
// ---------------------------------------------------------------------------------------------------------------
const float kDistilledSigmas[9] = {1.0f, 0.99375f, 0.9875f, 0.98125f, 0.975f, 0.909375f, 0.725f, 0.421875f, 0.0f};
const float kStage2DistilledSigmas[4] = {0.909375f, 0.725f, 0.421875f, 0.0f};

std::vector<float> compute_sigmas(bool distilled, int num_steps, int token_count) {
    const int BASE_SHIFT_ANCHOR = 1024, MAX_SHIFT_ANCHOR = 4096;
    const float maxShift = 2.05f, baseShift = 0.95f, terminal = 0.1f;
    std::vector<float> s;
    if (distilled) {
        for (float v : kDistilledSigmas)
            if (v > 0) s.push_back(v);
        if (token_count > 0) {
            const int clamped = token_count < MAX_SHIFT_ANCHOR ? token_count : MAX_SHIFT_ANCHOR;
            const float x1 = (float)BASE_SHIFT_ANCHOR, x2 = (float)MAX_SHIFT_ANCHOR;
            const float mm = (maxShift - baseShift) / (x2 - x1);
            const float b = baseShift - mm * x1;
            const float mu = (float)clamped * mm + b;
            const float expMu = expf(mu);
            for (float& sg : s) {
                if (sg == 0.0f || sg == 1.0f) continue;
                sg = expMu / (expMu + (1.0f / sg - 1.0f));
            }
            const float lastOneMinus = s.empty() ? 0.0f : 1.0f - s.back();
            if (lastOneMinus > 0) {
                const float scaleFactor = lastOneMinus / (1.0f - terminal);
                for (float& sg : s) {
                    if (sg == 0.0f) continue;
                    sg = 1.0f - ((1.0f - sg) / scaleFactor);
                }
            }
        }
        s.push_back(0.0f);
        return s;
    }
    int tc = token_count > 0 ? token_count : MAX_SHIFT_ANCHOR;
    if (tc > MAX_SHIFT_ANCHOR) tc = MAX_SHIFT_ANCHOR;
    for (int i = 0; i <= num_steps; ++i) s.push_back(1.0f - (float)i / (float)num_steps);
    const float x1 = (float)BASE_SHIFT_ANCHOR, x2 = (float)MAX_SHIFT_ANCHOR;
    const float mm = (maxShift - baseShift) / (x2 - x1);
    const float b = baseShift - mm * x1;
    const float sigmaShift = (float)tc * mm + b;
    const float expShift = expf(sigmaShift);
    for (float& sg : s) {
        if (sg == 0.0f) continue;
        sg = expShift / (expShift + powf(1.0f / sg - 1.0f, 1.0f));
    }
    if (num_steps > 0) {
        std::vector<float> om(s.size());
        for (size_t i = 0; i < s.size(); ++i) om[i] = 1.0f - s[i];
        const float lastOneMinus = om[num_steps - 1];
        const float scaleFactor = lastOneMinus / (1.0f - terminal);
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == 0.0f) continue;
            s[i] = 1.0f - (om[i] / scaleFactor);
        }
    }
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// R8 + R9: position grid and double-precision split-RoPE tables
// ---------------------------------------------------------------------------------------------------------------
void rope_tables(const TransformerConfig& cfg, int F, int H, int W, float fps, std::vector<float>* cos_out,
                 std::vector<float>* sin_out) {
    const int dim = cfg.inner_dim();
    const int nPosDims = 3;
    const int nElem = 2 * nPosDims;
    const int numIndices = dim / nElem > 1 ? dim / nElem : 1;
    const int freqDim = numIndices * nPosDims;
    const int expected = dim / 2;
    const int pad = expected - freqDim > 0 ? expected - freqDim : 0;
    const int per_token = pad + freqDim;
    const long T = (long)F * H * W;

    // createPositionGrid: f32 pixel-space mid coordinates, temporal causal fix, /fps
    const float tScale = 8.0f, sScale = 32.0f;
    std::vector<float> tc(F), hc(H), wc(W);
    for (int i = 0; i < F; ++i) {
        const float fi = (float)i;
        float start = fi * tScale, end = (fi + 1) * tScale;
        start = fmaxf(start + (1 - tScale), 0.0f);
        end = fmaxf(end + (1 - tScale), 0.0f);
        tc[i] = ((start + end) / 2.0f) / fps;
    }
    for (int i = 0; i < H; ++i) hc[i] = (float)i * sScale + sScale / 2.0f;
    for (int i = 0; i < W; ++i) wc[i] = (float)i * sScale + sScale / 2.0f;

    const double theta = (double)cfg.rope_theta;
    const double logStart = log(1.0) / log(theta);
    const double logEnd = log(theta) / log(theta);
    std::vector<double> idx(numIndices);
    for (int i = 0; i < numIndices; ++i) {
        const double t = numIndices > 1 ? logStart + (logEnd - logStart) * (double)i / (double)(numIndices - 1) : logStart;
        idx[i] = pow(theta, t) * (M_PI / 2.0);
    }
    cos_out->assign((size_t)T * per_token, 0.0f);
    sin_out->assign((size_t)T * per_token, 0.0f);
    for (int f = 0; f < F; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const long t = ((long)f * H + y) * W + x;
                const double g[3] = {(double)tc[f], (double)hc[y], (double)wc[x]};
                double sp[3];
                for (int d = 0; d < 3; ++d) {
                    const double frac = g[d] / (double)cfg.max_pos[d];
                    sp[d] = frac * 2.0 - 1.0;
                }
                float* c = cos_out->data() + t * per_token;
                float* s = sin_out->data() + t * per_token;
                for (int p = 0; p < pad; ++p) {
                    c[p] = 1.0f;
                    s[p] = 0.0f;
                }
                for (int fi = 0; fi < numIndices; ++fi)
                    for (int d = 0; d < 3; ++d) {
                        const double fr = idx[fi] * sp[d];
                        c[pad + fi * 3 + d] = (float)cos(fr);
                        s[pad + fi * 3 + d] = (float)sin(fr);
                    }
            }
}

// ---------------------------------------------------------------------------------------------------------------
// R19: temporal tiling plan
// ---------------------------------------------------------------------------------------------------------------
TilePlan vae_tile_plan(int latent_frames, int tile, int overlap) {
    TilePlan p;
    if (!(tile > 0 && latent_frames > tile)) {
        p.start.push_back(0);
        p.end.push_back(latent_frames);
        p.out_frames = 8 * (latent_frames - 1) + 1;
        return p;
    }
    const int stride = tile - overlap;
    if (stride <= 0) return p;  // the reference would never terminate; caller reports invalidConfiguration
    int start = 0;
    while (start < latent_frames) {
        const int end = start + tile < latent_frames ? start + tile : latent_frames;
        p.start.push_back(start);
        p.end.push_back(end);
        if (end >= latent_frames) break;
        start += stride;
    }
    const int pixelOverlap = 8 * overlap;
    int result = 8 * (p.end[0] - p.start[0] - 1) + 1;
    for (size_t i = 1; i < p.start.size(); ++i) {
        const int next = 8 * (p.end[i] - p.start[i] - 1) + 1;
        if (pixelOverlap > 0 && pixelOverlap < result && pixelOverlap < next)
            result = result + next - pixelOverlap;
        else
            result = result + next;
    }
    p.out_frames = result;
    return p;
}

// ---------------------------------------------------------------------------------------------------------------
// R20: key mapping
// ---------------------------------------------------------------------------------------------------------------
namespace {
bool has_prefix(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
bool has_suffix(const std::string& s, const char* p) {
    const size_t n = strlen(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}
bool contains(const std::string& s, const char* p) { return s.find(p) != std::string::npos; }
void replace_all(std::string& s, const char* from, const char* to) {
    const size_t nf = strlen(from), nt = strlen(to);
    size_t pos = 0;
    while ((pos = s.find(from, pos)) != std::string::npos) {
        s.replace(pos, nf, to);
        pos += nt;
    }
}
}  // namespace

bool map_transformer_file_key(const std::string& key, std::string* out) {
    // loadTransformerWeights filters (ModelDownloader.swift:617-629)
    if (has_suffix(key, ".weight_scale") || has_suffix(key, ".input_scale")) return false;
    if (contains(key, "audio") || has_prefix(key, "vocoder") || contains(key, "av_ca_")) return false;
    const char* diffusion = "model.diffusion_model.";
    if (!has_prefix(key, diffusion)) return false;
    if (has_prefix(key, "model.diffusion_model.video_embeddings_connector.")) return false;
    if (has_prefix(key, "model.diffusion_model.audio_embeddings_connector.")) return false;
    std::string k = key.substr(strlen(diffusion));
    // mapTransformerKey (ModelDownloader.swift:756-803)
    if (has_prefix(k, "audio_") || contains(k, ".audio_") || has_prefix(k, "av_cross_attn_") ||
        contains(k, "video_to_audio") || contains(k, "video_a2v") || contains(k, "a2v_ca") ||
        contains(k, "scale_shift_table_a2v"))
        return false;
    if (has_prefix(k, "proj_in.")) k = "patchify_proj." + k.substr(strlen("proj_in."));
    if (has_prefix(k, "time_embed.emb.timestep_embedder."))
        k = "adaln_single.emb." + k.substr(strlen("time_embed.emb.timestep_embedder."));
    else if (has_prefix(k, "time_embed.linear."))
        k = "adaln_single." + k.substr(strlen("time_embed."));
    else if (has_prefix(k, "adaln_single.emb.timestep_embedder."))
        k = "adaln_single.emb." + k.substr(strlen("adaln_single.emb.timestep_embedder."));
    replace_all(k, ".emb.timestep_embedder.", ".emb.");
    replace_all(k, ".norm_q.", ".q_norm.");
    replace_all(k, ".norm_k.", ".k_norm.");
    replace_all(k, ".to_out.0.", ".to_out.");
    replace_all(k, "ff.net.0.proj.", "ff.project_in.proj.");
    replace_all(k, "ff.net.2.", "ff.project_out.");
    *out = k;
    return true;
}

bool map_vae_file_key(const std::string& file_key, std::string* out) {
    std::string key = file_key;
    // splitUnifiedWeightsDict strips "vae." (ModelDownloader.swift:1386-1387)
    if (has_prefix(key, "vae.")) key = key.substr(4);
    if (has_prefix(key, "encoder.")) return false;
    if (contains(key, "per_channel_statistics")) {
        const size_t dot = key.rfind('.');
        const std::string base = dot == std::string::npos ? key : key.substr(dot + 1);
        if (base == "mean-of-means") { *out = "mean_of_means"; return true; }
        if (base == "std-of-means") { *out = "std_of_means"; return true; }
        return false;
    }
    if (key == "latents_mean") { *out = "mean_of_means"; return true; }
    if (key == "latents_std") { *out = "std_of_means"; return true; }
    std::string k = key;
    if (has_prefix(k, "decoder.")) k = k.substr(strlen("decoder."));
    if (has_prefix(k, "mid_block.")) {
        k = "up_blocks_0." + k.substr(strlen("mid_block."));
    } else {
        for (int i = 0; i <= 2; ++i) {
            char up[64], rs[64];
            snprintf(up, sizeof(up), "up_blocks.%d.upsamplers.0.", i);
            snprintf(rs, sizeof(rs), "up_blocks.%d.resnets.", i);
            if (has_prefix(k, up)) {
                char nb[64];
                snprintf(nb, sizeof(nb), "up_blocks_%d.", 2 * i + 1);
                k = std::string(nb) + k.substr(strlen(up));
                break;
            } else if (has_prefix(k, rs)) {
                char nb[64];
                snprintf(nb, sizeof(nb), "up_blocks_%d.resnets.", 2 * i + 2);
                k = std::string(nb) + k.substr(strlen(rs));
                break;
            }
        }
    }
    for (int i = 0; i <= 6; ++i) {
        char src[32];
        snprintf(src, sizeof(src), "up_blocks.%d.", i);
        if (has_prefix(k, src)) {
            char nb[32];
            snprintf(nb, sizeof(nb), "up_blocks_%d.", i);
            k = std::string(nb) + k.substr(strlen(src));
            break;
        }
    }
    replace_all(k, ".resnets.", ".res_blocks.");
    *out = k;
    return true;
}

bool map_lora_key(const std::string& lora_base, std::string* out) {
    // LoRAKeyMapper.loraKeyToModelKey (LoRALoader.swift:209-243)
    std::string k = lora_base;
    if (has_prefix(k, "diffusion_model.")) k = k.substr(strlen("diffusion_model."));
    replace_all(k, ".emb.timestep_embedder.", ".emb.");
    replace_all(k, ".to_out.0", ".to_out");
    replace_all(k, ".ff.net.0.proj", ".ff.project_in.proj");
    replace_all(k, ".ff.net.2", ".ff.project_out");
    *out = k + ".weight";
    return true;
}


// ---------------------------------------------------------------------------------------------------------------
// text-embedding connector helpers
// ---------------------------------------------------------------------------------------------------------------
namespace {
bool starts_with(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }
void replace_all(std::string& s, const std::string& a, const std::string& b) {
    size_t pos = 0;
    while ((pos = s.find(a, pos)) != std::string::npos) {
        s.replace(pos, a.size(), b);
        pos += b.size();
    }
}
std::string connector_internal(std::string k) {
    replace_all(k, "transformer_blocks.", "transformer_1d_blocks.");
    replace_all(k, ".norm_q.", ".q_norm.");
    replace_all(k, ".norm_k.", ".k_norm.");
    replace_all(k, ".to_out.0.", ".to_out.");
    replace_all(k, ".ff.net.0.proj.", ".ff.project_in.proj.");
    replace_all(k, ".ff.net.2.", ".ff.project_out.");
    return k;
}
}  // namespace

bool map_text_encoder_file_key(const std::string& file_key, std::string* module_key) {
    std::string k = file_key;
    static const char* pre[3][2] = {{"model.diffusion_model.video_embeddings_connector.", "video_embeddings_connector."},
                                    {"model.diffusion_model.audio_embeddings_connector.", "audio_embeddings_connector."},
                                    {"model.diffusion_model.text_embedding_projection.", "text_embedding_projection."}};
    for (auto& p : pre)
        if (starts_with(k, p[0])) {
            k = std::string(p[1]) + k.substr(strlen(p[0]));
            break;
        }
    if (starts_with(k, "text_proj_in.")) {
        replace_all(k, "text_proj_in.", "feature_extractor.aggregate_embed.");
    } else if (starts_with(k, "video_connector.")) {
        replace_all(k, "video_connector.", "embeddings_connector.");
        k = connector_internal(k);
    } else if (starts_with(k, "audio_connector.")) {
        replace_all(k, "audio_connector.", "audio_embeddings_connector.");
        k = connector_internal(k);
    } else if (starts_with(k, "text_embedding_projection.")) {
        replace_all(k, "text_embedding_projection.", "feature_extractor.");
    } else if (starts_with(k, "video_embeddings_connector.")) {
        replace_all(k, "video_embeddings_connector.", "embeddings_connector.");
        k = connector_internal(k);
    } else if (starts_with(k, "audio_embeddings_connector.")) {
        k = connector_internal(k);
    } else {
        return false;
    }
    *module_key = k;
    return true;
}

void rope_tables_1d(int T, int dim, double theta, int max_pos, std::vector<float>* cos_out, std::vector<float>* sin_out) {
    const int n_idx = dim / 2 > 1 ? dim / 2 : 1;
    const double log_start = log(1.0) / log(theta), log_end = log(theta) / log(theta);
    std::vector<double> idx(n_idx);
    for (int i = 0; i < n_idx; ++i) {
        const double t = n_idx > 1 ? log_start + (log_end - log_start) * (double)i / (double)(n_idx - 1) : log_start;
        idx[i] = pow(theta, t) * (M_PI / 2.0);
    }
    cos_out->assign((size_t)T * n_idx, 0.f);
    sin_out->assign((size_t)T * n_idx, 0.f);
    for (int t = 0; t < T; ++t) {
        const double scaled = ((double)(float)t / (double)max_pos) * 2.0 - 1.0;
        for (int i = 0; i < n_idx; ++i) {
            const double v = idx[i] * scaled;
            (*cos_out)[(size_t)t * n_idx + i] = (float)cos(v);
            (*sin_out)[(size_t)t * n_idx + i] = (float)sin(v);
        }
    }
}


bool map_vae_encoder_file_key(const std::string& file_key, std::string* module_key) {
    if (!starts_with(file_key, "encoder.")) return false;
    std::string k = file_key.substr(8);
    for (int i = 0; i < 4; ++i) {
        const std::string p = "down_blocks." + std::to_string(i) + ".";
        if (starts_with(k, p.c_str())) {
            k = "down_blocks_" + std::to_string(i) + "." + k.substr(p.size());
            break;
        }
    }
    for (int i = 0; i < 4; ++i) {
        const std::string rp = "down_blocks_" + std::to_string(i) + ".resnets.";
        if (starts_with(k, rp.c_str())) {
            const std::string suffix = k.substr(rp.size());
            if (!starts_with(suffix, "resnets.")) k = rp + "resnets." + suffix;  // EncoderDownBlock.resnets -> EncoderResBlockGroup.resnets
            break;
        }
    }
    for (int i = 0; i < 4; ++i) {
        const std::string dp = "down_blocks_" + std::to_string(i) + ".downsamplers.0.";
        if (starts_with(k, dp.c_str())) {
            k = "down_blocks_" + std::to_string(i) + ".downsamplers." + k.substr(dp.size());
            break;
        }
    }
    *module_key = k;
    return true;
}


// ---------------------------------------------------------------------------------------------------------------
// frame export
// ---------------------------------------------------------------------------------------------------------------
void frames_to_u8(const float* frames, long n, uint8_t* out) {
    for (long i = 0; i < n; ++i) {
        float v = frames[i];
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);  // MLX.clip; NaN compares false twice and converts to 0 below
        const float s = v * 255.0f;
        out[i] = (s == s) ? (uint8_t)s : 0;        // asType(.uint8): truncation toward zero
    }
}

namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void png_chunk(FILE* f, const char* type, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> hdr;
    put_be32(hdr, (uint32_t)data.size());
    fwrite(hdr.data(), 1, 4, f);
    fwrite(type, 1, 4, f);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
    uint32_t crc = crc32_update(0xFFFFFFFFu, (const uint8_t*)type, 4);
    if (!data.empty()) crc = crc32_update(crc, data.data(), data.size());
    std::vector<uint8_t> c;
    put_be32(c, crc ^ 0xFFFFFFFFu);
    fwrite(c.data(), 1, 4, f);
}
}  // namespace

bool write_png_rgb8(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width < 1 || height < 1) return false;
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit, RGB, deflate, no filter/interlace
    png_chunk(f, "IHDR", ihdr);
    // raw scanlines (filter byte 0) wrapped in stored deflate blocks of <= 65535 bytes, zlib header + Adler-32
    const size_t row = (size_t)width * 3 + 1, raw_n = row * height;
    std::vector<uint8_t> raw(raw_n);
    for (int y = 0; y < height; ++y) {
        raw[y * row] = 0;
        memcpy(&raw[y * row + 1], rgb + (size_t)y * width * 3, (size_t)width * 3);
    }
    std::vector<uint8_t> z;
    z.reserve(raw_n + raw_n / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw_n;) {
        const size_t len = raw_n - off > 65535 ? 65535 : raw_n - off;
        z.push_back(off + len == raw_n ? 1 : 0);
        z.push_back(len & 0xff); z.push_back(len >> 8);
        z.push_back(~len & 0xff); z.push_back((~len >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + len);
        for (size_t i = 0; i < len; ++i) {
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        off += len;
    }
    put_be32(z, (b << 16) | a);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
    const bool ok = ferror(f) == 0;
    fclose(f);
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// MLX-compatible noise (see hostmath.h). Restated from the published algorithm of mlx 0.30.x (random.cpp, threefry.cpp,
// erf.cpp); nothing of it is in /root/reference.
// ---------------------------------------------------------------------------------------------------------------
void threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
    static const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    const uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
    uint32_t x0 = ctr[0] + ks[0], x1 = ctr[1] + ks[1];
    for (int i = 0; i < 5; ++i) {  // 5 groups of 4 rounds = Threefry-2x32-20
        for (int r = 0; r < 4; ++r) {
            const int rr = rot[i & 1][r];
            x0 += x1;
            x1 = (x1 << rr) | (x1 >> (32 - rr));
            x1 ^= x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + uint32_t(i + 1);
    }
    out[0] = x0;
    out[1] = x1;
}

void mlx_random_bits(const uint32_t key[2], long n_words, uint32_t* out) {
    const long half = n_words / 2;
    const bool even = (n_words % 2) == 0;
    const long second = half + (even ? 0 : 1);  // where the second words of the pairs go
    for (long i = 0; i < half; ++i) {
        const uint32_t ctr[2] = {uint32_t(i), uint32_t(second + i)};
        uint32_t r[2];
        threefry2x32(key, ctr, r);
        out[i] = r[0];
        out[second + i] = r[1];
    }
    if (!even) {
        const uint32_t ctr[2] = {uint32_t(half), 0u};
        uint32_t r[2];
        threefry2x32(key, ctr, r);
        out[half] = r[0];
    }
}

static float mlx_erfinv(float a) {  // two-branch polynomial in t = log(1 - a^2), max error 2.4 ulp
    float t = std::fma(a, 0.0f - a, 1.0f);
    t = std::log(t);
    float p;
    if (std::fabs(t) > 6.125f) {
        p = 3.03697567e-10f;
        p = std::fma(p, t, 2.93243101e-8f);
        p = std::fma(p, t, 1.22150334e-6f);
        p = std::fma(p, t, 2.84108955e-5f);
        p = std::fma(p, t, 3.93552968e-4f);
        p = std::fma(p, t, 3.02698812e-3f);
        p = std::fma(p, t, 4.83185798e-3f);
        p = std::fma(p, t, -2.64646143e-1f);
        p = std::fma(p, t, 8.40016484e-1f);
    } else {
        p = 5.43877832e-9f;
        p = std::fma(p, t, 1.43285448e-7f);
        p = std::fma(p, t, 1.22774793e-6f);
        p = std::fma(p, t, 1.12963626e-7f);
        p = std::fma(p, t, -5.61530760e-5f);
        p = std::fma(p, t, -1.47697632e-4f);
        p = std::fma(p, t, 2.31468678e-3f);
        p = std::fma(p, t, 1.15392581e-2f);
        p = std::fma(p, t, -2.32015476e-1f);
        p = std::fma(p, t, 8.86226892e-1f);
    }
    return a * p;
}

void mlx_random_normal(uint64_t seed, int draw_index, long n, float* out) {
    uint32_t global[2] = {uint32_t(seed >> 32), uint32_t(seed & 0xffffffffu)};
    uint32_t sub[2] = {0, 0};
    for (int d = 0; d <= draw_index; ++d) {  // KeySequence::next(): split the global key in two, keep row 0, hand out row 1
        uint32_t w[4];
        mlx_random_bits(global, 4, w);
        global[0] = w[0];
        global[1] = w[1];
        sub[0] = w[2];
        sub[1] = w[3];
    }
    std::vector<uint32_t> bits((size_t)n);
    mlx_random_bits(sub, n, bits.data());
    const float lo = std::nextafter(-1.0f, 0.0f), hi = 1.0f;
    const float range = hi - lo;
    const float upper = std::nextafter(1.0f, 0.0f);
    const float maxval = float(UINT32_MAX);
    const float sqrt2 = float(std::sqrt(2.0));
    for (long i = 0; i < n; ++i) {
        float u = float(bits[(size_t)i]) / maxval;
        u = u < upper ? u : upper;
        u = range * u + lo;
        out[i] = sqrt2 * mlx_erfinv(u);
    }
}
