// LTXVideo.hpp - C++ host-side mirror of the reference's public `LTXVideo` Swift module surface for the hot path,
// written on top of the C ABI (include/ltxhip.h). The reference is compiled Swift; this image has no Swift toolchain,
// so the host side above the ABI is C++ with the same type / method names, argument meaning and error behaviour:
//
//   LTXModel, LTXVideoGenerationConfig (+validate)        Configuration/LTXConfig.swift:16-78,216-362
//   TransformerQuantization / LTXQuantizationConfig       Configuration/LTXQuantizationConfig.swift:19-115
//   MemoryOptimizationConfig presets                       Configuration/MemoryOptimizationConfig.swift:69-121
//   LTXError, GenerationTimings, VideoGenerationResult     LTXVideo.swift:66-141,255-348
//   GenerationProgress, PrecomputedEmbeddings, LTXPipeline Pipeline/LTXPipeline.swift:50-72,117-200,571-1076,2420-2741,3134
//
// Out of scope here: the Gemma text encoder and MP4 export. Text embeddings therefore enter through the
// `PrecomputedEmbeddings` hook the reference already has, and every noise tensor is an explicit input (or comes from
// the documented splitmix64/Box-Muller generator below - NOT MLX's threefry stream, which cannot be verified here).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ltxhip.h"

namespace ltx {

static const char* const version = "0.1.0";  // LTXVideo.version

// ---- LTXError (LTXVideo.swift:66-141) ----
struct LTXError : std::runtime_error {
    enum Case { modelNotLoaded = 1, invalidConfiguration, insufficientMemory, weightLoadingFailed, generationFailed,
                generationCancelled, invalidFrameCount, invalidDimensions, fileNotFound, invalidLoRA, hipError,
                textEncodingFailed = 100 };
    Case kind;
    LTXError(Case k, const std::string& what) : std::runtime_error(describe(k, what)), kind(k) {}
    static std::string describe(Case k, const std::string& m) {
        switch (k) {
            case modelNotLoaded: return "Model component not loaded: " + m;
            case invalidConfiguration: return "Invalid configuration: " + m;
            case weightLoadingFailed: return "Failed to load weights: " + m;
            case generationFailed: return "Generation failed: " + m;
            case generationCancelled: return "Generation was cancelled";
            case fileNotFound: return "File not found: " + m;
            case invalidLoRA: return "Invalid LoRA: " + m;
            case textEncodingFailed: return "Text encoding failed: " + m;
            default: return m;
        }
    }
};

// ---- LTXModel (LTXConfig.swift:16-78) ----
enum class LTXModel { dev, distilled };
inline bool isDistilled(LTXModel m) { return m == LTXModel::distilled; }
inline int defaultSteps(LTXModel m) { return m == LTXModel::dev ? 40 : 8; }
inline float defaultGuidance(LTXModel m) { return m == LTXModel::dev ? 4.0f : 1.0f; }
inline const char* unifiedWeightsFilename(LTXModel m) {
    return m == LTXModel::dev ? "ltx-2-19b-dev.safetensors" : "ltx-2-19b-distilled.safetensors";
}
inline std::optional<LTXModel> parseModel(const std::string& s) {
    if (s == "dev") return LTXModel::dev;
    if (s == "distilled") return LTXModel::distilled;
    return std::nullopt;
}

// ---- TransformerQuantization (LTXQuantizationConfig.swift:19-62) ----
enum class TransformerQuantization { bf16, qint8, int4 };
inline int bits(TransformerQuantization q) { return q == TransformerQuantization::bf16 ? 16 : (q == TransformerQuantization::qint8 ? 8 : 4); }
inline std::optional<TransformerQuantization> parseQuant(const std::string& s) {
    if (s == "bf16") return TransformerQuantization::bf16;
    if (s == "qint8") return TransformerQuantization::qint8;
    if (s == "int4") return TransformerQuantization::int4;
    return std::nullopt;
}
struct LTXQuantizationConfig {
    TransformerQuantization transformer = TransformerQuantization::bf16;
    int groupSize = 64;
};

// ---- MemoryOptimizationConfig (MemoryOptimizationConfig.swift:69-121); only the fields that change results ----
struct MemoryOptimizationConfig {
    int evalFrequency = 4;
    bool unloadAfterUse = true;  // on a 288 GB device both models stay resident; kept for API parity
    int vaeTemporalTileSize = 0;
    int vaeTemporalTileOverlap = 1;
    static MemoryOptimizationConfig disabled() { return {0, false, 0, 1}; }
    static MemoryOptimizationConfig light() { return {4, true, 0, 1}; }
    static MemoryOptimizationConfig moderate() { return {2, true, 8, 1}; }
    static MemoryOptimizationConfig aggressive() { return {1, true, 6, 1}; }
    static MemoryOptimizationConfig defaultConfig() { return light(); }
};

// ---- LTXVideoGenerationConfig (LTXConfig.swift:216-362) ----
struct LTXVideoGenerationConfig {
    int width = 704, height = 480, numFrames = 121, numSteps = 8;
    float cfgScale = 1.0f;
    std::optional<uint64_t> seed;
    // Which generator turns `seed` into the initial latent: false = this build's counter-based generator; true = the MLX
    // restatement (ltx_mlx_random_normal: MLXRandom.seed(seed); MLXRandom.normal(shape), LatentUtils.swift:69-83) so that a
    // seed reproduces the reference's noise - unverified against MLX itself, hence opt-in.
    bool mlxCompatibleNoise = false;
    float guidanceRescale = 0.0f, crossAttentionScale = 1.0f, geGamma = 0.0f, stgScale = 0.0f;
    float imageCondNoiseScale = 0.0f;  // LTXConfig.swift:271,289
    std::vector<int> stgBlocks{29};
    bool twoStage = false;
    void validate() const {
        char msg[256];
        if (ltx_validate_generation_config(width, height, numFrames, numSteps, cfgScale, twoStage ? 1 : 0, msg, sizeof(msg)) != 0)
            throw LTXError(LTXError::invalidConfiguration, msg);
    }
    int latentWidth() const { return width / 32; }
    int latentHeight() const { return height / 32; }
    int latentFrames() const { return (numFrames - 1) / 8 + 1; }
    int numLatentTokens() const { return latentFrames() * latentHeight() * latentWidth(); }
};

// ---- progress / timings / result (LTXPipeline.swift:50-72; LTXVideo.swift:255-348) ----
struct GenerationProgress {
    int currentStep, totalSteps;
    float sigma;
    double progress() const { return double(currentStep) / double(totalSteps); }
};
using GenerationProgressCallback = std::function<void(const GenerationProgress&)>;
// what the reference logs per step under --profile (LTXPipeline.swift:945-951): sigma -> sigmaNext, mean / std of the velocity and of the latent
struct StepDiagnostics {
    int step = 0;
    float sigma = 0, sigmaNext = 0, velocityMean = 0, velocityStd = 0, latentMean = 0, latentStd = 0;
};
struct GenerationTimings {
    double textEncoding = 0, vaeDecode = 0;
    std::vector<double> denoiseSteps;
    std::vector<StepDiagnostics> stepDiagnostics;  // filled when profiling
    double totalDenoise() const { double s = 0; for (double d : denoiseSteps) s += d; return s; }
};
struct VideoGenerationResult {
    std::vector<float> frames;  // (numFrames, height, width, 3) float in [0,1] (LTXPipeline.swift:1040-1041)
    int numFrames = 0, height = 0, width = 0;
    uint64_t seed = 0;
    double generationTime = 0;
    std::optional<GenerationTimings> timings;
};

// ---- PrecomputedEmbeddings (LTXPipeline.swift:571-584): the supported way to bypass Gemma ----
struct PrecomputedEmbeddings {
    std::vector<uint16_t> promptEmbeddings;  // [1][S][3840] bf16 bits
    std::vector<int32_t> promptMask;         // [1][S]
    std::vector<uint16_t> nullEmbeddings;    // optional (CFG)
    std::vector<int32_t> nullMask;           // optional; defaults to zeros (LTXPipeline.swift:642)
    int S = 0;
};

// Image-to-video conditioning (generateVideo(image:...), LTXPipeline.swift:2000-2125): the VAE-encoded image latent
// [1][128][1][H'][W'] is an input (the encoder is outside this path); injectionNoise = [numSteps][128][1][H'][W'] N(0,1) draws
// for the per-step re-noising of frame 0, used when config.imageCondNoiseScale > 0 (empty = none).
struct ImageConditioning {
    std::vector<float> imageLatent;
    std::vector<float> injectionNoise;
};

// Deterministic N(0,1) generator used when the caller supplies a seed instead of a noise tensor.
inline std::vector<float> generateNoise(size_t n, uint64_t seed) {
    std::vector<float> out(n);
    auto mix = [](uint64_t x) {
        x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    for (size_t i = 0; i < n; ++i) {
        const uint64_t r = mix(seed ^ (uint64_t(i) * 0xD6E8FEB86659FD93ull));
        const double u1 = (double((r >> 40) & 0xFFFFFF) + 1.0) / 16777217.0, u2 = double((r >> 8) & 0xFFFFFF) / 16777216.0;
        out[i] = float(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
    }
    return out;
}

// ---- LTXPipeline (LTXPipeline.swift:117) ----
class LTXPipeline {
   public:
    LTXPipeline(LTXModel model = LTXModel::distilled, LTXQuantizationConfig quantization = {},
                MemoryOptimizationConfig memoryOptimization = MemoryOptimizationConfig::defaultConfig(), int device = 0)
        : model_(model), quant_(quantization), mem_(memoryOptimization) {
        const int rc = ltx_ctx_create(device, &ctx_);
        if (rc != 0) throw LTXError(LTXError::hipError, ltx_last_error(nullptr));
    }
    ~LTXPipeline() { if (ctx_) ltx_ctx_destroy(ctx_); }
    LTXPipeline(const LTXPipeline&) = delete;
    LTXPipeline& operator=(const LTXPipeline&) = delete;

    bool isLoaded() const { return ditLoaded_ && vaeLoaded_; }  // LTXPipeline.swift:171-173

    // Launcher switches of libltxhip (ltx_ctx_set_option: the library never reads the environment; process-wide; `ltx-video options`
    // lists them). No reference counterpart. E.g. setOption("qk_f32", 1) + setOption("split_f32", 1): the reference's rounding points exactly.
    void setOption(const std::string& key, int value) { check(ltx_ctx_set_option(ctx_, key.c_str(), value)); }
    int option(const std::string& key) const {
        int v = 0;
        if (ltx_ctx_get_option(ctx_, key.c_str(), &v) != 0) throw LTXError(LTXError::invalidConfiguration, ltx_last_error(ctx_));
        return v;
    }

    // loadModels (LTXPipeline.swift:217-361), transformer + VAE parts. transformerConfig: nullptr = reference defaults.
    void loadModels(const std::string& ltxWeightsPath, const std::string& vaeWeightsPath,
                    const ltx_transformer_config* transformerConfig = nullptr) {
        check(ltx_dit_load(ctx_, ltxWeightsPath.c_str(), transformerConfig, bits(quant_.transformer), quant_.groupSize));
        ditLoaded_ = true;
        check(ltx_vae_load(ctx_, vaeWeightsPath.c_str(), nullptr));
        vaeLoaded_ = true;
    }
    // fuseLoRA(from:scale:) -> number of modified layers (LTXPipeline.swift:3134-3153)
    int fuseLoRA(const std::string& loraPath, float scale = 1.0f) {
        int n = 0;
        check(ltx_dit_fuse_lora(ctx_, loraPath.c_str(), scale, &n));
        return n;
    }
    void loadUpscaler(const std::string& path) { check(ltx_upscaler_load(ctx_, path.c_str())); }
    // loadVAEEncoder / encodeImage (LTXPipeline.swift:1871-1932) without the image file I/O: pixels [1][3][1][H][W] f32 in [-1,1]
    // -> ImageConditioning whose latent is already normalised with the decoder's statistics (loadModels first).
    void loadVAEEncoder(const std::string& vaeWeightsPath) { check(ltx_vae_encoder_load(ctx_, vaeWeightsPath.c_str(), 0)); }
    void unloadVAEEncoder() { check(ltx_vae_encoder_unload(ctx_)); }
    ImageConditioning encodeImage(const std::vector<float>& pixels, int width, int height) {
        if (pixels.size() != size_t(3) * width * height) throw LTXError(LTXError::invalidConfiguration, "image tensor must be [1][3][1][H][W]");
        ImageConditioning c;
        c.imageLatent.resize(size_t(128) * (height / 32) * (width / 32));
        check(ltx_vae_encode(ctx_, pixels.data(), 1, height, width, 1, c.imageLatent.data()));
        return c;
    }
    // Connector part of the text encoder (VideoGemmaTextEncoderModel, LTXTextEncoder.swift:535-643; loaded in
    // loadModels, LTXPipeline.swift:420-540): reads text_embedding_projection.* / video_embeddings_connector.* from the
    // unified checkpoint or text_proj_in.* / video_connector.* from a standalone connector file.
    void loadConnector(const std::string& path, const ltx_connector_config* cfg = nullptr) {
        check(ltx_connector_load(ctx_, path.c_str(), cfg));
        if (cfg) connCfg_ = *cfg; else ltx_connector_config_default(&connCfg_);
        connectorLoaded_ = true;
    }
    // encodeFromHiddenStates (LTXTextEncoder.swift:574-643): the 49 Gemma-3 hidden states [49][1][T][3840] (bf16 bits)
    // + the tokenizer's attention mask -> the PrecomputedEmbeddings the generate* calls take. Gemma itself is not part
    // of this library. nullHiddenStates (same layout) is optional, for CFG.
    PrecomputedEmbeddings encodeFromHiddenStates(const std::vector<uint16_t>& hiddenStates, const std::vector<int32_t>& attentionMask,
                                                 int T, bool paddingRight = false, const std::vector<uint16_t>* nullHiddenStates = nullptr,
                                                 const std::vector<int32_t>* nullAttentionMask = nullptr) {
        if (!connectorLoaded_) throw LTXError(LTXError::modelNotLoaded, "Model not loaded: text-embedding connector");
        const size_t need = size_t(connCfg_.states) * T * connCfg_.dim;
        if (hiddenStates.size() != need || attentionMask.size() != size_t(T))
            throw LTXError(LTXError::textEncodingFailed, "hidden states must be [states][1][T][dim] bf16 with a [1][T] mask");
        PrecomputedEmbeddings e;
        e.S = T;
        e.promptEmbeddings.resize(size_t(T) * connCfg_.dim);
        e.promptMask.resize(T);
        check(ltx_connector_encode(ctx_, hiddenStates.data(), attentionMask.data(), 1, T, paddingRight ? 1 : 0, e.promptEmbeddings.data(),
                                   e.promptMask.data()));
        if (nullHiddenStates && nullAttentionMask) {
            if (nullHiddenStates->size() != need || nullAttentionMask->size() != size_t(T))
                throw LTXError(LTXError::textEncodingFailed, "null-prompt hidden states must match the prompt's shape");
            e.nullEmbeddings.resize(size_t(T) * connCfg_.dim);
            e.nullMask.resize(T);
            check(ltx_connector_encode(ctx_, nullHiddenStates->data(), nullAttentionMask->data(), 1, T, paddingRight ? 1 : 0,
                                       e.nullEmbeddings.data(), e.nullMask.data()));
        }
        return e;
    }

    // generateVideo (LTXPipeline.swift:586-1046), text-to-video. `noise`: [1,128,F',H',W'] N(0,1) or empty (then
    // config.seed drives generateNoise above).
    VideoGenerationResult generateVideo(const LTXVideoGenerationConfig& config, const PrecomputedEmbeddings& emb,
                                        const std::vector<float>& noise = {}, GenerationProgressCallback onProgress = nullptr,
                                        bool profile = false, const std::vector<float>& vaeNoise = {},
                                        const ImageConditioning* image = nullptr) {
        config.validate();
        if (!isLoaded()) throw LTXError(LTXError::modelNotLoaded, "Models not loaded. Call loadModels() first.");
        const auto t0 = std::chrono::steady_clock::now();
        GenerationTimings timings;
        const bool useCFG = config.cfgScale > 1.0f;
        const int F = config.latentFrames(), H = config.latentHeight(), W = config.latentWidth();
        if (image) {
            if (image->imageLatent.size() != size_t(128) * H * W)
                throw LTXError(LTXError::invalidConfiguration, "image latent must be [1][128][1][H/32][W/32]");
            if (!image->injectionNoise.empty() && image->injectionNoise.size() != size_t(config.numSteps) * 128 * H * W)
                throw LTXError(LTXError::invalidConfiguration, "injection noise must hold numSteps draws of the image latent's shape");
        }
        const size_t n = size_t(128) * F * H * W;
        std::vector<float> latent = noise;
        if (latent.empty()) {
            if (config.mlxCompatibleNoise) {
                latent.resize(n);
                check(ltx_mlx_random_normal(config.seed.value_or(0), 0, latent.data(), long(n)));
            } else {
                latent = generateNoise(n, config.seed.value_or(0));
            }
        }
        if (latent.size() != n) throw LTXError(LTXError::invalidConfiguration, "noise tensor has the wrong size");
        // sigma schedule (LTXPipeline.swift:775-787); distilled ignores numSteps but the loop runs numSteps times
        float sig[128];
        const int ns = ltx_sigmas(isDistilled(model_) ? 1 : 0, config.numSteps, F * H * W, sig, 128);
        if (config.numSteps > ns - 1) throw LTXError(LTXError::invalidConfiguration, "numSteps exceeds the sigma schedule");
        for (float& v : latent) v *= sig[0];  // LTXPipeline.swift:793
        if (config.crossAttentionScale != 1.0f) check(ltx_dit_set_cross_attn_scale(ctx_, config.crossAttentionScale, 0, -1));
        std::vector<uint16_t> ctxBits;
        std::vector<int32_t> mask;
        buildContext(emb, useCFG, ctxBits, mask);
        ltx_denoise_options opt = LTX_DENOISE_OPTIONS_INIT;
        opt.cfg_scale = config.cfgScale;
        opt.guidance_rescale = config.guidanceRescale;
        opt.stg_scale = config.stgScale;
        opt.stg_blocks = config.stgBlocks.data();
        opt.n_stg_blocks = int(config.stgBlocks.size());
        opt.ge_gamma = config.geGamma;
        if (image) {  // per-token timesteps + frame-0 slice Euler (LTXPipeline.swift:2191-2401)
            opt.cond_latent = image->imageLatent.data();
            opt.image_cond_noise_scale = config.imageCondNoiseScale;
            opt.cond_noise = image->injectionNoise.empty() ? nullptr : image->injectionNoise.data();
        }
        struct Box { GenerationProgressCallback cb; GenerationTimings* t; std::chrono::steady_clock::time_point last; } box{onProgress, &timings, t0};
        auto thunk = [](int step, int total, float sigma, void* user) {
            Box* b = static_cast<Box*>(user);
            const auto now = std::chrono::steady_clock::now();
            if (step > 0) b->t->denoiseSteps.push_back(std::chrono::duration<double>(now - b->last).count());
            b->last = now;
            if (b->cb) b->cb(GenerationProgress{step, total, sigma});
        };
        box.last = std::chrono::steady_clock::now();
        std::vector<float> stepStats;
        if (profile) {  // the per-step diagnostics the reference logs when profiling (LTXPipeline.swift:945-951)
            stepStats.assign(size_t(config.numSteps) * 4, 0.0f);
            opt.step_stats = stepStats.data();
        }
        check(ltx_denoise(ctx_, latent.data(), F, H, W, sig, config.numSteps + 1, ctxBits.data(), mask.data(), emb.S, &opt, thunk, &box));
        timings.denoiseSteps.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - box.last).count());
        for (int i = 0; i < config.numSteps && profile; ++i)
            timings.stepDiagnostics.push_back(StepDiagnostics{i, sig[i], sig[i + 1], stepStats[4 * i], stepStats[4 * i + 1], stepStats[4 * i + 2],
                                                              stepStats[4 * i + 3]});
        VideoGenerationResult r = decode(latent, F, H, W, config, vaeNoise, timings);
        r.seed = config.seed.value_or(0);
        r.generationTime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (profile) r.timings = timings;
        return r;
    }

    // generateVideoTwoStage (LTXPipeline.swift:2420-2741), distilled-style (numSteps<=8, cfg<=1) T2V.
    VideoGenerationResult generateVideoTwoStage(const LTXVideoGenerationConfig& config, const PrecomputedEmbeddings& emb,
                                                const std::vector<float>& noise1, const std::vector<float>& noise2,
                                                GenerationProgressCallback onProgress = nullptr) {
        LTXVideoGenerationConfig c2 = config;
        c2.twoStage = true;
        c2.validate();
        if (!isLoaded()) throw LTXError(LTXError::modelNotLoaded, "Models not loaded. Call loadModels() first.");
        const auto t0 = std::chrono::steady_clock::now();
        GenerationTimings timings;
        int F1, H1, W1, F2, H2, W2;
        ltx_latent_shape(config.width / 2, config.height / 2, config.numFrames, &F1, &H1, &W1);
        ltx_latent_shape(config.width, config.height, config.numFrames, &F2, &H2, &W2);
        std::vector<float> lat = noise1;
        if (lat.size() != size_t(128) * F1 * H1 * W1) throw LTXError(LTXError::invalidConfiguration, "stage-1 noise has the wrong size");
        float sig[128];
        const int ns = ltx_sigmas(1, 8, F1 * H1 * W1, sig, 128);
        for (float& v : lat) v *= sig[0];
        std::vector<uint16_t> ctxBits;
        std::vector<int32_t> mask;
        buildContext(emb, false, ctxBits, mask);
        ltx_denoise_options opt = LTX_DENOISE_OPTIONS_INIT;
        struct Box { GenerationProgressCallback cb; } box{onProgress};
        auto thunk = [](int step, int total, float sigma, void* user) {
            Box* b = static_cast<Box*>(user);
            if (b->cb) b->cb(GenerationProgress{step, total, sigma});
        };
        check(ltx_denoise(ctx_, lat.data(), F1, H1, W1, sig, ns, ctxBits.data(), mask.data(), emb.S, &opt, thunk, &box));
        const std::vector<float> stage1 = lat;
        std::vector<float> up(size_t(128) * F2 * H2 * W2);
        check(ltx_upscale_latent(ctx_, lat.data(), F1, H1, W1, up.data()));
        check(ltx_adain_filter_latent(ctx_, up.data(), long(F2) * H2 * W2, stage1.data(), long(F1) * H1 * W1, 128, 1.0f));
        float s2[4];
        ltx_stage2_sigmas(s2, 4);
        if (noise2.size() != up.size()) throw LTXError(LTXError::invalidConfiguration, "stage-2 noise has the wrong size");
        for (size_t i = 0; i < up.size(); ++i) up[i] = s2[0] * noise2[i] + (1.0f - s2[0]) * up[i];  // LTXPipeline.swift:2644-2647
        check(ltx_denoise(ctx_, up.data(), F2, H2, W2, s2, 4, ctxBits.data(), mask.data(), emb.S, &opt, thunk, &box));
        VideoGenerationResult r = decode(up, F2, H2, W2, config, {}, timings);
        r.generationTime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return r;
    }

    ltx_ctx* context() { return ctx_; }

   private:
    void check(int rc) {
        if (rc != 0) throw LTXError(LTXError::Case(rc), ltx_last_error(ctx_));
    }
    static void buildContext(const PrecomputedEmbeddings& emb, bool useCFG, std::vector<uint16_t>& ctxBits, std::vector<int32_t>& mask) {
        if (emb.promptEmbeddings.empty() || emb.S <= 0)
            throw LTXError(LTXError::textEncodingFailed, "no PrecomputedEmbeddings supplied (the Gemma text encoder is outside this path)");
        const size_t per = emb.promptEmbeddings.size();
        if (useCFG) {  // batch order [negative, positive] (LTXPipeline.swift:715-716)
            ctxBits.assign(2 * per, 0);
            if (!emb.nullEmbeddings.empty()) std::memcpy(ctxBits.data(), emb.nullEmbeddings.data(), per * 2);
            std::memcpy(ctxBits.data() + per, emb.promptEmbeddings.data(), per * 2);
            mask.assign(2 * size_t(emb.S), 0);  // null mask defaults to zeros (LTXPipeline.swift:642)
            if (!emb.nullMask.empty()) std::memcpy(mask.data(), emb.nullMask.data(), size_t(emb.S) * 4);
            std::memcpy(mask.data() + emb.S, emb.promptMask.data(), size_t(emb.S) * 4);
        } else {
            ctxBits = emb.promptEmbeddings;
            mask = emb.promptMask;
        }
    }
    VideoGenerationResult decode(const std::vector<float>& latent, int F, int H, int W, const LTXVideoGenerationConfig& config,
                                 const std::vector<float>& vaeNoise, GenerationTimings& timings) {
        const auto tv = std::chrono::steady_clock::now();
        int outFrames = 0;
        ltx_vae_tile_plan(F, mem_.vaeTemporalTileSize, mem_.vaeTemporalTileOverlap, nullptr, nullptr, 0, &outFrames);
        VideoGenerationResult r;
        r.height = H * 32;
        r.width = W * 32;
        r.frames.resize(size_t(outFrames) * r.height * r.width * 3);
        const int useTs = ltx_vae_timestep_conditioning(ctx_) == 1 ? 1 : 0;  // LTXPipeline.swift:1004
        std::vector<float> vn = vaeNoise;
        if (useTs && vn.empty()) vn = generateNoise(latent.size(), config.seed.value_or(0) + 1);
        int nf = 0;
        check(ltx_vae_decode(ctx_, latent.data(), F, H, W, useTs, 0.05f, useTs ? vn.data() : nullptr, mem_.vaeTemporalTileSize,
                             mem_.vaeTemporalTileOverlap, r.frames.data(), long(r.frames.size()), &nf));
        timings.vaeDecode = std::chrono::duration<double>(std::chrono::steady_clock::now() - tv).count();
        // trim to the requested frame count only if longer (LTXPipeline.swift:1020-1026)
        r.numFrames = nf > config.numFrames ? config.numFrames : nf;
        r.frames.resize(size_t(r.numFrames) * r.height * r.width * 3);
        return r;
    }
    ltx_ctx* ctx_ = nullptr;
    LTXModel model_;
    LTXQuantizationConfig quant_;
    MemoryOptimizationConfig mem_;
    bool ditLoaded_ = false, vaeLoaded_ = false, connectorLoaded_ = false;
    ltx_connector_config connCfg_{};
};

}  // namespace ltx
