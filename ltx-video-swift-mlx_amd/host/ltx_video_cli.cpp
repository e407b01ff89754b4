// ltx_video_cli.cpp - `ltx-video` command line, same subcommands / flags / defaults / validation messages as the
// reference CLI (Sources/LTXVideoCLI/LTXVideoCLI.swift:21-449), on top of host/LTXVideo.hpp -> libltxhip.so.
//
// Differences forced by scope: the Gemma text encoder, prompt enhancement, audio, I2V, downloads and MP4
// export are not part of this path. Text conditioning comes from `--embeddings <file.safetensors>` (keys
// prompt_embeddings [1,S,3840], prompt_mask [1,S], optional negative_embeddings / negative_mask) - the CLI-level
// equivalent of the reference's PrecomputedEmbeddings hook - the VAE from `--vae-weights`, and frames are written as a
// raw float32 (F,H,W,3) file plus a JSON sidecar instead of an MP4.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "LTXVideo.hpp"

using namespace ltx;

namespace {

struct ValidationError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Args {
    std::string prompt, output = "output.mp4", model = "distilled", lora, modelsDir, gemmaPath, ltxWeights, image,
                negativePrompt, stgBlocks = "29", transformerQuant = "bf16", hfToken;
    int width = 512, height = 512, frames = 25;
    std::optional<int> steps;
    std::optional<float> guidance;
    std::optional<uint64_t> seed;
    float loraScale = 1.0f, imageCondNoise = 0.15f, guidanceRescale = 0.f, crossAttnScale = 1.f, geGamma = 0.f, stgScale = 0.f,
          audioGain = 1.f;
    bool twoStage = false, distilledLora = false, enhancePrompt = false, audio = false, debug = false, profile = false,
         dryRun = false;
    // additions of this build
    std::string embeddings, vaeWeights, upscalerWeights, distilledLoraPath, pngDir, hiddenStates, connectorWeights, imageTensor;
    std::string noiseRng = "native";
    int vaeTile = 0, vaeOverlap = 1;
    int numLayers = 0, numHeads = 0, captionChannels = 0;  // reduced architectures for tests (0 = reference default)
    std::vector<std::pair<std::string, int>> hipOptions;   // --hip-option name=value -> ltx_ctx_set_option (the library reads no environment)
};

[[noreturn]] void usage(int code) {
    std::cout << "OVERVIEW: LTX-2 video generation on MI355X (libltxhip)\n\n"
                 "USAGE: ltx-video <subcommand>\n\nSUBCOMMANDS:\n  generate    Generate a video from a text prompt\n"
                 "  download    (not available: no network on this path)\n  info (default)  Show version and model information\n\n"
                 "generate <prompt> [-o/--output] [-w/--width 512] [-h/--height 512] [-f/--frames 25] [-s/--steps] [-g/--guidance]\n"
                 "  [--seed] [-m/--model distilled|dev] [--lora] [--lora-scale] [--ltx-weights] [--negative-prompt]\n"
                 "  [--guidance-rescale] [--cross-attn-scale] [--ge-gamma] [--stg-scale] [--stg-blocks \"29\"]\n"
                 "  [--transformer-quant bf16|qint8|int4] [--two-stage] [--distilled-lora] [--profile] [--dry-run] [--debug]\n"
                 "  [--embeddings file] [--vae-weights file] [--upscaler-weights file] [--distilled-lora-path file]\n"
                 "  [--vae-tile N] [--vae-overlap N] [--png-dir dir] [--gemma-hidden-states file] [--connector-weights file]\n"
                 "  [--image-tensor file] [--noise-rng native|mlx] [--hip-option name=value]...\n"
                 "options     List the library's launcher switches (name, default, range, numerics flag) for --hip-option\n";
    std::exit(code);
}

Args parse_generate(int argc, char** argv, int start) {
    Args a;
    bool have_prompt = false;
    auto need = [&](int& i) -> std::string {
        if (i + 1 >= argc) throw ValidationError(std::string("Missing value for '") + argv[i] + "'");
        return argv[++i];
    };
    for (int i = start; i < argc; ++i) {
        const std::string k = argv[i];
        if (k == "-o" || k == "--output") a.output = need(i);
        else if (k == "-w" || k == "--width") a.width = std::stoi(need(i));
        else if (k == "-h" || k == "--height") a.height = std::stoi(need(i));
        else if (k == "-f" || k == "--frames") a.frames = std::stoi(need(i));
        else if (k == "-s" || k == "--steps") a.steps = std::stoi(need(i));
        else if (k == "-g" || k == "--guidance") a.guidance = std::stof(need(i));
        else if (k == "--seed") a.seed = std::stoull(need(i));
        else if (k == "-m" || k == "--model") a.model = need(i);
        else if (k == "--lora") a.lora = need(i);
        else if (k == "--lora-scale") a.loraScale = std::stof(need(i));
        else if (k == "--hf-token") a.hfToken = need(i);
        else if (k == "--models-dir") a.modelsDir = need(i);
        else if (k == "--gemma-path") a.gemmaPath = need(i);
        else if (k == "--ltx-weights") a.ltxWeights = need(i);
        else if (k == "--image") a.image = need(i);
        else if (k == "--image-cond-noise") a.imageCondNoise = std::stof(need(i));
        else if (k == "--negative-prompt") a.negativePrompt = need(i);
        else if (k == "--guidance-rescale") a.guidanceRescale = std::stof(need(i));
        else if (k == "--cross-attn-scale") a.crossAttnScale = std::stof(need(i));
        else if (k == "--ge-gamma") a.geGamma = std::stof(need(i));
        else if (k == "--stg-scale") a.stgScale = std::stof(need(i));
        else if (k == "--stg-blocks") a.stgBlocks = need(i);
        else if (k == "--transformer-quant") a.transformerQuant = need(i);
        else if (k == "--two-stage") a.twoStage = true;
        else if (k == "--distilled-lora") a.distilledLora = true;
        else if (k == "--enhance-prompt") a.enhancePrompt = true;
        else if (k == "--audio") a.audio = true;
        else if (k == "--audio-gain") a.audioGain = std::stof(need(i));
        else if (k == "--debug") a.debug = true;
        else if (k == "--profile") a.profile = true;
        else if (k == "--dry-run") a.dryRun = true;
        else if (k == "--embeddings") a.embeddings = need(i);
        else if (k == "--vae-weights") a.vaeWeights = need(i);
        else if (k == "--upscaler-weights") a.upscalerWeights = need(i);
        else if (k == "--distilled-lora-path") a.distilledLoraPath = need(i);
        else if (k == "--vae-tile") a.vaeTile = std::stoi(need(i));
        else if (k == "--png-dir") a.pngDir = need(i);
        else if (k == "--noise-rng") a.noiseRng = need(i);
        else if (k == "--hip-option") {
            const std::string kv = need(i);
            const size_t eq = kv.find('=');
            if (eq == std::string::npos || eq == 0 || eq + 1 >= kv.size()) throw ValidationError("--hip-option expects name=value, got '" + kv + "'");
            a.hipOptions.emplace_back(kv.substr(0, eq), std::stoi(kv.substr(eq + 1)));
        }
        else if (k == "--gemma-hidden-states") a.hiddenStates = need(i);
        else if (k == "--connector-weights") a.connectorWeights = need(i);
        else if (k == "--image-tensor") a.imageTensor = need(i);
        else if (k == "--vae-overlap") a.vaeOverlap = std::stoi(need(i));
        else if (k == "--num-layers") a.numLayers = std::stoi(need(i));
        else if (k == "--num-heads") a.numHeads = std::stoi(need(i));
        else if (k == "--caption-channels") a.captionChannels = std::stoi(need(i));
        else if (k == "--help") usage(0);
        else if (!k.empty() && k[0] == '-') throw ValidationError("Unknown option '" + k + "'");
        else if (!have_prompt) { a.prompt = k; have_prompt = true; }
        else throw ValidationError("Unexpected argument '" + k + "'");
    }
    if (!have_prompt) throw ValidationError("Missing expected argument '<prompt>'");
    return a;
}

std::vector<int> parse_blocks(const std::string& s) {
    std::vector<int> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        try { out.push_back(std::stoi(tok)); } catch (...) {}
    }
    return out;
}

PrecomputedEmbeddings read_embeddings(const std::string& path) {
    PrecomputedEmbeddings e;
    long shp[8] = {0};
    const int nd = ltx_st_info(path.c_str(), "prompt_embeddings", shp);
    if (nd < 2) throw LTXError(LTXError::textEncodingFailed, "cannot read prompt_embeddings from " + path + ": " + ltx_last_error(nullptr));
    e.S = int(shp[nd - 2]);
    long n = 1;
    for (int i = 0; i < nd; ++i) n *= shp[i];
    e.promptEmbeddings.resize(n);
    ltx_st_read(path.c_str(), "prompt_embeddings", 1, e.promptEmbeddings.data(), n);
    e.promptMask.assign(e.S, 1);
    if (ltx_st_info(path.c_str(), "prompt_mask", shp) > 0) ltx_st_read(path.c_str(), "prompt_mask", 2, e.promptMask.data(), e.S);
    if (ltx_st_info(path.c_str(), "negative_embeddings", shp) > 0) {
        e.nullEmbeddings.resize(n);
        ltx_st_read(path.c_str(), "negative_embeddings", 1, e.nullEmbeddings.data(), n);
        if (ltx_st_info(path.c_str(), "negative_mask", shp) > 0) {
            e.nullMask.resize(e.S);
            ltx_st_read(path.c_str(), "negative_mask", 2, e.nullMask.data(), e.S);
        }
    }
    return e;
}

int run_generate(const Args& a) {
    const std::vector<int> stg = parse_blocks(a.stgBlocks);
    for (const auto& kv : a.hipOptions)  // process-wide launcher switches (ltx-video options lists them); an unknown name or value is a usage error
        if (ltx_ctx_set_option(nullptr, kv.first.c_str(), kv.second) != LTX_OK)
            throw ValidationError("--hip-option " + kv.first + "=" + std::to_string(kv.second) + ": " + ltx_last_error(nullptr));
    // same banner as Generate.run (LTXVideoCLI.swift:138-166)
    std::cout << "LTX-2 Video Generation\n======================\n";
    std::cout << "Mode: " << (a.image.empty() ? "text-to-video" : "image-to-video") << "\n";
    std::cout << "Prompt: " << a.prompt << "\nOutput: " << a.output << "\nResolution: " << a.width << "x" << a.height
              << "\nFrames: " << a.frames << "\nModel: " << (a.distilledLora ? "dev" : a.model) << "\n";
    if (a.distilledLora) std::cout << "Distilled LoRA: enabled\n";
    if (a.seed) std::cout << "Seed: " << *a.seed << "\n";
    if (a.twoStage) std::cout << "Two-stage: enabled\n";
    if (a.transformerQuant != "bf16") std::cout << "Transformer quantization: " << a.transformerQuant << "\n";
    std::cout << "\n";
    // validation order and messages of LTXVideoCLI.swift:168-202
    if ((a.frames - 1) % 8 != 0)
        throw ValidationError("Frame count must be 8n+1 (e.g., 9, 17, 25, 33, ...). Got " + std::to_string(a.frames));
    if (a.width % 32 != 0 || a.height % 32 != 0)
        throw ValidationError("Width and height must be divisible by 32. Got " + std::to_string(a.width) + "x" + std::to_string(a.height));
    const auto quant = parseQuant(a.transformerQuant);
    if (!quant) throw ValidationError("Invalid transformer quantization: " + a.transformerQuant + ". Use: bf16, qint8, or int4");
    const std::string effModel = a.distilledLora ? "dev" : a.model;
    const auto model = parseModel(effModel);
    if (!model) throw ValidationError("Invalid model: " + a.model + ". Use: distilled or dev");
    if (a.noiseRng != "native" && a.noiseRng != "mlx") throw ValidationError("Invalid noise generator: " + a.noiseRng + ". Use: native or mlx");
    if (a.twoStage) {
        if (a.width % 64 != 0 || a.height % 64 != 0)
            throw ValidationError("Two-stage requires width and height divisible by 64. Got " + std::to_string(a.width) + "x" + std::to_string(a.height));
        std::cout << "Two-stage pipeline: " << a.width / 2 << "x" << a.height / 2 << " -> upscale 2x -> " << a.width << "x" << a.height << "\n";
    }
    if (a.distilledLora) std::cout << "Distilled LoRA: will fuse into dev model (8 steps, no CFG)\n";
    if (a.dryRun) {
        std::cout << "Validation passed (dry run mode)\n";
        return 0;
    }
    if (!a.image.empty() || a.audio || a.enhancePrompt)
        throw ValidationError("--image (file decoding) / --audio / --enhance-prompt are outside the MI355X hot path of this build; "
                              "pass the image as a float tensor with --image-tensor");
    if (a.ltxWeights.empty() || a.vaeWeights.empty() || (a.embeddings.empty() && a.hiddenStates.empty()))
        throw ValidationError("this build needs --ltx-weights, --vae-weights and --embeddings or --gemma-hidden-states (no downloads, no Gemma)");

    std::cout << "Creating pipeline...\n";
    MemoryOptimizationConfig mem = MemoryOptimizationConfig::defaultConfig();
    mem.vaeTemporalTileSize = a.vaeTile;
    mem.vaeTemporalTileOverlap = a.vaeOverlap;
    LTXPipeline pipeline(*model, LTXQuantizationConfig{*quant, 64}, mem);
    std::cout << "Pipeline created\nLoading models (this may take a while)...\n";
    ltx_transformer_config tc;
    ltx_transformer_config_default(&tc);
    if (a.numLayers) tc.num_layers = a.numLayers;
    if (a.numHeads) { tc.num_attention_heads = a.numHeads; tc.cross_attention_dim = a.numHeads * 128; }
    if (a.captionChannels) tc.caption_channels = a.captionChannels;
    pipeline.loadModels(a.ltxWeights, a.vaeWeights, &tc);
    std::cout << "Models loaded\n";
    if (a.distilledLora) {
        if (a.distilledLoraPath.empty()) throw ValidationError("--distilled-lora needs --distilled-lora-path (no downloads on this path)");
        std::cout << "Fusing distilled LoRA into transformer...\n";
        const int n = pipeline.fuseLoRA(a.distilledLoraPath, a.loraScale);
        std::cout << "  Fused " << n << " layers (scale=" << a.loraScale << ")\n";
    } else if (!a.lora.empty()) {
        // the reference's applyLoRA only counts layers and never changes weights (LoRAAdapter.swift:29-47)
        std::cout << "Applying LoRA from " << a.lora << "...\n  (applyLoRA does not modify weights in the reference; use --distilled-lora)\n";
    }
    LTXVideoGenerationConfig cfg;
    cfg.width = a.width; cfg.height = a.height; cfg.numFrames = a.frames;
    cfg.numSteps = a.distilledLora ? a.steps.value_or(8) : a.steps.value_or(defaultSteps(*model));
    cfg.cfgScale = a.distilledLora ? a.guidance.value_or(1.0f) : a.guidance.value_or(defaultGuidance(*model));
    cfg.seed = a.seed;
    cfg.mlxCompatibleNoise = a.noiseRng == "mlx";
    cfg.guidanceRescale = a.guidanceRescale; cfg.crossAttentionScale = a.crossAttnScale; cfg.geGamma = a.geGamma;
    cfg.stgScale = a.stgScale; cfg.stgBlocks = stg; cfg.twoStage = a.twoStage;
    PrecomputedEmbeddings emb;
    if (!a.hiddenStates.empty()) {
        // text encoding, connector part (LTXPipeline.swift:640-700): the 49 Gemma hidden states come from a file
        std::cout << "Encoding prompt from Gemma hidden states...\n";
        pipeline.loadConnector(a.connectorWeights.empty() ? a.ltxWeights : a.connectorWeights);
        long shp[8] = {0};
        const int nd = ltx_st_info(a.hiddenStates.c_str(), "hidden_states", shp);
        if (nd != 4) throw LTXError(LTXError::textEncodingFailed, "hidden_states must be [states][1][T][dim] in " + a.hiddenStates);
        const int T = int(shp[2]);
        std::vector<uint16_t> hs(size_t(shp[0]) * shp[1] * shp[2] * shp[3]);
        ltx_st_read(a.hiddenStates.c_str(), "hidden_states", 1, hs.data(), long(hs.size()));
        std::vector<int32_t> am(T, 1);
        if (ltx_st_info(a.hiddenStates.c_str(), "attention_mask", shp) > 0) ltx_st_read(a.hiddenStates.c_str(), "attention_mask", 2, am.data(), T);
        emb = pipeline.encodeFromHiddenStates(hs, am, T);
    } else {
        emb = read_embeddings(a.embeddings);
    }
    std::optional<ImageConditioning> image;
    if (!a.imageTensor.empty()) {
        // image-to-video: the resized image as a float tensor "pixels" [1][3][1][H][W] in [-1,1] -> VAE encoder -> frame-0 latent
        std::cout << "Encoding conditioning image...\n";
        long shp[8] = {0};
        if (ltx_st_info(a.imageTensor.c_str(), "pixels", shp) != 5 || shp[3] != a.height || shp[4] != a.width)
            throw LTXError(LTXError::invalidConfiguration, "pixels must be [1][3][1][height][width] in " + a.imageTensor);
        std::vector<float> px(size_t(3) * a.height * a.width);
        ltx_st_read(a.imageTensor.c_str(), "pixels", 0, px.data(), long(px.size()));
        pipeline.loadVAEEncoder(a.vaeWeights);
        image = pipeline.encodeImage(px, a.width, a.height);
        pipeline.unloadVAEEncoder();
        cfg.imageCondNoiseScale = a.imageCondNoise;
        if (a.imageCondNoise > 0) {  // one N(0,1) draw of the image latent's shape per step (LTXPipeline.swift:2226)
            int Fl, Hl, Wl;
            ltx_latent_shape(a.width, a.height, a.frames, &Fl, &Hl, &Wl);
            image->injectionNoise = generateNoise(size_t(cfg.numSteps) * 128 * Hl * Wl, a.seed.value_or(0) ^ 0x9e3779b9u);
        }
    }
    std::cout << "\nGenerating video...\n";
    auto progress = [](const GenerationProgress& p) {
        std::cout << "  Step " << p.currentStep + 1 << "/" << p.totalSteps << " (sigma=" << p.sigma << ")\n";
    };
    VideoGenerationResult r;
    if (a.twoStage) {
        if (a.upscalerWeights.empty()) throw ValidationError("--two-stage needs --upscaler-weights");
        pipeline.loadUpscaler(a.upscalerWeights);
        int F1, H1, W1, F2, H2, W2;
        ltx_latent_shape(a.width / 2, a.height / 2, a.frames, &F1, &H1, &W1);
        ltx_latent_shape(a.width, a.height, a.frames, &F2, &H2, &W2);
        const uint64_t sd = a.seed.value_or(0);
        r = pipeline.generateVideoTwoStage(cfg, emb, generateNoise(size_t(128) * F1 * H1 * W1, sd),
                                           generateNoise(size_t(128) * F2 * H2 * W2, sd + 0x5bd1e995u), progress);
    } else {
        r = pipeline.generateVideo(cfg, emb, {}, progress, a.profile, {}, image ? &*image : nullptr);
    }
    std::cout << "Generated " << r.numFrames << " frames (" << r.width << "x" << r.height << ") in " << r.generationTime << "s\n";
    std::ofstream f(a.output, std::ios::binary);
    f.write(reinterpret_cast<const char*>(r.frames.data()), std::streamsize(r.frames.size() * sizeof(float)));
    if (!a.pngDir.empty()) {  // tensorToImages' uint8 conversion (VideoExporter.swift:563-580), one PNG per frame
        std::vector<uint8_t> u8(r.frames.size());
        ltx_frames_to_u8(r.frames.data(), long(r.frames.size()), u8.data());
        const size_t fsz = size_t(r.height) * r.width * 3;
        for (int fi = 0; fi < r.numFrames; ++fi) {
            char name[64];
            snprintf(name, sizeof(name), "/frame_%04d.png", fi);
            if (ltx_write_png((a.pngDir + name).c_str(), u8.data() + fi * fsz, r.width, r.height) != 0)
                throw LTXError(LTXError::fileNotFound, a.pngDir + name);
        }
        std::cout << r.numFrames << " PNG frames written to " << a.pngDir << "\n";
    }
    std::ofstream j(a.output + ".json");
    j << "{\"frames\": " << r.numFrames << ", \"height\": " << r.height << ", \"width\": " << r.width
      << ", \"channels\": 3, \"dtype\": \"float32\", \"range\": [0, 1], \"seed\": " << r.seed << "}\n";
    std::cout << "Frames written to " << a.output << " (raw float32 F,H,W,3; see " << a.output << ".json)\n";
    if (a.profile && r.timings) {
        std::cout << "\nProfile:\n  denoise total: " << r.timings->totalDenoise() << "s over " << r.timings->denoiseSteps.size()
                  << " steps\n  VAE decode: " << r.timings->vaeDecode << "s\n";
        // the reference's per-step line, same format (LTXPipeline.swift:951), so that two --profile logs can be diffed
        for (const auto& d : r.timings->stepDiagnostics) {
            char line[256];
            snprintf(line, sizeof(line), "  Step %d: \xcf\x83=%.4f\xe2\x86\x92%.4f, vel mean=%.4f, std=%.4f, latent mean=%.4f, std=%.4f", d.step, d.sigma,
                     d.sigmaNext, d.velocityMean, d.velocityStd, d.latentMean, d.latentStd);
            std::cout << line << "\n";
        }
    }
    return 0;
}

// `ltx-video bench`: the headline measurement WITHOUT Python or PyTorch - this binary links libltxhip.so only and talks to it through
// the C ABI with host pointers (so the figure is PCIe-inclusive: per step 0.4 MB of latent in and out; the 7.9 MB text context crosses
// once, the library recognises unchanged context bytes). Synthetic weights of the reference architecture generated on the device, a
// seeded N(0,1) latent and context generated here. bench.py (HBM-resident inputs, roofline, CPU baseline) stays the number of record.
int run_bench(int argc, char** argv) {
    int width = 768, height = 512, frames = 25, steps = 20, warmup = 5, S = 1024, layers = 48, decodes = 3;
    for (int i = 2; i < argc; ++i) {
        const std::string k = argv[i];
        auto need = [&]() -> int { if (i + 1 >= argc) throw ValidationError("missing value for " + k); return std::stoi(argv[++i]); };
        if (k == "-w" || k == "--width") width = need();
        else if (k == "-h" || k == "--height") height = need();
        else if (k == "-f" || k == "--frames") frames = need();
        else if (k == "--steps") steps = need();
        else if (k == "--warmup") warmup = need();
        else if (k == "--text-keys") S = need();
        else if (k == "--num-layers") layers = need();
        else if (k == "--decodes") decodes = need();
        else throw ValidationError("Unknown bench option '" + k + "'");
    }
    ltx_ctx* ctx = nullptr;
    auto ck = [&](int st) { if (st != 0) throw LTXError(LTXError::generationFailed, ctx ? ltx_last_error(ctx) : "ltx_ctx_create failed"); };
    ck(ltx_ctx_create(0, &ctx));
    ltx_transformer_config cfg;
    ltx_transformer_config_default(&cfg);
    cfg.num_layers = layers;
    ck(ltx_dit_init_synthetic(ctx, &cfg, 1234));
    int F, H, W;
    if (ltx_latent_shape(width, height, frames, &F, &H, &W) != 0) throw ValidationError("bad width / height / frames");
    const size_t n = size_t(128) * F * H * W;
    std::vector<float> latent = generateNoise(n, 42);
    std::vector<float> cf = generateNoise(size_t(S) * cfg.caption_channels, 43);
    std::vector<uint16_t> context(cf.size());
    for (size_t i = 0; i < cf.size(); ++i) {  // f32 -> bf16 bits, round to nearest even
        uint32_t u;
        std::memcpy(&u, &cf[i], 4);
        context[i] = uint16_t((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    std::vector<int32_t> mask(S, 1);
    float sig[16];
    const int ns = ltx_sigmas(1, 8, F * H * W, sig, 16);
    if (ns != 9) throw LTXError(LTXError::generationFailed, "ltx_sigmas");
    // as a host uses the entry point: ONE call runs the steps of a generation (the distilled schedule has eight); `count` steps are
    // issued as calls of at most eight (patchify + forward + unpatchify + Euler per step inside the library)
    auto run_steps = [&](int count, uint64_t seed) {
        for (int done = 0; done < count;) {
            const int k = std::min(8, count - done);
            latent = generateNoise(n, seed + uint64_t(done));
            for (float& v : latent) v *= sig[0];
            ck(ltx_denoise(ctx, latent.data(), F, H, W, sig, k + 1, context.data(), mask.data(), S, nullptr, nullptr, nullptr));
            done += k;
        }
    };
    run_steps(warmup, 100);
    const auto t0 = std::chrono::steady_clock::now();
    run_steps(steps, 7000);
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double vae_ms = 0;
    if (decodes > 0) {
        ck(ltx_vae_init_synthetic(ctx, 77, 0));
        std::vector<float> vlat = generateNoise(n, 45);
        std::vector<float> out(size_t(8 * (F - 1) + 1) * H * 32 * W * 32 * 3);
        int nf = 0;
        ck(ltx_vae_decode(ctx, vlat.data(), F, H, W, 0, 0.f, nullptr, 0, 1, out.data(), long(out.size()), &nf));
        const auto v0 = std::chrono::steady_clock::now();
        for (int i = 0; i < decodes; ++i) ck(ltx_vae_decode(ctx, vlat.data(), F, H, W, 0, 0.f, nullptr, 0, 1, out.data(), long(out.size()), &nf));
        vae_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - v0).count() / decodes;
    }
    char line[1024];
    snprintf(line, sizeof(line),
             "{\"metric\": \"DiT denoise steps/sec, %dx%dx%d distilled (ltx-video bench: C ABI with host pointers, no Python, no PyTorch)\", "
             "\"value\": %.4f, \"unit\": \"steps/s\", \"ms_per_step\": %.3f, \"steps\": %d, \"warmup\": %d, \"tokens\": %d, \"text_keys\": %d, "
             "\"layers\": %d, \"dtype\": \"bf16\", \"data\": \"synthetic\", \"pcie_inclusive\": true, \"vae_decode_ms_host_pointers\": %.2f, "
             "\"build\": \"%s\"}",
             width, height, frames, steps / el, 1e3 * el / steps, steps, warmup, F * H * W, S, layers, vae_ms, ltx_build_info());
    std::cout << line << "\n";
    ltx_ctx_destroy(ctx);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        const std::string sub = argc > 1 ? argv[1] : "info";
        if (sub == "--version") {
            std::cout << version << "\n";
            return 0;
        }
        if (sub == "--help" || sub == "-h" || sub == "help") usage(0);
        if (sub == "info") {
            std::cout << "LTX-Video (MI355X / libltxhip) version " << ltx_version() << "\n"
                      << "Models: distilled (8 steps, CFG 1.0, " << unifiedWeightsFilename(LTXModel::distilled) << "), dev (40 steps, CFG 4.0, "
                      << unifiedWeightsFilename(LTXModel::dev) << ")\nConstraints: width/height % 32 == 0 (two-stage: % 64), frames = 8n+1\n";
            return 0;
        }
        if (sub == "options") {  // ltx_option_info: the one table behind ltx_ctx_set_option; no GPU needed
            const int n = ltx_option_info(-1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            std::cout << "libltxhip ABI revision " << ltx_abi_version() << ", " << n << " options (numerics: * = moves results by rounding / accumulation order)\n";
            for (int i = 0; i < n; ++i) {
                const char *name = nullptr, *doc = nullptr;
                int d = 0, lo = 0, hi = 0, num = 0;
                ltx_option_info(i, &name, &d, &lo, &hi, &num, &doc);
                std::cout << "  " << (num ? "* " : "  ") << name << " = " << d << "  [" << lo << ", " << hi << "]  " << doc << "\n";
            }
            return 0;
        }
        if (sub == "download") throw ValidationError("download is not available: this path has no network access");
        if (sub == "generate") return run_generate(parse_generate(argc, argv, 2));
        if (sub == "bench") return run_bench(argc, argv);
        throw ValidationError("Unknown subcommand '" + sub + "'");
    } catch (const ValidationError& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 64;  // EX_USAGE, as swift-argument-parser
    } catch (const LTXError& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
}
