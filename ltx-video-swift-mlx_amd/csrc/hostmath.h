// hostmath.h - pure host logic of the path: shapes, validation, sigma schedules, RoPE tables, VAE tiling plan and
// safetensors key mapping. No HIP calls here; everything is exported through the C ABI so the CPU test-suite can
// pin it against hand-derived known-answer vectors.
#pragma once
#include <string>
#include <vector>

struct TransformerConfig {
    // defaults = LTXTransformerConfig (reference LTXConfig.swift:83-177)
    int num_layers = 48;
    int num_heads = 32;
    int head_dim = 128;
    int in_channels = 128;
    int out_channels = 128;
    int cross_attention_dim = 4096;
    int caption_channels = 3840;
    float rope_theta = 10000.0f;
    int max_pos[3] = {20, 2048, 2048};
    float timestep_scale_multiplier = 1000.0f;
    float norm_eps = 1e-6f;
    int inner_dim() const { return num_heads * head_dim; }
};

// LTXVideoGenerationConfig.validate (LTXConfig.swift:310-353) + two-stage %64 rule (LTXPipeline.swift:2443).
// Returns 0 when valid, else an LTXS_* code; msg receives the reference's error text.
int validate_generation_config(int width, int height, int num_frames, int num_steps, float cfg_scale, int two_stage,
                               std::string* msg);

// F' = (F-1)/8+1, H' = H/32, W' = W/32 (VideoLatentShape.swift:34-41, LTXConfig.swift:356-361)
void latent_shape(int width, int height, int num_frames, int* F, int* H, int* W);

// LTXScheduler.setTimesteps (LTXScheduler.swift:74-182) in f32 scalar arithmetic.
std::vector<float> compute_sigmas(bool distilled, int num_steps, int token_count /* <=0: none */);
extern const float kDistilledSigmas[9];
extern const float kStage2DistilledSigmas[4];

// createPositionGrid + precomputeFreqsCisDoublePrecision (LTXRoPE.swift:552-610, 375-527), split RoPE.
// cos/sin are written as [T][D/2] f32 (token-major; head h owns columns h*64..h*64+63) - the same values as the
// reference's [B,H,T,64] tensors, batch-independent.
void rope_tables(const TransformerConfig& cfg, int F, int H, int W, float fps, std::vector<float>* cos_out,
                 std::vector<float>* sin_out);

// decodeWithTemporalTiling's chunk walk (VideoDecoder.swift:517-548) and output frame count (:561-592).
struct TilePlan {
    std::vector<int> start, end;  // latent-frame ranges [start,end)
    int out_frames = 0;
};
TilePlan vae_tile_plan(int latent_frames, int tile, int overlap);

// mapTransformerKey / loadTransformerWeights filters (ModelDownloader.swift:605-639,756-803).
// `file_key` is the raw safetensors key. Returns false when the key is skipped.
bool map_transformer_file_key(const std::string& file_key, std::string* module_key);
// mapVAEWeights (ModelDownloader.swift:808-899). Accepts keys with or without the unified file's "vae." prefix.
bool map_vae_file_key(const std::string& file_key, std::string* module_key);
// LoRA key -> module weight key (LoRALoader.swift:209-243)
bool map_lora_key(const std::string& lora_base, std::string* module_weight_key);
// VAE encoder tensors of the VAE file (mapVAEEncoderWeights, ModelDownloader.swift:1222-1283); false for everything else
bool map_vae_encoder_file_key(const std::string& file_key, std::string* module_key);
// text-embedding connector: file key -> module key of VideoGemmaTextEncoderModel
// (ModelDownloader.swift:911-968 after the unified-file prefix strip of :1353-1399); false = dropped
bool map_text_encoder_file_key(const std::string& file_key, std::string* module_key);
// connector RoPE: integer positions 0..T-1 on one axis, split type, f64 math (LTXTextEncoder.swift:482-497,
// LTXRoPE.swift:375-490). cos/sin [T][dim/2] f32 (no padding slots: one axis -> dim/2 frequencies).
void rope_tables_1d(int T, int dim, double theta, int max_pos, std::vector<float>* cos_out, std::vector<float>* sin_out);

// MLX-compatible noise (R2: generateNoise = MLXRandom.seed(seed); MLXRandom.normal(shape, f32),
// LatentUtils.swift:69-83). The arithmetic is mlx-swift 0.30.6's (absent from the reference tree), restated from its
// published algorithm: threefry2x32-20 counter hash; key(seed) = {seed >> 32, seed & 0xffffffff}; every keyless draw first
// splits the global key (bits of shape [2][2]: row 0 stays global, row 1 is the draw's key); bits(n words) =
// out[i], out[i + ceil(n/2)] = hash(key, {i, i + ceil(n/2)}); uniform = min(bits / float(UINT32_MAX), nextafter(1, 0)) mapped
// to [nextafter(-1, 0), 1); normal = sqrt(2) * erfinv(u) with the two-branch 9/10-term polynomial. The hash is pinned by the
// Random123 known-answer vectors; the draw pipeline is NOT verified against an MLX run (none is possible here).
void threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);
void mlx_random_bits(const uint32_t key[2], long n_words, uint32_t* out);
// the `draw_index`-th keyless MLXRandom.normal(shape) call after MLXRandom.seed(seed) (0 = the first), n f32 values
void mlx_random_normal(uint64_t seed, int draw_index, long n, float* out);

// frame export (VideoExporter.swift:563-580)
void frames_to_u8(const float* frames, long n, uint8_t* out);
bool write_png_rgb8(const char* path, const uint8_t* rgb, int width, int height);
