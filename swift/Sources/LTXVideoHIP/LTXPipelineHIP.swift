// LTXPipelineHIP.swift - SOURCE-ONLY Swift face of libltxhip.so (never compiled here: no Swift toolchain in the image).
// Mirrors the seams of the reference's `actor LTXPipeline` (Pipeline/LTXPipeline.swift:117-1076, 1902-2125, 2420-2741) that
// INTEGRATION.md lists; names and error cases follow the reference (LTXVideo.swift:66-141, LTXConfig.swift:216-361).
import CLTXHIP
import Foundation

public enum LTXError: Error, CustomStringConvertible {
    case modelNotLoaded(String), invalidConfiguration(String), insufficientMemory, weightLoadingFailed(String)
    case generationFailed(String), generationCancelled, fileNotFound(String), invalidLoRA(String), hip(String)

    public var description: String {
        switch self {
        case .modelNotLoaded(let m): return "Model component not loaded: \(m)"
        case .invalidConfiguration(let m): return "Invalid configuration: \(m)"
        case .insufficientMemory: return "Insufficient memory"
        case .weightLoadingFailed(let m): return "Failed to load weights: \(m)"
        case .generationFailed(let m): return "Generation failed: \(m)"
        case .generationCancelled: return "Generation was cancelled"
        case .fileNotFound(let m): return "File not found: \(m)"
        case .invalidLoRA(let m): return "Invalid LoRA: \(m)"
        case .hip(let m): return m
        }
    }
}

public struct GenerationProgress { public let currentStep: Int; public let totalSteps: Int; public let sigma: Float }

public struct LTXVideoGenerationConfig {
    public var width = 704, height = 480, numFrames = 121, numSteps = 8
    public var cfgScale: Float = 1.0, guidanceRescale: Float = 0.0, stgScale: Float = 0.0, geGamma: Float = 0.0
    public var stgBlocks: [Int32] = [29]
    public var imageCondNoiseScale: Float = 0.0
    public var vaeTemporalTileSize = 0, vaeTemporalTileOverlap = 1
    /// 0 = this GPU evaluates everything; 1 = CFG pair over the group (LTX_SHARD_CFG); 2 = one sample's tokens over the group
    /// (LTX_SHARD_SEQUENCE). Needs `joinGroup` first; every rank of the group calls generateVideo with the same arguments.
    public var shard = 0
    public init() {}
    public func validate() throws {
        var msg = [CChar](repeating: 0, count: 256)
        if ltx_validate_generation_config(Int32(width), Int32(height), Int32(numFrames), Int32(numSteps), cfgScale, 0, &msg, 256) != 0 {
            throw LTXError.invalidConfiguration(String(cString: msg))
        }
    }
}

/// Text conditioning as the loop takes it (the reference's `PrecomputedEmbeddings`, LTXPipeline.swift:571-584):
/// bf16 bit patterns [nb][S][3840] with nb = 2 ([negative, positive]) when CFG is on, and the [nb][S] mask.
public struct TextConditioning { public var embeddings: [UInt16]; public var mask: [Int32]; public var tokens: Int }

public final class LTXPipelineHIP {
    private var ctx: OpaquePointer?

    public init(device: Int32 = 0) throws {
        var h: OpaquePointer?
        let rc = ltx_ctx_create(device, &h)
        guard rc == 0 else { throw LTXError.hip(String(cString: ltx_last_error(nil))) }
        ctx = h
    }
    deinit { if let c = ctx { ltx_ctx_destroy(c) } }

    private func check(_ rc: Int32) throws {
        guard rc != 0 else { return }
        let msg = String(cString: ltx_last_error(ctx))
        switch rc {
        case 1: throw LTXError.modelNotLoaded(msg)
        case 2: throw LTXError.invalidConfiguration(msg)
        case 3: throw LTXError.insufficientMemory
        case 4: throw LTXError.weightLoadingFailed(msg)
        case 6: throw LTXError.generationCancelled
        case 9: throw LTXError.fileNotFound(msg)
        case 10: throw LTXError.invalidLoRA(msg)
        default: throw LTXError.generationFailed(msg)
        }
    }

    // loadModels (LTXPipeline.swift:217-361): transformer + VAE decoder (+ connector / VAE encoder from the same files)
    public func loadModels(ltxWeights: String, vaeWeights: String, quantBits: Int32 = 16) throws {
        try check(ltx_dit_load(ctx, ltxWeights, nil, quantBits, 64))
        try check(ltx_vae_load(ctx, vaeWeights, nil))
    }
    /// Launcher switches of libltxhip (`ltx_ctx_set_option`; the library never reads the environment; process-wide).
    /// No reference counterpart. `setOption("qk_f32", 1)` + `setOption("split_f32", 1)`: the reference's rounding points exactly.
    public func setOption(_ key: String, _ value: Int32) throws { try check(ltx_ctx_set_option(ctx, key, value)) }
    public func option(_ key: String) throws -> Int32 {
        var v: Int32 = 0
        try check(ltx_ctx_get_option(ctx, key, &v))
        return v
    }
    public func loadConnector(from unifiedWeights: String) throws { try check(ltx_connector_load(ctx, unifiedWeights, nil)) }
    public func loadVAEEncoder(from vaeWeights: String) throws { try check(ltx_vae_encoder_load(ctx, vaeWeights, 0)) }
    /// One process per GPU: the RCCL communicator of this pipeline's group (include/ltxhip.h, "Multi-GPU"). `uniqueId` = the 128
    /// bytes `LTXPipelineHIP.makeGroupId()` returned on the group's first rank, carried to the others by the host.
    public static func makeGroupId() throws -> [UInt8] {
        var id = [UInt8](repeating: 0, count: Int(LTX_DIST_ID_BYTES))
        guard ltx_dist_unique_id(&id) == 0 else { throw LTXError.hip(String(cString: ltx_last_error(nil))) }
        return id
    }

    public func joinGroup(rank: Int, size: Int, uniqueId: [UInt8]) throws {
        try check(ltx_dist_init(ctx, Int32(rank), Int32(size), uniqueId))
    }

    public func fuseLoRA(from path: String, scale: Float = 1.0) throws -> Int {
        var n: Int32 = 0
        try check(ltx_dit_fuse_lora(ctx, path, scale, &n))
        return Int(n)
    }

    /// encodeFromHiddenStates (LTXTextEncoder.swift:574-643): hidden = 49 x [1][T][3840] bf16 bit patterns, concatenated.
    public func encodeFromHiddenStates(hidden: [UInt16], attentionMask: [Int32], tokens: Int) throws -> TextConditioning {
        var out = [UInt16](repeating: 0, count: tokens * 3840)
        var mask = [Int32](repeating: 0, count: tokens)
        try check(ltx_connector_encode(ctx, hidden, attentionMask, 1, Int32(tokens), 0, &out, &mask))
        return TextConditioning(embeddings: out, mask: mask, tokens: tokens)
    }

    /// encodeImage (LTXPipeline.swift:1902-1932) without the file I/O: pixels [1][3][1][H][W] in [-1,1] -> normalised latent.
    public func encodeImage(pixels: [Float], width: Int, height: Int) throws -> [Float] {
        var lat = [Float](repeating: 0, count: 128 * (height / 32) * (width / 32))
        try check(ltx_vae_encode(ctx, pixels, 1, Int32(height), Int32(width), 1, &lat))
        return lat
    }

    /// The denoise loop + decode of generateVideo / generateVideoFromImage (LTXPipeline.swift:586-1046, 1953-2125).
    /// `noise` is the initial N(0,1) latent [1][128][F'][H'][W'] (draw it with MLXRandom so `seed` keeps its meaning);
    /// `imageLatent` (+ optional per-step `injectionNoise`) switches on image-to-video conditioning.
    public func generateVideo(config: LTXVideoGenerationConfig, text: TextConditioning, noise: [Float], distilled: Bool = true,
                              imageLatent: [Float]? = nil, injectionNoise: [Float]? = nil, vaeNoise: [Float]? = nil,
                              onProgress: ((GenerationProgress) -> Void)? = nil) throws -> (frames: [Float], count: Int) {
        try config.validate()
        var f: Int32 = 0, h: Int32 = 0, w: Int32 = 0
        try check(ltx_latent_shape(Int32(config.width), Int32(config.height), Int32(config.numFrames), &f, &h, &w))
        var sig = [Float](repeating: 0, count: 128)
        let ns = ltx_sigmas(distilled ? 1 : 0, Int32(config.numSteps), f * h * w, &sig, 128)
        guard ns >= 2, config.numSteps <= Int(ns) - 1 else { throw LTXError.invalidConfiguration("numSteps exceeds the sigma schedule") }
        var latent = noise.map { $0 * sig[0] }  // LTXPipeline.swift:793
        var stg = config.stgBlocks
        final class Box { var cb: ((GenerationProgress) -> Void)?; init(_ c: ((GenerationProgress) -> Void)?) { cb = c } }
        let box = Box(onProgress)
        let thunk: ltx_progress_cb = { step, total, sigma, user in
            Unmanaged<Box>.fromOpaque(user!).takeUnretainedValue().cb?(GenerationProgress(currentStep: Int(step), totalSteps: Int(total), sigma: sigma))
        }
        try stg.withUnsafeBufferPointer { stgPtr in
            try (imageLatent ?? []).withUnsafeBufferPointer { img in
                try (injectionNoise ?? []).withUnsafeBufferPointer { inj in
                    var opt = ltx_denoise_options(struct_size: UInt32(MemoryLayout<ltx_denoise_options>.size), cfg_scale: config.cfgScale, guidance_rescale: config.guidanceRescale, stg_scale: config.stgScale,
                                                  stg_blocks: stgPtr.baseAddress, n_stg_blocks: Int32(stgPtr.count), ge_gamma: config.geGamma,
                                                  cond_latent: imageLatent == nil ? nil : img.baseAddress,
                                                  image_cond_noise_scale: config.imageCondNoiseScale,
                                                  cond_noise: injectionNoise == nil ? nil : inj.baseAddress,
                                                  shard: Int32(config.shard),
                                                  step_stats: nil)  // per-step --profile diagnostics: pass a [Float](4 * numSteps) to get them
                    try check(ltx_denoise(ctx, &latent, f, h, w, sig, Int32(config.numSteps + 1), text.embeddings, text.mask,
                                          Int32(text.tokens), &opt, thunk, Unmanaged.passUnretained(box).toOpaque()))
                }
            }
        }
        let outFrames = 8 * (Int(f) - 1) + 1
        var frames = [Float](repeating: 0, count: outFrames * config.height * config.width * 3)
        var n: Int32 = 0
        let useTs: Int32 = ltx_vae_timestep_conditioning(ctx) == 1 ? 1 : 0
        try check(ltx_vae_decode(ctx, latent, f, h, w, useTs, 0.05, vaeNoise, Int32(config.vaeTemporalTileSize),
                                 Int32(config.vaeTemporalTileOverlap), &frames, frames.count, &n))
        return (frames, Int(n))
    }
}
