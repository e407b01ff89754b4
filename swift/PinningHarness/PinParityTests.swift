// PinParityTests.swift - the REFERENCE-side half of the pinning kit (tools/make_pinning_case.py; SURVEY 8(c) item 4).
// SOURCE ONLY: never compiled here (no Swift toolchain, no MLX in the build image). It is written against the reference's public
// surface as of VincentGourbin/ltx-video-swift-mlx (LTXPipeline.swift:189-232 init / loadModels, :571-584 PrecomputedEmbeddings,
// :586-594 generateVideo(..., precomputedEmbeddings:, profile:), :945-951 the per-step diagnostics) and is meant to be dropped into the
// reference checkout as Tests/LTXVideoTests/PinParityTests.swift by whoever has it running on a Mac.
//
// What it does: runs ONE generation of the seed-defined case written by
//     python tools/make_pinning_case.py --out /path/to/pin --layers 48            (numpy only; also runs on the Mac)
// through the reference with `profile: true`, so that its debug log carries, per step,
//     "  Step i: σ=a→b, vel mean=…, std=…, latent mean=…, std=…"
// and compares those lines with /path/to/pin/expected_oracle.txt (this repo's CPU restatement) - or with expected_hip.txt (the MI355X
// path). Agreement to the fourth decimal within the tolerance of the bf16 precision contract (DESIGN.md section 2) pins the oracle, and
// with it every parity number of this repo, to the reference itself. Until someone runs this, parity stays "unpinned".
//
// One change to the reference is needed, because generateVideo draws its initial noise from MLX's global generator
// (LTXPipeline.swift:755 -> LatentUtils.swift:69-83) and has no parameter for it. In LatentUtils.generateNoise, before the draw:
//
//     if let p = ProcessInfo.processInfo.environment["LTX_PIN_CASE"],
//        let t = try? MLX.loadArrays(url: URL(fileURLWithPath: p)), let n = t["noise"], n.shape == shape.shape { return n }
//
// (the MLX-compatible generator of this repo, `--noise-rng mlx`, would make even that unnecessary once IT is pinned - the same run can
// do that: print `MLXRandom.normal([8])` after `MLXRandom.seed(42)` and compare with `ltx_mlx_random_normal(42, 0, out, 8)`.)
import Foundation
import MLX
import XCTest

@testable import LTXVideo

final class PinParityTests: XCTestCase {
    /// LTX_PIN_DIR = the directory make_pinning_case.py wrote; LTX_PIN_GEMMA / LTX_PIN_TOKENIZER = local Gemma paths (the reference
    /// insists on a loaded text encoder even when the embeddings are precomputed: LTXPipeline.swift:603-608).
    func testDenoiseDiagnosticsMatchThePinningCase() async throws {
        let env = ProcessInfo.processInfo.environment
        guard let dir = env["LTX_PIN_DIR"] else { throw XCTSkip("set LTX_PIN_DIR to the pinning case directory") }
        let caseFile = URL(fileURLWithPath: dir).appendingPathComponent("case.safetensors")
        let tensors = try MLX.loadArrays(url: caseFile)
        let emb = tensors["prompt_embeddings"]!.asType(.bfloat16)          // [1, S, 3840]; the values are bf16-representable
        let mask = tensors["prompt_mask"]!                                  // [1, S] int32, all ones
        setenv("LTX_PIN_CASE", caseFile.path, 1)                            // read by the patched generateNoise (see the header)

        let pipeline = LTXPipeline(model: .distilled)                      // distilled 8-step schedule = the case's sigmas
        try await pipeline.loadModels(
            gemmaModelPath: env["LTX_PIN_GEMMA"], tokenizerPath: env["LTX_PIN_TOKENIZER"],
            ltxWeightsPath: URL(fileURLWithPath: dir).appendingPathComponent("ltx_transformer.safetensors").path)

        // the case's meta data (width / height / frames / seed) sits in expected_oracle.json
        let meta = try JSONSerialization.jsonObject(
            with: Data(contentsOf: URL(fileURLWithPath: dir).appendingPathComponent("expected_oracle.json"))) as! [String: Any]
        var config = LTXVideoGenerationConfig(
            width: meta["width"] as! Int, height: meta["height"] as! Int, numFrames: meta["frames"] as! Int, numSteps: 8, cfgScale: 1.0)
        config.seed = UInt64(meta["seed"] as! Int)

        // the diagnostics go through LTXDebug.log = print("[LTX] ...") on stdout (LTXVideo.swift:171-176): capture stdout in a file
        LTXDebug.enableDebugMode()
        let capture = FileManager.default.temporaryDirectory.appendingPathComponent("ltx_pin_stdout.txt")
        FileManager.default.createFile(atPath: capture.path, contents: nil)
        let saved = dup(STDOUT_FILENO)
        let fh = try FileHandle(forWritingTo: capture)
        fflush(stdout)
        dup2(fh.fileDescriptor, STDOUT_FILENO)
        defer { fflush(stdout); dup2(saved, STDOUT_FILENO); close(saved) }
        _ = try await pipeline.generateVideo(
            prompt: "pinning case", config: config,
            precomputedEmbeddings: .init(promptEmbeddings: emb, promptMask: mask), profile: true)
        fflush(stdout)
        dup2(saved, STDOUT_FILENO)

        let log = try String(contentsOf: capture).split(separator: "\n").map(String.init)
        let got = log.filter { $0.contains("Step ") && $0.contains("vel mean=") }
        let want = try String(contentsOf: URL(fileURLWithPath: dir).appendingPathComponent("expected_oracle.txt"))
            .split(separator: "\n").filter { $0.contains("vel mean=") }.map(String.init)
        XCTAssertEqual(got.count, want.count, "one diagnostics line per step")
        for (g, w) in zip(got, want) {
            // compare the six numbers of a line; the oracle and the MI355X path differ by < 1e-3 on each of them
            let a = numbers(in: g), b = numbers(in: w)
            XCTAssertEqual(a.count, b.count)
            for (x, y) in zip(a, b) { XCTAssertEqual(x, y, accuracy: 2e-3, "\(g)  vs  \(w)") }
        }
    }

    private func numbers(in line: String) -> [Double] {
        let re = try! NSRegularExpression(pattern: "-?[0-9]+\\.[0-9]+")
        return re.matches(in: line, range: NSRange(line.startIndex..., in: line)).map { Double((line as NSString).substring(with: $0.range))! }
    }
}
