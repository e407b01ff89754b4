// swift-tools-version:5.9
// SOURCE-ONLY (SURVEY 8(f) item 2): no Swift toolchain exists in the build image, so this package has never been compiled.
// It is the reference-side binding of include/ltxhip.h: a system-library target for libltxhip.so plus a thin `LTXPipelineHIP`
// that keeps the reference's public names (LTXVideoGenerationConfig, LTXError, GenerationProgress) and forwards the hot path -
// denoise loop, VAE decode/encode, connector - to the MI355X library. See INTEGRATION.md for the seam-by-seam mapping.
import PackageDescription

let package = Package(
    name: "LTXVideoHIP",
    products: [.library(name: "LTXVideoHIP", targets: ["LTXVideoHIP"])],
    targets: [
        // expects ltxhip.h on the header search path and libltxhip.so on the linker path, e.g.
        //   swift build -Xcc -I<repo>/include -Xlinker -L<repo>/ltx-video-swift-mlx_amd/csrc/build
        .systemLibrary(name: "CLTXHIP", path: "Sources/CLTXHIP"),
        .target(name: "LTXVideoHIP", dependencies: ["CLTXHIP"]),
    ]
)
