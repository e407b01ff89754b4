// gemm.h - launch interface of the bf16 MFMA GEMM family (gemm.hip).
//
// C[M,N] = A[M,K] * B[N,K]^T with both operands K-contiguous (activations [rows,K], weights [out,in] exactly as
// the reference's Linear stores them: LTXAttention.swift:152-157, "Linear weights are [out,in]" SURVEY R20).
// The A operand can also be an implicit im2col view of a channels-last VAE feature map (conv3d mode), which is
// how the 27-tap Conv3dFull of the reference (VideoConvolution.swift:238-347) is realised without materialising
// padded tensors.
#pragma once
#include "common.h"

enum { LTX_ACT_NONE = 0, LTX_ACT_GELU_TANH = 1, LTX_ACT_SILU = 2 };

struct GemmEpilogue {
    float* out_f32 = nullptr;     // [M, ld_f32]
    long ld_f32 = 0;
    bf16_t* out_bf16 = nullptr;   // [M, ld_bf16]
    long ld_bf16 = 0;
    // transposed output: out_bf16_t[n * ld_bf16_t + m] = bf16(acc + bias_n[n]) and nothing else (no activation, residual or other
    // output). Stored straight from the accumulator layout - a lane holds four consecutive rows m of one column n, i.e. 8 contiguous
    // bytes of the transposed matrix. Used for V^T = (X.Wv^T)^T with the tokens as rows (the attention kernels take V^T).
    bf16_t* out_bf16_t = nullptr;  // [N, ld_bf16_t], ld_bf16_t % 4 == 0
    long ld_bf16_t = 0;
    const float* bias_n = nullptr;  // per output column (the usual Linear bias)
    const float* bias_m = nullptr;  // per output row (used when operands are swapped to emit C^T)
    // gate_scalar FIRST of four 4-byte fields: hipcc reads it with a 16-byte vector load (the splat to four lanes), and when those 16
    // bytes also hold a pointer the kernels' by-value argument copy is not promoted to registers - every epilogue read of the
    // neighbouring fields then becomes a scratch load with a vmcnt(0) behind it
    float gate_scalar = 1.0f;
    int act = LTX_ACT_NONE;
    int round_bf16 = 0;  // round (acc+bias, activated) through bf16 before any f32 store/residual use
    // residual mode: out_f32[m][n] = resid_src[m][n] + gate(m,n) * (acc + bias); gate(m,n) =
    //   gate ? gate[(m / rows_per_batch) * gate_bstride + n] : gate_scalar
    // (reference residualGate: LTXTransformerBlock.swift:86-92; cross-attn scale :211-214)
    int resid = 0;
    const float* gate = nullptr;
    long gate_bstride = 0;
    int rows_per_batch = 1;
    const int32_t* gate_rowmap = nullptr;  // optional: gate row of output row m = gate_rowmap[m] (per-token timestep groups, I2V)
    const float* resid_src = nullptr;  // defaults to out_f32 (in-place) when null
    long ld_resid = 0;
    // depth-to-space store. 1: VAE upsampler (2,2,2) with first-frame drop and tiled D2S residual
    // (VideoDecoder.swift:201-251); 2: per-frame pixel shuffle (1,2,2) of the latent upscaler
    // (SpatialUpscaler.swift:116-131). Conv output channels are stored sub-position-major (permuted at load).
    // 3: the VAE decoder's final un-patchify (VideoDecoder.swift:257-275) as the store: 48 output channels, permuted at load to
    // n' = b*12 + a*3 + c (original (c*4 + a)*4 + b), go to frames (F, 4H, 4W, 3) at pixel (4y + b, 4x + a), colour c: 16-byte
    // chunks of one pixel row. With clip01 the stored value is clamp((v + 1) / 2, 0, 1) (VideoDecoder.swift:501-507).
    int d2s = 0;
    int clip01 = 0;
    // fused PixelNorm + SiLU second output of a conv launch whose ONE column tile holds every channel (N == the tile's 128 columns):
    // pn_out[m][n] = bf16(silu(v[m][n] / sqrt(mean_n(v[m][:]^2) + 1e-8) * pn_scale[n] + pn_shift[n])), v = the value the f32 output
    // gets (after bias and residual). The VAE's 128-channel stage: the next conv's input without a pass over the f32 stream
    // (vaePixelNorm + modulation + SiLU: VideoDecoder.swift:29-32,93-113). out_f32 may be null (the res-block's inner conv).
    bf16_t* pn_out = nullptr;
    long ld_pn = 0;
    const float* pn_scale = nullptr;  // [N], the "1 + scale" row of the block's modulation table
    const float* pn_shift = nullptr;  // [N]
};

// Geometry of an implicit-GEMM conv3d A operand: x is [F][H][W][C] bf16 (channels-last), 3x3x3 taps,
// reflect padding in H/W and replicate padding in T (non-causal) or causal (first frame repeated twice).
struct Conv3dGeom {
    int F = 0, H = 0, W = 0, C = 0;
    int causal = 0;
    int pad_mode = 0;  // 0: reflect H/W + replicate T (VAE Conv3dFull); 1: zeros in every dim (MLXNN.Conv3d/Conv2d  /* 3: zeros in H/W (zero row at index P) + replicated frames in T: the VAE encoder's CausalConv3dFull */
                       // padding=1, SpatialUpscaler.swift:78-92,139-145): out-of-range taps read the all-zero row that
                       // the caller keeps at position index F*H*W of the input tensor; 2: replicate H/W/T
    int kt = 3;        // temporal taps: 3 (3x3x3) or 1 (per-frame 3x3 conv2d)
    // Tile order of launches with ONE column tile (N <= 128): row-groups (192-position tiles) per frame when H * W % 192 == 0, else 0.
    // The workgroups that run together on an XCD then cover a compact block of 8 row-groups x 4 frames instead of 32 consecutive
    // row-groups of one frame: their 27-tap halo working set is (8 + 2) x (4 + 2) rows (2.9 MB at 128 channels x 192 voxels) instead
    // of (32 + 2) x 3 (5 MB), i.e. inside the XCD's 4 MB L2. Speed only: every tile is computed exactly once either way.
    int blk_rg = 0;
};

struct GemmArgs {
    const bf16_t* A = nullptr;
    long lda = 0;
    const bf16_t* B = nullptr;
    long ldb = 0;
    int M = 0, N = 0, K = 0;
    int conv = 0;  // 1: A is an implicit conv3d view described by geom, K = 27*C
    Conv3dGeom geom;
    GemmEpilogue ep;
    // split-K (ring kernels): grid.y = split_k workgroups share one output tile, each walks 1/split_k of the K-tiles and
    // stores its raw f32 partial tile to split_ws[z][M][N]; splitk_finish then sums the partials in a fixed order and applies
    // the epilogue (deterministic - no float atomics). Used when a launch has too few output tiles to fill the chip.
    int split_k = 1;         // 0 with a workspace: let launch_gemm_bf16 decide (dense GEMMs)
    float* split_ws = nullptr;
    long split_ws_elems = 0;  // capacity of split_ws in floats (only read when split_k == 0)
    // Tile window (ring kernels): the launch covers the linear tiles [tile0, tile0 + tile_count) of the problem's tile order
    // (tile_count 0 = all). With split-K, win_row0 / win_rows name the rows those tiles cover: the partial slices and the finish
    // pass are compact over that row range. Used by the conv launcher to run the last partial round of a launch as a split-K
    // launch over every CU instead of one K-long tile on a quarter of them.
    int tile0 = 0, tile_count = 0;
    int win_row0 = 0, win_rows = 0;
    int group_m = 4;  // row-tiles per supertile of the workgroup order (0 = column-major tile order); see tile_coords()
    // B as affine-quantised codes: 8-bit codes [N][K] (one per byte) and bf16 scale / bias per 64-wide group along K, [N][K/64];
    // w' = bf16(q * scale + bias). Few-row launches only (gemm_takes_codes: M <= 256, with a split-K workspace): the 128x64 ring
    // kernel de-quantises in its B stage. When set, B may be null.
    const uint8_t* Bq = nullptr;
    const bf16_t* Bqs = nullptr;
    const bf16_t* Bqb = nullptr;
    // B's LDS-DMA loads with the non-temporal policy (once-read weights of a few-row launch: they should not displace the activations
    // every column tile re-reads from the L2). Set by launch_gemm_bf16 for its few-row path; 128x64 ring instance only.
    int b_nt = 0;
    // 8-wave ring kernels (halo-staged conv, dense and conv ring GEMM): the second wave of every SIMD issues its LDS-DMA pieces half a
    // barrier interval after the first (round 5; set by the launchers from options "conv_stagger" / "gemm_stagger"; bit-neutral)
    int conv_stagger = 0;
    // 192x256 kernel's split-K only: the CALLER allows the partial tiles to cross the workspace as bf16 (round 4; half the bytes of the
    // partial round trip). An extra rounding of each K range's share before the f32 finish that a plain f32-accumulating GEMM (the
    // reference's included) does not have - so it is off unless a call site asks (the DiT's FFN-down Linear does; ABI / generic launches
    // do not) and option "split_f32" = 1 overrides it.
    int split_bf16 = 0;
};

// tile_cfg 90 (experiments build only): a weight-streaming kernel for M <= 128 that loads MFMA fragments straight from global memory
// (bf16 or 8-bit codes de-quantised in registers). Measured 1.4x SLOWER than the ring kernel's split-K path at 128 tokens (its four
// waves each re-read the activations: 2 bytes of A per byte of W through the vector-memory path) - kept as a measured negative.

// Number of K splits that fills the chip for a launch with few output tiles (1 = do not split). The caller provides
// split_ws with room for split_k * M * N floats.
int gemm_suggest_split_k(int M, int N, int K);

// Launches on `stream`. Picks the tile shape from (M,N). Throws LtxError on invalid shapes.
void launch_gemm_bf16(const GemmArgs& args, hipStream_t stream);

// The adaLN pass that reads this GEMM's f32 output next (launch_norm_mod's arguments; x = args.ep.out_f32): handed to the launcher so that a
// launch which ends in a split-K finish pass - a row-wise pass over the same stream - applies the norm there instead of in a launch of
// its own (round 4: the FFN's second GEMM + the next block's first adaLN pass). Every other launch runs launch_norm_mod behind the GEMM:
// the caller never launches the pass itself.
struct NormAfter {
    const float* scale = nullptr;
    const float* shift = nullptr;
    long mod_bstride = 0;
    int rows_per_batch = 1;
    bf16_t* out = nullptr;
    long ldo = 0;
    float eps = 1e-6f;
    int norm_kind = 0;  // LTX_NORM_RMS
    int round_norm_bf16 = 0;
    const int32_t* row_map = nullptr;
};
void launch_gemm_bf16(const GemmArgs& args, hipStream_t stream, const NormAfter* norm_after);
// can a launch of this shape take B as 8-bit codes (GemmArgs::Bq)?
bool gemm_takes_codes(int M, int N, int K);
bool gemm_fewrow_takes(int M, int N, int K, int splits);
// Force a tile config (v1: 0 128x128, 1 192x128, 3 96x128, 4 128x96; v2 ring: 21 192x128, 23 256x128, 25 128x192) -
// used by tests/bench sweeps.
void launch_gemm_bf16_cfg(const GemmArgs& args, int cfg, hipStream_t stream);

// Small-M path (M <= 8): out[m][n] = act_out( sum_k in_act(a[m][k]) * W[n][k] + bias[n] ), f32 activations x bf16
// weights -> f32, exactly the reference's promotion on the timestep path.
void launch_gemv_f32(const float* a, long lda, const bf16_t* W, long ldw, const float* bias, float* out, long ldo,
                     int M, int N, int K, int in_act, hipStream_t stream);
