// options.h - the library's tuning / A-B switches in ONE table (round 5; round-4 verdict "Harden the hooks").
//
// Until round 4 the launchers read 26 environment variables (`LTX_QK_F32`, `LTX_SPLIT_F32`, ...): a drop-in library whose numerics change
// with the caller's environment is not what a maintainer expects. Now:
//   * the PRODUCT build never reads the environment. The only way to move a switch is `ltx_ctx_set_option(ctx, "name", value)`
//     (include/ltxhip.h), which the Swift / C++ wrappers can expose; defaults are the measured best.
//   * the -DLTX_EXPERIMENTS build (tools/, the `experiments` tests) additionally seeds the table ONCE from `LTX_<NAME>` so that the
//     micro-benchmarks and A/B scripts keep working from a shell.
// The table is process-wide (the kernel launchers are shared by every context of a process) and read with relaxed atomics per launch -
// nothing is cached in function statics any more, so a test can flip a switch between two launches of one process.
#pragma once

enum LtxOpt {
    // ---- numerics: each of these changes roundings or the accumulation order (never the algorithm) ----
    OPT_QK_F32,           // 1 = q|k and cross-q projections stored f32 before RMSNorm + RoPE (default 0: bf16 store, one extra rounding)
    OPT_SPLIT_F32,        // 1 = split-K partial tiles of the DiT's FFN-down GEMM cross the workspace as f32 (default 0: bf16 partials)
    OPT_DTL_SPLITK,       // 0 = no split-K on the 192x256 kernel (FFN-down on the ring kernel)
    OPT_DTL_SPLITK_MINK,  // least K per range of that split (8192)
    OPT_GEMM_ROWSPLIT,    // 0 = no row split of a ragged last round of 192x256 tiles
    OPT_SMALLM_CFG,       // tile cfg of few-row launches (29; 30 = the few-row kernel of the experiments build)
    OPT_CONV_CFG,         // 0 = launcher's choice; else the tile cfg of implicit-GEMM convs (21 / 23 / 27)
    OPT_CONV_NO_TAIL,     // 1 = no split-K launch for the last partial round of a conv
    OPT_ATTN_IMPL,        // 0 = launcher's choice; 1 / 2 / 4 force the 4-wave / ping-pong / assembly kernel (3, 5: experiments build)
    OPT_ATTN_NO_SPLIT,    // 1 = no key split for few-query launches
    OPT_QB_OFF,           // 1 = quantised Linears always through the de-quantised scratch matrix
    // ---- bit-neutral: same results, different launches ----
    OPT_FINISH_NORM,      // 0 = the next block's adaLN pass as its own launch (default 1: rides on the split-K finish)
    OPT_FINISH_ROWS,      // rows per workgroup of the fused finish + norm pass: 1 (default), 2, 4 - clamped to what divides M and rows_per_batch
    OPT_NORM_ROWS,        // rows per workgroup of norm_mod_rows_kernel: 0 = launcher's choice, 2, 4
    OPT_QKNORM_NO_PAIR,   // 1 = q and k RMSNorm + RoPE as two launches
    OPT_CONV_HALO,        // 0 = no halo-staged conv kernel
    OPT_CONV_PERSIST,     // 0 = halo kernel one workgroup per tile
    OPT_CONV_BLOCK,       // 0 = plain tile order for single-column convs
    OPT_CONV_TALL,        // 0 = no 384 x 128 tiles of whole image rows (conv_halo2.inc) for the W == 384 / 192 / 96 convs; 1 = launches of more than
                          // half a round of them; 2 = the same, not for W == 384; 3 = every launch whose shape allows (tests)
    OPT_GEMM_STAGGER,     // the same for the dense 8-wave ring GEMM kernels
    OPT_CONV_D2S_PN,      // 0 = the 128-channel stage's first PixelNorm as a row pass of its own (default 1: in the upsampler conv's epilogue)
    OPT_CONV_STAGGER,     // 0 = both waves of a SIMD issue their LDS-DMA pieces at the same point of the halo kernel's K loop (as before round 5)
    OPT_B_NT,             // -1 = launcher's choice; 0 / 1 = non-temporal weight loads of the few-row GEMM off / on
    OPT_ATTN_PLAIN_ORDER, // 1 = (query block, head, batch) workgroup order as before round 3
    OPT_SP_OVERLAP,       // 1 = sequence-parallel V^T gather on a side stream (opt-in until a >= 2-rank RCCL run exists)
    OPT_SP_SELFTEST,      // 1 = one-rank self-test of the side-stream branch
    OPT_ABL_ROWS,         // ablation of row passes (tools/bench_rows.py; experiments)
    OPT_COUNT
};

struct LtxOptInfo {
    const char* name;  // lower case, no prefix: "qk_f32"
    int def, lo, hi;
    int numerics;      // 1 = moves results by rounding / accumulation order
    const char* doc;
};

int ltx_opt(LtxOpt o);                          // current value (relaxed atomic load)
const LtxOptInfo& ltx_opt_info(int index);      // index < OPT_COUNT
int ltx_opt_find(const char* name);             // -1 = unknown
bool ltx_opt_set(int index, int value);         // false = out of range
