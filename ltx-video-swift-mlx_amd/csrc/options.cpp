// options.cpp - the one table behind ltx_ctx_set_option / ltx_ctx_get_option (see options.h).
#include "options.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>

namespace {

const LtxOptInfo kInfo[OPT_COUNT] = {
    {"qk_f32", 0, 0, 1, 1, "1: store the q|k and cross-attention q projections as f32 (as the reference hands them to RMSNorm); 0: bf16 store"},
    {"split_f32", 0, 0, 1, 1, "1: f32 split-K partial tiles for the DiT's FFN-down GEMM; 0: bf16 partials"},
    {"dtl_splitk", 1, 0, 1, 1, "0: no split-K on the 192x256 GEMM kernel"},
    {"dtl_splitk_mink", 8192, 64, 1 << 20, 1, "least K per range of the 192x256 split-K"},
    {"gemm_rowsplit", 1, 0, 1, 1, "0: no row split of a ragged last round of 192x256 tiles"},
    {"smallm_cfg", 29, 0, 99, 1, "tile configuration of few-row GEMM launches"},
    {"conv_cfg", 0, 0, 99, 1, "0: launcher's choice; else the tile configuration of implicit-GEMM convs"},
    {"conv_no_tail", 0, 0, 1, 1, "1: no split-K launch for the last partial round of a conv"},
    {"attn_impl", 0, 0, 5, 1, "0: launcher's choice; 1 / 2 / 4: 4-wave / ping-pong / assembly attention kernel"},
    {"attn_no_split", 0, 0, 1, 1, "1: no key split for few-query attention launches"},
    {"qb_off", 0, 0, 1, 1, "1: quantised Linears always through the de-quantised scratch matrix"},
    {"finish_norm", 1, 0, 1, 0, "0: the next block's adaLN pass as its own launch"},
    {"finish_rows", 1, 1, 4, 0, "rows per workgroup of the fused split-K finish + norm pass (1, 2, 4)"},
    {"norm_rows", 0, 0, 4, 0, "rows per workgroup of the norm + modulation pass (0 = launcher's choice, 2, 4)"},
    {"qknorm_no_pair", 0, 0, 1, 0, "1: q and k RMSNorm + RoPE as two launches"},
    {"conv_halo", 1, 0, 1, 0, "0: no halo-staged conv kernel"},
    {"conv_persist", 1, 0, 1, 0, "0: halo-staged conv kernel with one workgroup per tile"},
    {"conv_block", 1, 0, 1, 0, "0: plain tile order for single-column convs"},
    {"conv_tall", 1, 0, 3, 1, "0: no 384 x 128 conv tiles of whole image rows for W == 384 / 192 / 96 (results then equal the 192-row kernel's bit for bit); 1: where they cover more than half the chip; 2: the same, not for W == 384; 3: wherever the shape allows"},
    {"gemm_stagger", 0, 0, 1, 0, "1: the second wave of every SIMD issues its LDS-DMA pieces half a barrier interval late in the dense ring GEMM kernel (measured: no gain); 0: both waves of a SIMD issue their LDS-DMA pieces at the same point of the dense ring GEMM kernel's K loop"},
    {"conv_d2s_pn", 1, 0, 1, 1, "0: the first PixelNorm of the VAE decoder's 128-channel stage as a row pass of its own instead of in the upsampler conv's epilogue"},
    {"conv_stagger", 1, 0, 1, 0, "0: both waves of a SIMD issue their LDS-DMA pieces at the same point of the halo conv kernel's K loop"},
    {"b_nt", -1, -1, 1, 0, "-1: launcher's choice; 0 / 1: non-temporal weight loads of the few-row GEMM off / on"},
    {"attn_plain_order", 0, 0, 1, 0, "1: (query block, head, batch) workgroup order"},
    {"sp_overlap", 0, 0, 1, 0, "1: sequence-parallel V^T gather on a side stream"},
    {"sp_selftest", 0, 0, 1, 0, "1: one-rank self-test of the sequence-parallel side-stream branch"},
    {"abl_rows", 0, 0, 7, 0, "row-pass ablation mask (tools only)"},
};

std::atomic<int> g_val[OPT_COUNT];
std::once_flag g_once;

void init_table() {
    for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kInfo[i].def, std::memory_order_relaxed);
#ifdef LTX_EXPERIMENTS
    // experiments build only: LTX_<NAME> seeds the table once (tools/, A/B scripts). The product library ignores the environment.
    for (int i = 0; i < OPT_COUNT; ++i) {
        std::string env = "LTX_";
        for (const char* c = kInfo[i].name; *c; ++c) env += (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
        if (const char* v = getenv(env.c_str())) {
            const int x = *v ? atoi(v) : 1;  // a bare `LTX_FOO=` counted as "set" for the presence-style hooks
            if (x >= kInfo[i].lo && x <= kInfo[i].hi) g_val[i].store(x, std::memory_order_relaxed);
        }
    }
#endif
}

}  // namespace

int ltx_opt(LtxOpt o) {
    std::call_once(g_once, init_table);
    return g_val[o].load(std::memory_order_relaxed);
}
const LtxOptInfo& ltx_opt_info(int index) { return kInfo[index]; }
int ltx_opt_find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(kInfo[i].name, name) == 0) return i;
    return -1;
}
bool ltx_opt_set(int index, int value) {
    std::call_once(g_once, init_table);
    if (index < 0 || index >= OPT_COUNT || value < kInfo[index].lo || value > kInfo[index].hi) return false;
    g_val[index].store(value, std::memory_order_relaxed);
    return true;
}
