// linear_ops.h - the two GEMM call shapes the transformer graphs (DiT, text-embedding connector) are built from.
#pragma once
#include <stdlib.h>

#include "dit.h"
#include "gemm.h"
#include "options.h"

// optional split-K workspace (floats): when given, launches with few output tiles and a long K split the reduction
struct SplitWs {
    float* p = nullptr;
    long elems = 0;
    // the caller accepts bf16 partial tiles when the launcher picks the 192x256 split-K (GemmArgs::split_bf16): an extra rounding the
    // reference does not have, so it is requested per call site (the DiT's FFN-down Linear) and never by the generic / ABI launches
    bool bf16_partials = false;
};

// A quantised Linear whose launch has few rows hands its 8-bit codes to the GEMM, which de-quantises them in its B stage (no scratch
// matrix: the codes are read once, half the bytes of the bf16 weights); every other launch gets bf16 weights (dit_linear_weights).
inline bool linear_codes_in_gemm(const LinearW& w, int M, const SplitWs& ws) {
    const bool off = ltx_opt(OPT_QB_OFF) != 0;  // A/B option "qb_off": every launch through the scratch matrix
    return !off && !w.w && w.q && w.qbits == 8 && ws.p && gemm_takes_codes(M, w.out, w.in);
}
inline void set_weights(GemmArgs& g, const LinearW& w, bool codes, hipStream_t s) {
    if (codes) {
        g.Bq = w.q;
        g.Bqs = w.qs;
        g.Bqb = w.qb;
    } else {
        g.B = dit_linear_weights(w, s);
    }
}

// Y = X . W^T (+ bias unless the epilogue brings its own)
inline void gemm_linear(const bf16_t* A, long lda, const LinearW& w, int M, GemmEpilogue ep, hipStream_t s, SplitWs ws = {}, const NormAfter* norm_after = nullptr) {
    GemmArgs g;
    g.A = A;
    g.lda = lda;
    g.ldb = w.in;
    g.M = M;
    g.N = w.out;
    g.K = w.in;
    set_weights(g, w, linear_codes_in_gemm(w, M, ws), s);
    if (!ep.bias_n && !ep.bias_m) ep.bias_n = w.b;
    g.ep = ep;
    if (ws.p) {
        g.split_k = 0;
        g.split_ws = ws.p;
        g.split_ws_elems = ws.elems;
        g.split_bf16 = ws.bf16_partials ? 1 : 0;
    }
    if (norm_after) launch_gemm_bf16(g, s, norm_after); else launch_gemm_bf16(g, s);
}

// V^T[d][token] = W_v[d][:] . X[token][:] + b_v[d]  (swapped operands -> the attention kernel's Vt layout)
inline void gemm_vt(const bf16_t* X, long ldx, int tokens, const LinearW& wv, bf16_t* vt, long ldvt, hipStream_t s, SplitWs ws = {}) {
    if (ldvt % 4 == 0 && linear_codes_in_gemm(wv, tokens, ws)) {
        // few tokens on a quantised Linear: tokens as rows so that the codes are the column operand the B stage de-quantises; the
        // split-K finish pass stores transposed
        GemmArgs g;
        g.A = X;
        g.lda = ldx;
        set_weights(g, wv, true, s);
        g.ldb = wv.in;
        g.M = tokens;
        g.N = wv.out;
        g.K = wv.in;
        g.ep.out_bf16_t = vt;
        g.ep.ld_bf16_t = ldvt;
        g.ep.bias_n = wv.b;
        g.split_k = 0;
        g.split_ws = ws.p;
        g.split_ws_elems = ws.elems;
        launch_gemm_bf16(g, s);
        return;
    }
    if (ldvt % 4 == 0 && gemm_suggest_split_k(tokens, wv.out, wv.in) <= 1) {
        // enough output tiles without a K split: tokens as rows (the weights stay the column operand, as in every other launch -
        // 50.4 us against 56.5 us at 1536 tokens with HBM-cold weights) and the epilogue stores transposed
        GemmArgs g;
        g.A = X;
        g.lda = ldx;
        g.B = dit_linear_weights(wv, s);
        g.ldb = wv.in;
        g.M = tokens;
        g.N = wv.out;
        g.K = wv.in;
        g.ep.out_bf16_t = vt;
        g.ep.ld_bf16_t = ldvt;
        g.ep.bias_n = wv.b;
        launch_gemm_bf16(g, s);
        return;
    }
    GemmArgs g;
    g.A = dit_linear_weights(wv, s);
    g.lda = wv.in;
    g.B = X;
    g.ldb = ldx;
    g.M = wv.out;
    g.N = tokens;
    g.K = wv.in;
    g.ep.out_bf16 = vt;
    g.ep.ld_bf16 = ldvt;
    g.ep.bias_m = wv.b;
    if (ws.p && tokens % 4 == 0) {
        g.split_k = 0;
        g.split_ws = ws.p;
        g.split_ws_elems = ws.elems;
    }
    launch_gemm_bf16(g, s);
}
