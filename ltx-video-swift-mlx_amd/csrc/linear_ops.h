// linear_ops.h - the two GEMM call shapes the transformer graphs (DiT, text-embedding connector) are built from.
#pragma once
#include "dit.h"
#include "gemm.h"

// optional split-K workspace (floats): when given, launches with few output tiles and a long K split the reduction
struct SplitWs {
    float* p = nullptr;
    long elems = 0;
};

// Y = X . W^T (+ bias unless the epilogue brings its own)
inline void gemm_linear(const bf16_t* A, long lda, const LinearW& w, int M, GemmEpilogue ep, hipStream_t s, SplitWs ws = {}) {
    GemmArgs g;
    g.A = A;
    g.lda = lda;
    g.ldb = w.in;
    g.M = M;
    g.N = w.out;
    g.K = w.in;
    g.B = dit_linear_weights(w, s);
    if (!ep.bias_n && !ep.bias_m) ep.bias_n = w.b;
    g.ep = ep;
    if (ws.p) {
        g.split_k = 0;
        g.split_ws = ws.p;
        g.split_ws_elems = ws.elems;
    }
    launch_gemm_bf16(g, s);
}

// V^T[d][token] = W_v[d][:] . X[token][:] + b_v[d]  (swapped operands -> the attention kernel's Vt layout)
inline void gemm_vt(const bf16_t* X, long ldx, int tokens, const LinearW& wv, bf16_t* vt, long ldvt, hipStream_t s, SplitWs ws = {}) {
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
