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
    g.B = w.w;
    g.ldb = w.in;
    g.M = M;
    g.N = w.out;
    g.K = w.in;
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
    GemmArgs g;
    g.A = wv.w;
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
