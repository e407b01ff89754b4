// linear_ops.h - the two GEMM call shapes the transformer graphs (DiT, text-embedding connector) are built from.
#pragma once
#include "dit.h"
#include "gemm.h"

// Y = X . W^T (+ bias unless the epilogue brings its own)
inline void gemm_linear(const bf16_t* A, long lda, const LinearW& w, int M, GemmEpilogue ep, hipStream_t s) {
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
    launch_gemm_bf16(g, s);
}

// V^T[d][token] = W_v[d][:] . X[token][:] + b_v[d]  (swapped operands -> the attention kernel's Vt layout)
inline void gemm_vt(const bf16_t* X, long ldx, int tokens, const LinearW& wv, bf16_t* vt, long ldvt, hipStream_t s) {
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
    launch_gemm_bf16(g, s);
}
