// attention.hip - flash-style non-causal attention forward for gfx950, head_dim = 128, bf16 in / f32 softmax.
//
// Replaces the reference's MLXFast.scaledDotProductAttention call (LTXAttention.swift:207-211): self-attention
// over the spatio-temporal video tokens and cross-attention over the projected caption keys with an additive
// key mask (LTXTransformer.swift:141-156). scale = 1/sqrt(128).
//
// Layouts (chosen for this kernel, not inherited): Q,K are token-major [B][T][H*128] exactly as the projection
// GEMM + q/k-norm + RoPE kernels leave them (no head transposes anywhere); V arrives TRANSPOSED,
// Vt[B][H*128][ld] with keys contiguous, which the V projection emits for free by running the GEMM with swapped
// operands. With that, both MFMA products read their LDS tiles as plain K-contiguous GEMM operands:
//   S^T = K * Q^T      (A = K tile rows=keys, B = Q fragments held in registers)  -> each lane owns one query column
//   O^T = V^T * P^T    (A = Vt tile rows=d,   B = P converted in registers)       -> same query column per lane
// so the online-softmax statistics (m, l) are per-lane scalars, the O rescale is a per-lane multiply and P never
// touches LDS. The key order inside S^T's MFMA rows is permuted (bits 2<->3) when the K fragment is addressed, so
// that the 8 accumulator registers of one k-step are 8 CONSECUTIVE keys - exactly the B-operand layout the PV
// product needs (guide: "An accumulator tile as the next MFMA's operand").
#include "attention.h"
#include "runtime.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int K_TILE_BYTES = KV_TILE * 256;  // [64 keys][128 d] bf16
constexpr int V_TILE_BYTES = 128 * 128;      // [128 d][64 keys] bf16
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y;
    const int b = blockIdx.z;
    const int r = lane & 31, h = lane >> 5;

    const bf16_t* Qb = a.Q + (long)b * a.q_bstride + head * 128;
    const bf16_t* Kb = a.K + (long)b * a.k_bstride + head * 128;
    const bf16_t* Vb = a.Vt + (long)b * a.vt_bstride + (long)head * 128 * a.ldvt;
    const float* biasb = HAS_BIAS ? a.bias + (long)b * a.bias_bstride : nullptr;

    const int q_base = blockIdx.x * 128 + wave * 32;
    const int qi = q_base + r;
    const int qrow = qi < a.Tq ? qi : a.Tq - 1;

    // Q fragments: B operand of S^T = K*Q^T. lane (r,h) holds Q[q=r][d = 16*ks + 8*h + j]
    s16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const s16x8*)(Qb + (long)qrow * a.ldq + ks * 16 + h * 8);

    // ---- staging geometry ----
    // K tile: wave-instruction = 4 keys x 256 B; physical chunk p of key k holds logical chunk p ^ (k & 15)
    const int k_key = lane >> 4, k_pch = lane & 15;
    // Vt tile: wave-instruction = 8 d-rows x 128 B; physical chunk p of row d holds logical chunk p ^ ((d>>1)&7)
    const int v_row = lane >> 3, v_pch = lane & 7;
    const bf16_t* v_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = (wave + 4 * i) * 8 + v_row;
        v_src[i] = Vb + (long)d * a.ldvt + ((v_pch ^ ((d >> 1) & 7)) << 3);
    }
    auto stage = [&](int s, int t) {
        char* kbuf = smem + s * STAGE_BYTES;
        char* vbuf = kbuf + K_TILE_BYTES;
        const int k0 = t * KV_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kl = (wave + 4 * i) * 4 + k_key;
            int gk = k0 + kl;
            gk = gk < a.Tk ? gk : a.Tk - 1;
            const bf16_t* src = Kb + (long)gk * a.ldk + ((k_pch ^ (kl & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(kbuf + (wave + 4 * i) * 1024),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(v_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(vbuf + (wave + 4 * i) * 1024),
                                             16, 0, 0);
        }
    };

    // ---- fragment addressing ----
    // K fragment (A operand of S^T): MFMA row r reads key pi(r) = r with bits 2 and 3 swapped.
    const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    const int k_frag_row = pr * 256;
    const int k_sw = pr & 15;
    // Vt fragment (A operand of O^T): MFMA row r reads d-row r of the 32-row d-block.
    const int v_frag_row = r * 128;
    const int v_sw = (r >> 1) & 7;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float LOG2E = 1.4426950408889634f;
    const float c = HAS_BIAS ? LOG2E : a.scale * LOG2E;

    const int nt = (a.Tk + KV_TILE - 1) / KV_TILE;
    stage(0, 0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const char* kbuf = smem + cur * STAGE_BYTES;
        const char* vbuf = kbuf + K_TILE_BYTES;
        const int k0 = t * KV_TILE;

        // S^T = K * Q^T : two 32-key blocks, 8 k-steps of 16 over head_dim
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const s16x8 kf = *(const s16x8*)(kbuf + kb * 32 * 256 + k_frag_row + ((((ks << 1) + h) ^ k_sw) << 4));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf),
                                                                __builtin_bit_cast(bf16x8_t, qf[ks]), s[kb], 0, 0, 0);
            }
        }
        // register e of s[kb] is key k0 + kb*32 + 16*(e>>3) + 8*h + (e&7) for query column r
        if (HAS_BIAS) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int key = k0 + kb * 32 + 16 * (e >> 3) + 8 * h + (e & 7);
                    key = key < a.Tk ? key : a.Tk - 1;
                    s[kb][e] = s[kb][e] * a.scale + biasb[key];
                }
        }
        if (k0 + KV_TILE > a.Tk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = k0 + kb * 32 + 16 * (e >> 3) + 8 * h + (e & 7);
                    if (key >= a.Tk) s[kb][e] = -INFINITY;
                }
        }
        float mloc = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, s[kb][e]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * c);
        m_run = m_new;
        const float mc = m_safe * c;
        float psum = 0.f;
        s16x8 pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p[j] = __builtin_amdgcn_exp2f(s[kb][s2 * 8 + j] * c - mc);
                    psum += p[j];
                }
                s16x8 pk;
#pragma unroll
                for (int j = 0; j < 8; ++j) pk[j] = (short)f32_to_bf16(p[j]);
                pf[kb * 2 + s2] = pk;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[i][e] *= alpha;

        // O^T += Vt * P^T : four 32-row d-blocks, 4 k-steps of 16 keys
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int kst = 0; kst < 4; ++kst) {
                const s16x8 vf = *(const s16x8*)(vbuf + db * 32 * 128 + v_frag_row + ((((kst << 1) + h) ^ v_sw) << 4));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf),
                                                                __builtin_bit_cast(bf16x8_t, pf[kst]), o[db], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T / l ; register e of o[db] is d = db*32 + (e&3) + 8*(e>>2) + 4*h ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < a.Tq) {
        bf16_t* orow = a.O + (long)b * a.o_bstride + (long)qi * a.ldo + head * 128;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 pk;
                pk.x = pack_bf16x2(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv);
                pk.y = pack_bf16x2(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv);
                *(uint2*)(orow + db * 32 + g4 * 8 + h * 4) = pk;
            }
    }
}

}  // namespace

void launch_attention(const AttnArgs& a, hipStream_t stream) {
    LTX_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: empty problem");
    LTX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 4 == 0, "attention: leading dims");
    LTX_REQUIRE(a.ldvt >= ((a.Tk + 63) / 64) * 64, "attention: Vt row stride %ld must cover Tk=%d rounded up to 64", a.ldvt, a.Tk);
    LTX_REQUIRE(((uintptr_t)a.Q & 15) == 0 && ((uintptr_t)a.K & 15) == 0 && ((uintptr_t)a.Vt & 15) == 0 && ((uintptr_t)a.O & 7) == 0,
                "attention: pointer alignment");
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        attr_set = true;
    }
    ProfScope prof(PROF_ATTN, 4.0 * a.B * a.H * (double)a.Tq * a.Tk * 128, stream);
    dim3 grid((a.Tq + 127) / 128, a.H, a.B);
    if (a.bias)
        hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
    HIP_CHECK(hipGetLastError());
}
