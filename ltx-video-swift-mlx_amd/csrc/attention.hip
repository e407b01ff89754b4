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
#include <stdlib.h>

#include <type_traits>

#include "attention.h"
#include "options.h"
#include "runtime.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int K_TILE_BYTES = KV_TILE * 256;  // [64 keys][128 d] bf16
constexpr int V_TILE_BYTES = 128 * 128;      // [128 d][64 keys] bf16
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Workgroup -> (query block, head, batch element) for the one-workgroup-per-CU kernels. Workgroups are dealt round-robin over the
// 8 XCDs (blocks b and b + 8 share an L2: observed, for speed only), so with the plain (x, head, b) grid the query blocks of ONE head
// land on eight different L2s and every XCD pulls every head's K / V^T through the fabric: 8 x the bytes. Here a 1-D grid is decoded
// so that all query blocks of a head instance sit on one XCD (heads xcd, xcd + 8, ... per XCD), when the head count allows it.
struct AttnBlock {
    int x, head, b;
};
LTX_DEVFN AttnBlock attn_block(int nqb, int H, int B, int plain_order) {
    const int id = blockIdx.x;
    const int G = H * B;
    int x, hb;
    if ((G & 7) == 0 && !plain_order) {
        const int xcd = id & 7, j = id >> 3;
        hb = xcd + 8 * (j / nqb);
        x = j % nqb;
    } else {
        hb = id / nqb;
        x = id % nqb;
    }
    return AttnBlock{x, hb % H, hb / H};
}

// One online-softmax update for this lane's query column over a 64-key tile: scores s (register e of s[kb] is key
// k0 + kb*32 + 16*(e>>3) + 8*h + (e&7)) -> P as the PV product's bf16 B operand, running max / sum, O rescale.
//
// Written for VALU instruction COUNT: on gfx950 an MFMA wave and a VALU wave on the same SIMD overlap only for transcendentals;
// every plain VALU instruction costs ~3/4 of its 4 issue cycles even under another wave's MFMAs (tools/ubench/overlap.hip:
// 32 MFMA = 512 ns alone, +141 ns with 96 FMAs from the other wave, +4 ns with 32 exp2). So: packed-f32 FMA / add for the
// exponent argument and the row sum (2 values per instruction), and the 64-register O rescale is skipped whenever no lane's
// maximum moved (alpha == 1 exactly) - the common case after the first few tiles.
template <bool HAS_BIAS>
LTX_DEVFN void softmax_tile(f32x16 (&s)[2], f32x16 (&o)[4], s16x8 (&pf)[4], float& m_run, float& l_run, float c, float scale,
                            const float* biasb, int k0, int Tk, int h) {
    if (HAS_BIAS) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int key = k0 + kb * 32 + 16 * (e >> 3) + 8 * h + (e & 7);
                key = key < Tk ? key : Tk - 1;
                s[kb][e] = s[kb][e] * scale + biasb[key];
            }
    }
    if (k0 + KV_TILE > Tk) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + kb * 32 + 16 * (e >> 3) + 8 * h + (e & 7);
                if (key >= Tk) s[kb][e] = -INFINITY;
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
    const f32x2 c2 = {c, c};
    const f32x2 nmc2 = {-m_safe * c, -m_safe * c};
    f32x2 psum2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            s16x8 pk;
#pragma unroll
            for (int jj = 0; jj < 8; jj += 2) {
                const f32x2 sv = {s[kb][s2 * 8 + jj], s[kb][s2 * 8 + jj + 1]};
                const f32x2 t = __builtin_elementwise_fma(sv, c2, nmc2);
                const f32x2 pv = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                psum2 += pv;
                pk[jj] = (short)f32_to_bf16(pv.x);
                pk[jj + 1] = (short)f32_to_bf16(pv.y);
            }
            pf[kb * 2 + s2] = pk;
        }
    l_run = l_run * alpha + (psum2.x + psum2.y);
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const AttnBlock blk = attn_block((a.Tq + 127) / 128, a.H, a.B, a.plain_order);
    const int head = blk.head;
    const int b = blk.b;
    const int r = lane & 31, h = lane >> 5;

    const bf16_t* Qb = a.Q + (long)b * a.q_bstride + head * 128;
    const bf16_t* Kb = a.K + (long)b * a.k_bstride + head * 128;
    const bf16_t* Vb = a.Vt + (long)b * a.vt_bstride + (long)head * 128 * a.ldvt;
    const float* biasb = HAS_BIAS ? a.bias + (long)b * a.bias_bstride : nullptr;

    const int q_base = blk.x * 128 + wave * 32;
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
        s16x8 pf[4];
        softmax_tile<HAS_BIAS>(s, o, pf, m_run, l_run, c, a.scale, biasb, k0, a.Tk, h);

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


// ---------------------------------------------------------------------------------------------------------------
// Ping-pong variant: 8 waves x 32 queries per workgroup, one workgroup per CU, TWO waves per SIMD that alternate roles.
//
// Why: with two independent waves per SIMD the kernel above takes the SUM of its MFMA time (32 x 32 cycles per 64-key tile) and
// its softmax VALU time (~200 instructions, ~1200 cycles) per wave: measured 2.3 us per tile step with two workgroups on a CU
// against 1.5 us with one - the two waves of a SIMD drift into the same phase (both in the MFMA block, then both in the softmax)
// and the matrix pipe idles while both do VALU work. Here the two wave groups (waves 0-3 / 4-7, one of each per SIMD) are held
// exactly one phase apart by the workgroup barrier: every wave runs  [softmax cluster] barrier [MFMA cluster] barrier ...,
// group 1 enters the loop one barrier late, so while group 0 is in its MFMA cluster group 1 is in its softmax cluster and
// vice versa. For that a wave's MFMA cluster must not depend on its own softmax of the same tile, so the loop is skewed:
//     softmax cluster j : P(j) = softmax(S(j)), rescale O;       also issues the LDS-DMA of stage j+2
//     MFMA cluster j    : S(j+1) = K(j+1) Q^T   and   O += Vt(j) P(j)^T
// "stage s" = {K tile s+1, Vt tile s}: what MFMA cluster s reads. A stage is issued two steps ahead, waited for (counted vmcnt)
// one step ahead, and read by group 0 / group 1 in consecutive barrier intervals; with 4 ring slots per operand (128 KB LDS) a
// slot is rewritten two intervals after its last reader.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PP_Q = 256;
constexpr int PP_SLOTS = 4;
constexpr int PP_RA = 4;  // fragment reads in flight ahead of their MFMA (8 measured the same)
constexpr int PP_LDS = PP_SLOTS * (K_TILE_BYTES + V_TILE_BYTES);

// Phase boundaries. The barriers are inline asm that takes the phase's RESULTS as read-write operands: a plain "memory" clobber
// orders memory operations only, and the compiler otherwise sinks most of the softmax below the barrier (seen in the ISA), which
// puts both wave groups' VALU work into the same interval.
#define PP_END_SOFTMAX(VM)                                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_barrier"                                                                     \
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]),     \
                   "+v"(l_run), "+v"(m_run)::"memory")
#define PP_END_MFMA()                                                                                                       \
    asm volatile("s_barrier" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(s[0]), "+v"(s[1])::"memory")
LTX_DEVFN void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

#ifdef PP_STAMPS  // tools/ubench/attn_stamps.hip: cycle stamps at the phase boundaries of one steady-state tile step
__device__ unsigned long long g_pp_stamps[8][8];
#define PP_STAMP(i) \
    if (j == 9) st_[i] = __builtin_readcyclecounter()  // kept in SGPRs until the kernel's end: a store here would join vmcnt
#else
#define PP_STAMP(i)
#endif

template <bool HAS_BIAS>
__global__ __launch_bounds__(512, 1) void attn_fwd_kernel_pp(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;
    const AttnBlock blk = attn_block((a.Tq + PP_Q - 1) / PP_Q, a.H, a.B, a.plain_order);
    const int head = blk.head;
    const int b = blk.b;
    const int r = lane & 31, h = lane >> 5;

    const bf16_t* Qb = a.Q + (long)b * a.q_bstride + head * 128;
    const bf16_t* Kb = a.K + (long)b * a.k_bstride + head * 128;
    const bf16_t* Vb = a.Vt + (long)b * a.vt_bstride + (long)head * 128 * a.ldvt;
    const float* biasb = HAS_BIAS ? a.bias + (long)b * a.bias_bstride : nullptr;

    const int qi = blk.x * PP_Q + wave * 32 + r;
    const int qrow = qi < a.Tq ? qi : a.Tq - 1;
    s16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const s16x8*)(Qb + (long)qrow * a.ldq + ks * 16 + h * 8);

    const int nt = (a.Tk + KV_TILE - 1) / KV_TILE;
    char* const kring = smem;
    char* const vring = smem + PP_SLOTS * K_TILE_BYTES;

    // staging: a tile is 16 wave-instructions of 1 KB; wave w issues chunks w and w+8 of each operand (same swizzles as above).
    // Buffer loads: an SGPR descriptor per operand, a 32-bit per-lane byte offset (4 loop-invariant VGPRs in all) and the tile
    // advance as the scalar offset - no 64-bit address arithmetic in the loop. The K descriptor ends after key Tk-1, so the keys
    // of a ragged last tile that lie beyond it read as zeros (their scores are masked to -inf below).
    const int k_key = lane >> 4, k_pch = lane & 15;
    const int v_row = lane >> 3, v_pch = lane & 7;
    const __amdgpu_buffer_rsrc_t k_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((long)(a.Tk - 1) * a.ldk + 128) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)((long)128 * a.ldvt * 2), 0x00020000);
    int koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kl = (wave + 8 * i) * 4 + k_key;
        koff[i] = (kl * a.ldk + ((k_pch ^ (kl & 15)) << 3)) * 2;
        const int d = (wave + 8 * i) * 8 + v_row;
        voff[i] = (d * a.ldvt + ((v_pch ^ ((d >> 1) & 7)) << 3)) * 2;
    }
    auto stage_k = [&](int t) {  // K tile t (a tile index past the end reads zeros into a free slot; nobody reads them)
        char* kbuf = kring + (t & (PP_SLOTS - 1)) * K_TILE_BYTES;
        const int soff = t * KV_TILE * a.ldk * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(kbuf + (wave + 8 * i) * 1024), 16, koff[i], soff, 0, 0);
    };
    auto stage_v = [&](int t) {
        char* vbuf = vring + (t & (PP_SLOTS - 1)) * V_TILE_BYTES;
        const int soff = t * KV_TILE * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(vbuf + (wave + 8 * i) * 1024), 16, voff[i], soff, 0, 0);
    };

    const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    const int k_frag_row = pr * 256;
    const int k_sw = pr & 15;
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

#ifdef PP_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
#endif
    f32x16 s[2];
    s16x8 pf[4];
    // One MFMA cluster: S(t_k) = K(t_k) Q^T (if QK) and O += Vt(t_v) P^T, as ONE scheduling region in which every LDS fragment
    // read is issued RA MFMAs ahead of its consumer (read latency ~100+ cycles against 32 cycles per MFMA; without this the
    // compiler pairs read->wait->MFMA and the cluster runs at LDS latency, 2.4x its MFMA time). Accumulators are interleaved
    // (2 key blocks / 4 d-blocks round-robin) so consecutive MFMAs are independent.
    // Fragment addresses. The XOR swizzle only permutes the 32-byte k-step groups of a row, so a fragment address is
    //   base + ((kstep << 5) ^ x5)   with two loop-invariant VGPRs per operand (one v_xad_u32 per read, issued between MFMAs),
    // and ring slots are compile-time (the tile loop is unrolled by PP_SLOTS) so slot and block offsets are instruction immediates.
    const int kf_base = k_frag_row + ((h ^ (k_sw & 1)) << 4);
    int kf_x5 = (k_sw >> 1) << 5;
    const int vf_base = PP_SLOTS * K_TILE_BYTES + v_frag_row + ((h ^ (v_sw & 1)) << 4);
    int vf_x5 = (v_sw >> 1) << 5;
    auto mfma_cluster = [&](auto kslot_tag, auto vslot_tag, auto qk_tag, auto pv_tag) {
        constexpr bool QK = decltype(qk_tag)::value, PV = decltype(pv_tag)::value;
        constexpr int KS = decltype(kslot_tag)::value & (PP_SLOTS - 1), VS = decltype(vslot_tag)::value & (PP_SLOTS - 1);
        constexpr int RA = PP_RA;
        constexpr int NM = (QK ? 16 : 0) + (PV ? 16 : 0);
        if constexpr (QK) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
        }
        // keep the 12 fragment addresses from being hoisted out of the tile loop as 12 live VGPRs (the kernel sits at the 256-VGPR
        // limit of two waves per SIMD): the compiler must treat the XOR terms as changed here
        asm volatile("" : "+v"(kf_x5), "+v"(vf_x5));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (QK) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const s16x8 kf = *(const s16x8*)(smem + (((ks << 5) ^ kf_x5) + kf_base) + (KS * K_TILE_BYTES + kb * 32 * 256));
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]), s[kb], 0, 0, 0);
                }
        }
        if constexpr (PV) {
#pragma unroll
            for (int kst = 0; kst < 4; ++kst)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const s16x8 vf = *(const s16x8*)(smem + (((kst << 5) ^ vf_x5) + vf_base) + (VS * V_TILE_BYTES + db * 32 * 128));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[kst]), o[db], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);
#pragma unroll
        for (int i = 0; i < NM - RA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, RA, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using I0_ = std::integral_constant<int, 0>;

    // ---- prologue: stages -1 (K0), 0 (K1, V0), 1 (K2, V1); S(0) ----
    stage_k(0);
    stage_k(1);
    stage_v(0);
    stage_k(2);
    stage_v(1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    mfma_cluster(I0_{}, I0_{}, T_{}, F_{});
    asm volatile("" : "+v"(s[0]), "+v"(s[1]));
    if (group == 1) pp_barrier();  // group 1 runs one barrier interval behind group 0 from here on

    auto step = [&](int j, auto slot_tag) {
        constexpr int SL = decltype(slot_tag)::value;  // == j & 3
        // ================= softmax cluster j =================
        PP_STAMP(0);
        // stage j+2, unconditionally: a tile past the end lands in a free slot (K reads zeros past its descriptor, Vt whatever
        // follows within its own) and is never read; that keeps the loop branch-free and the vmcnt count below constant
        stage_k(j + 3);
        stage_v(j + 2);
        PP_STAMP(1);
        const int k0 = j * KV_TILE;
        softmax_tile<HAS_BIAS>(s, o, pf, m_run, l_run, c, a.scale, biasb, k0, a.Tk, h);
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(2);
        // stage j+1 (issued one step ago) must have landed before anyone's MFMA cluster j+1; the 4 loads just issued may fly
        PP_END_SOFTMAX(4);
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(3);

        // ================= MFMA cluster j =================
        __builtin_amdgcn_s_setprio(1);
        if (j + 1 < nt)
            mfma_cluster(std::integral_constant<int, SL + 1>{}, slot_tag, T_{}, T_{});
        else
            mfma_cluster(slot_tag, slot_tag, F_{}, T_{});
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(4);
        PP_END_MFMA();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(5);
    };
    for (int j = 0; j < nt; j += 4) {
        step(j, std::integral_constant<int, 0>{});
        if (j + 1 < nt) step(j + 1, std::integral_constant<int, 1>{});
        if (j + 2 < nt) step(j + 2, std::integral_constant<int, 2>{});
        if (j + 3 < nt) step(j + 3, std::integral_constant<int, 3>{});
    }
    if (group == 0) pp_barrier();  // pairs with group 1's last loop barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two stages issued past the end must not outlive the workgroup's LDS
#ifdef PP_STAMPS
    if (blk.x == 1 && head == 3 && lane == 0) {
        for (int i = 0; i < 6; ++i) g_pp_stamps[wave][i] = st_[i];
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_pp_stamps[wave][6] = hwid;
    }
#endif

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


// (Tried and removed: a one-wave-per-SIMD kernel, 48 queries per wave as three 16-query blocks of v_mfma_f32_16x16x32_bf16, 192
// queries per workgroup = exactly 256 workgroups at 1536 tokens x 32 heads, softmax of tile t interleaved with the QK^T MFMAs
// of tile t+1 in ONE instruction stream - tools/ubench/overlap.hip shows that in-wave interleaving hides the VALU work
// completely (503 ns against 495 ns for the MFMAs alone) where a second wave does not. Written in HIP C++ it does not survive
// the compiler: O (96 registers) + double-buffered S (96) + Q (48) + P (24) need the AGPR half of the register file, and the
// allocator moves 250-300 registers between the two halves per tile step (v_accvgpr_read/write), ignores or over-serialises the
// sched_group_barrier interleave (softmax ends up after the MFMAs) and spills 1.1-1.5 KB to scratch. The shape is right for
// this problem size; it needs an assembly-level main loop with hand-assigned registers.)

// ---------------------------------------------------------------------------------------------------------------
// 48 queries per wave, one wave per SIMD (192 queries per workgroup): layout reference in plain HIP C++.
// The shape that fits 1536 tokens x 32 heads exactly (256 workgroups, every SIMD loaded alike), on v_mfma_f32_16x16x32_bf16:
// A/B lane (r = lane&15, g = lane>>4) holds k = 8g..8g+7 of row/col r; C lane (c = lane&15, g) holds rows 4g..4g+3 of col c.
// MFMA row rho of key block kb reads tile key 32*(kb>>1) + 4*(kb&1) + 8*(rho>>2) + (rho&3), so a lane's 4+4 accumulators of
// blocks (2i, 2i+1) are the 8 consecutive keys 32i + 8g .. +7 = the B operand of the PV product's k-step i. K rows are
// XOR-swizzled by rho (bits {3,4,0,1} of the key), Vt rows by (d>>1)&7. This version leaves scheduling and register assignment
// to the compiler (which parks half the state in AGPRs and moves it back and forth): it exists to pin the layout with the
// parity tests; the fast path is the assembly main loop generated from the same layout (attention_w48_asm.inc).
// ---------------------------------------------------------------------------------------------------------------
constexpr int W48_Q = 192;
constexpr int W48_SLOTS = 4;
constexpr int W48_LDS = W48_SLOTS * STAGE_BYTES;

struct W48Lane {  // per-lane constants shared by the C++ and the assembly kernel
    int koff[4], voff[4];  // buffer-load byte offsets of this lane's 16-byte pieces of a K / Vt tile (pieces w, w+4, w+8, w+12)
    int kaddr[4], vaddr[2];  // LDS byte address of the K fragment for k-step ks / the Vt fragment for k-step i (slot 0, block 0)
};
LTX_DEVFN W48Lane w48_lane(int lane, int wave, long ldk, long ldvt) {
    W48Lane L;
    const int c16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = (wave + 4 * i) * 4 + (lane >> 4);
        const int sw = (((key >> 3) & 3) << 2) | (key & 3);
        L.koff[i] = (key * (int)ldk + (((lane & 15) ^ sw) << 3)) * 2;
        const int d = (wave + 4 * i) * 8 + (lane >> 3);
        L.voff[i] = (d * (int)ldvt + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2;
    }
    const int kf_base = (8 * (c16 >> 2) + (c16 & 3)) * 256 + ((g ^ (c16 & 3)) << 4);
    const int kf_x6 = (c16 & 12) << 4;
    const int vf_base = K_TILE_BYTES + c16 * 128 + ((g ^ ((c16 >> 1) & 3)) << 4);
    const int vf_x6 = ((c16 >> 1) & 4) << 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) L.kaddr[ks] = ((ks << 6) ^ kf_x6) + kf_base;
#pragma unroll
    for (int i = 0; i < 2; ++i) L.vaddr[i] = ((i << 6) ^ vf_x6) + vf_base;
    return L;
}
constexpr int w48_kblock_off(int kb) { return (32 * (kb >> 1) + 4 * (kb & 1)) * 256; }  // LDS offset of key block kb's row 0
constexpr int w48_vblock_off(int db) { return db * 16 * 128; }

#ifdef LTX_EXPERIMENTS  // plain-HIP layout reference of the 48-query kernel (LTX_ATTN_IMPL=3): pins the LDS images and the MFMA operand mapping
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_w48_ref(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const bf16_t* Qb = a.Q + (long)b * a.q_bstride + head * 128;
    const bf16_t* Kb = a.K + (long)b * a.k_bstride + head * 128;
    const bf16_t* Vb = a.Vt + (long)b * a.vt_bstride + (long)head * 128 * a.ldvt;
    const int q0 = blockIdx.x * W48_Q + wave * 48;
    const int nt = a.Tk / KV_TILE;
    const W48Lane L = w48_lane(lane, wave, a.ldk, a.ldvt);
    s16x8 qf[3][4];
    for (int qb = 0; qb < 3; ++qb)
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const s16x8*)(Qb + (long)(q0 + 16 * qb + c16) * a.ldq + ks * 32 + g * 8);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((long)(a.Tk - 1) * a.ldk + 128) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)((long)128 * a.ldvt * 2), 0x00020000);
    f32x4 o[8][3];
    for (int db = 0; db < 8; ++db)
        for (int qb = 0; qb < 3; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[3] = {-INFINITY, -INFINITY, -INFINITY}, l_run[3] = {0.f, 0.f, 0.f};
    const float c = a.scale * 1.4426950408889634f;
    for (int t = 0; t < nt; ++t) {  // single-buffered, fully synchronous: a layout check, not a fast kernel
        __syncthreads();
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(smem + (wave + 4 * i) * 1024), 16, L.koff[i], t * KV_TILE * (int)a.ldk * 2, 0, 0);
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(smem + K_TILE_BYTES + (wave + 4 * i) * 1024), 16, L.voff[i], t * KV_TILE * 2, 0, 0);
        __syncthreads();
        f32x4 s[4][3];
        for (int kb = 0; kb < 4; ++kb)
            for (int qb = 0; qb < 3; ++qb) s[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < 4; ++kb)
            for (int ks = 0; ks < 4; ++ks) {
                const s16x8 kf = *(const s16x8*)(smem + L.kaddr[ks] + w48_kblock_off(kb));
                for (int qb = 0; qb < 3; ++qb)
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[qb][ks]), s[kb][qb], 0, 0, 0);
            }
        s16x8 pf[3][2];
        for (int qb = 0; qb < 3; ++qb) {
            float mloc = s[0][qb][0];
            for (int kb = 0; kb < 4; ++kb)
                for (int j = 0; j < 4; ++j) mloc = fmaxf(mloc, s[kb][qb][j]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run[qb], mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
            m_run[qb] = m_new;
            float psum = 0.f;
            for (int i = 0; i < 2; ++i) {
                s16x8 pk;
                for (int hh = 0; hh < 2; ++hh)
                    for (int j = 0; j < 4; ++j) {
                        const float p = __builtin_amdgcn_exp2f(s[2 * i + hh][qb][j] * c - m_new * c);
                        psum += p;
                        pk[4 * hh + j] = (short)f32_to_bf16(p);
                    }
                pf[qb][i] = pk;
            }
            l_run[qb] = l_run[qb] * alpha + psum;
            for (int db = 0; db < 8; ++db) o[db][qb] *= alpha;
        }
        for (int db = 0; db < 8; ++db)
            for (int i = 0; i < 2; ++i) {
                const s16x8 vf = *(const s16x8*)(smem + L.vaddr[i] + w48_vblock_off(db));
                for (int qb = 0; qb < 3; ++qb)
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[qb][i]), o[db][qb], 0, 0, 0);
            }
    }
    for (int qb = 0; qb < 3; ++qb) {
        float l = l_run[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        bf16_t* orow = a.O + (long)b * a.o_bstride + (long)(q0 + 16 * qb + c16) * a.ldo + head * 128;
        for (int db = 0; db < 8; ++db) {
            uint2 pk;
            pk.x = pack_bf16x2(o[db][qb][0] * inv, o[db][qb][1] * inv);
            pk.y = pack_bf16x2(o[db][qb][2] * inv, o[db][qb][3] * inv);
            *(uint2*)(orow + db * 16 + g * 4) = pk;
        }
    }
}
#endif  // LTX_EXPERIMENTS


#ifdef W48_STAMPS
__device__ unsigned long long g_w48_stamps[8][32];
#endif
// The same kernel with the main loop in assembly (generated by tools/gen_attn_w48.py from the layout above; register map and
// schedule in that script's header). C++ only prepares the per-lane offsets and the uniform operands.
template <bool HAS_BIAS, bool PRESCALED = false>
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_w48_asm(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const AttnBlock blk = attn_block((a.Tq + W48_Q - 1) / W48_Q, a.H, a.B, a.plain_order);
    const int head = blk.head, b = blk.b;
    // key split: blockIdx.y = the key range of this workgroup; everything below sees that range as the whole problem
    const int zs = a.key_splits > 1 ? (int)blockIdx.y : 0;
    const int s0 = zs * a.split_keys;
    const int Tk = a.key_splits > 1 ? (a.Tk - s0 < a.split_keys ? a.Tk - s0 : a.split_keys) : a.Tk;
    const long ldo = a.key_splits > 1 ? (long)a.H * 128 : a.ldo;
    const bf16_t* Qb = a.Q + (long)b * a.q_bstride + head * 128;
    const bf16_t* Kb = a.K + (long)b * a.k_bstride + head * 128 + (long)s0 * a.ldk;
    const bf16_t* Vb = a.Vt + (long)b * a.vt_bstride + (long)head * 128 * a.ldvt + s0;
    bf16_t* Ob = a.key_splits > 1 ? a.o_part + ((long)zs * a.B + b) * a.Tq * ldo + head * 128 : a.O + (long)b * a.o_bstride + head * 128;
    const int q0 = blk.x * W48_Q + wave * 48;
    const float* lse_w = a.key_splits > 1 ? a.lse + (((long)zs * a.B + b) * a.H + head) * ((a.Tq + W48_Q - 1) / W48_Q * W48_Q) + q0 : nullptr;
    const W48Lane L = w48_lane(lane, wave, a.ldk, a.ldvt);
    // any Tq / Tk: query rows past Tq read row Tq-1 and their stores fall outside the O descriptor; key rows past Tk read zeros
    // through the K descriptor and are masked to -inf in the last tile (tmask: bit kb*4+j = this lane's key (kb, j) is invalid)
    int qo[3];
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        const int qi = q0 + 16 * qb + c16;
        qo[qb] = ((qi < a.Tq ? qi : a.Tq - 1) * (int)a.ldq + g * 8) * 2;
    }
    // epilogue: the wave's [48 rows][256 B] output block goes through LDS so that the stores are whole rows (the assembly derives the
    // lane geometry itself); eso = byte offset of the wave's first output row, o1 = one row, o4 = four rows
    const uint32_t eso = (uint32_t)((long)q0 * ldo * 2), o1 = (uint32_t)(ldo * 2), o4 = (uint32_t)(4 * ldo * 2);
    const uint32_t nt = (uint32_t)((Tk + KV_TILE - 1) / KV_TILE);
    const uint32_t rag = (uint32_t)__builtin_amdgcn_readfirstlane((Tk % KV_TILE) != 0 ? 1 : 0);
    uint32_t tmask = 0;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((int)(nt - 1) * KV_TILE + 32 * (kb >> 1) + 4 * (kb & 1) + 8 * g + j >= Tk) tmask |= 1u << (kb * 4 + j);
    const uint32_t oblo = (uint32_t)(uintptr_t)Ob, obhi = (uint32_t)((uintptr_t)Ob >> 32);
    const uint32_t orec = (uint32_t)(((long)(a.Tq - 1) * ldo + 128) * 2);
    const uint32_t kblo = (uint32_t)(uintptr_t)Kb, kbhi = (uint32_t)((uintptr_t)Kb >> 32);
    const uint32_t vblo = (uint32_t)(uintptr_t)Vb, vbhi = (uint32_t)((uintptr_t)Vb >> 32);
    const uint32_t krec = (uint32_t)(((long)(Tk - 1) * a.ldk + 128) * 2), vrec = (uint32_t)(((long)128 * a.ldvt - s0) * 2);
    const uint32_t ktb = (uint32_t)(KV_TILE * a.ldk * 2);
    // PRESCALED: Q carries scale * log2(e) already, the scores are base-2 exponents (c = 1, the stream has no multiplies)
    const float c = PRESCALED ? 1.0f : a.scale * 1.4426950408889634f;
    const uint32_t wlds = (uint32_t)wave * 1024u;
    const float tau = 8.0f / c;  // raw-score threshold: the reference maximum of a query is raised only when exp2((s - ref)*c) > 2^8
    // masked variant: the bias vector of this batch element goes to LDS once (16 KB after the ring, Tk <= 4096); a lane's 16 values
    // of a tile sit at ba + 256 * tile
    const float* biasb = HAS_BIAS ? a.bias + (long)b * a.bias_bstride + s0 : nullptr;
    const uint32_t bilo = (uint32_t)(uintptr_t)biasb, bihi = (uint32_t)((uintptr_t)biasb >> 32), birec = (uint32_t)(Tk * 4);
    const float isc = PRESCALED ? 1.4426950408889634f : 1.0f / a.scale;  // bias (post-scale, natural log) -> score units
    const int bvo = lane * 16;
    const int ba = W48_LDS + g * 32;
#ifdef W48_STAMPS  // tools/ubench/attn_stamps.hip: per-wave s_memtime stamps of one tile step -> g_w48_stamps[wave][5]
    unsigned long long* dbg = (blk.x == 1 && head == 3) ? &g_w48_stamps[wave][0] : &g_w48_stamps[4 + (wave & 3)][0];
#endif
#define W48_OPERANDS                                                                                                                  \
    [qbase] "s"(Qb), [oblo] "s"(oblo), [obhi] "s"(obhi), [orec] "s"(orec), [rag] "s"(rag), [tmask] "v"(tmask), [kblo] "s"(kblo),         \
        [kbhi] "s"(kbhi), [vblo] "s"(vblo), [vbhi] "s"(vbhi), [krec] "s"(krec), [vrec] "s"(vrec), [ktb] "s"(ktb), [nt] "s"(nt),          \
        [c] "s"(c), [tau] "s"(tau), [wlds] "s"(wlds), [qo0] "v"(qo[0]), [qo1] "v"(qo[1]), [qo2] "v"(qo[2]), [eso] "s"(eso),               \
        [o1] "s"(o1), [o4] "s"(o4), [ko0] "v"(L.koff[0]), [ko1] "v"(L.koff[1]), [ko2] "v"(L.koff[2]), [ko3] "v"(L.koff[3]),       \
        [vo0] "v"(L.voff[0]), [vo1] "v"(L.voff[1]), [vo2] "v"(L.voff[2]), [vo3] "v"(L.voff[3]), [ka0] "v"(L.kaddr[0]),                  \
        [ka1] "v"(L.kaddr[1]), [ka2] "v"(L.kaddr[2]), [ka3] "v"(L.kaddr[3]), [va0] "v"(L.vaddr[0]), [va1] "v"(L.vaddr[1]),              \
        [lse] "s"(lse_w)
    if constexpr (HAS_BIAS) {
        if constexpr (PRESCALED) {
            asm volatile(
#include "attention_w48_asm_bias_ps.inc"
                :
                : W48_OPERANDS, [bilo] "s"(bilo), [bihi] "s"(bihi), [birec] "s"(birec), [isc] "s"(isc), [bvo] "v"(bvo), [ba] "v"(ba)
                :
#include "attention_w48_bias_clobbers.inc"
            );
        } else {
            asm volatile(
#include "attention_w48_asm_bias.inc"
                :
                : W48_OPERANDS, [bilo] "s"(bilo), [bihi] "s"(bihi), [birec] "s"(birec), [isc] "s"(isc), [bvo] "v"(bvo), [ba] "v"(ba)
                :
#include "attention_w48_bias_clobbers.inc"
            );
        }
    } else if constexpr (PRESCALED) {
        asm volatile(
#ifdef W48_STAMPS
#include "attention_w48_asm_ps_stamps.inc"
#else
#include "attention_w48_asm_ps.inc"
#endif
            :
            : W48_OPERANDS
#ifdef W48_STAMPS
              , [dbg] "s"(dbg)
#endif
            :
#include "attention_w48_clobbers.inc"
        );
    } else {
        asm volatile(
#ifdef W48_STAMPS
#include "attention_w48_asm_stamps.inc"
#else
#include "attention_w48_asm.inc"
#endif
            :
            : W48_OPERANDS
#ifdef W48_STAMPS
              , [dbg] "s"(dbg)
#endif
            :
#include "attention_w48_clobbers.inc"
        );
    }
#undef W48_OPERANDS
    (void)smem;
}

// Key-split launches: O[b][t][h][:] = sum_z w_z O_z / sum_z w_z, w_z = exp2(lse_z - max_z lse_z) - the softmax over all keys from
// the per-range softmaxes (each O_z is normalised over its own keys, lse_z = log2 of that range's denominator in absolute units).
// One thread per 8 output channels (16 bytes of bf16); the slices are summed in ascending z.
__global__ __launch_bounds__(256) void attn_combine_kernel(const bf16_t* __restrict__ o_part, const float* __restrict__ lse, int Z, int B, int H,
                                                           int Tq, int Tq_pad, bf16_t* __restrict__ O, long ldo, long o_bstride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // ((b * Tq + t) * H + h) * 16 + c
    const long total = (long)B * Tq * H * 16;
    if (i >= total) return;
    const int c = (int)(i & 15);
    const int h = (int)((i >> 4) % H);
    const long bt = (i >> 4) / H;
    const int t = (int)(bt % Tq), b = (int)(bt / Tq);
    float ls[8];
    float mx = -INFINITY;
#pragma unroll
    for (int z = 0; z < 8; ++z) {   // Z <= 8 (attn_split_plan); unrolled so that ls[] stays in registers
        ls[z] = z < Z ? lse[(((long)z * B + b) * H + h) * Tq_pad + t] : -INFINITY;
        mx = fmaxf(mx, ls[z]);
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        if (z < Z) {
            const float w = exp2f(ls[z] - mx);
            den += w;
            const s16x8 v = *(const s16x8*)(o_part + (((long)z * B + b) * Tq + t) * ((long)H * 128) + h * 128 + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * bf16_to_f32((bf16_t)v[e]);
        }
    }
    const float inv = 1.0f / den;
    uint4 pk;
    pk.x = pack_bf16x2(acc[0] * inv, acc[1] * inv);
    pk.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
    pk.z = pack_bf16x2(acc[4] * inv, acc[5] * inv);
    pk.w = pack_bf16x2(acc[6] * inv, acc[7] * inv);
    *(uint4*)(O + (long)b * o_bstride + (long)t * ldo + h * 128 + c * 8) = pk;
}

#ifdef LTX_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------
// 32x32x16 stream, 48 queries per wave (round 3): one 32-query block per wave + half of a block shared by a wave pair, split by
// keys (tools/gen_attn_x32.py: layout, register map, pipeline). The LDS images and the K-row permutation are those of
// attn_fwd_kernel above (the plain-HIP 32x32 kernel pins them through the parity tests); C++ only prepares per-lane offsets.
// Takes unmasked launches with prescaled Q and Tk % 64 == 0; any Tq (rows past Tq are clamped on load and dropped by the O
// descriptor). MEASURED, NOT SELECTED (experiments build, option "attn_impl" = 5): its tile step takes 2071 cycles against 2428 of the
// 16x16x32 stream, but the part holds 1.47 GHz under it against 1.86 GHz (32x32x16 MFMAs draw more per FLOP): 495 vs 459 us at
// T = 6144, 40.2 vs 38.5 us at T = 1536, 3.65 vs 3.40 ms per DiT step (profiles/r03_attn_x32_stamps.txt).
// ---------------------------------------------------------------------------------------------------------------
constexpr int X32_Q = 192;
constexpr int X32_LDS = 4 * STAGE_BYTES;
constexpr int X32_XSZ = 0x2400;  // epilogue exchange area per wave (gen_attn_x32.py XSZ)
#ifdef X32_STAMPS
__device__ unsigned long long g_x32_stamps[8][16];
#endif

__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_x32_asm(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int hwv = wave & 1, pair = wave >> 1;
    const AttnBlock blk = attn_block((a.Tq + X32_Q - 1) / X32_Q, a.H, a.B, a.plain_order);
    const int head = blk.head, b = blk.b;
    const bf16_t* Qb = a.Q + (long)b * a.q_bstride + head * 128;
    const bf16_t* Kb = a.K + (long)b * a.k_bstride + head * 128;
    const bf16_t* Vb = a.Vt + (long)b * a.vt_bstride + (long)head * 128 * a.ldvt;
    bf16_t* Ob = a.O + (long)b * a.o_bstride + head * 128;
    const int q0 = blk.x * X32_Q;
    const int qi_own = q0 + 32 * wave + r, qi_sh = q0 + 128 + 32 * pair + r;
    const int qoo = ((qi_own < a.Tq ? qi_own : a.Tq - 1) * (int)a.ldq + h * 8) * 2;
    const int qos = ((qi_sh < a.Tq ? qi_sh : a.Tq - 1) * (int)a.ldq + h * 8) * 2;
    const int oow = (qi_own * (int)a.ldo + h * 4) * 2, oos = (qi_sh * (int)a.ldo + h * 4) * 2;
    // LDS-DMA pieces: wave w stages pieces w, w+4, w+8, w+12 of each image (1 KB each: 4 keys x 256 B / 8 d-rows x 128 B); the bank
    // swizzle sits on the SOURCE address: physical chunk p of key k holds logical chunk p ^ (k & 15), of d-row d chunk p ^ ((d>>1)&7)
    int ko[4], vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kl = (wave + 4 * i) * 4 + (lane >> 4);
        ko[i] = (kl * (int)a.ldk + (((lane & 15) ^ (kl & 15)) << 3)) * 2;
        const int d = (wave + 4 * i) * 8 + (lane >> 3);
        vo[i] = (d * (int)a.ldvt + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2;
    }
    // fragment addresses. K: MFMA row r reads key pi(r) (bits 2 and 3 swapped) of a 32-key half; half A = this wave's own (hw)
    const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
    const int ksw = pr & 15;
    const int kbase = pr * 256 + ((h ^ (ksw & 1)) << 4);
    const int kbA = kbase + hwv * 8192, kbB = kbase + (1 - hwv) * 8192;
    const int kx5 = (ksw >> 1) << 5;
    const int vsw = (r >> 1) & 7;
    int va[4];  // A0 A1 B0 B1: k-step (2 * half + s2) of the PV product
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int half = (i >> 1) ? 1 - hwv : hwv;
        const int kst = 2 * half + (i & 1);
        va[i] = K_TILE_BYTES + r * 128 + ((((kst << 1) + h) ^ vsw) << 4);
    }
    const int xout = wave * X32_XSZ + lane * 16, xin = (wave ^ 1) * X32_XSZ + lane * 16;
    const uint32_t nt = (uint32_t)(a.Tk / KV_TILE);
    const uint32_t oblo = (uint32_t)(uintptr_t)Ob, obhi = (uint32_t)((uintptr_t)Ob >> 32);
    const uint32_t orec = (uint32_t)(((long)(a.Tq - 1) * a.ldo + 128) * 2);
    const uint32_t kblo = (uint32_t)(uintptr_t)Kb, kbhi = (uint32_t)((uintptr_t)Kb >> 32);
    const uint32_t vblo = (uint32_t)(uintptr_t)Vb, vbhi = (uint32_t)((uintptr_t)Vb >> 32);
    const uint32_t krec = (uint32_t)(((long)(a.Tk - 1) * a.ldk + 128) * 2), vrec = (uint32_t)((long)128 * a.ldvt * 2);
    const uint32_t ktb = (uint32_t)(KV_TILE * a.ldk * 2);
    const uint32_t wlds = (uint32_t)wave * 1024u;
    const uint32_t hws = (uint32_t)hwv;
    const float tau = 8.0f;
#ifdef X32_STAMPS
    unsigned long long* dbg = (blk.x == 1 && head == 3) ? &g_x32_stamps[wave][0] : &g_x32_stamps[4 + (wave & 3)][0];
#endif
    asm volatile(
#ifdef X32_STAMPS
#include "attention_x32_asm_stamps.inc"
#else
#include "attention_x32_asm.inc"
#endif
        :
        : [qbase] "s"(Qb), [oblo] "s"(oblo), [obhi] "s"(obhi), [orec] "s"(orec), [kblo] "s"(kblo), [kbhi] "s"(kbhi), [vblo] "s"(vblo),
          [vbhi] "s"(vbhi), [krec] "s"(krec), [vrec] "s"(vrec), [ktb] "s"(ktb), [nt] "s"(nt), [tau] "s"(tau), [wlds] "s"(wlds),
          [hw] "s"(hws), [qoo] "v"(qoo), [qos] "v"(qos), [oow] "v"(oow), [oos] "v"(oos), [ko0] "v"(ko[0]), [ko1] "v"(ko[1]),
          [ko2] "v"(ko[2]), [ko3] "v"(ko[3]), [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [kbA] "v"(kbA),
          [kbB] "v"(kbB), [kx5] "v"(kx5), [vaA0] "v"(va[0]), [vaA1] "v"(va[1]), [vaB0] "v"(va[2]), [vaB1] "v"(va[3]), [xout] "v"(xout),
          [xin] "v"(xin)
#ifdef X32_STAMPS
          , [dbg] "s"(dbg)
#endif
        :
#include "attention_x32_clobbers.inc"
    );
    (void)smem;
}

#endif  // LTX_EXPERIMENTS

}  // namespace

// Key split of a launch of the 48-query kernel: only when its (query block, head, batch) workgroups leave most of the chip idle and
// every range still has two key tiles; at most 8 ranges (attn_combine_kernel's register array), as many as fill 256 CUs.
struct AttnSplitPlan {
    int splits = 1, keys = 0;
    long o_bytes = 0, bytes = 0;
};
static AttnSplitPlan attn_split_plan(int B, int H, int Tq, int Tk) {
    AttnSplitPlan p;
    const long wgs = (long)((Tq + W48_Q - 1) / W48_Q) * H * B;
    if (wgs > 128 || Tk < 4 * KV_TILE) return p;
    int z = (int)(256 / wgs);
    if (z > 8) z = 8;
    if (z > Tk / (2 * KV_TILE)) z = Tk / (2 * KV_TILE);
    if (z < 2) return p;
    const int keys = ((Tk + z - 1) / z + KV_TILE - 1) / KV_TILE * KV_TILE;
    z = (Tk + keys - 1) / keys;  // no empty range
    if (z < 2) return p;
    p.splits = z;
    p.keys = keys;
    p.o_bytes = ((long)z * B * Tq * H * 128 * 2 + 255) / 256 * 256;
    p.bytes = p.o_bytes + (long)z * B * H * ((Tq + W48_Q - 1) / W48_Q * W48_Q) * 4;
    return p;
}
long attn_split_ws_bytes(int B, int H, int Tq, int Tk) { return attn_split_plan(B, H, Tq, Tk).splits > 1 ? attn_split_plan(B, H, Tq, Tk).bytes : 0; }
int attn_key_splits(int B, int H, int Tq, int Tk) { return attn_split_plan(B, H, Tq, Tk).splits; }

void launch_attention(const AttnArgs& a_in, hipStream_t stream) {
    AttnArgs a = a_in;
    // q_prescaled: every kernel but the prescaled assembly stream computes exp2(score * scale * log2(e)); with Q carrying
    // scale * log2(e) already that factor must be 1, i.e. scale = ln 2
    if (a.q_prescaled) a.scale = 0.6931471805599453f;
    const bool plain = ltx_opt(OPT_ATTN_PLAIN_ORDER) != 0;  // A/B option "attn_plain_order": (query block, head, batch) workgroup order as before round 3
    if (plain) a.plain_order = 1;
    LTX_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: empty problem");
    LTX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 4 == 0, "attention: leading dims");
    LTX_REQUIRE(a.ldvt >= ((a.Tk + 63) / 64) * 64, "attention: Vt row stride %ld must cover Tk=%d rounded up to 64", a.ldvt, a.Tk);
    LTX_REQUIRE(((uintptr_t)a.Q & 15) == 0 && ((uintptr_t)a.K & 15) == 0 && ((uintptr_t)a.Vt & 15) == 0 && ((uintptr_t)a.O & 7) == 0,
                "attention: pointer alignment");
    static PerDeviceOnce attr_set;
    attr_set.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
    });
    ProfScope prof(PROF_ATTN, 4.0 * a.B * a.H * (double)a.Tq * a.Tk * 128, stream);
    // Kernel choice by grid fill. Tile-step costs measured on MI355X at T=6144: 2.3 us with two 128-query workgroups on a CU,
    // 1.5 us with one; 2.18 us for one 256-query ping-pong workgroup; 1.25 us for one 192-query workgroup of the 48-query
    // assembly kernel. The launch lasts as long as its fullest CU:
    //   T=1536, B=1: 384 4-wave workgroups (2.3) | 192 ping-pong (2.18) | 256 x 192-query (1.25)     -> assembly (42 vs 59 / 65 us)
    //   T=1536, B=2: 768 (2.3 + 1.5)             | 384 (4.36)           | 512 (2.5)                   -> assembly (82 vs 114 / 100 us)
    //   T=6144:      1536 (6.9)                  | 768 (6.54)           | 1024 (5.0)                  -> assembly (491 vs 641 / 658 us)
    // The assembly kernel takes unmasked launches with Tq % 192 == 0 and Tk % 256 == 0 (the DiT's self- and cross-attention at
    // every BASELINE configuration); masked launches stay on the 4-wave kernel (its per-key bias loads sit in the softmax phase
    // of the ping-pong kernel, measured slower). Option "attn_impl" = 1 / 2 / 4 forces the 4-wave / ping-pong / assembly kernel, 3 the
    // plain-HIP layout reference of the assembly kernel (same-process A/B runs).
    {
        const long wg4 = (long)((a.Tq + 127) / 128) * a.H * a.B;
        const long rem4 = wg4 % 512;
        const double cost4 = (double)(wg4 / 512) * 2.3 + (rem4 == 0 ? 0.0 : (rem4 <= 256 ? 1.5 : 2.3));
        const long wgpp = (long)((a.Tq + PP_Q - 1) / PP_Q) * a.H * a.B;
        const double costpp = (double)((wgpp + 255) / 256) * 2.18;
        const int impl_opt = ltx_opt(OPT_ATTN_IMPL);  // option "attn_impl"
        const char impl_c[2] = {(char)('0' + impl_opt), 0};
        const char* impl = impl_opt ? impl_c : nullptr;
        const bool forced = impl && impl[0] >= '1' && impl[0] <= '4';
        bool use_pp = forced ? impl[0] == '2' : (!a.bias && costpp < cost4);
        // the 48-query kernels cover unmasked launches whose query count is a multiple of 192 and key count a multiple of 256
        const bool w48_ok = !a.bias && a.Tq % W48_Q == 0 && a.Tk % (4 * KV_TILE) == 0;
#ifdef LTX_EXPERIMENTS
        if (impl && impl[0] == '3') {
            LTX_REQUIRE(w48_ok, "attention: attn_impl = 3 needs Tq %% 192 == 0, Tk %% 256 == 0 and no mask (Tq=%d Tk=%d)", a.Tq, a.Tk);
            static PerDeviceOnce attr3_set;
            attr3_set.run([&] {
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_w48_ref, hipFuncAttributeMaxDynamicSharedMemorySize, W48_LDS));
            });
            hipLaunchKernelGGL(attn_fwd_kernel_w48_ref, dim3(a.Tq / W48_Q, a.H, a.B), dim3(256), W48_LDS, stream, a);
            HIP_CHECK(hipGetLastError());
            return;
        }
#endif
        const long wg48 = (long)((a.Tq + W48_Q - 1) / W48_Q) * a.H * a.B;
        const double cost48 = (double)((wg48 + 255) / 256) * 1.25;
#ifdef LTX_EXPERIMENTS
        // the 32x32 stream ("attn_impl" = 5 only: measured slower than the 16x16 stream on every BASELINE shape)
        if (impl && impl[0] == '5') {
            const bool x32_ok = !a.bias && a.q_prescaled && a.Tk % KV_TILE == 0;
            LTX_REQUIRE(x32_ok, "attention: attn_impl = 5 takes unmasked launches with prescaled Q and Tk %% 64 == 0 (Tq=%d Tk=%d)", a.Tq, a.Tk);
            static PerDeviceOnce attr5_set;
            attr5_set.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_x32_asm, hipFuncAttributeMaxDynamicSharedMemorySize, X32_LDS)); });
            hipLaunchKernelGGL(attn_fwd_kernel_x32_asm, dim3(((a.Tq + X32_Q - 1) / X32_Q) * a.H * a.B), dim3(256), X32_LDS, stream, a);
            HIP_CHECK(hipGetLastError());
            return;
        }
#endif
        const bool asm_ok = !a.bias || a.Tk <= 4096;  // any Tq, Tk (ragged tails in the kernel); masked: the bias vector must fit 16 KB of LDS
        const bool use_asm = forced ? impl[0] == '4' : (asm_ok && cost48 < cost4 && cost48 < costpp);
        if (use_asm) {
            LTX_REQUIRE(asm_ok, "attention: attn_impl = 4 takes masked launches up to 4096 keys only (Tq=%d Tk=%d)", a.Tq, a.Tk);
            static PerDeviceOnce attr4_set;
            attr4_set.run([&] {
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_w48_asm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W48_LDS));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_w48_asm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W48_LDS + 16384));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_w48_asm<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W48_LDS));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_w48_asm<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W48_LDS + 16384));
            });
            // few queries against many keys: divide the keys over workgroups (attn_split_plan) when the caller lent a workspace
            AttnArgs k = a;
            k.key_splits = 1;
            const AttnSplitPlan plan = attn_split_plan(a.B, a.H, a.Tq, a.Tk);
            const bool no_split = ltx_opt(OPT_ATTN_NO_SPLIT) != 0;  // A/B option "attn_no_split"
            if (plan.splits > 1 && a.split_ws && a.split_ws_bytes >= plan.bytes && !no_split && ((uintptr_t)a.split_ws & 15) == 0 &&
                ((uintptr_t)a.O & 15) == 0 && a.ldo % 8 == 0 && a.o_bstride % 8 == 0) {
                k.key_splits = plan.splits;
                k.split_keys = plan.keys;
                k.o_part = (bf16_t*)a.split_ws;
                k.lse = (float*)((char*)a.split_ws + plan.o_bytes);
            }
            const dim3 grid4(((a.Tq + W48_Q - 1) / W48_Q) * a.H * a.B, k.key_splits);
            if (a.bias && a.q_prescaled)
                hipLaunchKernelGGL((attn_fwd_kernel_w48_asm<true, true>), grid4, dim3(256), W48_LDS + 16384, stream, k);
            else if (a.bias)
                hipLaunchKernelGGL((attn_fwd_kernel_w48_asm<true, false>), grid4, dim3(256), W48_LDS + 16384, stream, k);
            else if (a.q_prescaled)
                hipLaunchKernelGGL((attn_fwd_kernel_w48_asm<false, true>), grid4, dim3(256), W48_LDS, stream, k);
            else
                hipLaunchKernelGGL((attn_fwd_kernel_w48_asm<false, false>), grid4, dim3(256), W48_LDS, stream, k);
            HIP_CHECK(hipGetLastError());
            if (k.key_splits > 1) {
                const long total = (long)a.B * a.Tq * a.H * 16;
                hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, k.o_part, k.lse, k.key_splits, a.B, a.H,
                                   a.Tq, (a.Tq + W48_Q - 1) / W48_Q * W48_Q, a.O, a.ldo, a.o_bstride);
                HIP_CHECK(hipGetLastError());
            }
            return;
        }
        if (use_pp) {
            static PerDeviceOnce attr2_set;
            attr2_set.run([&] {
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_pp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel_pp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
            });
            dim3 grid_pp(((a.Tq + PP_Q - 1) / PP_Q) * a.H * a.B);
            if (a.bias)
                hipLaunchKernelGGL((attn_fwd_kernel_pp<true>), grid_pp, dim3(512), PP_LDS, stream, a);
            else
                hipLaunchKernelGGL((attn_fwd_kernel_pp<false>), grid_pp, dim3(512), PP_LDS, stream, a);
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    dim3 grid(((a.Tq + 127) / 128) * a.H * a.B);
    const int lds = 2 * STAGE_BYTES;
    if (a.bias)
        hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, dim3(256), lds, stream, a);
    HIP_CHECK(hipGetLastError());
}
