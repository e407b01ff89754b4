// gemm.hip - bf16 MFMA GEMM for gfx950 (CDNA4): C[M,N] = A[M,K] * B[N,K]^T, f32 accumulate, fused epilogues.
//
// This is the kernel that carries ~92 % of the DiT step's FLOPs (reference call sites: every `Linear` of
// LTXAttention.swift:171-175,217, LTXFeedForward.swift:26,49, LTXTransformer.swift:257,223) and, through the
// implicit-im2col A loader, every Conv3dFull of the VAE decoder (VideoConvolution.swift:238-347).
//
// Design (MI355X-first, not a translation of anything):
//   * 256-thread workgroup = 4 waves (2x2), each wave owns a (BM/2)x(BN/2) output sub-tile as
//     v_mfma_f32_16x16x32_bf16 accumulators.
//   * K is walked in 64-element (128-byte) tiles. Both operands are K-contiguous, so A and B tiles are the same
//     LDS image: [rows][128 B], filled by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction = 8 rows)
//     with the bank swizzle applied on the per-lane SOURCE address (16-B chunk c of row r lives at chunk
//     c ^ ((r>>1)&7)); fragments come back with conflict-free ds_read_b128.
//   * two LDS stages, one barrier per K-tile (prefetch tile t+1 while the MFMAs of tile t run).
//   * epilogue through a per-wave LDS transpose so global stores are 16 B/lane, row-contiguous.
//   * workgroup ids are remapped so that each XCD's L2 sees a contiguous chunk of the tile grid.
#include <type_traits>

#include "gemm.h"
#include "elementwise.h"

#include <stdlib.h>
#include "options.h"
#include "runtime.h"

namespace {

constexpr int BK = 64;            // bf16 elements per K-tile
constexpr int ROW_BYTES = BK * 2;  // 128 B per tile row

struct RowPos {  // conv3d: decoded output position of an A row
    int f, y, x;
};

LTX_DEVFN int reflect_idx(int i, int n) {
    // reflect padding by 1 (VideoConvolution.swift:258-266): -1 -> 1, n -> n-2
    i = (i < 0) ? -i : i;
    return (i >= n) ? (2 * n - 2 - i) : i;
}
LTX_DEVFN int clamp_idx(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
// x / d == __umulhi(x, magic_u32(d)) for every x with x * d < 2^32 (d >= 2): the runtime divisors of the conv index arithmetic (W, H,
// slab rows) as multiply-high constants - an integer division by a runtime value is ~35 VALU instructions, and a tile needs dozens
// (d == 1 has no 32-bit constant - callers divide by 1 themselves; W, H >= 2 is checked by the launcher)
LTX_DEVFN unsigned magic_u32(int d) { return (unsigned)((0x100000000ull + (unsigned long long)(d - 1)) / (unsigned long long)d); }

// compile-time loop: the body gets an integral_constant, so accumulator arrays are only ever indexed with constants
// (a `#pragma unroll` that the optimiser declines leaves a runtime index and demotes the array to scratch memory)
template <int I, int N, class F>
LTX_DEVFN void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Workgroup id -> output tile. Supertiles of group_m row-tiles x all column-tiles, row-tile fastest inside: the chunk
// of consecutive workgroups one XCD receives (xcd_remap) then spans group_m row-tiles x (chunk/group_m) column-tiles
// instead of every row-tile x few columns. Every XCD's L2 fetches each operand tile it touches once from the fabric, so
// this trades re-reads of A (rows) against re-reads of B (columns): measured on MI355X with group_m = 4 vs column-major:
// 6144x4096x4096 833 -> 1039 TFLOP/s, 4096^3 897 -> 1017, 1536x4096x16384 1099 -> 1146 (PMC: A was fetched 8x).
// conv launches with one column tile: position L of the launch order -> row tile, blocked 8 row-groups x 4 frames (Conv3dGeom::blk_rg)
LTX_DEVFN int conv_block_order(int L, int F, int RG) {
    constexpr int FB = 4;
    const int RB = RG < 8 ? RG : 8;
    const int fgi = L / (FB * RG);
    const int f0 = fgi * FB;
    const int nf = (F - f0) < FB ? (F - f0) : FB;
    const int w = L - fgi * FB * RG;
    const int per = nf * RB;
    const int rbi = w / per, v = w - rbi * per;
    const int fo = v / RB, ro = v - fo * RB;
    return (f0 + fo) * RG + rbi * RB + ro;
}

LTX_DEVFN void tile_coords(const GemmArgs& g, int bid, int tiles_m, int BN_, int& tm, int& tn) {
    if (g.group_m > 0 && g.group_m < tiles_m) {
        const int tiles_n = (g.N + BN_ - 1) / BN_;
        const int per = g.group_m * tiles_n;
        const int grp = bid / per, rem = bid - grp * per;
        const int gm = (tiles_m - grp * g.group_m) < g.group_m ? (tiles_m - grp * g.group_m) : g.group_m;
        tn = rem / gm;
        tm = grp * g.group_m + (rem - tn * gm);
    } else {
        tm = bid % tiles_m;
        tn = bid / tiles_m;
    }
}

// ---- epilogue: per-wave LDS transpose (16 rows at a time), then 16-B row-contiguous global accesses ----
// `get(mi_c, slab)` hands over the 16-row slab mi of this wave's accumulators (f32x4 slab[NI]) - an array copy for the kernels
// that accumulate in VGPRs, an AGPR read-out for the assembly kernel (whose 192 accumulator registers must never be live in
// VGPRs all at once next to this function's prefetch buffers).
// Residual-stream values of this wave's whole tile, loaded by the kernel under its last K-tiles (`gemm_residual_prefetch`) so that
// the epilogue of a gated-residual GEMM - one workgroup per CU, nothing else resident - does not start with an exposed HBM read.
// NSL: how many of the wave's 16-row slabs are held (all of them by default). A kernel whose registers are too few to hold the whole tile
// next to its accumulators keeps slab 0 only (the persistent conv kernel: 16 instead of 48 registers) and the epilogue fetches the
// others itself, one slab ahead.
template <int BM, int BN, int WGM, int WGN, int NSL = BM / WGM / 16>
struct ResidualTile {
    static constexpr int MI = NSL, NSLABS = NSL, NIT = (16 * (BN / WGN / 4)) / 64;
    f32x4 v[MI][NIT];
    f32x4 bias[NIT];          // the column bias: what the epilogue's first slab waits on in launches without a residual
    bool valid = false;       // v holds data
    bool bias_valid = false;  // bias holds data
};

// Column bias of this wave's tile (every launch with interior columns), fetched under the last K-tiles as well.
template <int BM, int BN, int WGM, int WGN, int NSL>
LTX_DEVFN void gemm_bias_prefetch(const GemmArgs& g, int n0, int wc, int lane, ResidualTile<BM, BN, WGM, WGN, NSL>& rt) {
    constexpr int WN = BN / WGN, LPR = WN / 4, NIT = (16 * LPR) / 64;
    const int gn_w = n0 + wc * WN;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c4 = ((it * 64 + lane) % LPR) * 4;
        rt.bias[it] = g.ep.bias_n ? *(const f32x4*)(g.ep.bias_n + gn_w + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    rt.bias_valid = true;
}

template <int BM, int BN, int WGM, int WGN, int NSL>
LTX_DEVFN void gemm_residual_prefetch(const GemmArgs& g, int m0, int n0, int wr, int wc, int lane, ResidualTile<BM, BN, WGM, WGN, NSL>& rt) {
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = NSL, LPR = WN / 4, NIT = (16 * LPR) / 64;
    const GemmEpilogue& ep = g.ep;
    const float* rbase = ep.resid_src ? ep.resid_src : ep.out_f32;
    const long rld = ep.resid_src ? ep.ld_resid : ep.ld_f32;
    const int gn_w = n0 + wc * WN;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int chunk = it * 64 + lane;
            int gm = m0 + wr * WM + mi * 16 + chunk / LPR;
            gm = gm < g.M ? gm : g.M - 1;
            rt.v[mi][it] = *(const f32x4*)(rbase + (long)gm * rld + gn_w + (chunk % LPR) * 4);
        }
    rt.valid = true;
}

// PN: the fused PixelNorm + SiLU second output (GemmEpilogue::pn_out) is compiled in (conv instantiations only: it adds a
// workgroup barrier per 16-row slab and registers the dense kernels' epilogues do not have to spare).
// SCR: `get(mi_c, scr)` writes the 16 x WN slab row-major into this wave's LDS scratch itself (accumulators that are not in the
// 16x16 MFMA layout: the 32x32x16 kernel); the transposed store, which works from the 16x16 register layout, is then not available.
// SGATE: the launch never carries a gate VECTOR (conv launches: residual gate = gate_scalar) - the interior path then keeps no gate
// registers (32 per lane in the general form, which the persistent conv kernel does not have to spare).
// `hook()`: called once, at the point of the epilogue behind which it issues no further global LOAD (the interior path: right after the
// residual fetch of the last slab has been requested). A persistent kernel requests its next tile's first operands there: vmcnt retires
// loads in order, so anything requested earlier would be waited for by every later residual fetch of this epilogue.
struct NoHook { LTX_DEVFN void operator()() const {} };
template <int BM, int BN, int WGM = 2, int WGN = 2, bool PN = false, bool SCR = false, bool SGATE = false, class Get, class Pre = ResidualTile<BM, BN, WGM, WGN>,
          class Hook = NoHook>
LTX_DEVFN void gemm_epilogue_with(Get&& get, const GemmArgs& g, int m0, int n0, int wr, int wc, int lane, int wave, char* smem,
                                  const Pre* pre = nullptr, Hook&& hook = Hook{}) {
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    float* scr = (float*)(smem + wave * (16 * WN * 4));
    constexpr int LPR = WN / 4;    // lanes per output row
    const GemmEpilogue& ep = g.ep;
    if constexpr (!SCR)
    if (ep.out_bf16_t) {
        hook();
        // transposed bf16 store from the accumulator layout: acc[mi][ni][r] = C[16 mi + 4 (lane >> 4) + r][16 ni + (lane & 15)]
        static_for<0, MI>([&](auto mi_c) {
            constexpr int mi = decltype(mi_c)::value;
            f32x4 slab[NI];
            get(mi_c, slab);
            const int gm0 = m0 + wr * WM + mi * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int gn = n0 + wc * WN + ni * 16 + (lane & 15);
                if (gn >= g.N || gm0 >= g.M) continue;
                const float bn = ep.bias_n ? ep.bias_n[gn] : 0.f;
                bf16_t* o = ep.out_bf16_t + (long)gn * ep.ld_bf16_t + gm0;
                if (gm0 + 3 < g.M) {
                    uint2 pk;
                    pk.x = pack_bf16x2(slab[ni][0] + bn, slab[ni][1] + bn);
                    pk.y = pack_bf16x2(slab[ni][2] + bn, slab[ni][3] + bn);
                    *(uint2*)o = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gm0 + r < g.M) o[r] = f32_to_bf16(slab[ni][r] + bn);
                }
            }
        });
        return;
    }
    if (!ep.d2s && n0 + BN <= g.N) {
        // Interior columns (every DiT GEMM): branch-free 16-B accesses, and the residual-stream / gate reads of slab
        // mi+1 are issued BEFORE slab mi's LDS transpose so that their HBM/MALL latency overlaps it. One workgroup
        // per CU runs this tail with nothing else resident, so a dependent load -> fma -> store chain per 16-row slab
        // (the general path below) was costing 12-20 us per gated-residual GEMM (76.8 vs 57.1 us in the DiT block trace).
        constexpr int NIT = (16 * LPR) / 64;
        const bool has_res = ep.resid != 0;
        const bool pre_res = pre && pre->valid;  // workgroup-uniform
        const int gn_w = n0 + wc * WN;
        f32x4 bias[NIT], rs[2][NIT], gt[2][NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = ((it * 64 + lane) % LPR) * 4;
            if (pre && pre->bias_valid)
                bias[it] = pre->bias[it];
            else
                bias[it] = ep.bias_n ? *(const f32x4*)(ep.bias_n + gn_w + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* rbase = ep.resid_src ? ep.resid_src : ep.out_f32;
        const long rld = ep.resid_src ? ep.ld_resid : ep.ld_f32;
        // the usual case - all rows of this wave's tile in one batch element - needs one gate vector, not an integer division
        // per lane and slab: load it once, beside the bias
        const int row_lo = (m0 + wr * WM) < g.M ? (m0 + wr * WM) : g.M - 1;
        const int row_hi = (m0 + wr * WM + WM - 1) < g.M ? (m0 + wr * WM + WM - 1) : g.M - 1;
        const int gb_lo = row_lo / ep.rows_per_batch;
        const bool gate_uniform = !SGATE && has_res && ep.gate && !ep.gate_rowmap && gb_lo == row_hi / ep.rows_per_batch;
        f32x4 gtu[NIT];
        if (gate_uniform) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) gtu[it] = *(const f32x4*)(ep.gate + (long)gb_lo * ep.gate_bstride + gn_w + ((it * 64 + lane) % LPR) * 4);
        }
        auto prefetch = [&](auto mi_c, auto buf_c) {
            constexpr int mi = decltype(mi_c)::value, buf = decltype(buf_c)::value;
            if (!has_res) return;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int chunk = it * 64 + lane;
                const int c4 = (chunk % LPR) * 4;
                int gm = m0 + wr * WM + mi * 16 + chunk / LPR;
                gm = gm < g.M ? gm : g.M - 1;
                bool held = false;
                if constexpr (mi < Pre::NSLABS) {
                    if (pre_res) {
                        rs[buf][it] = pre->v[mi][it];
                        held = true;
                    }
                }
                if (!held) rs[buf][it] = *(const f32x4*)(rbase + (long)gm * rld + gn_w + c4);
                if constexpr (SGATE) {
                    // (nothing: the scalar gate is applied from ep.gate_scalar below)
                } else if (gate_uniform)
                    gt[buf][it] = gtu[it];
                else if (ep.gate)
                    gt[buf][it] = *(const f32x4*)(ep.gate + (long)(ep.gate_rowmap ? ep.gate_rowmap[gm] : gm / ep.rows_per_batch) * ep.gate_bstride + gn_w + c4);
                else
                    gt[buf][it] = f32x4{ep.gate_scalar, ep.gate_scalar, ep.gate_scalar, ep.gate_scalar};
            }
        };
        prefetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        // fused PixelNorm: the two waves of a tile row-block (wc = 0 / 1) each hold 64 of a row's 128 channels; they trade their
        // partial sums of squares through LDS, double-buffered by slab parity so that one barrier per slab is enough
        const bool pn = PN && ep.pn_out != nullptr;  // workgroup-uniform
        float* pns = (float*)(smem + WGM * WGN * (16 * WN * 4));  // [2][waves][16 rows], behind the transpose scratch
        f32x4 pn_s4 = f32x4{1.f, 1.f, 1.f, 1.f}, pn_h4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PN) {
            static_assert(WGN == 2 && LPR == 16, "fused PixelNorm: two waves per tile row, 16 lanes per 64-column half row");
            if (pn && ep.pn_scale) {
                pn_s4 = *(const f32x4*)(ep.pn_scale + gn_w + (lane % LPR) * 4);
                pn_h4 = *(const f32x4*)(ep.pn_shift + gn_w + (lane % LPR) * 4);
            }
        }
        static_for<0, MI>([&](auto mi_c) {
            constexpr int mi = decltype(mi_c)::value;
            constexpr int buf = mi & 1;
            if constexpr (mi + 1 < MI) prefetch(std::integral_constant<int, mi + 1>{}, std::integral_constant<int, (mi + 1) & 1>{});
            if constexpr (mi == (MI >= 2 ? MI - 2 : 0)) hook();  // the last residual fetch has just been requested
            if constexpr (SCR) {
                get(mi_c, scr);
            } else {
                f32x4 slab[NI];
                get(mi_c, slab);
                static_for<0, NI>([&](auto ni_c) {
                    constexpr int ni = decltype(ni_c)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[((lane >> 4) * 4 + r) * WN + ni * 16 + (lane & 15)] = slab[ni][r];
                });
            }
            f32x4 pn_v[PN ? NIT : 1];
            float pn_s2[PN ? NIT : 1];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int chunk = it * 64 + lane;
                const int row = chunk / LPR;
                const int c4 = (chunk % LPR) * 4;
                f32x4 v = *(const f32x4*)(scr + row * WN + c4);
                const int gm = m0 + wr * WM + mi * 16 + row;
                const int gmc = gm < g.M ? gm : g.M - 1;
                v += bias[it];
                if (ep.bias_m) {
                    const float bm = ep.bias_m[gmc];
                    v += f32x4{bm, bm, bm, bm};
                }
                if (ep.act == LTX_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                } else if (ep.act == LTX_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                if (ep.round_bf16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = rs[buf][it][e] + (SGATE ? ep.gate_scalar : gt[buf][it][e]) * v[e];
                }
#ifdef EPI_NO_STORE  // tools/ubench/gemm_stamps.hip timing experiment
                if (gm < 0) {
#else
                if (gm < g.M) {
#endif
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    if (ep.out_f32) *(f32x4*)(ep.out_f32 + (long)gm * ep.ld_f32 + gn_w + c4) = v;
                    if (ep.out_bf16) *(uint2*)(ep.out_bf16 + (long)gm * ep.ld_bf16 + gn_w + c4) = pk;
                }
                if constexpr (PN) {
                    if (pn) {
                        pn_v[it] = v;
                        pn_s2[it] = row16_allsum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);  // this wave's 64 channels of row it*4 + lane/16
                    }
                }
            }
            if constexpr (PN) {
                if (pn) {
                    float* mine = pns + (buf * WGM * WGN + wave) * 16;
                    const float* other = pns + (buf * WGM * WGN + (wave ^ 1)) * 16;
                    if ((lane & (LPR - 1)) == 0) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) mine[it * (64 / LPR) + (lane / LPR)] = pn_s2[it];
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int chunk = it * 64 + lane;
                        const int row = chunk / LPR;
                        const int c4 = (chunk % LPR) * 4;
                        const int gm = m0 + wr * WM + mi * 16 + row;
                        const float tot = pn_s2[it] + other[row];
                        const float inv = __builtin_amdgcn_rsqf(tot * (1.0f / (float)BN) + 1e-8f);
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = silu_f(pn_v[it][e] * inv * pn_s4[e] + pn_h4[e]);
                        if (gm < g.M) {
                            uint2 pk;
                            pk.x = pack_bf16x2(y[0], y[1]);
                            pk.y = pack_bf16x2(y[2], y[3]);
                            *(uint2*)(ep.pn_out + (long)gm * ep.ld_pn + gn_w + c4) = pk;
                        }
                    }
                }
            }
        });
        return;
    }
    // ---- depth-to-space store of the VAE upsampler (d2s == 1), interior columns, a wave's columns inside ONE sub-position: the
    // general path below spends ~25 us per tile here (three runtime divisions per 4-wide chunk, four dependent gather loads of the tiled
    // D2S residual, nothing prefetched - tools/ubench/conv_stamps.hip, CONV_D2S=1: epilogue 32.5 us against 7.2 us of a plain store).
    // Here the column part (sub-position, channel, residual column) is per-lane constant, a row splits into (frame, y, x) with two
    // multiply-high forms, and the residual gathers of slab mi + 1 are requested before slab mi is transposed.
    if (ep.d2s == 1 && n0 + BN <= g.N && ep.out_f32 && !ep.out_bf16 && !ep.bias_m && ep.act == LTX_ACT_NONE && !ep.round_bf16 &&
        ((g.N >> 3) % WN) == 0 && ((g.geom.C >> 3) & 3) == 0 && (long)g.M * g.geom.W < (1L << 32) &&
        (long)g.geom.F * g.geom.H * g.geom.H < (1L << 32) && g.geom.H >= 2 && g.geom.W >= 2) {  // (magic_u32(1) does not exist: round-4 advice)
        constexpr int NIT = (16 * LPR) / 64, RPI = 64 / LPR;
        const int cout = g.N >> 3, cd2s = g.geom.C >> 3;
        const int H = g.geom.H, W = g.geom.W;
        const unsigned mg_w = magic_u32(W), mg_h = magic_u32(H);
        const int gn = n0 + wc * WN + (lane % LPR) * 4;
        const int sub = gn / cout, c = gn - sub * cout;  // the wave's 64 columns share `sub` (cout is a multiple of WN)
        const int dt = sub >> 2, dh = (sub >> 1) & 1, dw = sub & 1;
        const int cm = (cd2s & (cd2s - 1)) == 0 ? (c & (cd2s - 1)) : (c % cd2s);
        // (if / else on VALUES: a ?: between `pre->bias[0]` and a global load becomes a select between two pointers, and the kernel's
        // ResidualTile then cannot be promoted to registers - the whole struct went to scratch, in every kernel that shares this function)
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ep.bias_n) bias = *(const f32x4*)(ep.bias_n + gn);
        const float* rcol = ep.resid_src ? ep.resid_src + cm * 8 + sub : nullptr;  // + gm * ld_resid + 8 e
        float* ocol = ep.out_f32 + c;
        f32x4 rs[2][NIT];
        long orow[2][NIT];  // -1: nothing to store (row past M, or the dropped first frame)
        auto fetch = [&](auto mi_c, auto buf_c) {
            constexpr int mi = decltype(mi_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int gm = m0 + wr * WM + mi * 16 + it * RPI + lane / LPR;
                const int gmc = gm < g.M ? gm : g.M - 1;
                const int rowi = (int)__umulhi((unsigned)gmc, mg_w);  // image row over all frames
                const int x = gmc - rowi * W;
                const int f = (int)__umulhi((unsigned)rowi, mg_h);
                const int y = rowi - f * H;
                const int fo = 2 * f + dt - 1;  // first frame after D2S is dropped
                orow[buf][it] = (gm < g.M && fo >= 0) ? ((long)fo * (2 * H) + (2 * y + dh)) * (2 * W) + (2 * x + dw) : -1;
                if (rcol) {
                    const float* rp = rcol + (long)gmc * ep.ld_resid;
                    rs[buf][it] = f32x4{rp[0], rp[8], rp[16], rp[24]};
                } else {
                    rs[buf][it] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        // fused PixelNorm of the upsampled voxel (round 5): with cout == BN a tile's 128 columns are ALL channels of one sub-position's
        // voxel, so the first res-block of the next stage gets its input (PixelNorm + modulation + SiLU of the value just stored) from
        // here instead of from a row pass over the whole f32 stream; same exchange between the two waves of a row block as above
        const bool pn = PN && ep.pn_out != nullptr && cout == BN;  // workgroup-uniform
        float* pns = (float*)(smem + WGM * WGN * (16 * WN * 4));
        f32x4 pn_s4 = f32x4{1.f, 1.f, 1.f, 1.f}, pn_h4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PN) {
            if (pn && ep.pn_scale) {
                pn_s4 = *(const f32x4*)(ep.pn_scale + c);
                pn_h4 = *(const f32x4*)(ep.pn_shift + c);
            }
        }
        static_for<0, MI>([&](auto mi_c) {
            constexpr int mi = decltype(mi_c)::value;
            constexpr int buf = mi & 1;
            if constexpr (mi + 1 < MI) fetch(std::integral_constant<int, mi + 1>{}, std::integral_constant<int, (mi + 1) & 1>{});
            if constexpr (mi == (MI >= 2 ? MI - 2 : 0)) hook();  // the last residual gather has just been requested
            if constexpr (SCR) {
                get(mi_c, scr);
            } else {
                f32x4 slab[NI];
                get(mi_c, slab);
                static_for<0, NI>([&](auto ni_c) {
                    constexpr int ni = decltype(ni_c)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[((lane >> 4) * 4 + r) * WN + ni * 16 + (lane & 15)] = slab[ni][r];
                });
            }
            f32x4 pn_v[PN ? NIT : 1];
            float pn_s2[PN ? NIT : 1];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + lane / LPR;
                f32x4 v = *(const f32x4*)(scr + row * WN + (lane % LPR) * 4);
                v += bias;
                v += rs[buf][it];
                if (orow[buf][it] >= 0) *(f32x4*)(ocol + orow[buf][it] * ep.ld_f32) = v;
                if constexpr (PN) {
                    if (pn) {
                        pn_v[it] = v;
                        pn_s2[it] = row16_allsum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
                    }
                }
            }
            if constexpr (PN) {
                if (pn) {
                    static_assert(!PN || (WGN == 2 && LPR == 16), "fused PixelNorm: two waves per tile row, 16 lanes per 64-column half row");
                    float* mine = pns + (buf * WGM * WGN + wave) * 16;
                    const float* other = pns + (buf * WGM * WGN + (wave ^ 1)) * 16;
                    if ((lane & (LPR - 1)) == 0) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) mine[it * RPI + (lane / LPR)] = pn_s2[it];
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int row = it * RPI + lane / LPR;
                        const float inv = __builtin_amdgcn_rsqf((pn_s2[it] + other[row]) * (1.0f / (float)BN) + 1e-8f);
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = silu_f(pn_v[it][e] * inv * pn_s4[e] + pn_h4[e]);
                        if (orow[buf][it] >= 0) {
                            uint2 pk;
                            pk.x = pack_bf16x2(y[0], y[1]);
                            pk.y = pack_bf16x2(y[2], y[3]);
                            *(uint2*)(ep.pn_out + orow[buf][it] * ep.ld_pn + c) = pk;
                        }
                    }
                }
            }
        });
        return;
    }
    // ---- the decoder's final un-patchify store (d2s == 3; conv_out: 48 of the tile's 64 columns exist): the same treatment - the
    // column part (pixel row b, offset inside its 12 floats) is per-lane constant, rows split with two multiply-high forms
    if (ep.d2s == 3 && (g.N & 3) == 0 && ep.out_f32 && !ep.out_bf16 && !ep.bias_m && ep.act == LTX_ACT_NONE && !ep.round_bf16 && !ep.resid &&
        (long)g.M * g.geom.W < (1L << 32) && (long)g.geom.F * g.geom.H * g.geom.H < (1L << 32) && g.geom.H >= 2 && g.geom.W >= 2) {
        constexpr int NIT = (16 * LPR) / 64, RPI = 64 / LPR;
        hook();
        const int H = g.geom.H, W = g.geom.W;
        const unsigned mg_w = magic_u32(W), mg_h = magic_u32(H);
        const int gn = n0 + wc * WN + (lane % LPR) * 4;
        const bool col_ok = gn < g.N;  // N % 4 == 0: a 4-wide chunk exists entirely or not at all
        const int b = gn / 12, oc = gn - 12 * b;
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ep.bias_n && col_ok) bias = *(const f32x4*)(ep.bias_n + gn);
        static_for<0, MI>([&](auto mi_c) {
            constexpr int mi = decltype(mi_c)::value;
            if constexpr (SCR) {
                get(mi_c, scr);
            } else {
                f32x4 slab[NI];
                get(mi_c, slab);
                static_for<0, NI>([&](auto ni_c) {
                    constexpr int ni = decltype(ni_c)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[((lane >> 4) * 4 + r) * WN + ni * 16 + (lane & 15)] = slab[ni][r];
                });
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + lane / LPR;
                const int gm = m0 + wr * WM + mi * 16 + row;
                f32x4 v = *(const f32x4*)(scr + row * WN + (lane % LPR) * 4);
                v += bias;
                if (ep.clip01) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf((v[e] + 1.0f) * 0.5f, 0.f), 1.f);
                }
                if (gm < g.M && col_ok) {
                    const int rowi = (int)__umulhi((unsigned)gm, mg_w);
                    const int x = gm - rowi * W;
                    const int f = (int)__umulhi((unsigned)rowi, mg_h);
                    const int y = rowi - f * H;
                    const long orow = (((long)f * (4 * H) + (4 * y + b)) * (4 * W) + 4 * x) * 3;  // ld_f32 is 1
                    *(f32x4*)(ep.out_f32 + orow * ep.ld_f32 + oc) = v;
                }
            }
        });
        return;
    }
    hook();
    static_for<0, MI>([&](auto mi_c) {
        constexpr int mi = decltype(mi_c)::value;
        if constexpr (SCR) {
            get(mi_c, scr);
        } else {
            f32x4 slab[NI];
            get(mi_c, slab);
            static_for<0, NI>([&](auto ni_c) {
                constexpr int ni = decltype(ni_c)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[((lane >> 4) * 4 + r) * WN + ni * 16 + (lane & 15)] = slab[ni][r];
            });
        }
#pragma unroll
        for (int it = 0; it < (16 * LPR) / 64; ++it) {
            const int chunk = it * 64 + lane;  // 16 rows x LPR float4 chunks, row-major
            const int row = chunk / LPR;
            const int c4 = (chunk % LPR) * 4;
            f32x4 v = *(const f32x4*)(scr + row * WN + c4);
            const int gm = m0 + wr * WM + mi * 16 + row;
            const int gn = n0 + wc * WN + c4;
            if (gm >= g.M || gn >= g.N) continue;
            const int nv = (g.N - gn) < 4 ? (g.N - gn) : 4;
            if (ep.bias_n) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nv) v[e] += ep.bias_n[gn + e];
            }
            if (ep.bias_m) {
                const float bm = ep.bias_m[gm];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bm;
            }
            if (ep.act == LTX_ACT_GELU_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            } else if (ep.act == LTX_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            }
            if (ep.round_bf16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
            }
            long orow = gm;  // output row (remapped for depth-to-space)
            int ocol = gn;
            if (ep.d2s == 2) {
                // latent upscaler: per-frame pixel shuffle (1,2,2); channels stored (i,j)-major: n' = sub*Cout + c
                const int cout = g.N >> 2;
                const int sub = gn / cout;
                const int c = gn - sub * cout;
                const int hw = g.geom.H * g.geom.W;
                const int f = gm / hw;
                const int rem = gm - f * hw;
                const int y = rem / g.geom.W;
                const int x = rem - y * g.geom.W;
                orow = ((long)f * (2 * g.geom.H) + (2 * y + (sub >> 1))) * (2 * g.geom.W) + (2 * x + (sub & 1));
                ocol = c;
            } else if (ep.d2s == 3) {
                // un-patchify store: this 4-wide chunk is one pixel row b, floats a*3 + c of its 12
                const int hw = g.geom.H * g.geom.W;
                const int f = gm / hw;
                const int rem = gm - f * hw;
                const int y = rem / g.geom.W;
                const int x = rem - y * g.geom.W;
                const int b = gn / 12;
                orow = (((long)f * (4 * g.geom.H) + (4 * y + b)) * (4 * g.geom.W) + 4 * x) * 3;  // ld_f32 is 1
                ocol = gn - b * 12;
                if (ep.clip01) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf((v[e] + 1.0f) * 0.5f, 0.f), 1.f);
                }
            } else if (ep.d2s) {
                // VAE upsampler (VideoDecoder.swift:201-251). Conv output channels were permuted at load time to
                // n' = sub*Cout + c (sub = dt*4+dh*2+dw), so this 4-wide chunk has one `sub` and consecutive c.
                const int cout = g.N >> 3;
                const int sub = gn / cout;
                const int c = gn - sub * cout;
                const int dt = sub >> 2, dh = (sub >> 1) & 1, dw = sub & 1;
                const int hw = g.geom.H * g.geom.W;
                const int f = gm / hw;
                const int rem = gm - f * hw;
                const int y = rem / g.geom.W;
                const int x = rem - y * g.geom.W;
                const int fo = 2 * f + dt - 1;  // first frame after D2S is dropped
                if (fo < 0) continue;
                orow = ((long)fo * (2 * g.geom.H) + (2 * y + dh)) * (2 * g.geom.W) + (2 * x + dw);
                ocol = c;
                if (ep.resid_src) {
                    // residual = D2S(x)[c mod C/8], tiled along channels (VideoDecoder.swift:219-234)
                    // c is a multiple of 4 and so is C/8: the chunk's four channels wrap together - one modulo, not four
                    const int cd2s = g.geom.C >> 3;
                    const int cm = (cd2s & (cd2s - 1)) == 0 ? (c & (cd2s - 1)) : (c % cd2s);
                    const float* rs = ep.resid_src + (long)gm * ep.ld_resid + cm * 8 + sub;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < nv) v[e] += rs[e * 8];
                }
            } else if (ep.resid) {
                const float* rs = (ep.resid_src ? ep.resid_src + (long)gm * ep.ld_resid : ep.out_f32 + (long)gm * ep.ld_f32) + gn;
                f32x4 gt;
                if (ep.gate) {
                    const float* gp = ep.gate + (long)(ep.gate_rowmap ? ep.gate_rowmap[gm] : gm / ep.rows_per_batch) * ep.gate_bstride + gn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gt[e] = (e < nv) ? gp[e] : 0.f;
                } else {
                    gt = f32x4{ep.gate_scalar, ep.gate_scalar, ep.gate_scalar, ep.gate_scalar};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nv) v[e] = rs[e] + gt[e] * v[e];
            }
            if (ep.out_f32) {
                float* o = ep.out_f32 + orow * ep.ld_f32 + ocol;
                if (nv == 4) {
                    *(f32x4*)o = v;
                } else {
                    for (int e = 0; e < nv; ++e) o[e] = v[e];
                }
            }
            if (ep.out_bf16) {
                bf16_t* o = ep.out_bf16 + orow * ep.ld_bf16 + ocol;
                if (nv == 4) {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)o = pk;
                } else {
                    for (int e = 0; e < nv; ++e) o[e] = f32_to_bf16(v[e]);
                }
            }
        }
    });
}

template <int BM, int BN, int WGM = 2, int WGN = 2, bool PN = false, bool SGATE = false, class Pre = ResidualTile<BM, BN, WGM, WGN>, class Hook = NoHook>
LTX_DEVFN void gemm_epilogue(f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], const GemmArgs& g, int m0, int n0, int wr, int wc,
                             int lane, int wave, char* smem, const Pre* pre = nullptr, Hook&& hook = Hook{}) {
    constexpr int NI = BN / WGN / 16;
    gemm_epilogue_with<BM, BN, WGM, WGN, PN, false, SGATE>(
        [&](auto mi_c, f32x4(&slab)[NI]) {
            constexpr int mi = decltype(mi_c)::value;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) slab[ni] = acc[mi][ni];
        },
        g, m0, n0, wr, wc, lane, wave, smem, pre, hook);
}

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PER_WAVE = BM / 32, B_PER_WAVE = BN / 32;  // wave-instructions (8 rows each) per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(g, bid, tiles_m, BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: per-lane source pointers (swizzle on the source side, LDS image stays lane-linear) ----
    const int srow = lane >> 3;  // row inside an 8-row wave-instruction
    const int pch = lane & 7;    // physical 16-B chunk the lane writes
    const bf16_t* a_src[A_PER_WAVE];
    const bf16_t* b_src[B_PER_WAVE];
    RowPos a_pos[A_PER_WAVE];
    int a_lch[A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave + 4 * i) * 8 + srow;
        const int lch = pch ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < g.M ? gm : g.M - 1;
        a_lch[i] = lch;
        if constexpr (CONV) {
            const int hw = g.geom.H * g.geom.W;
            a_pos[i].f = gm / hw;
            const int rem = gm - a_pos[i].f * hw;
            a_pos[i].y = rem / g.geom.W;
            a_pos[i].x = rem - a_pos[i].y * g.geom.W;
            a_src[i] = g.A;
        } else {
            a_src[i] = g.A + (long)gm * g.lda + lch * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        const int row = (wave + 4 * i) * 8 + srow;
        const int lch = pch ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < g.N ? gn : g.N - 1;
        b_src[i] = g.B + (long)gn * g.ldb + lch * 8;
    }
    const int nk = g.K / BK;
    const int cpt = CONV ? (g.geom.C / BK) : 1;  // K-tiles per conv tap

    auto stage = [&](int s, int kt) {
        char* base = smem + s * STAGE;
        if constexpr (CONV) {
            const int tap = kt / cpt;
            const int cc = kt - tap * cpt;
            const int dt = (g.geom.kt == 3) ? tap / 9 : 1;
            const int t9 = (g.geom.kt == 3) ? tap - dt * 9 : tap;
            const int dy = t9 / 3, dx = t9 - dy * 3;
#pragma unroll
            for (int i = 0; i < A_PER_WAVE; ++i) {
                int fi = g.geom.causal ? (a_pos[i].f + dt - 2) : (a_pos[i].f + dt - 1);
                int yi = a_pos[i].y + dy - 1, xi = a_pos[i].x + dx - 1;
                long pos;
                if (g.geom.pad_mode == 1 || g.geom.pad_mode == 3) {
                    if (g.geom.pad_mode == 3) fi = clamp_idx(fi, g.geom.F);  // zeros in H/W, replicated frames in T
                    const bool ok = fi >= 0 && fi < g.geom.F && yi >= 0 && yi < g.geom.H && xi >= 0 && xi < g.geom.W;
                    pos = ok ? ((long)fi * g.geom.H + yi) * g.geom.W + xi : (long)g.geom.F * g.geom.H * g.geom.W;
                } else {
                    fi = clamp_idx(fi, g.geom.F);
                    if (g.geom.pad_mode == 0) {
                        yi = reflect_idx(yi, g.geom.H);
                        xi = reflect_idx(xi, g.geom.W);
                    } else {
                        yi = clamp_idx(yi, g.geom.H);
                        xi = clamp_idx(xi, g.geom.W);
                    }
                    pos = ((long)fi * g.geom.H + yi) * g.geom.W + xi;
                }
                const bf16_t* src = g.A + pos * g.geom.C + cc * BK + a_lch[i] * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + (wave + 4 * i) * 1024),
                                                 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER_WAVE; ++i) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(a_src[i] + (long)kt * BK),
                    (__attribute__((address_space(3))) void*)(base + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(b_src[i] + (long)kt * BK),
                (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (per lane constants; the row-dependent XOR term only depends on lane&15) ----
    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int foff0 = frow * ROW_BYTES + ((((lane >> 4) + 0) ^ fsw) << 4);
    const int foff1 = frow * ROW_BYTES + ((((lane >> 4) + 4) ^ fsw) << 4);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* abase = smem + cur * STAGE + (wr * WM) * ROW_BYTES;
        const char* bbase = smem + cur * STAGE + A_BYTES + (wc * WN) * ROW_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? foff1 : foff0;
            s16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const s16x8*)(abase + i * 16 * ROW_BYTES + fo);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *(const s16x8*)(bbase + j * 16 * ROW_BYTES + fo);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                        __builtin_bit_cast(bf16x8_t, bfr[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // LDS-DMA of the next stage has landed (vmcnt(0)) and every wave is done reading `cur`
    }

    gemm_epilogue<BM, BN>(acc, g, m0, n0, wr, wc, lane, wave, smem);
}


// ---------------------------------------------------------------------------------------------------------------
// v2 main loop: NSTAGE-deep LDS ring filled NSTAGE-1 tiles ahead by LDS-DMA, counted `s_waitcnt vmcnt(N)` + raw
// `s_barrier` (never vmcnt(0) in the steady state), and a skewed software pipeline on the register fragments:
//
//   iteration t:   ds_read frags(t, kk=1) -> set B      | MFMA(set A = frags(t, kk=0))      <- LDS latency hidden
//                  wait tile t+1 landed ; s_barrier
//                  LDS-DMA tile t+NSTAGE-1 -> ring slot of tile t-1 (every wave is past its reads of t-1)
//                  ds_read frags(t+1, kk=0) -> set A    | MFMA(set B)                       <- barrier + LDS latency hidden
//
// One barrier per K-tile, loads in flight across it (guide "Pipelining across barriers": counted vmcnt + raw barrier).
// ---------------------------------------------------------------------------------------------------------------
template <int N>
LTX_DEVFN void wait_lgkm_vmcnt_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int N>
LTX_DEVFN void wait_lgkm_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
LTX_DEVFN void wait_lgkm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
LTX_DEVFN void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N>
LTX_DEVFN void wait_vmcnt_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// (Tried and removed: a ninth wave per workgroup that touched the B lines eight K-tiles ahead of the ring, one dword per
// 128-B line, to hide the HBM latency of the once-streamed DiT weights. Same-process A/B on MI355X: 779 vs 963 TFLOP/s
// cold and 932 vs 1040 warm at 1536x8192x4096 - every touched line crosses the CU's vector L1, +40 % bytes through the
// texture path that this loop already saturates. The ring depth stays the only latency cover.)
#if defined(GEMM_ASM_STAMPS) || defined(GEMM_V2_STAMPS) || defined(DTL_STAMPS)
__device__ unsigned long long g_gemm_stamps[5][8];  // diagnostic builds only (tools/ubench/gemm_stamps.hip)
#endif
// QB (few-row launches on a quantised Linear, round 3): B arrives as 8-bit codes + bf16 scale / bias per 64-wide group (one group =
// one K-tile) and is de-quantised IN the B stage: stage(t) brings A(t), the codes of tile t+1 (BN x 64 B, a quarter of the bf16
// bytes plus its 2 x BN scalars) into a code ring; the first half of K-tile t converts the codes of tile t+1 into the bf16 image of
// ring slot t+1 (w' = bf16(q * scale + bias), the arithmetic of quant_decode_kernel: the result is bit-identical to the scratch
// path), and the mid-tile barrier that already orders tile t+1's LDS-DMA makes the converted image visible before its first
// fragment read. The codes are read from HBM once and nothing is written back: a few-row GEMM streams half the bytes of the bf16 one.
template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int WGN = 2, bool QB = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_bf16_kernel_v2(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WGM * WGN;  // waves per workgroup
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PER_WAVE = BM / 8 / NW, B_PER_WAVE = QB ? BN / 16 / NW + 1 : BN / 8 / NW;  // QB: code pieces of 16 rows + one scale / bias piece
    static_assert(A_PER_WAVE * 8 * NW == BM && (QB || B_PER_WAVE * 8 * NW == BN), "tile rows must split evenly over the waves");
    static_assert(!QB || (!CONV && BN % (16 * NW) == 0 && NW * 64 * 16 == BN * 64), "QB: one 16-byte conversion unit per thread");
    constexpr int NQ = NSTAGE + 1;                          // code / scalar ring slots (a tile's codes live from stage(t-1) to K-tile t-1)
    constexpr int Q_BYTES = BN * 64, SB_BYTES = NW * 256;   // codes [BN][64 B]; scalars: one dword per lane of one piece per wave
    constexpr int Q_OFF = NSTAGE * STAGE, SB_OFF = Q_OFF + NQ * Q_BYTES;
    constexpr int LPT = A_PER_WAVE + B_PER_WAVE;  // LDS-DMA instructions per wave per K-tile
    constexpr int PD = NSTAGE - 1;                // prefetch distance in tiles
    static_assert(NSTAGE >= 3, "ring needs >= 3 slots");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;

    const int tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x) + g.tile0;
    int tm, tn;
    tile_coords(g, bid, tiles_m, BN, tm, tn);
    if constexpr (CONV && BM == 192) {
        if (g.geom.blk_rg > 0 && g.tile_count == 0 && g.split_k <= 1 && g.N <= BN) tm = conv_block_order(tm, g.geom.F, g.geom.blk_rg);
    }
    const int m0 = tm * BM, n0 = tn * BN;
#ifdef GEMM_V2_STAMPS  // tools/ubench/gemm_stamps.hip: 100 MHz wall-clock stamps of one wave (prologue / main loop / epilogue)
    const unsigned long long st0 = wall_clock64();
#endif

    const int srow = lane >> 3;
    const int pch = lane & 7;
    const bf16_t* a_src[A_PER_WAVE];
    const bf16_t* b_src[B_PER_WAVE];
    int a_lch[A_PER_WAVE];
    // conv mode, separable tap tables: the voxel index of tap (dt, dy, dx) of a row is tf[dt] + ty[dy] + tx[dx] with the padding
    // rule (reflect / clamp / replicate) already applied per axis, and `bad` has one bit per (axis, tap index) that falls into ZERO
    // padding (those taps read the zero voxel behind the tensor). Built once per tile, branch-free; a tap change then costs a few
    // selects and adds per row instead of re-deriving the neighbour from (f, y, x) (the 128-channel convs change tap every second
    // K-tile).
    struct TapTab {
        int f0, f1, f2, y0, y1, y2, x0, x1, x2, bad;
    };
    TapTab a_tab[A_PER_WAVE];
    struct TapStep {
        int x01, x12, y01, y12, t01, t12;
    };
    TapStep a_step[CONV ? A_PER_WAVE : 1];
    const bool conv_steps = CONV && !(g.geom.pad_mode == 1 || g.geom.pad_mode == 3);  // no zero padding: taps are positions
    const bf16_t* a_row0[A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        const int lch = pch ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < g.M ? gm : g.M - 1;
        a_lch[i] = lch;
        if constexpr (CONV) {
            const int F = g.geom.F, H = g.geom.H, W = g.geom.W, hw = H * W;
            const int pf = gm / hw;
            const int rem = gm - pf * hw;
            const int py = rem / W;
            const int px = rem - py * W;
            a_src[i] = g.A;
            a_row0[i] = g.A + lch * 8;
            const bool zero_t = g.geom.pad_mode == 1;
            const bool zero_hw = g.geom.pad_mode == 1 || g.geom.pad_mode == 3;
            const bool refl_hw = g.geom.pad_mode == 0;
            int bad = 0;
            auto axis_t = [&](int d) {
                const int fi = g.geom.causal ? (pf + d - 2) : (pf + d - 1);
                const bool out = fi < 0 || fi >= F;
                bad |= (zero_t && out) ? (1 << d) : 0;
                return clamp_idx(fi, F) * hw;
            };
            auto axis_s = [&](int p, int n, int d, int bit) {
                const int v = p + d - 1;
                const bool out = v < 0 || v >= n;
                bad |= (zero_hw && out) ? (bit << d) : 0;
                return refl_hw ? reflect_idx(v, n) : clamp_idx(v, n);
            };
            TapTab& t = a_tab[i];
            t.f0 = axis_t(0); t.f1 = axis_t(1); t.f2 = axis_t(2);
            t.y0 = axis_s(py, H, 0, 8) * W; t.y1 = axis_s(py, H, 1, 8) * W; t.y2 = axis_s(py, H, 2, 8) * W;
            t.x0 = axis_s(px, W, 0, 64); t.x1 = axis_s(px, W, 1, 64); t.x2 = axis_s(px, W, 2, 64);
            t.bad = bad;
            // without zero padding a tap is a position, and the next tap in (dt, dy, dx) order is that position plus one of six
            // per-row steps (elements): a tap change is then a 64-bit add per row
            TapStep& st = a_step[i];
            const int C = g.geom.C;
            st.x01 = (t.x1 - t.x0) * C;
            st.x12 = (t.x2 - t.x1) * C;
            st.y01 = (t.x0 - t.x2 + t.y1 - t.y0) * C;
            st.y12 = (t.x0 - t.x2 + t.y2 - t.y1) * C;
            st.t01 = (t.x0 - t.x2 + t.y0 - t.y2 + t.f1 - t.f0) * C;
            st.t12 = (t.x0 - t.x2 + t.y0 - t.y2 + t.f2 - t.f1) * C;
        } else {
            a_src[i] = g.A + (long)gm * g.lda + lch * 8;
        }
    }
    const uint8_t* q_src[QB ? BN / 16 / NW : 1];  // QB: this lane's 16 codes of a K-tile: row 16 (wave + NW i) + lane / 4, bytes 16 (lane % 4) ..
    const bf16_t* sb_src = nullptr;               // QB: this lane's scalar of a K-tile: lanes 0-15 scale, 16-31 bias of row 16 wave + (lane & 15)
    const int groups = g.K / BK;
    if constexpr (QB) {
#pragma unroll
        for (int i = 0; i < BN / 16 / NW; ++i) {
            int gn = n0 + (wave + NW * i) * 16 + (lane >> 2);
            gn = gn < g.N ? gn : g.N - 1;
            q_src[i] = g.Bq + (long)gn * g.K + (lane & 3) * 16;
        }
        int gn = n0 + wave * 16 + (lane & 15);
        gn = gn < g.N ? gn : g.N - 1;
        sb_src = ((lane & 16) ? g.Bqb : g.Bqs) + (long)gn * groups;
    } else {
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const int row = (wave + NW * i) * 8 + srow;
            const int lch = pch ^ ((row >> 1) & 7);
            int gn = n0 + row;
            gn = gn < g.N ? gn : g.N - 1;
            b_src[i] = g.B + (long)gn * g.ldb + lch * 8;
        }
    }
    int nk = g.K / BK, kt0 = 0;  // this workgroup's K-tiles: [kt0, kt0 + nk)
    if (g.split_k > 1) {
        const int z = blockIdx.y;
        kt0 = (int)((long)nk * z / g.split_k);
        nk = (int)((long)nk * (z + 1) / g.split_k) - kt0;
    }
    const int cpt = CONV ? (g.geom.C / BK) : 1;
    // QB: codes + scalars of absolute K-tile `kt` (clamped: a tile past the end is staged into a free slot and never converted)
    auto stage_q = [&](int kt) {
        if constexpr (QB) {
            const int t = kt < groups ? kt : groups - 1;
            const int qslot = (kt - kt0 + NQ) % NQ;
#pragma unroll
            for (int i = 0; i < BN / 16 / NW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q_src[i] + (long)t * BK),
                                                 (__attribute__((address_space(3))) void*)(smem + Q_OFF + qslot * Q_BYTES + (wave + NW * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb_src + t),
                                             (__attribute__((address_space(3))) void*)(smem + SB_OFF + qslot * SB_BYTES + wave * 256), 2, 0, 0);
        }
    };
    // QB: codes of tile index `j` (relative to kt0) -> bf16 image of ring slot `slot` (B part). Thread = (row tid / 4, 16 codes tid % 4).
    auto convert_q = [&](int slot, int j) {
        if constexpr (QB) {
            const int qslot = (j + NQ) % NQ;
            const int row = tid >> 2, c4 = tid & 3;
            const uint4 raw = *(const uint4*)(smem + Q_OFF + qslot * Q_BYTES + row * 64 + c4 * 16);
            const char* sbp = smem + SB_OFF + qslot * SB_BYTES + (row >> 4) * 256 + (row & 15) * 4;
            const float scale = bf16_to_f32((bf16_t)(*(const uint32_t*)sbp & 0xffffu));
            const float bias = bf16_to_f32((bf16_t)(*(const uint32_t*)(sbp + 64) & 0xffffu));
            const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
            uint32_t out[8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float q0 = (float)((wd[i] >> (16 * jj)) & 0xffu), q1 = (float)((wd[i] >> (16 * jj + 8)) & 0xffu);
                    out[i * 2 + jj] = pack_bf16x2(q0 * scale + bias, q1 * scale + bias);
                }
            char* brow = smem + slot * STAGE + A_BYTES + row * ROW_BYTES;
            const int sw = (row >> 1) & 7;
            *(uint4*)(brow + (((2 * c4) ^ sw) << 4)) = uint4{out[0], out[1], out[2], out[3]};
            *(uint4*)(brow + (((2 * c4 + 1) ^ sw) << 4)) = uint4{out[4], out[5], out[6], out[7]};
        }
    };

    // conv mode: the gathered source pointers only change when the 3x3x3 tap changes (every C/64 K-tiles); inside a
    // tap they just advance by one K-tile. K-tiles are staged in increasing order, so the pointers are loop state.
    int a_fy[A_PER_WAVE];  // tf[dt] + ty[dy] of the current (dt, dy): changes every third tap
    int fy_of = -1;        // tap / 3 that a_fy belongs to
    auto conv_tap_ptrs = [&](int tap) {
        const int dt = (g.geom.kt == 3) ? tap / 9 : 1;
        const int t9 = (g.geom.kt == 3) ? tap - dt * 9 : tap;
        const int dy = t9 / 3, dx = t9 - dy * 3;
        // uniform all-ones / zero masks instead of `dt == 0 ? f0 : ...`: the compiler turns such a chain over struct fields into an
        // indexed scratch array (a scratch load + vmcnt(0) in the middle of the ring's counted waits)
        if (tap / 3 != fy_of) {
            fy_of = tap / 3;
            const int mt1 = -(dt == 1), mt2 = -(dt == 2), my1 = -(dy == 1), my2 = -(dy == 2);
#pragma unroll
            for (int i = 0; i < A_PER_WAVE; ++i) {
                const TapTab& t = a_tab[i];
                a_fy[i] = t.f0 + t.y0 + (((t.f1 - t.f0) & mt1) + ((t.f2 - t.f0) & mt2)) + (((t.y1 - t.y0) & my1) + ((t.y2 - t.y0) & my2));
            }
        }
        const int mx1 = -(dx == 1), mx2 = -(dx == 2);
        const int mask = (1 << dt) | (8 << dy) | (64 << dx);
        const int zero_vox = g.geom.F * g.geom.H * g.geom.W;  // the zero voxel the caller keeps behind the tensor
        const bool zero_pad = g.geom.pad_mode == 1 || g.geom.pad_mode == 3;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) {
            const TapTab& t = a_tab[i];
            int vox = a_fy[i] + t.x0 + (((t.x1 - t.x0) & mx1) + ((t.x2 - t.x0) & mx2));
            if (zero_pad) {
                const int hit = -(int)((t.bad & mask) != 0);
                vox = (zero_vox & hit) | (vox & ~hit);
            }
            a_src[i] = a_row0[i] + (long)vox * g.geom.C;
        }
    };
    // tap `from` -> from + 1 where taps are positions: one uniform branch picks the step, each arm is a 64-bit add per row
    auto conv_tap_step = [&](int from) {
        const int t9 = (g.geom.kt == 3) ? from % 9 : from;
        const int dt = (g.geom.kt == 3) ? from / 9 : 1;
        const int dy = t9 / 3, dx = t9 - dy * 3;
#define LTX_TAP_STEP(FIELD)                                                   \
    _Pragma("unroll") for (int i = 0; i < A_PER_WAVE; ++i) a_src[i] += (long)a_step[i].FIELD
        if (dx == 0) { LTX_TAP_STEP(x01); }
        else if (dx == 1) { LTX_TAP_STEP(x12); }
        else if (dy == 0) { LTX_TAP_STEP(y01); }
        else if (dy == 1) { LTX_TAP_STEP(y12); }
        else if (dt == 0) { LTX_TAP_STEP(t01); }
        else { LTX_TAP_STEP(t12); }
#undef LTX_TAP_STEP
    };
    int conv_tap = kt0 / cpt, conv_cc = kt0 - conv_tap * cpt;  // position of the NEXT K-tile to stage
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * STAGE;
        if constexpr (CONV) {
#pragma unroll
            for (int i = 0; i < A_PER_WAVE; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + conv_cc * BK),
                                                 (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024),
                                                 16, 0, 0);
            }
            if (++conv_cc == cpt) {  // the pointers of the next tap, behind the loads of this one
                conv_cc = 0;
                if (++conv_tap < 9 * g.geom.kt) {
                    if (conv_steps) conv_tap_step(conv_tap - 1); else conv_tap_ptrs(conv_tap);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER_WAVE; ++i) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(a_src[i] + (long)kt * BK),
                    (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, 0);
            }
        }
        if constexpr (QB) {
            stage_q(kt + 1);  // the codes run one tile ahead of the activations: they are converted during K-tile kt
        } else {
            if (BM == 128 && BN == 64 && !CONV && g.b_nt) {  // uniform branch; the policy is an immediate of the instruction
#pragma unroll
                for (int i = 0; i < B_PER_WAVE; ++i) {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(b_src[i] + (long)kt * BK),
                        (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + NW * i) * 1024), 16, 0, 2);
                }
            } else {
#pragma unroll
                for (int i = 0; i < B_PER_WAVE; ++i) {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(b_src[i] + (long)kt * BK),
                        (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + NW * i) * 1024), 16, 0, 0);
                }
            }
        }
    };

    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int foff0 = frow * ROW_BYTES + ((((lane >> 4) + 0) ^ fsw) << 4);
    const int foff1 = frow * ROW_BYTES + ((((lane >> 4) + 4) ^ fsw) << 4);
    const int a_wave_off = (wr * WM) * ROW_BYTES;
    const int b_wave_off = A_BYTES + (wc * WN) * ROW_BYTES;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    s16x8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
    auto load_frags = [&](int slot, int fo, s16x8(&fa)[MI], s16x8(&fb)[NI]) {
        const char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const s16x8*)(base + a_wave_off + i * 16 * ROW_BYTES + fo);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = *(const s16x8*)(base + b_wave_off + j * 16 * ROW_BYTES + fo);
    };
    // MFMAs idx in [FROM, TO) of the MIxNI grid (row-major) - split so that the compiler-inserted lgkmcnt wait sits
    // in front of the FIRST product only, while nothing younger is outstanding (see the loop below)
    auto mfma_first = [&](const s16x8(&fa)[MI], const s16x8(&fb)[NI]) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[0]),
                                                            __builtin_bit_cast(bf16x8_t, fb[0]), acc[0][0], 0, 0, 0);
    };
    auto mfma_rest = [&](const s16x8(&fa)[MI], const s16x8(&fb)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (i + j > 0)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                        __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
    };

    // prologue: fill PD ring slots, wait for tile 0, fetch its first fragments
    if constexpr (CONV) {
        conv_tap_ptrs(conv_tap);  // absolute pointers of the first tap (a K split may begin at any tap, in the middle of one)
    }
    if constexpr (QB) stage_q(kt0);  // codes of the first tile (stage(t) brings the codes of tile t + 1)
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (s < nk) stage(s, kt0 + s);
    if (nk >= PD) wait_vmcnt_barrier<(PD - 1) * LPT>(); else wait_vmcnt_barrier<0>();
    if constexpr (QB) {
        convert_q(0, 0);  // the first tile's weights, visible behind a barrier of their own
        wait_lgkm_barrier();
    }
#ifdef GEMM_V2_STAMPS
    const unsigned long long st1 = wall_clock64();
    const unsigned long long cy1 = __builtin_amdgcn_s_memtime();  // shader cycles next to wall time: the clock the part holds in the loop (round 5)
#endif
    load_frags(0, foff0, fa0, fb0);
    const bool late = NW == 8 && g.conv_stagger && wave >= NW / 2;  // wave-uniform (GemmArgs::conv_stagger: option "gemm_stagger" for dense launches)

    // One K-tile. STEADY iterations are branch-free so that each half is ONE scheduling region in which the LDS
    // fragment reads and the LDS-DMA issues are interleaved one-for-one with MFMAs (sched_group_barrier): their issue
    // cost then hides under the previous MFMA's execution instead of serialising in front of the MFMA cluster
    // (ablation on MI355X: MFMA 98 us, LDS reads+barriers 74 us, DMA issue 57 us were ADDING UP to 207 us).
    int slot = 0;  // ring slot of tile kt
    bool drained = false;
    auto ktile = [&](int kt, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        int nslot = slot + 1;
        nslot = nslot == NSTAGE ? 0 : nslot;
        int pslot = slot - 1;  // slot of tile kt-1 == slot of tile kt+PD
        pslot = pslot < 0 ? NSTAGE - 1 : pslot;
        // ---- first half: the only LDS reads outstanding at the first MFMA are set A's (issued half a tile ago)
        mfma_first(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(slot, foff1, fa1, fb1);
        if constexpr (QB) {
            if (kt + 1 < nk) convert_q(nslot, kt + 1);  // codes of tile kt+1 (visible since the previous mid-tile barrier) -> its bf16 image
        }
        mfma_rest(fa0, fb0);
        if constexpr (!QB) {
#pragma unroll
            for (int q = 0; q < MI + NI; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one ds_read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - 1 - (MI + NI), 0);
        }
        // ---- tile kt+1 must have landed for every wave; tiles kt+2 .. kt+PD-1 may stay in flight. The first tail tile drains the
        // ring (every remaining tile visible to every wave); the tail tiles after it stage nothing and need neither wait nor barrier.
        // QB: the barrier also publishes the image just converted (its ds_writes are retired first), so every tile has one.
        if constexpr (QB) {
            if constexpr (STEADY) wait_lgkm_vmcnt_barrier<(PD - 2) * LPT>(); else wait_lgkm_vmcnt_barrier<0>();
        } else {
            if constexpr (STEADY) wait_vmcnt_barrier<(PD - 2) * LPT>(); else if (!drained) wait_vmcnt_barrier<0>();
        }
        // ---- second half
        mfma_first(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STEADY) {
            // staggered staging (round 5, as conv_halo.inc): of the two waves that share a SIMD (w, w + NW / 2) the second issues its
            // LDS-DMA pieces behind this half's MFMAs instead of in front of them, so that the two are not held by their pieces at the
            // same time. Same loads, same per-wave order, same barrier interval: the counted waits do not change.
            if (!late) stage(pslot, kt0 + kt + PD);
            load_frags(nslot, foff0, fa0, fb0);
            mfma_rest(fa1, fb1);
            if constexpr (!QB) {
#pragma unroll
                for (int q = 0; q < LPT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);  // one LDS-DMA (VMEM read)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                }
#pragma unroll
                for (int q = 0; q < MI + NI; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - 1 - (MI + NI) - LPT, 1);
            }
            if (late) stage(pslot, kt0 + kt + PD);
        } else {
            if (kt + 1 < nk) load_frags(nslot, foff0, fa0, fb0);
            mfma_rest(fa1, fb1);
        }
        slot = nslot;
    };
    int kt = 0;
    for (; kt < nk - PD; ++kt) ktile(kt, std::true_type{});
    ResidualTile<BM, BN, WGM, WGN> rt;
    if (kt < nk) {
        ktile(kt, std::false_type{});
        ++kt;
        drained = true;
        // gated-residual launches (the DiT's attention-out and FFN-out projections): fetch the residual-stream tile now, under the
        // MFMAs of the last PD-1 K-tiles, instead of at the top of the epilogue
        if (!g.ep.d2s && g.split_k <= 1 && n0 + BN <= g.N && kt < nk) {
            gemm_bias_prefetch(g, n0, wc, lane, rt);
            if (g.ep.resid) gemm_residual_prefetch(g, m0, n0, wr, wc, lane, rt);
        }
    }
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
    __syncthreads();
#ifdef GEMM_V2_STAMPS
    const unsigned long long st2 = wall_clock64();
    const unsigned long long cy2 = __builtin_amdgcn_s_memtime();
#endif
    if (g.split_k > 1) {
        GemmArgs gs = g;  // raw partial tile -> workspace slice of this split
        gs.ep = GemmEpilogue{};
        gs.ep.out_f32 = g.split_ws + ((long)blockIdx.y * (g.win_rows ? g.win_rows : g.M) - g.win_row0) * g.N;  // rows are absolute
        gs.ep.ld_f32 = g.N;
        gemm_epilogue<BM, BN, WGM, WGN>(acc, gs, m0, n0, wr, wc, lane, wave, smem);
        return;
    }
    gemm_epilogue<BM, BN, WGM, WGN, (CONV && BN == 128 && WGN == 2)>(acc, g, m0, n0, wr, wc, lane, wave, smem, &rt);
#ifdef GEMM_V2_STAMPS
    {
        const unsigned long long st3 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st4 = wall_clock64();
        if (lane == 0 && (blockIdx.x == 7 || blockIdx.x == 200) && wave < 4) {
            unsigned long long* d = &g_gemm_stamps[blockIdx.x == 7 ? wave : 4][0];
            if (blockIdx.x == 7 || wave == 0) {
                d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = st4; d[5] = cy2 - cy1;
            }
        }
    }
#endif
}


#include "conv_halo.inc"
#include "conv_halo2.inc"

#ifdef LTX_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------
// Few-row launches (M <= 128: the DiT at 256x256x9 = 128 tokens), round 3. Such a GEMM streams its weights once and is bound by
// how many bytes a CU keeps in flight against the HBM latency, not by MFMAs (16 per wave and K-tile). The ring kernel above stages
// activations and weights through ONE ring with one in-order counter: three K-tiles = 72 KB in flight, two thirds of them the
// activation tile, which comes from L2 and needs no such cover. Here the two operands have rings of their own and waves of their own:
//   waves 0, 1 stage the activation tile (128 x 64 bf16 = 16 pieces) two K-tiles ahead: 3 slots, 48 KB;
//   waves 2, 3 stage the weights in MACRO-tiles of four K-tiles, two macro-tiles ahead: a piece is 2 rows x 512 contiguous bytes
//              (8 rows x 128 B pieces fetch a quarter of a DRAM page per row visit: the ring kernel streams weights at ~3 TB/s);
//              3 slots of 32 KB - or, QB, the 8-bit codes of a quantised Linear: pieces of 4 rows x 256 B, 4 slots of 16 KB, three
//              macro-tiles ahead, plus the group scales / biases of the macro-tile (two dwords per row and wave);
// each role waits on its own vmcnt (the counter is per wave), one workgroup barrier per K-tile publishes both. All four waves run the
// MFMAs (wave tile 64 x 32). QB: the codes of tile t+1 are converted to the bf16 image (two slots) during K-tile t, w' = bf16(q * scale
// + bias) - quant_decode_kernel's arithmetic, bit-identical to the scratch path - so a quantised few-row GEMM reads half the bytes.
// Split-K (grid.y) and the epilogue are the ring kernel's.
// MEASURED, NOT SELECTED (tile_cfg 30, experiments build): bit-exact (tests/test_experiments_gpu.py), and 9 % SLOWER than the 128x64
// ring kernel at 128 tokens - 22.9 / 25.9 / 45.0 us against 20.95 / 23.3 / 41.1 us for the N = 4096 / 8192 / 16384 launches of a block
// (rocprofv3, profiles/r03_config1_fewrow.txt); forward 12.7 against 11.9 ms. Both kernels stream the weights at ~3 TB/s whatever the
// run length per DRAM row (128 B or 512 B). The activation tile's re-fetch (16 KB per K-tile from L2 beside 8 KB of weights) was this
// kernel's premise and is NOT the bound: 16-row launches of the ring kernel are only 7 % faster than 128-row ones
// (profiles/r03_fewrow_bounds.txt); a six-slot ring there gained 5 % and is what the product now runs.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
LTX_DEVFN void wait_vmcnt_only() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for a run-time n (immediates only): n is wave-uniform, <= 45
LTX_DEVFN void wait_vmcnt_dyn(int n) {
    switch (n) {
#define LTX_W(K) case K: wait_vmcnt_only<K>(); break;
        LTX_W(0) LTX_W(1) LTX_W(2) LTX_W(3) LTX_W(4) LTX_W(5) LTX_W(6) LTX_W(7) LTX_W(8) LTX_W(9) LTX_W(10) LTX_W(11) LTX_W(12) LTX_W(13) LTX_W(14)
        LTX_W(15) LTX_W(16) LTX_W(17) LTX_W(18) LTX_W(19) LTX_W(20) LTX_W(21) LTX_W(22) LTX_W(23) LTX_W(24) LTX_W(25) LTX_W(26) LTX_W(27) LTX_W(28)
        LTX_W(29) LTX_W(30) LTX_W(31) LTX_W(32) LTX_W(33) LTX_W(34) LTX_W(35) LTX_W(36) LTX_W(37) LTX_W(38) LTX_W(39) LTX_W(40) LTX_W(41) LTX_W(42)
        LTX_W(43) LTX_W(44) LTX_W(45)
#undef LTX_W
        default: wait_vmcnt_only<0>(); break;
    }
}

template <bool QB>
__global__ __launch_bounds__(256, 1) void gemm_fewrow_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 64, WGM = 2, WGN = 2, WM = 64, WN = 32, MI = 4, NI = 2;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    constexpr int NA = 3, PA = NA - 1;                       // activation ring (K-tiles)
    constexpr int MT = 4;                                    // K-tiles per weight macro-tile: 256 elements = 512 B (256 B of codes) of a row
    constexpr int NBM = QB ? 4 : 3, PBM = NBM - 1;           // weight ring (macro-tiles)
    constexpr int BM_BYTES = QB ? BN * MT * 64 : BN * MT * ROW_BYTES;   // 16 KB of codes / 32 KB of bf16 weights
    constexpr int SB_BYTES = 1024;                           // QB: [scale dword 0][scale dword 1][bias dword 0][bias dword 1] x 64 rows
    constexpr int A_OFF = 0, B_OFF = NA * A_BYTES;           // bf16 macro ring, or (QB) the two-slot bf16 image of single K-tiles
    constexpr int Q_OFF = B_OFF + 2 * B_BYTES, SB_OFF = Q_OFF + NBM * BM_BYTES;
    constexpr int APW = 8;                                   // LDS-DMA pieces per K-tile of an activation wave
    constexpr int BPW = QB ? 10 : 16;                        // ... per MACRO-tile of a weight wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const bool a_role = wave < 2;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = bid * BN, m0 = 0;
    int nk = g.K / BK, kt0 = 0;
    const int groups = g.K / BK;
    if (g.split_k > 1) {  // the launcher guarantees whole macro-tiles per split
        const int z = blockIdx.y;
        kt0 = (int)((long)nk * z / g.split_k);
        nk = (int)((long)nk * (z + 1) / g.split_k) - kt0;
    }
    const int nm = nk / MT;  // macro-tiles of this workgroup
    // ---- staging sources ----
    const bf16_t* asrc[APW];
    const char* bsrc = nullptr;   // weight wave: this lane's 16 bytes of piece 0 of macro-tile 0; piece i is `bstep` rows further down
    long bstep = 0;
    const bf16_t* sbsrc = nullptr;
    if (a_role) {
        const int srow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int row = (wave + 2 * i) * 8 + srow;
            const int gm = row < g.M ? row : g.M - 1;
            asrc[i] = g.A + (long)gm * g.lda + (pch ^ ((row >> 1) & 7)) * 8;
        }
    } else {
#pragma unroll
        for (int i = 0; i < APW; ++i) asrc[i] = nullptr;
        if constexpr (QB) {
            // piece = 4 rows x 256 B of codes; wave w-2 takes pieces (w-2), (w-2)+2, ...: rows 4 piece + lane / 16, chunk lane % 16
            const int row = (wave - 2) * 4 + (lane >> 4);
            int gn = n0 + row;
            gn = gn < g.N ? gn : g.N - 1;
            bsrc = (const char*)(g.Bq + (long)gn * g.K + (long)kt0 * BK + (lane & 15) * 16);
            bstep = 8L * g.K;   // two pieces = 8 rows further (ragged N is handled by the caller: N % 64 == 0 for this instance)
            int gs = n0 + lane;
            gs = gs < g.N ? gs : g.N - 1;
            sbsrc = (wave == 2 ? g.Bqs : g.Bqb) + (long)gs * groups + kt0;
        } else {
            // piece = 2 rows x 512 B; physical chunk p of row r holds logical chunk p ^ (r & 15) (fragment reads: 16 rows, one chunk);
            // the source addresses are formed in stage_b
        }
    }
    auto stage_a = [&](int j) {  // relative K-tile j -> activation slot j % NA
        char* base = smem + A_OFF + (j % NA) * A_BYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)(kt0 + j) * BK),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 2 * i) * 1024), 16, 0, 0);
    };
    auto stage_b = [&](int m) {  // relative macro-tile m -> weight slot m % NBM
        if constexpr (QB) {
            char* base = smem + Q_OFF + (m % NBM) * BM_BYTES;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + i * bstep + (long)m * MT * BK),
                                                 (__attribute__((address_space(3))) void*)(base + ((wave - 2) + 2 * i) * 1024), 16, 0, 0);
            char* sb = smem + SB_OFF + (m % NBM) * SB_BYTES + (wave - 2) * 512;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbsrc + m * MT + 2 * i),
                                                 (__attribute__((address_space(3))) void*)(sb + i * 256), 4, 0, 0);
        } else {
            char* base = smem + B_OFF + (m % NBM) * BM_BYTES;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // piece (wave - 2) + 2 i = rows 2 piece, 2 piece + 1; the swizzle term (row & 15) changes from piece to piece
                const int row = ((wave - 2) + 2 * i) * 2 + (lane >> 5);
                const bf16_t* src = g.B + (long)(n0 + row) * g.ldb + (long)(kt0 + m * MT) * BK + (((lane & 31) ^ (row & 15)) * 8);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + ((wave - 2) + 2 * i) * 1024), 16, 0, 0);
            }
        }
    };
    // QB: codes of relative K-tile j -> bf16 image slot j & 1. Thread = (row tid / 4, 16 codes tid % 4).
    auto convert_q = [&](int j) {
        if constexpr (QB) {
            const int row = tid >> 2, c4 = tid & 3;
            const int m = j / MT, sub = j % MT;
            const uint4 raw = *(const uint4*)(smem + Q_OFF + (m % NBM) * BM_BYTES + row * 256 + sub * 64 + c4 * 16);
            const char* sbp = smem + SB_OFF + (m % NBM) * SB_BYTES + (sub >> 1) * 256 + row * 4;
            const uint32_t sw_ = *(const uint32_t*)sbp, bw_ = *(const uint32_t*)(sbp + 512);
            const float scale = bf16_to_f32((bf16_t)((sub & 1) ? (sw_ >> 16) : (sw_ & 0xffffu)));
            const float bias = bf16_to_f32((bf16_t)((sub & 1) ? (bw_ >> 16) : (bw_ & 0xffffu)));
            const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
            uint32_t out[8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float q0 = (float)((wd[i] >> (16 * jj)) & 0xffu), q1 = (float)((wd[i] >> (16 * jj + 8)) & 0xffu);
                    out[i * 2 + jj] = pack_bf16x2(q0 * scale + bias, q1 * scale + bias);
                }
            char* brow = smem + B_OFF + (j & 1) * B_BYTES + row * ROW_BYTES;
            const int sw = (row >> 1) & 7;
            *(uint4*)(brow + (((2 * c4) ^ sw) << 4)) = uint4{out[0], out[1], out[2], out[3]};
            *(uint4*)(brow + (((2 * c4 + 1) ^ sw) << 4)) = uint4{out[4], out[5], out[6], out[7]};
        }
    };
    auto role_wait = [&](int need_through, int issued_through, int per) {
        int keep = issued_through - need_through;
        keep = keep < 0 ? 0 : keep;
        wait_vmcnt_dyn(keep * per);
    };

    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int foff0 = frow * ROW_BYTES + ((((lane >> 4) + 0) ^ fsw) << 4);
    const int foff1 = frow * ROW_BYTES + ((((lane >> 4) + 4) ^ fsw) << 4);
    const int a_wave_off = (wr * WM) * ROW_BYTES;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue ----
    int a_issued = -1, b_issued = -1;  // last relative K-tile / macro-tile this role has staged
    if (a_role) {
        for (int j = 0; j < PA && j < nk; ++j) { stage_a(j); a_issued = j; }
    } else {
        for (int m = 0; m < PBM && m < nm; ++m) { stage_b(m); b_issued = m; }
    }
    if constexpr (QB) {  // the first tile's image needs a barrier of its own
        if (!a_role) role_wait(0, b_issued, BPW);
        raw_barrier();
        convert_q(0);
        wait_lgkm_barrier();
    }
    ResidualTile<BM, BN, WGM, WGN> rt;
    for (int t = 0; t < nk; ++t) {
        const int m = t / MT, sub = t % MT;
        // K-tile t of the activations and macro-tile m of the weights (QB: the macro-tile of K-tile t+1, converted below) must have landed
        if (a_role) role_wait(t, a_issued, APW);
        else role_wait(QB ? ((t + 1 < nk ? t + 1 : t) / MT) : m, b_issued, BPW);
        if constexpr (QB) wait_lgkm_barrier(); else raw_barrier();   // QB: also publishes the image converted during K-tile t-1
        // refill what K-tile t-1 / macro-tile m-1 occupied: every wave is past its reads (QB: past the conversion of its last K-tile)
        if (a_role) {
            if (t + PA < nk) { stage_a(t + PA); a_issued = t + PA; }
        } else if (sub == (QB ? 1 : 0)) {
            if (m + PBM < nm) { stage_b(m + PBM); b_issued = m + PBM; }
        }
        if (t == nk - 1 && !g.ep.d2s && g.split_k <= 1 && n0 + BN <= g.N) {  // residual / bias under the last tile's MFMAs
            gemm_bias_prefetch(g, n0, wc, lane, rt);
            if (g.ep.resid) gemm_residual_prefetch(g, m0, n0, wr, wc, lane, rt);
        }
        const char* abase = smem + A_OFF + (t % NA) * A_BYTES + a_wave_off;
        s16x8 fa[2][MI], fb[2][NI];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[k][i] = *(const s16x8*)(abase + i * 16 * ROW_BYTES + (k ? foff1 : foff0));
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if constexpr (QB) {
                    fb[k][j] = *(const s16x8*)(smem + B_OFF + (t & 1) * B_BYTES + (wc * WN + j * 16) * ROW_BYTES + (k ? foff1 : foff0));
                } else {
                    // macro image: row stride 512 B, logical chunk sub*8 + 4k + (lane >> 4) at physical chunk ^ (row & 15)
                    const int row = wc * WN + j * 16 + frow;
                    fb[k][j] = *(const s16x8*)(smem + B_OFF + (m % NBM) * BM_BYTES + row * (MT * ROW_BYTES) +
                                               (((sub * 8 + 4 * k + (lane >> 4)) ^ frow) << 4));
                }
            }
        }
        if constexpr (QB) {
            if (t + 1 < nk) convert_q(t + 1);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[k][i]), __builtin_bit_cast(bf16x8_t, fb[k][j]),
                                                                        acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the epilogue reuses the LDS
    if (g.split_k > 1) {
        GemmArgs gs = g;  // raw partial tile -> workspace slice of this split
        gs.ep = GemmEpilogue{};
        gs.ep.out_f32 = g.split_ws + (long)blockIdx.y * g.M * g.N;
        gs.ep.ld_f32 = g.N;
        gemm_epilogue<BM, BN, WGM, WGN>(acc, gs, m0, n0, wr, wc, lane, wave, smem);
        return;
    }
    gemm_epilogue<BM, BN, WGM, WGN>(acc, g, m0, n0, wr, wc, lane, wave, smem, &rt);
}
#endif  // LTX_EXPERIMENTS (few-row kernel)

#ifdef LTX_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------
// The ring kernel on v_mfma_f32_32x32x16_bf16 (dense launches). Same LDS ring, counted waits and skewed fragment pipeline as _v2;
// what changes is the issue budget: an MFMA holds the SIMD's vector issue port for 8 cycles whether it computes for 16 or for 32
// (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), so the 12 MFMAs of a K-tile leave 24 free cycles per gap where the 24 of
// the 16x16x32 form leave 8 - and the LDS-DMA pieces and fragment reads that the 16x16x32 loop cannot hide fit. Waves 2 x 4, per
// wave 96 x 32 = 3 x 1 blocks of 32 x 32 (48 accumulator registers, as before).
// MEASURED, NOT SELECTED (tile_cfg 31, experiments build): bit-identical to the 16x16x32 ring kernel on integer data, and slower -
// 1536x4096x4096 951 against 1021 TFLOP/s, K = 16384 1083 against 1178 (HBM-cold weights, same process): a 96 x 32 wave tile reads
// 16 KB of fragments per K-tile where 48 x 64 reads 14, and the compiler bunches the LDS-DMA pieces in one gap.
//   operands: lane l holds row (l & 31), k = 8 (l >> 5) .. +7 of the 16-deep k-step  -> one ds_read_b128 per block and k-step
//   result:   lane l holds column (l & 31), rows 8 (v / 4) + 4 (l >> 5) + (v % 4), v = 0..15
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int NSTAGE>
__global__ __launch_bounds__(512) void gemm_bf16_kernel_m32(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WGM = 2, WGN = 4, NW = 8;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 32, NI = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "32x32 blocks");
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PER_WAVE = BM / 8 / NW, B_PER_WAVE = BN / 8 / NW;
    static_assert(A_PER_WAVE * 8 * NW == BM && B_PER_WAVE * 8 * NW == BN, "tile rows must split evenly over the waves");
    constexpr int LPT = A_PER_WAVE + B_PER_WAVE;
    constexpr int PD = NSTAGE - 1;
    constexpr int NF = 2 * (MI + NI);  // fragment reads per half K-tile (two 16-deep k-steps)
    constexpr int NM = 2 * MI * NI;    // MFMAs per half K-tile
    static_assert(NSTAGE >= 3, "ring needs >= 3 slots");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x) + g.tile0;
    int tm, tn;
    tile_coords(g, bid, tiles_m, BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int srow = lane >> 3, pch = lane & 7;
    const bf16_t* a_src[A_PER_WAVE];
    const bf16_t* b_src[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        int gm = m0 + row;
        gm = gm < g.M ? gm : g.M - 1;
        a_src[i] = g.A + (long)gm * g.lda + (pch ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        int gn = n0 + row;
        gn = gn < g.N ? gn : g.N - 1;
        b_src[i] = g.B + (long)gn * g.ldb + (pch ^ ((row >> 1) & 7)) * 8;
    }
    int nk = g.K / BK, kt0 = 0;
    if (g.split_k > 1) {
        const int z = blockIdx.y;
        kt0 = (int)((long)nk * z / g.split_k);
        nk = (int)((long)nk * (z + 1) / g.split_k) - kt0;
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (long)kt * BK),
                                             (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (long)kt * BK),
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + NW * i) * 1024), 16, 0, 0);
    };

    // fragment read offsets: row (lane & 31) of a 32-row block, 16-byte chunk (lane >> 5) + 2 ks of the row, XOR-swizzled by the row
    const int frow = lane & 31;
    const int fsw = (lane >> 1) & 7;  // ((row >> 1) & 7) for row = 32 k + (lane & 31)
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * ROW_BYTES + ((((lane >> 5) + 2 * ks) ^ fsw) << 4);
    const int a_wave_off = (wr * WM) * ROW_BYTES;
    const int b_wave_off = A_BYTES + (wc * WN) * ROW_BYTES;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // two fragment sets (a half K-tile each = two k-steps): [k-step within the half][block]
    s16x8 fa0[2][MI], fb0[2][NI], fa1[2][MI], fb1[2][NI];
    auto load_frags = [&](int slot, int half, s16x8(&fa)[2][MI], s16x8(&fb)[2][NI]) {
        const char* base = smem + slot * STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[k][i] = *(const s16x8*)(base + a_wave_off + i * 32 * ROW_BYTES + foff[2 * half + k]);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[k][j] = *(const s16x8*)(base + b_wave_off + j * 32 * ROW_BYTES + foff[2 * half + k]);
        }
    };
    auto mfma_one = [&](const s16x8(&fa)[2][MI], const s16x8(&fb)[2][NI], int k, int i, int j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[k][i]), __builtin_bit_cast(bf16x8_t, fb[k][j]),
                                                            acc[i][j], 0, 0, 0);
    };
    auto mfma_first = [&](const s16x8(&fa)[2][MI], const s16x8(&fb)[2][NI]) { mfma_one(fa, fb, 0, 0, 0); };
    auto mfma_rest = [&](const s16x8(&fa)[2][MI], const s16x8(&fb)[2][NI]) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (k + i + j > 0) mfma_one(fa, fb, k, i, j);
    };

#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (s < nk) stage(s, kt0 + s);
    if (nk >= PD) wait_vmcnt_barrier<(PD - 1) * LPT>(); else wait_vmcnt_barrier<0>();
    load_frags(0, 0, fa0, fb0);

    int slot = 0;
    bool drained = false;
    auto ktile = [&](int kt, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        int nslot = slot + 1;
        nslot = nslot == NSTAGE ? 0 : nslot;
        int pslot = slot - 1;
        pslot = pslot < 0 ? NSTAGE - 1 : pslot;
        // ---- first half: MFMAs of k-steps 0, 1; the fragment reads of k-steps 2, 3 go two per MFMA gap
        mfma_first(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(slot, 1, fa1, fb1);
        mfma_rest(fa0, fb0);
#pragma unroll
        for (int q = 0; q < (NF + 1) / 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // two ds_reads
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1 - (NF + 1) / 2, 0);
        if constexpr (STEADY) wait_vmcnt_barrier<(PD - 2) * LPT>(); else if (!drained) wait_vmcnt_barrier<0>();
        // ---- second half
        mfma_first(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STEADY) {
            stage(pslot, kt0 + kt + PD);
            load_frags(nslot, 0, fa0, fb0);
            mfma_rest(fa1, fb1);
            // five MFMAs to go: each gap takes one LDS-DMA piece and two of the next tile's fragment reads
#pragma unroll
            for (int q = 0; q < NM - 1; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x020, (LPT + NM - 2) / (NM - 1), 1);
                __builtin_amdgcn_sched_group_barrier(0x100, (NF + NM - 2) / (NM - 1), 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            }
        } else {
            if (kt + 1 < nk) load_frags(nslot, 0, fa0, fb0);
            mfma_rest(fa1, fb1);
        }
        slot = nslot;
    };
    int kt = 0;
    for (; kt < nk - PD; ++kt) ktile(kt, std::true_type{});
    ResidualTile<BM, BN, WGM, WGN> rt;
    if (kt < nk) {
        ktile(kt, std::false_type{});
        ++kt;
        drained = true;
        if (!g.ep.d2s && g.split_k <= 1 && n0 + BN <= g.N && kt < nk) {
            gemm_bias_prefetch(g, n0, wc, lane, rt);
            if (g.ep.resid) gemm_residual_prefetch(g, m0, n0, wr, wc, lane, rt);
        }
    }
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
    __syncthreads();
    // the 16-row slab s of this wave's tile = half (s & 1) of block s / 2: values v = 8 (s & 1) + t, t = 0..7, at row
    // 8 (t / 4) + 4 (lane >> 5) + (t % 4) of the slab, column 32 j + (lane & 31)
    auto put = [&](auto s_c, float* scr) {
        constexpr int sidx = decltype(s_c)::value;
        constexpr int blk = sidx / 2, h = sidx % 2;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t)
                scr[(8 * (t / 4) + 4 * (lane >> 5) + (t % 4)) * WN + 32 * j + (lane & 31)] = acc[blk][j][8 * h + t];
    };
    if (g.split_k > 1) {
        GemmArgs gs = g;
        gs.ep = GemmEpilogue{};
        gs.ep.out_f32 = g.split_ws + ((long)blockIdx.y * (g.win_rows ? g.win_rows : g.M) - g.win_row0) * g.N;
        gs.ep.ld_f32 = g.N;
        gemm_epilogue_with<BM, BN, WGM, WGN, false, true>(put, gs, m0, n0, wr, wc, lane, wave, smem);
        return;
    }
    gemm_epilogue_with<BM, BN, WGM, WGN, false, true>(put, g, m0, n0, wr, wc, lane, wave, smem, &rt);
}

#endif  // LTX_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------------
// small-M path: one wave per output column, f32 activations x bf16 weights
// ---------------------------------------------------------------------------------------------------------------
template <int MMAX>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ a, long lda,
                                                       const bf16_t* __restrict__ W, long ldw,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       long ldo, int M, int N, int K, int in_act) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float accv[MMAX];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) accv[m] = 0.f;
    const bf16_t* wrow = W + (long)n * ldw;
    for (int k = lane * 8; k < K; k += 512) {
        const s16x8 wv = *(const s16x8*)(wrow + k);
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = bf16_to_f32((bf16_t)wv[e]);
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            if (m < M) {
                const f32x4 a0 = *(const f32x4*)(a + (long)m * lda + k);
                const f32x4 a1 = *(const f32x4*)(a + (long)m * lda + k + 4);
                float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = av[e];
                    if (in_act == LTX_ACT_SILU) x = silu_f(x);
                    // an explicit fma: left to contraction, the compiler fused this for some rows m and emitted (packed) mul + add for
                    // others, so identical rows of a batch differed by an ulp (round 4: found by the batch-consistency test)
                    accv[m] = __builtin_fmaf(x, wf[e], accv[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        const float s = wave_reduce_sum(accv[m]);
        if (lane == 0 && m < M) out[(long)m * ldo + n] = s + (bias ? bias[n] : 0.f);
    }
}

#ifdef LTX_EXPERIMENTS
#define LTX_GEMM_EXPERIMENTS_PART 1
#include "gemm_experiments.inc"
#endif

// split-K finish: out = epilogue(sum_z ws[z]) - every epilogue option except the depth-to-space stores; partials are summed in
// ascending z, so the result does not depend on scheduling.
// one 16-byte (f32) or 8-byte (bf16) group of a partial slice
LTX_DEVFN f32x4 ws_load4(const float* ws, long idx, int ws_bf16) {
    if (ws_bf16) {
        const uint2 u = *(const uint2*)((const bf16_t*)ws + idx);
        return f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    }
    return *(const f32x4*)(ws + idx);
}
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int S, int M, int N, GemmEpilogue ep, int ws_bf16 = 0) {
    const long n4 = N >> 2;
    const long total = (long)M * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / n4;
        const int c = (int)(i - m * n4) * 4;
        f32x4 v = ws_load4(ws, m * N + c, ws_bf16);
        for (int z = 1; z < S; ++z) v += ws_load4(ws, ((long)z * M + m) * N + c, ws_bf16);
        if (ep.bias_n) v += *(const f32x4*)(ep.bias_n + c);
        if (ep.bias_m) {
            const float bm = ep.bias_m[m];
            v += f32x4{bm, bm, bm, bm};
        }
        if (ep.act == LTX_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        } else if (ep.act == LTX_ACT_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
        }
        if (ep.round_bf16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
        }
        if (ep.resid) {
            const float* rs = (ep.resid_src ? ep.resid_src + m * ep.ld_resid : ep.out_f32 + m * ep.ld_f32) + c;
            const f32x4 r4 = *(const f32x4*)rs;
            f32x4 gt = f32x4{ep.gate_scalar, ep.gate_scalar, ep.gate_scalar, ep.gate_scalar};
            if (ep.gate) gt = *(const f32x4*)(ep.gate + (long)(ep.gate_rowmap ? ep.gate_rowmap[m] : m / ep.rows_per_batch) * ep.gate_bstride + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r4[e] + gt[e] * v[e];
        }
        if (ep.out_f32) *(f32x4*)(ep.out_f32 + m * ep.ld_f32 + c) = v;
        if (ep.out_bf16) {
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)(ep.out_bf16 + m * ep.ld_bf16 + c) = pk;
        }
        if (ep.out_bf16_t) {  // transposed output (V^T of a few-row launch): 2-byte stores, a few hundred KB in all
#pragma unroll
            for (int e = 0; e < 4; ++e) ep.out_bf16_t[(long)(c + e) * ep.ld_bf16_t + m] = f32_to_bf16(v[e]);
        }
    }
}

// split-K finish of R whole rows of N = 4096 per workgroup + the RMS norm / adaLN modulation of the finished rows (NormAfter): the rows
// are in registers when their f32 values are stored, so the pass that would read them back next is applied here. The finish arithmetic is
// splitk_finish_kernel's and the norm arithmetic norm_mod_rows_kernel's (elementwise.hip), in the same order: the bf16 rows are
// bit-identical to the two launches (tests/test_kernels_gpu.py).
template <int R>
__global__ __launch_bounds__(256) void splitk_finish_norm_kernel(const float* __restrict__ ws, int S, int M, GemmEpilogue ep, NormAfter na, int ws_bf16) {
    constexpr int N = 4096, NV = 4;
    __shared__ float red[4][R];
    const int row0 = blockIdx.x * R;
    const long nb = row0 / na.rows_per_batch;  // the launcher guarantees that the R rows lie in one batch element
    const float* sc = na.scale + nb * na.mod_bstride;
    const float* sh = na.shift + nb * na.mod_bstride;
    f32x4 v[R][NV], s4[NV], h4[NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long m = (row0 + r) < M ? (row0 + r) : M - 1;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (threadIdx.x + j * 256) * 4;
            f32x4 t = ws_load4(ws, m * N + c, ws_bf16);
            for (int z = 1; z < S; ++z) t += ws_load4(ws, ((long)z * M + m) * N + c, ws_bf16);
            if (ep.bias_n) t += *(const f32x4*)(ep.bias_n + c);
            if (ep.resid) {
                const float* rs = (ep.resid_src ? ep.resid_src + m * ep.ld_resid : ep.out_f32 + m * ep.ld_f32) + c;
                const f32x4 r4 = *(const f32x4*)rs;
                f32x4 gt = f32x4{ep.gate_scalar, ep.gate_scalar, ep.gate_scalar, ep.gate_scalar};
                if (ep.gate) gt = *(const f32x4*)(ep.gate + (m / ep.rows_per_batch) * ep.gate_bstride + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = r4[e] + gt[e] * t[e];
            }
            v[r][j] = t;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        s4[j] = *(const f32x4*)(sc + (threadIdx.x + j * 256) * 4);
        h4[j] = *(const f32x4*)(sh + (threadIdx.x + j * 256) * 4);
    }
    float ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += v[r][j][e] * v[r][j][e];
        ss[r] = wave_reduce_sum(a);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[w][r] = ss[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= M) break;
        const long m = row0 + r;
        const float ms = (red[0][r] + red[1][r] + red[2][r] + red[3][r]) / (float)N;
        const float rstd = rsqrtf(ms + na.eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (threadIdx.x + j * 256) * 4;
            if (ep.out_f32) *(f32x4*)(ep.out_f32 + m * ep.ld_f32 + c) = v[r][j];
            if (ep.out_bf16) {
                uint2 pk;
                pk.x = pack_bf16x2(v[r][j][0], v[r][j][1]);
                pk.y = pack_bf16x2(v[r][j][2], v[r][j][3]);
                *(uint2*)(ep.out_bf16 + m * ep.ld_bf16 + c) = pk;
            }
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = v[r][j][e] * rstd;
                if (na.round_norm_bf16) y[e] = bf16_to_f32(f32_to_bf16(y[e]));
                y[e] = y[e] * (1.0f + s4[j][e]) + h4[j][e];
            }
            uint2 pk;
            pk.x = pack_bf16x2(y[0], y[1]);
            pk.y = pack_bf16x2(y[2], y[3]);
            *(uint2*)(na.out + m * na.ldo + c) = pk;
        }
    }
}
// what splitk_finish_norm_kernel covers of the two passes' options
static bool finish_takes_norm(const GemmArgs& a, const NormAfter& na) {
    const GemmEpilogue& e = a.ep;
    return a.N == 4096 && e.out_f32 && !e.out_bf16_t && !e.bias_m && e.act == LTX_ACT_NONE && !e.round_bf16 && !e.gate_rowmap && !e.d2s &&
           !e.pn_out && na.norm_kind == 0 && na.scale && na.shift && !na.row_map && na.out && na.ldo % 4 == 0 && e.ld_f32 % 4 == 0 &&
           na.rows_per_batch % 2 == 0 && (!e.gate || e.rows_per_batch % 2 == 0) && a.M % 2 == 0;
}

template <int BM, int BN, bool CONV>
void launch_one(const GemmArgs& a, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
    static PerDeviceOnce attr_set;
    attr_set.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, CONV>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    });
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, CONV>), dim3(tiles), dim3(256), smem, stream, a);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// 192 x 256 tile, ONE wave per SIMD, assembly main loop generated by tools/gen_gemm_asm_dtl.py (schedule, register map and the
// static proof of its LDS-DMA / ds_read ordering are in that script). The wide DiT GEMMs (fused q|k: N = 8192, FFN up: N = 16384)
// at 1536 tokens are exactly one / two rounds of 256 such tiles, where 192 x 128 tiles need two / four rounds and pay the fixed
// cost of a round (ring fill, epilogue with every workgroup storing at once) twice as often. A 24 + 32 KB K-tile leaves room for
// two activation slots and three weight slots, so both k-steps' fragments are held in registers and the LDS-DMA of A tile t+2 /
// B tile t+3 reuses the slots of tile t as soon as every wave has read them (the loop runs at bytes in flight / HBM latency). Dense A.B^T only, M % 192 == 0, N % 256 == 0, K % 64 == 0. C++ prepares the per-lane offsets / LDS
// addresses and runs the epilogue on the accumulators the assembly leaves in a[0:191].
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void gemm_bf16_kernel_dtl(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 192, BN = 256, WGN = 2;
    constexpr int WM = 96, WN = 128;
    constexpr int A_BYTES = BM * ROW_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const int tiles_m = g.M / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(g, bid, tiles_m, BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    // piece i of this wave = rows (wave + 4 i) * 8 .. +7 of the tile: the per-lane part of its source address (row inside the piece,
    // swizzled 16-byte chunk) does not depend on i ((row >> 1) & 7 is the same for rows 32 apart), the i part is a scalar offset
    const int srow = lane >> 3, pch = lane & 7;
    const int row0 = wave * 8 + srow;
    const int ao = (row0 * (int)g.lda + ((pch ^ ((row0 >> 1) & 7)) << 3)) * 2;
    const int bo = (row0 * (int)g.ldb + ((pch ^ ((row0 >> 1) & 7)) << 3)) * 2;
    const uint32_t sa = (uint32_t)(32 * g.lda * 2), sb = (uint32_t)(32 * g.ldb * 2);
    // deterministic split-K (round 4): grid.y = K range; each range writes its raw partial tile to its workspace slice, the finish pass
    // adds the slices in order and applies the epilogue
    int kt0 = 0, nkt = g.K / BK;
    if (g.split_k > 1) {
        const int z = blockIdx.y;
        kt0 = (int)((long)nkt * z / g.split_k);
        nkt = (int)((long)nkt * (z + 1) / g.split_k) - kt0;
    }
    const bf16_t* At = g.A + (long)m0 * g.lda + (long)kt0 * BK;
    const bf16_t* Bt = g.B + (long)n0 * g.ldb + (long)kt0 * BK;
    const uint32_t alo = (uint32_t)(uintptr_t)At, ahi = (uint32_t)((uintptr_t)At >> 32);
    const uint32_t blo = (uint32_t)(uintptr_t)Bt, bhi = (uint32_t)((uintptr_t)Bt >> 32);
    const uint32_t arec = (uint32_t)(((long)(BM - 1) * g.lda + (long)nkt * BK) * 2), brec = (uint32_t)(((long)(BN - 1) * g.ldb + (long)nkt * BK) * 2);
    const uint32_t nk = (uint32_t)nkt;
    const uint32_t wlds = (uint32_t)wave * 1024u;
    const int frow = lane & 15, fsw = (lane >> 1) & 7;
    const int foff0 = frow * ROW_BYTES + ((((lane >> 4) + 0) ^ fsw) << 4);
    const int foff1 = frow * ROW_BYTES + ((((lane >> 4) + 4) ^ fsw) << 4);
    // LDS: [A slot 0][A slot 1][B slot 0][B slot 1][B slot 2] (tools/gen_gemm_asm_dtl.py)
    const int a_wave_off = (wr * WM) * ROW_BYTES, b_wave_off = 2 * A_BYTES + (wc * WN) * ROW_BYTES;
    const int fa0 = a_wave_off + foff0, fa1 = a_wave_off + foff1, fb0 = b_wave_off + foff0, fb1 = b_wave_off + foff1;
#ifdef DTL_STAMPS  // tools/ubench/gemm_dtl_stamps.hip
    unsigned long long* dbg = (blockIdx.x == 7) ? &g_gemm_stamps[wave][0] : &g_gemm_stamps[4][0];
#endif
    asm volatile(
#ifdef DTL_STAMPS
#include "gemm_dtl_192x256_stamps.inc"
#else
#include "gemm_dtl_192x256.inc"
#endif
        :
        : [alo] "s"(alo), [ahi] "s"(ahi), [arec] "s"(arec), [blo] "s"(blo), [bhi] "s"(bhi), [brec] "s"(brec), [nk] "s"(nk), [sa] "s"(sa),
          [sb] "s"(sb), [wlds] "s"(wlds), [ao] "v"(ao), [bo] "v"(bo), [fa0] "v"(fa0), [fa1] "v"(fa1), [fb0] "v"(fb0), [fb1] "v"(fb1)
#ifdef DTL_STAMPS
          , [dbg] "s"(dbg)
#endif
        :
#include "gemm_dtl_192x256_clobbers.inc"
    );
    __syncthreads();  // every wave is done with the K-tile slots before the epilogue scratch reuses them
    // Epilogue: 32 rows x 128 columns at a time through the wave's LDS scratch (assembly dump shared with tools/gen_gemm_asm.py:
    // same accumulator map), then a rolled loop: one 16-byte column group per lane, two rows per iteration.
    constexpr int LPR = WN / 4, RPI = 64 / LPR;  // lanes per row, rows per iteration
    float* scr = (float*)(smem + wave * (32 * WN * 4));
    // LDS byte address of this lane's first scratch element (the dynamic LDS of this kernel starts at 0, as the main loop assumes)
    const unsigned scr_lane = (unsigned)(wave * (32 * WN * 4) + ((((lane >> 4) * 4) * WN + (lane & 15)) * 4));
    const int gn = n0 + wc * WN + (lane % LPR) * 4;
    if (g.split_k > 1) {  // raw partial tile -> this K range's slice of the workspace
        float* ws = g.split_ws + (long)blockIdx.y * g.M * g.N;
        bf16_t* wsb = (bf16_t*)g.split_ws + (long)blockIdx.y * g.M * g.N;
        static_for<0, 3>([&](auto grp_c) {
            constexpr int grp = decltype(grp_c)::value;
#include "gemm_asm_192x256_dump.inc"
#pragma unroll 4
            for (int it = 0; it < 32 / RPI; ++it) {
                const int row = it * RPI + lane / LPR;
                const f32x4 v = *(const f32x4*)(scr + row * WN + (lane % LPR) * 4);
                if (g.split_bf16) {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)(wsb + (long)(m0 + wr * WM + grp * 32 + row) * g.N + gn) = pk;
                } else {
                    *(f32x4*)(ws + (long)(m0 + wr * WM + grp * 32 + row) * g.N + gn) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
        return;
    }
    const GemmEpilogue& ep = g.ep;
    const f32x4 bias = ep.bias_n ? *(const f32x4*)(ep.bias_n + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* rbase = ep.resid_src ? ep.resid_src : ep.out_f32;
    const long rld = ep.resid_src ? ep.ld_resid : ep.ld_f32;
    if (ep.resid) {
        // Residual launches (round 4; the long-sequence configurations route their gated-residual GEMMs here). The general loop below reads
        // the residual row and the gate with one dependent load per two-row iteration and divides by rows_per_batch for every row: at
        // 9984 tokens a gated launch took 426 us against 276 us for the same product with a plain store. Here the 16 residual loads of a
        // 32-row group are requested together, one group ahead of the accumulator dump, and the gate vector is loaded once when the wave's
        // 96 rows lie in one batch element (always, for one sample per GPU).
        constexpr int NITG = 32 / RPI;
        const int row_lo = m0 + wr * WM, row_hi = row_lo + WM - 1;
        const bool gate_uniform = ep.gate && !ep.gate_rowmap && row_lo / ep.rows_per_batch == row_hi / ep.rows_per_batch;
        f32x4 gtu = f32x4{ep.gate_scalar, ep.gate_scalar, ep.gate_scalar, ep.gate_scalar};
        if (gate_uniform) gtu = *(const f32x4*)(ep.gate + (long)(row_lo / ep.rows_per_batch) * ep.gate_bstride + gn);
        f32x4 rs[2][NITG];
        auto fetch = [&](auto grp_c, auto buf_c) {
            constexpr int grp = decltype(grp_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
            for (int it = 0; it < NITG; ++it) rs[buf][it] = *(const f32x4*)(rbase + (long)(m0 + wr * WM + grp * 32 + it * RPI + lane / LPR) * rld + gn);
        };
        fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 3>([&](auto grp_c) {
            constexpr int grp = decltype(grp_c)::value;
            constexpr int buf = grp & 1;
            if constexpr (grp + 1 < 3) fetch(std::integral_constant<int, grp + 1>{}, std::integral_constant<int, (grp + 1) & 1>{});
#include "gemm_asm_192x256_dump.inc"
            static_for<0, 2>([&](auto half_c) {
                constexpr int half = decltype(half_c)::value;
                f32x4 vv[NITG / 2];  // the scratch reads of eight two-row steps issued together
#pragma unroll
                for (int i2 = 0; i2 < NITG / 2; ++i2) vv[i2] = *(const f32x4*)(scr + ((half * (NITG / 2) + i2) * RPI + lane / LPR) * WN + (lane % LPR) * 4);
#pragma unroll
                for (int i2 = 0; i2 < NITG / 2; ++i2) {
                    const int it = half * (NITG / 2) + i2;
                    const int row = it * RPI + lane / LPR;
                    const int gm = m0 + wr * WM + grp * 32 + row;
                    f32x4 v = vv[i2];
                    v += bias;
                    if (ep.bias_m) {
                        const float bm = ep.bias_m[gm];
                        v += f32x4{bm, bm, bm, bm};
                    }
                    if (ep.act == LTX_ACT_GELU_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                    } else if (ep.act == LTX_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
                    if (ep.round_bf16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
                    }
                    f32x4 gt = gtu;
                    if (ep.gate && !gate_uniform)
                        gt = *(const f32x4*)(ep.gate + (long)(ep.gate_rowmap ? ep.gate_rowmap[gm] : gm / ep.rows_per_batch) * ep.gate_bstride + gn);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = rs[buf][it][e] + gt[e] * v[e];
                    if (ep.out_f32) *(f32x4*)(ep.out_f32 + (long)gm * ep.ld_f32 + gn) = v;
                    if (ep.out_bf16) {
                        uint2 pk;
                        pk.x = pack_bf16x2(v[0], v[1]);
                        pk.y = pack_bf16x2(v[2], v[3]);
                        *(uint2*)(ep.out_bf16 + (long)gm * ep.ld_bf16 + gn) = pk;
                    }
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch is rewritten by the next dump
        });
        return;
    }
    // plain stores (residual launches returned above): the 16 scratch reads of a group are issued together (round 4: the rolled loop
    // waited for one ds_read_b128 per two rows - ~100 cycles x 48 per tile with nothing else to issue on a one-wave SIMD)
    static_for<0, 3>([&](auto grp_c) {
        constexpr int grp = decltype(grp_c)::value;
        constexpr int NITG = 32 / RPI;
#include "gemm_asm_192x256_dump.inc"
        f32x4 vv[NITG];
#pragma unroll
        for (int it = 0; it < NITG; ++it) vv[it] = *(const f32x4*)(scr + (it * RPI + lane / LPR) * WN + (lane % LPR) * 4);
#pragma unroll
        for (int it = 0; it < NITG; ++it) {
            const int gm = m0 + wr * WM + grp * 32 + it * RPI + lane / LPR;
            f32x4 v = vv[it] + bias;
            if (ep.bias_m) {
                const float bm = ep.bias_m[gm];
                v += f32x4{bm, bm, bm, bm};
            }
            if (ep.act == LTX_ACT_GELU_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            } else if (ep.act == LTX_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            }
            if (ep.round_bf16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
            }
            if (ep.out_f32) *(f32x4*)(ep.out_f32 + (long)gm * ep.ld_f32 + gn) = v;
            if (ep.out_bf16) {
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)(ep.out_bf16 + (long)gm * ep.ld_bf16 + gn) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch is rewritten by the next dump
    });
}

#ifdef LTX_EXPERIMENTS
template <int BM, int BN, int NSTAGE>
void launch_m32(const GemmArgs& a, hipStream_t stream) {
    constexpr int smem = NSTAGE * (BM + BN) * ROW_BYTES;
    static PerDeviceOnce attr_set;
    attr_set.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_kernel_m32<BM, BN, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    });
    LTX_REQUIRE(!a.conv && !a.ep.out_bf16_t && !a.ep.pn_out && !a.tile_count, "gemm: the 32x32x16 ring kernel takes dense launches without a transposed output");
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    if (a.split_k > 1) {
        LTX_REQUIRE(a.split_ws && a.N % 4 == 0 && !a.ep.d2s && a.split_k <= a.K / BK, "gemm split-K: workspace, N %% 4 == 0, no depth-to-space");
    }
    hipLaunchKernelGGL((gemm_bf16_kernel_m32<BM, BN, NSTAGE>), dim3(tiles, a.split_k > 1 ? a.split_k : 1), dim3(512), smem, stream, a);
    HIP_CHECK(hipGetLastError());
    if (a.split_k > 1) {
        const long total = (long)a.M * (a.N / 4);
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.N, a.ep, 0);
        HIP_CHECK(hipGetLastError());
    }
}
#endif

template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int WGN = 2, bool QB = false>
void launch_v2(const GemmArgs& a, hipStream_t stream) {
    constexpr int smem = NSTAGE * (BM + BN) * ROW_BYTES + (QB ? (NSTAGE + 1) * (BN * 64 + WGM * WGN * 256) : 0);
    static PerDeviceOnce attr_set;
    attr_set.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_kernel_v2<BM, BN, NSTAGE, CONV, WGM, WGN, QB>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    });
    LTX_REQUIRE(QB == (a.Bq != nullptr), "gemm: quantised codes need the de-quantising instance of the ring kernel");
    const int all_tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int tiles = a.tile_count ? a.tile_count : all_tiles;
    LTX_REQUIRE(a.tile0 >= 0 && a.tile0 + tiles <= all_tiles, "gemm: tile window [%d, %d) outside %d tiles", a.tile0, a.tile0 + tiles, all_tiles);
    if (a.split_k > 1) {
        const GemmEpilogue& e = a.ep;
        LTX_REQUIRE(a.split_ws && a.N % 4 == 0 && !e.d2s, "gemm split-K: needs a workspace, N %% 4 == 0 and no depth-to-space epilogue");
        LTX_REQUIRE(a.split_k <= a.K / BK, "gemm split-K: %d splits for %d K-tiles", a.split_k, a.K / BK);
        LTX_REQUIRE(!a.win_rows || (!e.gate && !e.bias_m && !e.out_bf16_t && a.win_row0 + a.win_rows <= a.M),
                    "gemm split-K: a row window takes neither per-row gates / biases nor the transposed output");
    }
    GemmArgs ab = a;
    ab.conv_stagger = ltx_opt(CONV ? OPT_CONV_STAGGER : OPT_GEMM_STAGGER) != 0;
    hipLaunchKernelGGL((gemm_bf16_kernel_v2<BM, BN, NSTAGE, CONV, WGM, WGN, QB>), dim3(tiles, a.split_k > 1 ? a.split_k : 1),
                       dim3(WGM * WGN * 64), smem, stream, ab);
    HIP_CHECK(hipGetLastError());
    if (a.split_k > 1) {
        const int rows = a.win_rows ? a.win_rows : a.M;
        GemmEpilogue e = a.ep;  // the finish pass indexes rows from the window's first row
        if (a.win_rows) {
            if (e.out_f32) e.out_f32 += (long)a.win_row0 * e.ld_f32;
            if (e.out_bf16) e.out_bf16 += (long)a.win_row0 * e.ld_bf16;
            if (e.resid_src) e.resid_src += (long)a.win_row0 * e.ld_resid;
        }
        const long total = (long)rows * (a.N / 4);
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid), dim3(256), 0, stream, a.split_ws, a.split_k, rows, a.N, e, 0);
        HIP_CHECK(hipGetLastError());
    }
}

#ifdef LTX_EXPERIMENTS
template <bool QB>
void launch_fewrow(const GemmArgs& a, hipStream_t stream) {
    constexpr int smem = 3 * 128 * ROW_BYTES + (QB ? 2 * 64 * ROW_BYTES + 4 * (64 * 256 + 1024) : 3 * 64 * 4 * ROW_BYTES);
    static PerDeviceOnce attr_set;
    attr_set.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)gemm_fewrow_kernel<QB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); });
    LTX_REQUIRE(QB == (a.Bq != nullptr), "gemm: quantised codes need the de-quantising instance of the few-row kernel");
    LTX_REQUIRE(gemm_fewrow_takes(a.M, a.N, a.K, a.split_k) && !a.conv && !a.tile_count && !a.win_rows && !a.ep.d2s && !a.ep.pn_out,
                "gemm: the few-row kernel takes dense launches with M <= 128, N %% 64 == 0 and whole 256-wide macro-tiles of K per split "
                "(M=%d N=%d K=%d splits=%d)", a.M, a.N, a.K, a.split_k);
    const int tiles = (a.N + 63) / 64;
    if (a.split_k > 1) LTX_REQUIRE(a.split_ws && a.N % 4 == 0 && a.split_k <= a.K / BK, "gemm split-K: needs a workspace and N %% 4 == 0");
    hipLaunchKernelGGL(gemm_fewrow_kernel<QB>, dim3(tiles, a.split_k > 1 ? a.split_k : 1), dim3(256), smem, stream, a);
    HIP_CHECK(hipGetLastError());
    if (a.split_k > 1) {
        const long total = (long)a.M * (a.N / 4);
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.N, a.ep, 0);
        HIP_CHECK(hipGetLastError());
    }
}
#endif

void validate(const GemmArgs& a) {
    LTX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    LTX_REQUIRE(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    LTX_REQUIRE(a.A && (a.B || (a.Bq && a.Bqs && a.Bqb)), "gemm: null operand");
    LTX_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && ((uintptr_t)a.Bq & 15) == 0, "gemm: operands must be 16-B aligned");
    LTX_REQUIRE(a.ldb % 8 == 0, "gemm: ldb=%ld must be a multiple of 8", a.ldb);
    if (a.conv) {
        LTX_REQUIRE(a.geom.kt == 1 || a.geom.kt == 3, "gemm/conv3d: kt=%d", a.geom.kt);
        LTX_REQUIRE(a.geom.C % BK == 0 && a.K == 9 * a.geom.kt * a.geom.C, "gemm/conv3d: C=%d K=%d kt=%d", a.geom.C, a.K, a.geom.kt);
        LTX_REQUIRE(a.M == a.geom.F * a.geom.H * a.geom.W, "gemm/conv3d: M=%d != F*H*W", a.M);
        LTX_REQUIRE(a.geom.pad_mode != 0 || (a.geom.H >= 2 && a.geom.W >= 2), "gemm/conv3d: reflect padding needs H,W >= 2");
    } else {
        LTX_REQUIRE(a.lda % 8 == 0, "gemm: lda=%ld must be a multiple of 8", a.lda);
    }
    const GemmEpilogue& e = a.ep;
    LTX_REQUIRE(e.out_f32 || e.out_bf16 || e.out_bf16_t || e.pn_out, "gemm: no output");
    if (e.pn_out) LTX_REQUIRE(e.ld_pn % 4 == 0 && ((uintptr_t)e.pn_out & 7) == 0, "gemm: PixelNorm output alignment");
    if (e.out_f32) LTX_REQUIRE((e.ld_f32 % 4 == 0 || e.d2s == 3) && ((uintptr_t)e.out_f32 & 15) == 0, "gemm: f32 output alignment");
    if (e.out_bf16) LTX_REQUIRE(e.ld_bf16 % 4 == 0 && ((uintptr_t)e.out_bf16 & 7) == 0, "gemm: bf16 output alignment");
    if (e.d2s == 3) LTX_REQUIRE(a.conv && a.N == 48 && e.out_f32 && e.ld_f32 == 1 && !e.out_bf16 && !e.resid && a.split_k <= 1, "gemm: the un-patchify store is for the 48-channel conv_out with one f32 output");
    if (e.resid && !e.d2s) LTX_REQUIRE(e.out_f32 || e.resid_src, "gemm: residual mode needs an f32 stream");
    if (!e.d2s) {  // the interior-column epilogue reads these with 16-B accesses
        LTX_REQUIRE(((uintptr_t)e.bias_n & 15) == 0, "gemm: bias must be 16-B aligned");
        LTX_REQUIRE(((uintptr_t)e.gate & 15) == 0 && e.gate_bstride % 4 == 0, "gemm: gate must be 16-B aligned (stride %ld)", (long)e.gate_bstride);
        LTX_REQUIRE(((uintptr_t)e.resid_src & 15) == 0 && (!e.resid_src || e.ld_resid % 4 == 0), "gemm: residual source alignment");
    }
    if (e.d2s && e.d2s != 3) LTX_REQUIRE(a.conv && a.N % 32 == 0, "gemm: d2s epilogue needs conv mode and N%%32==0");
    if (e.d2s == 1 && e.resid_src) LTX_REQUIRE(a.geom.C % 32 == 0, "gemm: the depth-to-space residual needs C %% 32 == 0 (C=%d)", a.geom.C);
}

}  // namespace

#ifdef LTX_EXPERIMENTS
template <int BN>
static bool gemm_asm_takes(const GemmArgs& a) {
    return !a.conv && a.split_k <= 1 && a.M % 192 == 0 && a.N % BN == 0 && a.K % 64 == 0 && a.K >= 64 && !a.ep.d2s;
}
template <int BN, int RING = 0>
static void launch_asm(const GemmArgs& a, hipStream_t stream) {
    LTX_REQUIRE(gemm_asm_takes<BN>(a), "gemm: the assembly kernel needs a dense A.B^T with M %% 192 == 0, N %% %d == 0, K %% 64 == 0 (M=%d N=%d K=%d)",
                BN, a.M, a.N, a.K);
    constexpr int smem = RING == 2 ? 4 * 192 * ROW_BYTES : (RING ? 4 : 2) * (192 + BN) * ROW_BYTES;
    static PerDeviceOnce attr_set;
    attr_set.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_kernel_asm<BN, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    });
    hipLaunchKernelGGL((gemm_bf16_kernel_asm<BN, RING>), dim3((a.M / 192) * (a.N / BN)), dim3(256), smem, stream, a);
    HIP_CHECK(hipGetLastError());
}

#endif

static bool gemm_dtl_takes(const GemmArgs& a) {
    // lda / ldb % 8: the LDS-DMA pieces are 16-byte loads; the epilogue of this kernel has no clip / PixelNorm / tile-window form
    return !a.conv && a.M % 192 == 0 && a.N % 256 == 0 && a.K % 64 == 0 && a.K >= 64 && !a.ep.d2s && !a.ep.out_bf16_t &&
           a.lda % 8 == 0 && a.ldb % 8 == 0 && !a.ep.pn_out && !a.ep.clip01 && a.tile_count == 0 && a.win_rows == 0 &&
           (a.split_k <= 1 || (a.split_ws && a.split_k <= a.K / 64 && (long)a.split_k * a.M * a.N <= a.split_ws_elems));
}
static void launch_dtl(const GemmArgs& a, hipStream_t stream, const NormAfter* na = nullptr) {
    LTX_REQUIRE(gemm_dtl_takes(a), "gemm: the 192x256 kernel needs a dense A.B^T with M %% 192 == 0, N %% 256 == 0, K %% 64 == 0 (M=%d N=%d K=%d)",
                a.M, a.N, a.K);
    constexpr int smem = (2 * 192 + 3 * 256) * ROW_BYTES;  // two activation slots, three weight slots
    static PerDeviceOnce attr_set;
    attr_set.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_kernel_dtl, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); });
    hipLaunchKernelGGL(gemm_bf16_kernel_dtl, dim3((a.M / 192) * (a.N / 256), a.split_k > 1 ? a.split_k : 1), dim3(256), smem, stream, a);
    HIP_CHECK(hipGetLastError());
    if (a.split_k > 1) {
        if (na) {
            LTX_REQUIRE(finish_takes_norm(a, *na), "gemm: the fused finish + norm pass needs N = 4096, an RMS norm with modulation and no row maps");
            // rows per workgroup (A/B hook: 1, 2 or 4): one row = 1536 workgroups, six per CU in flight - 34.77 / 34.85 / 35.26 ms per step
            // with 1 / 2 / 4 on one box (the pass loads three streams per row, unlike norm_mod_rows_kernel, where two rows won)
            // R rows need M % R == 0 and R | rows_per_batch of both the norm and the gate (a group must not straddle two batch elements,
            // and dim3(M / R) must cover every row): anything else falls back to one row per workgroup (round-4 verdict, Weak 9)
            int fr = ltx_opt(OPT_FINISH_ROWS);
            if ((fr != 2 && fr != 4) || a.M % fr != 0 || na->rows_per_batch % fr != 0 || (a.ep.gate && a.ep.rows_per_batch % fr != 0)) fr = 1;
            if (fr == 1) hipLaunchKernelGGL(splitk_finish_norm_kernel<1>, dim3(a.M), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.ep, *na, a.split_bf16);
            else if (fr == 4)
                hipLaunchKernelGGL(splitk_finish_norm_kernel<4>, dim3(a.M / 4), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.ep, *na, a.split_bf16);
            else hipLaunchKernelGGL(splitk_finish_norm_kernel<2>, dim3(a.M / 2), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.ep, *na, a.split_bf16);
        } else {
            const long total = (long)a.M * (a.N / 4);
            const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.N, a.ep, a.split_bf16);
        }
        HIP_CHECK(hipGetLastError());
    } else {
        LTX_REQUIRE(!na, "gemm: a norm pass can only ride on a split-K finish");
    }
}

#ifdef LTX_EXPERIMENTS
bool gemm_stream_takes(const GemmArgs& a) {
    const GemmEpilogue& e = a.ep;
    return !a.conv && a.M >= 1 && a.M <= 128 && a.K % 64 == 0 && a.K >= 64 && a.N >= 16 && a.N % 4 == 0 && !e.d2s && a.lda % 8 == 0 &&
           (a.Bq ? a.ldb % 16 == 0 : a.ldb % 8 == 0);
}
// K split of the weight-streaming kernel: aim at one to two workgroups per CU, each wave with >= 4 K-tiles
static int stream_split_k(const GemmArgs& a) {
    const int tiles_n = (a.N + 63) / 64, nk = a.K / BK;
    int s = 1;
    while (tiles_n * s * 2 <= 512 && nk / (s * 2) >= 16 && s < 8) s *= 2;
    return s;
}
static void launch_stream(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    LTX_REQUIRE(gemm_stream_takes(a), "gemm: the weight-streaming kernel needs M <= 128, K %% 64 == 0, N %% 4 == 0 (M=%d N=%d K=%d)", a.M, a.N, a.K);
    if (a.split_k == 0) {  // workspace given, split left to the launcher
        a.split_k = stream_split_k(a);
        while (a.split_k > 1 && (long)a.split_k * a.M * a.N > a.split_ws_elems) a.split_k /= 2;
    }
    if (a.split_k < 1 || !a.split_ws) a.split_k = 1;
    constexpr int smem = 4 * 8 * 4 * 64 * 16;  // the four waves' 128 x 64 f32 tiles
    static PerDeviceOnce attr_set;
    attr_set.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_stream_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    });
    const dim3 grid((a.N + 63) / 64, a.split_k);
    if (a.Bq)
        hipLaunchKernelGGL(gemm_stream_kernel<8>, grid, dim3(256), smem, stream, a);
    else
        hipLaunchKernelGGL(gemm_stream_kernel<16>, grid, dim3(256), smem, stream, a);
    HIP_CHECK(hipGetLastError());
    if (a.split_k > 1) {
        const long total = (long)a.M * (a.N / 4);
        const int fgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(fgrid), dim3(256), 0, stream, a.split_ws, a.split_k, a.M, a.N, a.ep, 0);
        HIP_CHECK(hipGetLastError());
    }
}

#endif

// conv_halo.inc: which conv launches the halo-staged kernel takes (everything else stays on the ring kernel)
static bool conv_halo_takes(const GemmArgs& a) {
    const bool off = ltx_opt(OPT_CONV_HALO) == 0;  // A/B option "conv_halo"
    const Conv3dGeom& q = a.geom;
    // (win_rows on a launch without split-K is work accounting only: the head launch of a window pair)
    if (off || !a.conv || q.kt != 3 || !(q.pad_mode == 0 || q.pad_mode == 2) || q.C % 64 != 0 || a.split_k > 1 || a.tile0 != 0) return false;
    if (a.K != 27 * q.C || a.M != q.F * q.H * q.W || a.M < 192 || a.ldb % 8 != 0) return false;
    if ((long)a.M * q.C >= (1L << 31) || (long)a.N * a.ldb >= (1L << 31)) return false;  // the kernel addresses both operands with 32-bit element offsets
    if ((long)a.M * q.W >= (1L << 32) || (long)q.F * q.H * q.H >= (1L << 32) || q.H < 2 || q.W < 2) return false;  // ... and divides by W and H with multiply-high forms
    const bool w_ok = (q.W <= 192 && q.W >= 48 && 192 % q.W == 0) || (q.W % 192 == 0);
    return w_ok && !a.ep.out_bf16_t && !a.ep.gate;  // (its epilogue is the scalar-gate form)
}
template <int BN>
static void launch_conv_halo(const GemmArgs& a, hipStream_t stream);
// conv_halo2.inc: the 384 x 128 tile of NRT image rows -> 0 = not this launch, else NRT (1: W == 384, 2: W == 192, 4: W == 96)
static int conv_halo2_takes(const GemmArgs& a) {
    if (ltx_opt(OPT_CONV_TALL) == 0 || !conv_halo_takes(a)) return 0;
    const Conv3dGeom& q = a.geom;
    const int nrt = (q.W == 384 && ltx_opt(OPT_CONV_TALL) != 2) ? 1 : (q.W == 192 ? 2 : (q.W == 96 ? 4 : 0));  // (2: A/B without the W == 384 instance)
    if (!nrt || q.H % nrt != 0 || a.N % 128 != 0 || a.M % 384 != 0 || a.tile_count != 0 || a.tile0 != 0 || a.ldb % 8 != 0) return 0;
    if (nrt == 4 && q.C % 128 != 0) return 0;                       // (the row-slot rotation must come back to slot 0 or 4 at a tile boundary)
    if (a.ep.d2s != 0 && a.ep.d2s != 1) return 0;                   // plain / fused-PixelNorm / depth-to-space epilogues (conv_out's d2s == 3 has N = 48)
    if (a.ep.pn_out && !(a.N == 128 || (a.ep.d2s == 1 && a.N == 8 * 128))) return 0;
    if ((long)q.F * q.H * q.W * q.C >= (1L << 31) || (long)a.N * a.ldb >= (1L << 31)) return 0;  // 32-bit element offsets
    // A tall tile is two 192-row tiles on ONE CU: a launch whose tall tiles cover at most half the chip finishes sooner as 192-row tiles
    // on twice the CUs (small clips; the stages of a 256 x 256 decode). Option value 3 takes the tall kernel regardless (tests of small shapes).
    if (ltx_opt(OPT_CONV_TALL) != 3 && 2L * (a.M / 384) * (a.N / 128) <= device_cu_count()) return 0;
    return nrt;
}
template <int NRT>
static void launch_conv_halo2_nrt(const GemmArgs& a, hipStream_t stream) {
    constexpr int RSP = (384 / NRT + 2 + 7) / 8, NS = NRT == 1 ? 2 : (NRT == 2 ? 4 : 8);
    constexpr int smem = NS * RSP * 1024 + 1024 + 3 * 128 * ROW_BYTES;
    static PerDeviceOnce attr_set;
    attr_set.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)conv3d_halo2_kernel<NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); });
    const int tiles_n = a.N / 128, rows_t = a.M / 384, tall = rows_t * tiles_n;
    const int ncu = device_cu_count() & ~7;
    // Whole rounds of tall tiles, one workgroup per CU walking its XCD's chunk; what is left over (768 x 512, 128 channels: 1600 = 6 x 256
    // + 64 tall tiles) would keep a quarter of the chip busy for a seventh round, so it runs as 192-row tiles of conv_halo.inc behind the
    // main launch (128 tiles = half a round of tiles half the size). The window starts at a whole number of that kernel's supertiles
    // (group_m row tiles x all column tiles) and of this kernel's row tiles.
    int main_tiles = tall;
    if (ncu >= 8 && tall > ncu && tall % ncu != 0 && tall % ncu <= ncu / 2) {
        const int m = tall / ncu * ncu;
        const int gm = a.group_m > 0 ? a.group_m : 1;
        if (m % tiles_n == 0 && (2 * (m / tiles_n)) % gm == 0) main_tiles = m;
    }
    GemmArgs b = a;
    b.tile_count = main_tiles == tall ? 0 : main_tiles;
    const int grid = (main_tiles > ncu && ncu >= 8) ? ncu : main_tiles;
    hipLaunchKernelGGL(conv3d_halo2_kernel<NRT>, dim3(grid), dim3(512), smem, stream, b);
    HIP_CHECK(hipGetLastError());
    if (main_tiles < tall) {
        GemmArgs t = a;
        t.tile0 = 2 * (main_tiles / tiles_n) * tiles_n;  // 192-row tiles in front of the window (conv_halo.inc's order: supertiles of group_m row tiles)
        t.tile_count = 2 * rows_t * tiles_n - t.tile0;
        launch_conv_halo<128>(t, stream);
    }
}
static void launch_conv_halo2(const GemmArgs& a, hipStream_t stream) {
    const int nrt = conv_halo2_takes(a);
    if (nrt == 1) launch_conv_halo2_nrt<1>(a, stream);
    else if (nrt == 2) launch_conv_halo2_nrt<2>(a, stream);
    else launch_conv_halo2_nrt<4>(a, stream);
}
template <int BN>
static void launch_conv_halo(const GemmArgs& a, hipStream_t stream) {
    constexpr int smem = 2 * 32 * 1024 + 6 * BN * ROW_BYTES;
    static PerDeviceOnce attr_set;
    attr_set.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)conv3d_halo_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); });
    const int all_tiles = ((a.M + 191) / 192) * ((a.N + BN - 1) / BN);
    const int tiles = a.tile_count ? a.tile_count : all_tiles;
    // (a window with tile0 > 0 - the tail behind the tall kernel's whole rounds - starts at a whole number of supertiles: launch_conv_halo2_nrt)
    LTX_REQUIRE(a.tile0 >= 0 && a.tile0 + tiles <= all_tiles, "conv halo: tile window [%d, %d) outside %d", a.tile0, a.tile0 + tiles, all_tiles);
    // persistent above one round: one workgroup per CU walks the tiles of its XCD's chunk and requests the next tile's first operands
    // before the epilogue of the current one (conv_halo.inc); option "conv_persist" = 0 restores one workgroup per tile (A/B)
    const bool persist = ltx_opt(OPT_CONV_PERSIST) != 0;
    const int ncu = device_cu_count() & ~7;
    const int grid = (persist && tiles > ncu && ncu >= 8) ? ncu : tiles;
    GemmArgs b = a;
    b.conv_stagger = ltx_opt(OPT_CONV_STAGGER) != 0;
    hipLaunchKernelGGL(conv3d_halo_kernel<BN>, dim3(grid), dim3(512), smem, stream, b);
    HIP_CHECK(hipGetLastError());
}

void launch_gemm_bf16_cfg(const GemmArgs& a, int cfg, hipStream_t stream) {
    if (a.ep.out_bf16_t) {
        const GemmEpilogue& e = a.ep;
        LTX_REQUIRE(!e.out_f32 && !e.out_bf16 && !e.resid && !e.d2s && !e.bias_m && e.act == LTX_ACT_NONE && !e.round_bf16 &&
                        (cfg == 90 || cfg == 29 || cfg == 30 || (a.split_k <= 1 && !a.split_ws)) && e.ld_bf16_t % 4 == 0 && e.ld_bf16_t >= a.M && (cfg < 41 || cfg == 90),
                    "gemm: the transposed bf16 output takes bias_n only, no split-K, ld %% 4 == 0 (ld=%ld M=%d cfg=%d)", e.ld_bf16_t, a.M, cfg);
    }
    validate(a);
    LTX_REQUIRE(a.split_k <= 1 || (cfg >= 20 && cfg < 32) || cfg == 90 || cfg == 75, "gemm: split-K needs a ring kernel, the 192x256 kernel or the weight-streaming kernel (tile cfg %d)", cfg);
    LTX_REQUIRE(!a.Bq || cfg == 90 || cfg == 29 || cfg == 30, "gemm: quantised codes are read by the few-row ring kernel only (tile cfg %d)", cfg);
    ProfScope prof(a.conv ? PROF_CONV : PROF_GEMM, 2.0 * (a.win_rows ? a.win_rows : a.M) * a.N * a.K, stream);  // a tile window's rows
    // cfg 0..2: v1 (2-stage, one barrier + full drain per K-tile); cfg 10..: v2 (ring + counted vmcnt)
    if (a.conv) {
        switch (cfg) {
            case 21:  // 8 waves (4x2), per-wave 48x64: the halo-staged kernel where it applies (its tall form where THAT applies), else the 4-slot ring
                if (conv_halo2_takes(a)) launch_conv_halo2(a, stream);
                else if (conv_halo_takes(a)) launch_conv_halo<128>(a, stream);
                else launch_v2<192, 128, 4, true, 4, 2>(a, stream);
                break;
#ifdef LTX_EXPERIMENTS  // measured, not selected for convs (VAE decode 19.4 ms with the 256x128 ring, 22.8 with the two-stage kernel, against 17.0)
            case 0: launch_one<128, 128, true>(a, stream); break;
            case 1: launch_one<192, 128, true>(a, stream); break;
            case 3: launch_one<96, 128, true>(a, stream); break;
            case 4: launch_one<128, 96, true>(a, stream); break;
            case 23: launch_v2<256, 128, 3, true, 4, 2>(a, stream); break;  // 8 waves, per-wave 64x64
            case 25: launch_v2<128, 192, 4, true, 2, 4>(a, stream); break;
#endif
            case 27:  // narrow outputs (the VAE's 128 -> 48 conv_out): the halo-staged kernel's 192x64 instance, else the 256x64 ring (per-wave 64x32)
                if (a.N <= 64 && conv_halo_takes(a)) launch_conv_halo<64>(a, stream); else launch_v2<256, 64, 4, true, 4, 2>(a, stream);
                break;
            default: LTX_THROW(LTXS_INVALID_CONFIGURATION, "gemm: unknown tile cfg %d", cfg);
        }
    } else {
        switch (cfg) {
            case 0: launch_one<128, 128, false>(a, stream); break;
            case 1: launch_one<192, 128, false>(a, stream); break;
            case 3: launch_one<96, 128, false>(a, stream); break;
            case 4: launch_one<128, 96, false>(a, stream); break;
            case 21: launch_v2<192, 128, 4, false, 4, 2>(a, stream); break;  // 8 waves (4x2), per-wave 48x64, 4-slot ring
            case 22: { GemmArgs b = a; b.group_m = 0; launch_v2<192, 128, 4, false, 4, 2>(b, stream); break; }  // A/B: column-major order
            case 23: launch_v2<256, 128, 3, false, 4, 2>(a, stream); break;  // 8 waves, per-wave 64x64
            case 25: launch_v2<128, 192, 4, false, 2, 4>(a, stream); break;
            case 29:  // few rows: narrow column tiles, 4 waves, one workgroup per CU; 8-bit codes de-quantised in the B stage.
                      // bf16 weights: SIX ring slots (144 KB) - five K-tiles of weights in flight per CU instead of three: the
                      // weight stream of a 128-token launch goes from 3.45 to 3.65 TB/s (same-box A/B, tools/fewrow_stride.py)
                if (a.Bq) {
                    launch_v2<128, 64, 4, false, 2, 2, true>(a, stream);
                } else {
                    const int nt_env = ltx_opt(OPT_B_NT);  // A/B option "b_nt": 0 / 1 overrides the launcher
                    if (nt_env >= 0 && a.b_nt != nt_env) {
                        GemmArgs b = a;
                        b.b_nt = nt_env;
                        launch_v2<128, 64, 6, false, 2, 2>(b, stream);
                    } else {
                        launch_v2<128, 64, 6, false, 2, 2>(a, stream);
                    }
                }
                break;
#ifdef LTX_EXPERIMENTS
            case 28: launch_v2<128, 64, 4, false, 2, 2>(a, stream); break;  // cfg 29 with the 4-slot ring it had until round 3 (A/B)
            case 30:  // few rows (M <= 128), operand rings of their own, weights in macro-tiles: measured, not selected
                if (a.Bq) launch_fewrow<true>(a, stream); else launch_fewrow<false>(a, stream);
                break;
#endif
            case 75: launch_dtl(a, stream); break;  // 192x256, one wave per SIMD, fragments of a whole K-tile in registers, 2 + 3 LDS slots
#ifdef LTX_EXPERIMENTS  // measured, not selected (gemm_experiments.inc)
            case 31: launch_m32<192, 128, 4>(a, stream); break;  // the 192x128 ring tile on 32x32x16 MFMAs (2 x 4 waves of 96 x 32): 951 vs 1021 TFLOP/s
            case 90: launch_stream(a, stream); break;  // weight-streaming kernel for M <= 128 (bf16 weights or 8-bit codes): 1.4x slower than the ring split-K path
            case 71: launch_asm<256>(a, stream); break;      // one wave per SIMD, assembly main loop, 192x256
            case 72: launch_asm<128>(a, stream); break;      // the same, 192x128
            case 73: launch_asm<128, 1>(a, stream); break;  // 192x128, one wave per SIMD, LDS-DMA ring of four slots
            case 74: launch_asm<128, 2>(a, stream); break;  // the same with B as fragment-layout loads straight to registers
            case 41: launch_v4<192, 256>(a, stream); break;  // ping-pong, 8 waves (2x4), per-wave 96x64
            case 42: launch_v4<256, 256>(a, stream); break;  // ping-pong, per-wave 128x64
#endif
            default: LTX_THROW(LTXS_INVALID_CONFIGURATION, "gemm: unknown tile cfg %d", cfg);
        }
    }
}

bool gemm_takes_codes(int M, int N, int K) { return M >= 1 && M <= 256 && K % BK == 0 && K >= 4 * BK && N % 4 == 0; }
// the few-row kernel (tile_cfg 30): one 128-row tile, whole 64-column tiles, K of every split a whole number of 256-wide macro-tiles
bool gemm_fewrow_takes(int M, int N, int K, int splits) {
    if (splits < 1) splits = 1;
    return M >= 1 && M <= 128 && N % 64 == 0 && K % 256 == 0 && (K / 256) % splits == 0;
}

int gemm_suggest_split_k(int M, int N, int K) {
    const long tiles = (long)((M + 191) / 192) * ((N + 127) / 128);  // the 192x128 ring kernel's tiles
    const int nk = K / BK;
    if (tiles >= 160 || nk < 32) return 1;
    long s = 256 / tiles;
    if (s > 8) s = 8;
    if (s > nk / 8) s = nk / 8;
    return s < 1 ? 1 : (int)s;
}

// At most one round of 192x128 ring tiles with a long reduction (the FFN's second GEMM at 1536 tokens: 256 tiles, K = 16384) runs as K
// ranges of the 192x256 kernel, whose main loop reads half the fragments per MFMA, plus the finish pass: 36.46 -> 35.75 ms per headline
// step on one box. -> number of ranges, 0 = not this launch. Option "dtl_splitk" = 0 turns it off; "dtl_splitk_mink" = least K per range
// (8192; with 2048 the three K = 4096 launches of a block split too and the step LOSES 1.3 ms).
// Partial tiles of such a launch cross the workspace as bf16 ONLY when the caller asked for it (GemmArgs::split_bf16 - the DiT sets it for
// its FFN-down Linear, nothing else does: a generic or ABI launch keeps f32 partials, round-4 advice) and option "split_f32" is 0.
static int dtl_split_bf16(const GemmArgs& a) { return a.split_bf16 && ltx_opt(OPT_SPLIT_F32) == 0; }
static int dtl_split_for(const GemmArgs& a) {
    const bool dtl_split = ltx_opt(OPT_DTL_SPLITK) != 0;
    const int dtl_mink = ltx_opt(OPT_DTL_SPLITK_MINK);
    if (!dtl_split || dtl_mink < 64 || a.conv || a.Bq || !a.split_ws || a.split_k != 0 || a.M % 192 != 0 || a.N % 256 != 0 || a.ep.d2s) return 0;
    if (gemm_suggest_split_k(a.M, a.N, a.K) > 1) return 0;  // fewer than 160 ring tiles: the ring kernel's own split-K
    const long t1 = (long)(a.M / 192) * (a.N / 256);
    const int ncu = device_cu_count();
    long s = ncu / t1;
    if (s > a.K / dtl_mink) s = a.K / dtl_mink;
    while (s > 1 && s * a.M * a.N > a.split_ws_elems) --s;
    if (s < 2 || s * t1 <= ncu / 2) return 0;
    GemmArgs b = a;
    b.split_k = (int)s;
    return gemm_dtl_takes(b) ? (int)s : 0;
}

void launch_gemm_bf16(const GemmArgs& a, hipStream_t stream, const NormAfter* na) {
    LTX_REQUIRE(na && a.ep.out_f32 && !a.conv, "gemm: norm_after needs a dense launch with an f32 output");
    const bool fuse_on = ltx_opt(OPT_FINISH_NORM) != 0;  // A/B option "finish_norm"
    const int s2 = (fuse_on && finish_takes_norm(a, *na)) ? dtl_split_for(a) : 0;
    if (s2) {
        GemmArgs b = a;
        b.split_k = s2;
        b.split_bf16 = dtl_split_bf16(a);
        validate(b);
        ProfScope prof(PROF_GEMM, 2.0 * a.M * a.N * a.K, stream);
        launch_dtl(b, stream, na);
        return;
    }
    launch_gemm_bf16(a, stream);
    launch_norm_mod(a.ep.out_f32, a.ep.ld_f32, na->scale, na->shift, na->mod_bstride, na->rows_per_batch, na->out, na->ldo, a.M, a.N, na->norm_kind,
                    na->eps, na->round_norm_bf16, stream, na->row_map);
}

void launch_gemm_bf16(const GemmArgs& a_in, hipStream_t stream) {
    // A/B hook: LTX_GEMM_GROUP_M="N:g,..." overrides the supertile height of dense launches with that N (tile_coords)
    GemmArgs a = a_in;
#ifdef LTX_EXPERIMENTS  // string-valued tile experiments stay environment hooks of the experiments build only
    if (const char* f = getenv("LTX_GEMM_GROUP_M")) {
        for (const char* q = f; *q;) {
            const long n = strtol(q, (char**)&q, 10);
            if (*q != ':') break;
            const long gm = strtol(q + 1, (char**)&q, 10);
            if (n == a.N && !a.conv) a.group_m = (int)gm;
            if (*q == ',') ++q;
        }
    }
#endif
    // Tile choice = workgroup-count quantisation x structure efficiency. Two structures:
    //   v1 (cfg 0,1,3,4): 4 waves, 2 LDS stages, TWO workgroups resident per CU (512 slots) - two waves per SIMD from
    //       co-residency; best when a launch has >= 512 tiles. Its single-tile prefetch exposes HBM latency on
    //       weights that are streamed once (the DiT case), hence the lower factor for the small tiles.
    //   v2 (cfg 21,23,25): 8 waves (two per SIMD inside ONE workgroup per CU, 256 slots), 3/4-slot LDS ring filled
    //       2-3 K-tiles ahead with counted vmcnt - HBM-cold weights cost nothing; best at exactly k*256 tiles.
    // Measured on MI355X, HBM-cold B operand, random data (tools/bench_gemm.py --cold), TFLOP/s:
    //   M=1536 N=4096 K=4096 : cfg21 944 | cfg0 754 | cfg1 608      N=8192: cfg1 1026 | cfg21 949
    //   N=16384: cfg1 1049 | cfg21 991         K=16384: cfg21 1094 | cfg0 864       M=4096 N=1536: cfg25 876 | cfg0 629
    struct Cand { int cfg, bm, bn, slots; double f; };
    //   cfg 75: 192x256 tile, one wave per SIMD, assembly main loop (tools/gen_gemm_asm_dtl.py): half the rounds of a 192x128 tile.
    //   Same measurement, round 2 (profiles/r02_gemm_microbench.txt): N=8192: cfg75 1165 | library GEMM 1116 | cfg1 936 | cfg21 959;
    //   N=16384: 1274 | 1294 | 1054 | 1009; 6144x4096x4096: 1290 | 1325 | 1093 | 1033; 1536x4096x4096 (128 tiles): 740 -> not chosen.
    static const Cand cands[] = {{75, 192, 256, 256, 1.22}, {1, 192, 128, 512, 1.00}, {21, 192, 128, 256, 0.95}, {25, 128, 192, 256, 0.90},
                                 {0, 128, 128, 512, 0.80},  {3, 96, 128, 512, 0.72},  {4, 128, 96, 512, 0.72}};
    if (a.conv) {
        // implicit-GEMM convs: the per-tap gather arithmetic must hide under MFMAs (8-wave ring kernels interleave it;
        // the two-stage 4-wave kernel serialises it and ran the 256-channel VAE stage at 150 TFLOP/s). M is huge, so
        // tile-count quantisation does not matter; N <= 128 wants the 192x128 tile, wide N the same (B re-use).
        const int cc_opt = ltx_opt(OPT_CONV_CFG);  // A/B option "conv_cfg" for tile experiments (21 = 192x128 ring; 23 = 256x128 ring in the experiments build)
        const bool cc = cc_opt != 0;
        // N <= 64 (the decoder's conv_out, 128 -> 48 channels): a 192x128 tile spends 62 % of its MFMAs on padding columns
        const int conv_default = (a.N <= 64 && a.split_k <= 1) ? 27 : 21;
        if (a.ep.pn_out) {
            // the fused PixelNorm output needs every channel of a row in ONE 128-column tile and the epilogue in the GEMM launch
            // (round 5: or, with the depth-to-space store of an upsampler, N == 8 x 128: a tile is one sub-position's 128 channels; only the
            // halo-staged kernels carry that epilogue)
            LTX_REQUIRE(((a.N == 128 && !a.ep.d2s) || (a.ep.d2s == 1 && a.N == 8 * 128 && conv_halo_takes(a))) && a.split_k <= 1 &&
                            (a.ep.pn_scale == nullptr) == (a.ep.pn_shift == nullptr),
                        "gemm: fused PixelNorm output needs N == 128 (got %d; 1024 with a depth-to-space store on a halo-staged launch) and no split-K", a.N);
            // (A last partial round - the 128-channel stage at 768x512 is 3200 tiles = 12.5 rounds - is NOT worth a launch of its own here:
            // running the 128 remaining tiles as 256 half-height tiles behind the whole rounds measured +0.28 ms per decode, round 3.
            // Tiles are dispatched as CUs free up, so the ragged end costs half a tile, less than a launch boundary.)
            launch_gemm_bf16_cfg(a, 21, stream);
            return;
        }
        // Last partial round: T tiles = q full rounds of 256 + r. With r <= 128 the r tiles of the last round would run K-long on
        // r CUs while the others idle (the VAE's 256-channel stage: 832 tiles = 3.25 rounds, 19 % of every conv). Run the full
        // rounds as one launch and the remainder as a split-K launch over all CUs (window fields of GemmArgs).
        const bool no_tail = ltx_opt(OPT_CONV_NO_TAIL) != 0;  // A/B option "conv_no_tail"
        if (!cc && conv_default == 21 && conv_halo2_takes(a)) {  // the tall kernel balances its own last round (launch_conv_halo2)
            launch_gemm_bf16_cfg(a, 21, stream);
            return;
        }
        if (!no_tail && !cc && conv_default == 21 && a.split_k <= 1 && a.split_ws && !a.ep.d2s && a.N % 4 == 0 && !a.ep.gate && !a.ep.bias_m &&
            !a.ep.out_bf16_t && a.group_m > 0) {
            const int tiles_m = (a.M + 191) / 192, tiles_n = (a.N + 127) / 128, tiles = tiles_m * tiles_n;
            const int full = tiles / 256 * 256, tail = tiles - full, per = a.group_m * tiles_n, nk = a.K / BK;
            if (full >= 512 && tail > 0 && tail <= 128 && full % per == 0) {
                const int row0 = full / per * a.group_m * 192;  // whole supertiles before the window: it covers complete rows
                int sk = 256 / tail;
                if (sk > 8) sk = 8;
                if (sk > nk / 8) sk = nk / 8;
                while (sk > 1 && (long)sk * (a.M - row0) * a.N > a.split_ws_elems) --sk;
                if (sk > 1) {
                    GemmArgs head = a, rest = a;
                    head.tile_count = full;
                    head.win_rows = row0;  // rows [0, row0): only the work accounting reads it on a launch without split-K
                    head.split_ws = nullptr;
                    rest.tile0 = full;
                    rest.tile_count = tail;
                    rest.split_k = sk;
                    rest.win_row0 = row0;
                    rest.win_rows = a.M - row0;
                    launch_gemm_bf16_cfg(head, 21, stream);
                    launch_gemm_bf16_cfg(rest, 21, stream);
                    return;
                }
            }
        }
        launch_gemm_bf16_cfg(a, (cc && a.split_k <= 1) ? cc_opt : conv_default, stream);
        return;
    }
    LTX_REQUIRE(!a.ep.pn_out, "gemm: the fused PixelNorm output exists for conv launches only");
    LTX_REQUIRE(!a.Bq || (gemm_takes_codes(a.M, a.N, a.K) && a.split_ws && a.split_k == 0),
                "gemm: 8-bit codes are de-quantised in the B stage of few-row launches only (M=%d N=%d K=%d)", a.M, a.N, a.K);
    if (a.split_ws && a.split_k == 0) {
        // caller-provided workspace, split count left to the launcher: few output tiles and a long reduction (the DiT at
        // small token counts, e.g. 256x256x9 -> 128 tokens: 32 tiles of weights to stream with 256 CUs) -> split K over the idle CUs
        GemmArgs b = a;
        b.split_k = gemm_suggest_split_k(a.M, a.N, a.K);
        // few rows (<= 128, e.g. 256x256x9): narrow 128x64 ring tiles - 64 column tiles at N = 4096, so 4 splits fill the chip and
        // the partial slices are a quarter of the weights (128x192 tiles needed 16 splits: as many partial bytes as weight bytes),
        // and the FFN's first GEMM (256 column tiles) needs no split at all. Config 1: 12.04 ms per forward against 12.25 (cfg 25).
        const int smallm_cfg = ltx_opt(OPT_SMALLM_CFG);  // 30 (experiments build): the few-row kernel, 9 % slower
        // codes in the B stage exist only in the 128x64 ring (29) and the experiments build's few-row kernel (30): the hook cannot move a
        // quantised launch anywhere else
        const int smallm_q = (smallm_cfg == 29 || smallm_cfg == 30) ? smallm_cfg : 29;
        int cfg = a.Bq ? (a.M <= 128 ? smallm_q : 29) : ((a.M <= 128 && b.split_k > 1) ? smallm_cfg : 21);
        if (cfg == 29 || cfg == 30) {
            const long tiles = (long)((a.N + 63) / 64);
            long sk = 256 / tiles;
            const int nk = a.K / BK;
            if (sk > 16) sk = 16;
            if (sk > nk / 8) sk = nk / 8;
            b.split_k = sk < 1 ? 1 : (int)sk;
            while (b.split_k > 1 && (long)b.split_k * a.M * a.N > a.split_ws_elems) --b.split_k;
            if (cfg == 30 && !gemm_fewrow_takes(a.M, a.N, a.K, b.split_k)) cfg = 29;
            // once-read weights, non-temporal - from 64 MB up: the q|k, FFN-up and FFN-down launches gain 3-6 %, the 33.5 MB ones
            // (N = K = 4096, four splits) LOSE 12 % with it (profiles/r03_fewrow_bounds.txt)
            b.b_nt = (long)a.N * a.K >= (32L << 20) ? 1 : 0;
        }
        if (cfg == 25) {
            const long tiles = (long)((a.N + 191) / 192);
            long sk = 256 / tiles;
            const int nk = a.K / BK;
            if (sk > 16) sk = 16;
            if (sk > nk / 8) sk = nk / 8;
            b.split_k = sk < 1 ? 1 : (int)sk;
        }
        while (b.split_k > 1 && (long)b.split_k * a.M * a.N > a.split_ws_elems) --b.split_k;
        LTX_REQUIRE(!a.Bq || (a.N % 4 == 0 && !a.ep.d2s), "gemm: quantised few-row launch with N %% 4 != 0");
        if ((b.split_k > 1 || cfg == 29 || cfg == 30) && a.N % 4 == 0 && !a.ep.d2s) {
            if (b.split_k <= 1) {
                b.split_k = 1;  // enough column tiles to fill the chip without a split (the FFN's first GEMM)
                b.split_ws = nullptr;
            }
            launch_gemm_bf16_cfg(b, cfg, stream);
            return;
        }
        if (const int s2 = dtl_split_for(a)) {
            b.split_k = s2;
            b.split_bf16 = dtl_split_bf16(a);
            launch_gemm_bf16_cfg(b, 75, stream);
            return;
        }
        b.split_k = 1;
        b.split_ws = nullptr;
        launch_gemm_bf16(b, stream);
        return;
    }
#ifdef LTX_EXPERIMENTS
    // The assembly kernel (tile_cfg 71: 192 x 256 tile, one wave per SIMD, generated main loop) is NOT in the default choice.
    // Measured on MI355X, HBM-cold weights, random data, same process (tools/bench_gemm.py --cold --cfgs=-1,21,1,71), TFLOP/s:
    // 1536x16384x4096 1057 vs 1009 (two-stage) / 987 (ring); 6144x4096x4096 1069 vs 1053 / 1015; 1536x8192x4096 895 vs 930 (ring);
    // but inside the DiT step (same-box A/B of bench.py) routing the FFN's first GEMM to it costs 0.3 ms per step (41.6 vs 41.3).
    // Its loop runs 2380 cycles per K-tile against 1730 of MFMA issue (14 buffer loads + 14 ds_write_b128 per wave and tile are
    // not free) and its epilogue (10 us for 50 MB of output, every workgroup at the same time) is exposed once per launch.
    // LTX_GEMM_ASM=1 opts in for launches that fill the chip at least twice with 192x256 tiles (A/B runs).
    static const bool asm_on = getenv("LTX_GEMM_ASM") && getenv("LTX_GEMM_ASM")[0] == '1';
    if (asm_on && gemm_asm_takes<256>(a)) {
        const long tiles_asm = (long)(a.M / 192) * (a.N / 256);
        if (tiles_asm >= 512 && tiles_asm % 256 == 0 && a.K <= 8192) {
            launch_gemm_bf16_cfg(a, 71, stream);
            return;
        }
    }
#endif
#ifdef LTX_EXPERIMENTS
    // A/B hook for in-pipeline tile experiments (experiments build only): LTX_GEMM_FORCE="8192:21,16384:21" forces a tile cfg for dense launches by N
    if (const char* f = getenv("LTX_GEMM_FORCE")) {
        for (const char* q = f; *q;) {
            const long n = strtol(q, (char**)&q, 10);
            if (*q != ':') break;
            const long c = strtol(q + 1, (char**)&q, 10);
            const bool takes = c == 71 ? gemm_asm_takes<256>(a) : (c >= 72 && c <= 74) ? gemm_asm_takes<128>(a) : c == 75 ? gemm_dtl_takes(a) : true;
            if (n == a.N && takes) {
                launch_gemm_bf16_cfg(a, (int)c, stream);
                return;
            }
            if (*q == ',') ++q;
        }
    }
#endif
    int best = 0;
    double be = -1;
    for (const Cand& c : cands) {
        if (c.cfg == 75 && !gemm_dtl_takes(a)) continue;
        const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
        const long rounds = (tiles + c.slots - 1) / c.slots;
        const double fill = ((double)a.M * a.N) / ((double)tiles * c.bm * c.bn);  // ragged edges waste MFMA work
        const double e = (double)tiles / (double)(rounds * c.slots) * c.f * fill;
        if (e > be + 1e-9) { be = e; best = c.cfg; }
    }
    // Ragged last round of the 192x256 kernel (round 4). With T = 9984 tokens the N = 4096 launches are 52 x 16 = 832 tiles = 3.25 rounds:
    // the fourth round runs 64 workgroups, K-long, on a quarter of the chip (19 % of the launch). Rows are independent, so such a launch is
    // split by rows into a head whose tiles are a whole number of rounds (48 row tiles = 768 tiles) and a tail (4 row tiles = 768 rows)
    // that goes through this function again - as 128 tiles of 192x128 on the ring kernel it takes half the time of the K-long quarter
    // round. Same kernels per output element as an unsplit launch of each part: bit-identical to those; option "gemm_rowsplit" = 0 = off (A/B).
    const bool rowsplit_on = ltx_opt(OPT_GEMM_ROWSPLIT) != 0;
    if (best == 75 && rowsplit_on && !a.ep.out_bf16_t && !a.ep.bias_m && a.split_k <= 1) {
        const int ncu = device_cu_count();
        const int tm = a.M / 192, tn = a.N / 256;
        const long tiles = (long)tm * tn;
        const long rem = tiles % ncu;
        // the gate vector of a residual launch is indexed by m / rows_per_batch (or a row map): a row offset keeps that meaning only inside
        // one batch element
        const bool gate_ok = !a.ep.gate || a.ep.gate_rowmap || a.ep.rows_per_batch >= a.M;
        if (tiles > ncu && rem > 0 && rem <= ncu / 2 && gate_ok) {
            int g = tn, h = ncu;
            while (h) { const int t = g % h; g = h; h = t; }  // gcd(tn, ncu)
            const int step = ncu / g;                          // row tiles per whole number of rounds
            const int tm_head = (tm / step) * step;
            if (tm_head > 0 && tm_head < tm) {
                auto rows = [&](int row0, int nrows) {
                    GemmArgs s = a;
                    GemmEpilogue& e = s.ep;
                    s.A = a.A + (long)row0 * a.lda;
                    s.M = nrows;
                    if (e.out_f32) e.out_f32 += (long)row0 * e.ld_f32;
                    if (e.out_bf16) e.out_bf16 += (long)row0 * e.ld_bf16;
                    if (e.resid_src) e.resid_src += (long)row0 * e.ld_resid;
                    if (e.gate_rowmap) e.gate_rowmap += row0;
                    return s;
                };
                launch_gemm_bf16_cfg(rows(0, tm_head * 192), 75, stream);
                launch_gemm_bf16(rows(tm_head * 192, a.M - tm_head * 192), stream);
                return;
            }
        }
    }
    launch_gemm_bf16_cfg(a, best, stream);
}

void launch_gemv_f32(const float* a, long lda, const bf16_t* W, long ldw, const float* bias, float* out, long ldo,
                     int M, int N, int K, int in_act, hipStream_t stream) {
    LTX_REQUIRE(M >= 1 && M <= 8, "gemv: M=%d out of range (1..8)", M);
    LTX_REQUIRE(K % 8 == 0 && lda % 4 == 0 && ldw % 8 == 0, "gemv: K=%d lda=%ld ldw=%ld alignment", K, lda, ldw);
    const int grid = (N + 3) / 4;
    if (M <= 2)
        hipLaunchKernelGGL((gemv_f32_kernel<2>), dim3(grid), dim3(256), 0, stream, a, lda, W, ldw, bias, out, ldo, M, N, K, in_act);
    else
        hipLaunchKernelGGL((gemv_f32_kernel<8>), dim3(grid), dim3(256), 0, stream, a, lda, W, ldw, bias, out, ldo, M, N, K, in_act);
    HIP_CHECK(hipGetLastError());
}
