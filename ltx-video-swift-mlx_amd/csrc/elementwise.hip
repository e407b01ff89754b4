// elementwise.hip - HBM-bound kernels of the DiT / VAE path (see elementwise.h for the reference call sites).
// All are streaming kernels: 16-B (or 8-B bf16) accesses per lane, one pass over the data, reductions via
// wave shuffles + one LDS hop. Roofline for each is HBM bytes / 8 TB/s; none of them is reshaped into a GEMM.
#include "elementwise.h"

#include <stdlib.h>

#include "options.h"
#include "runtime.h"

namespace {

constexpr int MAXV = 8;  // float4 chunks per thread kept in registers by the row kernels (D <= 8192)

LTX_DEVFN float block_reduce_sum(float v, float* red) {
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// one workgroup (256 threads) per row; NV = float4 chunks per thread (D <= NV*1024). The modulation vectors are
// fetched together with the row, BEFORE the reduction: every workgroup of the launch is resident at once, so the kernel
// costs one memory latency plus the transfer, and a second dependent load after the barrier doubled it (17 -> ~9 us).
template <int NV>
__global__ __launch_bounds__(256) void norm_mod_kernel(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, long mod_bstride,
                                                       int rows_per_batch, bf16_t* __restrict__ out, long ldo, int D,
                                                       int norm_kind, float eps, int round_norm_bf16,
                                                       const int32_t* __restrict__ row_map) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (long)row * ldx;
    const long b = row_map ? row_map[row] : row / rows_per_batch;  // modulation row (per-token timestep groups: I2V)
    const float* sc = scale ? scale + b * mod_bstride : nullptr;
    const float* sh = shift ? shift + b * mod_bstride : nullptr;
    f32x4 v[NV], s4[NV], h4[NV];
    float s1 = 0.f, s2 = 0.f;
    const int nchunk = D >> 2;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + j * 256;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nchunk) v[j] = *(const f32x4*)(xr + c * 4);
    }
    if (sc) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = threadIdx.x + j * 256;
            if (c < nchunk) {
                s4[j] = *(const f32x4*)(sc + c * 4);
                h4[j] = *(const f32x4*)(sh + c * 4);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1 += v[j][e];
            s2 += v[j][e] * v[j][e];
        }
    float mean = 0.f, rstd;
    if (norm_kind == LTX_NORM_LAYER) {
        mean = block_reduce_sum(s1, red) / (float)D;
        // two-pass variance on the register copy (population variance, as MLXNN.LayerNorm)
        float sv = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = threadIdx.x + j * 256;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[j][e] - mean;
                    sv += d * d;
                }
            }
        }
        const float var = block_reduce_sum(sv, red) / (float)D;
        rstd = rsqrtf(var + eps);
    } else {
        const float ms = block_reduce_sum(s2, red) / (float)D;
        rstd = rsqrtf(ms + eps);
    }
    bf16_t* orow = out + (long)row * ldo;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < nchunk) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (v[j][e] - mean) * rstd;
                if (round_norm_bf16) y[e] = bf16_to_f32(f32_to_bf16(y[e]));
            }
            if (sc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = y[e] * (1.0f + s4[j][e]) + h4[j][e];
            }
            uint2 pk;
            pk.x = pack_bf16x2(y[0], y[1]);
            pk.y = pack_bf16x2(y[2], y[3]);
            *(uint2*)(orow + c * 4) = pk;
        }
    }
}

// R rows per workgroup (RMS norm, D = NV * 1024 exactly, rows of one batch element): the modulation vectors are the same for every
// row of a batch element, and one workgroup per row re-read their 2 x 4 D bytes from L2 for each row - twice the bytes of the row
// itself. Here they are fetched once per R rows, all R rows are loaded before the first reduction (one memory latency per
// workgroup, as before), and the R sums of squares share one barrier pair.
template <int NV, int R>
__global__ __launch_bounds__(256) void norm_mod_rows_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, long mod_bstride, int rows_per_batch,
                                                            bf16_t* __restrict__ out, long ldo, int rows, float eps, int round_norm_bf16) {
    __shared__ float red[4][R];
    const int row0 = blockIdx.x * R;
    const long b = row0 / rows_per_batch;   // the launcher guarantees that the R rows do not straddle a batch element
    const float* sc = scale + b * mod_bstride;
    const float* sh = shift + b * mod_bstride;
    constexpr int D = NV * 1024;
    f32x4 v[R][NV], s4[NV], h4[NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
#pragma unroll
        for (int j = 0; j < NV; ++j) v[r][j] = *(const f32x4*)(x + (long)row * ldx + (threadIdx.x + j * 256) * 4);
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
        if (row0 + r >= rows) break;
        // same summation order as norm_mod_kernel's block_reduce_sum: wave sums added in wave order
        const float ms = (red[0][r] + red[1][r] + red[2][r] + red[3][r]) / (float)D;
        const float rstd = rsqrtf(ms + eps);
        bf16_t* orow = out + (long)(row0 + r) * ldo;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = v[r][j][e] * rstd;
                if (round_norm_bf16) y[e] = bf16_to_f32(f32_to_bf16(y[e]));
                y[e] = y[e] * (1.0f + s4[j][e]) + h4[j][e];
            }
            uint2 pk;
            pk.x = pack_bf16x2(y[0], y[1]);
            pk.y = pack_bf16x2(y[2], y[3]);
            *(uint2*)(orow + (threadIdx.x + j * 256) * 4) = pk;
        }
    }
}

// one workgroup per row; thread handles float4 chunks of the first half `a` of a head and the matching `b` chunk.
// blockIdx.y selects the job (q or k of the fused projection): one launch normalises and rotates both. The weight and
// cos/sin chunks are fetched with the row, before the reduction (same reason as norm_mod_kernel).
struct QkJob {
    const float* x;
    const float* w;
    bf16_t* out;
    long ldx, ldo;
    float out_scale;  // multiplies the normed (and rotated) row before its ONE rounding to bf16 (1 = plain)
};
// four consecutive elements of a q / k projection row as f32: the row is f32, or (XB) bf16 at the same element offsets
template <bool XB>
LTX_DEVFN f32x4 qk_load4(const float* x_row, int col) {
    if constexpr (XB) {
        const uint2 u = *(const uint2*)((const bf16_t*)x_row + col);
        return f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    } else {
        return *(const f32x4*)(x_row + col);
    }
}
template <int NP, bool XB = false>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(QkJob j0, QkJob j1, const float* __restrict__ cosT,
                                                          const float* __restrict__ sinT, int T, int D, float eps) {
    __shared__ float red[4];
    const QkJob job = blockIdx.y ? j1 : j0;
    const int row = blockIdx.x;
    const float* xr = XB ? (const float*)((const bf16_t*)job.x + (long)row * job.ldx) : job.x + (long)row * job.ldx;
    // pair chunk p (0 .. D/8-1): head = p / 16, i4 = p % 16 ; a at head*128 + i4*4, b at +64
    const int npair = D >> 3;
    const int t = row % T;
    f32x4 va[NP], vb[NP], wa[NP], wb[NP], c4[NP], s4[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = threadIdx.x + j * 256;
        va[j] = vb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p < npair) {
            const int col = (p >> 4) * 128 + (p & 15) * 4;
            va[j] = qk_load4<XB>(xr, col);
            vb[j] = qk_load4<XB>(xr, col + 64);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = threadIdx.x + j * 256;
        if (p < npair) {
            const int col = (p >> 4) * 128 + (p & 15) * 4;
            wa[j] = *(const f32x4*)(job.w + col);
            wb[j] = *(const f32x4*)(job.w + col + 64);
            if (cosT) {
                // cos/sin rows are [T][D/2]; head h uses columns h*64 .. h*64+63
                const int fc = (p >> 4) * 64 + (p & 15) * 4;
                c4[j] = *(const f32x4*)(cosT + (long)t * (D >> 1) + fc);
                s4[j] = *(const f32x4*)(sinT + (long)t * (D >> 1) + fc);
            }
        }
    }
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s2 += va[j][e] * va[j][e] + vb[j][e] * vb[j][e];
    const float rstd = rsqrtf(block_reduce_sum(s2, red) / (float)D + eps);
    const float osc = job.out_scale;
    bf16_t* orow = job.out + (long)row * job.ldo;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = threadIdx.x + j * 256;
        if (p < npair) {
            const int col = (p >> 4) * 128 + (p & 15) * 4;
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = va[j][e] * rstd * wa[j][e];
                b[e] = vb[j][e] * rstd * wb[j][e];
            }
            if (cosT) {
                // the reference rounds the normed q/k to its storage dtype before the f32 rotation only when that
                // dtype is bf16 (cross-modal case); in the DiT q,k are f32 here (LTXRoPE.swift:90-92).
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = a[e], b0 = b[e];
                    a[e] = a0 * c4[j][e] - b0 * s4[j][e];
                    b[e] = b0 * c4[j][e] + a0 * s4[j][e];
                }
            }
            uint2 pa, pb;
            pa.x = pack_bf16x2(a[0] * osc, a[1] * osc);
            pa.y = pack_bf16x2(a[2] * osc, a[3] * osc);
            pb.x = pack_bf16x2(b[0] * osc, b[1] * osc);
            pb.y = pack_bf16x2(b[2] * osc, b[3] * osc);
            *(uint2*)(orow + col) = pa;
            *(uint2*)(orow + col + 64) = pb;
        }
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const f32x4 v = *(const f32x4*)(x + i);
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)(out + i) = pk;
    }
    if (i < n && i + 3 >= n)
        for (long k = i; k < n; ++k) out[k] = f32_to_bf16(x[k]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = bf16_to_f32(x[i]);
}

__global__ void timestep_embedding_kernel(const float* __restrict__ ts, float mult, float* __restrict__ out, int n,
                                          int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int r = i / half, k = i - r * half;
    const float t = ts[r] * mult;
    const float freq = expf(-logf(10000.0f) * ((float)k / (float)half));
    const float arg = t * freq;
    out[(long)r * dim + k] = cosf(arg);
    out[(long)r * dim + half + k] = sinf(arg);
}

__global__ void make_mod_kernel(const float* __restrict__ tables, const float* __restrict__ ada,
                                float* __restrict__ mod, int B, int L, int J, int D) {
    const long n = (long)B * L * J * D;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int d = i % D;
        long r = i / D;
        const int j = r % J;
        r /= J;
        const int l = r % L;
        const int b = r / L;
        mod[i] = tables[((long)l * J + j) * D + d] + ada[((long)b * J + j) * D + d];
    }
}

__global__ void mask_to_bias_kernel(const int32_t* __restrict__ mask, float* __restrict__ bias, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bias[i] = (1.0f - (float)mask[i]) * -10000.0f;
}

// latent [B][C][T] f32 -> tokens [B][T][C] bf16 through a 32x32 LDS transpose tile
__global__ __launch_bounds__(256) void patchify_bf16_kernel(const float* __restrict__ latent,
                                                            bf16_t* __restrict__ tokens, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, t = t0 + tx;
        tile[k][tx] = (c < C && t < T) ? latent[((long)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        if (t < T && c < C) tokens[((long)b * T + t) * C + c] = f32_to_bf16(tile[tx][k]);
    }
}
__global__ __launch_bounds__(256) void unpatchify_f32_kernel(const float* __restrict__ tokens,
                                                             float* __restrict__ latent, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        tile[k][tx] = (t < T && c < C) ? tokens[((long)b * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, t = t0 + tx;
        if (c < C && t < T) latent[((long)b * C + c) * T + t] = tile[tx][k];
    }
}

// skip_hw > 0: image-to-video slice step - frame 0 of every channel ([C][F][H*W] layout, skip_hw = H*W, frames = F) keeps its
// value (LTXPipeline.swift:2344-2357)
__global__ void euler_step_kernel(float* __restrict__ latent, const float* __restrict__ vel, float sigma,
                                  float sigma_next, long n, int skip_hw, int frames) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (skip_hw > 0 && (i / skip_hw) % frames == 0) return;
    const float x = latent[i];
    const float den = x - sigma * vel[i];
    latent[i] = (sigma_next > 0.f) ? den + sigma_next * (x - den) / sigma : den;
}
__global__ void cfg_combine_kernel(const float* __restrict__ uncond, const float* __restrict__ cond, float sm1,
                                   float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cond[i] + sm1 * (cond[i] - uncond[i]);
}
// out = ca*(a-b) + cb_sel, where the two guidance forms are expressed as out = a*pa + b*pb
__global__ void lincomb_kernel(const float* __restrict__ a, const float* __restrict__ b, float ca, float cb,
                               float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ca * a[i] + cb * b[i];
}
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float ca, float cb,
                             float* __restrict__ out, long n) {
    // out = a*1 ... kept separate from lincomb to preserve the reference's operation order:
    // STG: v + s*(v - vp)  (ca = s, cb = 1 -> out = cb*a + ca*(a-b))
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cb * a[i] + ca * (a[i] - b[i]);
}
__global__ void ge_kernel(const float* __restrict__ a, const float* __restrict__ b, float g, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g * (a[i] - b[i]) + b[i];
}
__global__ void scale_inplace_kernel(float* __restrict__ x, float s, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}

// one workgroup per segment: population mean / variance (two-pass over a segment of n elements)
__global__ __launch_bounds__(1024) void mean_var_kernel(const float* __restrict__ x, long n, float* __restrict__ stats) {
    __shared__ float red[16];
    const float* xs = x + (long)blockIdx.x * n;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += xs[i];
    const float mean = block_reduce_sum(s, red) / (float)n;
    float v = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = xs[i] - mean;
        v += d * d;
    }
    const float var = block_reduce_sum(v, red) / (float)n;
    if (threadIdx.x == 0) {
        stats[blockIdx.x * 2 + 0] = mean;
        stats[blockIdx.x * 2 + 1] = var;
    }
}
__global__ void guidance_rescale_kernel(float* __restrict__ cfg, const float* __restrict__ st_cfg,
                                        const float* __restrict__ st_cond, float phi, long n_per_batch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n_per_batch) return;
    const float eps = 1e-8f;
    const float cfg_std = sqrtf(st_cfg[b * 2 + 1] + eps);
    const float cond_std = sqrtf(st_cond[b * 2 + 1] + eps);
    const float v = cfg[(long)b * n_per_batch + i];
    const float rescaled = v * (cond_std / cfg_std);
    cfg[(long)b * n_per_batch + i] = phi * rescaled + (1.0f - phi) * v;
}
__global__ void adain_kernel(float* __restrict__ x, const float* __restrict__ st_x, const float* __restrict__ st_ref,
                             float factor, long n_per_chan) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int bc = blockIdx.y;
    if (i >= n_per_chan) return;
    const float mu = st_x[bc * 2], sd = sqrtf(st_x[bc * 2 + 1]);
    const float mur = st_ref[bc * 2], sdr = sqrtf(st_ref[bc * 2 + 1]);
    const float v = x[(long)bc * n_per_chan + i];
    const float res = (v - mu) / (sd + 1e-8f) * sdr + mur;
    x[(long)bc * n_per_chan + i] = (factor >= 1.0f) ? res : factor * res + (1.0f - factor) * v;
}

// ---- VAE ----
// latent [C][P] f32 -> channels-last bf16 [P][C], with optional noise blend and per-channel denormalisation
__global__ __launch_bounds__(256) void vae_prepare_kernel(const float* __restrict__ latent, long chan_stride,
                                                          const float* __restrict__ noise, float noise_scale,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ std_, bf16_t* __restrict__ out,
                                                          int C, long P) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const long p = p0 + tx;
        float v = 0.f;
        if (c < C && p < P) {
            v = latent[(long)c * chan_stride + p];
            if (noise) v = noise[(long)c * chan_stride + p] * noise_scale + (1.0f - noise_scale) * v;
            v = v * std_[c] + mean[c];
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long p = p0 + k;
        const int c = c0 + tx;
        if (p < P && c < C) out[p * C + c] = f32_to_bf16(tile[tx][k]);
    }
}

// one wave per position when C <= 256*... generic: a group of G = C/4 lanes (<= 256) per position
template <int LANES_PER_POS, int NJ = 4>
__global__ __launch_bounds__(256) void pixelnorm_silu_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             bf16_t* __restrict__ out, long P, int C) {
    // each lane owns C/(4*LANES_PER_POS) float4 chunks; LANES_PER_POS lanes cooperate via shuffles (<= 64)
    constexpr int POS_PER_BLOCK = 256 / LANES_PER_POS;
    const int sub = threadIdx.x % LANES_PER_POS;
    const long p = (long)blockIdx.x * POS_PER_BLOCK + threadIdx.x / LANES_PER_POS;
    const int nchunk = C >> 2;
    const bool active = p < P;
    const float* xr = x + (active ? p : 0) * (long)C;
    f32x4 v[NJ];
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = sub + j * LANES_PER_POS;
        if (c < nchunk) {
            v[j] = *(const f32x4*)(xr + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s2 += v[j][e] * v[j][e];
        }
    }
#pragma unroll
    for (int off = LANES_PER_POS >> 1; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
    const float inv = 1.0f / sqrtf(s2 / (float)C + 1e-8f);
    if (!active) return;
    bf16_t* orow = out + p * (long)C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = sub + j * LANES_PER_POS;
        if (c < nchunk) {
            f32x4 s4 = f32x4{1.f, 1.f, 1.f, 1.f}, h4 = f32x4{0.f, 0.f, 0.f, 0.f};  // no modulation: the VAE encoder's blocks
            if (scale) {
                s4 = *(const f32x4*)(scale + c * 4);
                h4 = *(const f32x4*)(shift + c * 4);
            }
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = silu_f(v[j][e] * inv * s4[e] + h4[e]);
            uint2 pk;
            pk.x = pack_bf16x2(y[0], y[1]);
            pk.y = pack_bf16x2(y[2], y[3]);
            *(uint2*)(orow + c * 4) = pk;
        }
    }
}

// every modulation table of a decode in one launch (21 of them at a few KB each were 21 launches of ~5 us)
__global__ void vae_make_mods_batch_kernel(const VaeModsBatch b) {
    const VaeModsJob& j = b.job[blockIdx.y];
    const long n = (long)j.rows * j.C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = i / j.C;
        j.out[i] = j.table[i] + (j.te ? j.te[i] : 0.f) + ((r & 1) ? 1.0f : 0.f);
    }
}
__global__ void blend_frames_kernel(float* __restrict__ r, const float* __restrict__ nx, int n_frames, long frame_elems) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (i >= frame_elems) return;
    const float w = (float)f / (float)n_frames;
    const long o = (long)f * frame_elems + i;
    r[o] = r[o] * (1.0f - w) + nx[o] * w;
}
__global__ void clip01_kernel(float* __restrict__ x, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = fminf(fmaxf((x[i] + 1.0f) * 0.5f, 0.f), 1.f);
}

// one workgroup per group: two passes (mean, then variance) over P positions x (C/G) channels
__global__ __launch_bounds__(1024) void groupnorm_stats_kernel(const float* __restrict__ x, long P, int C, int G, float eps,
                                                               float* __restrict__ stats) {
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int cpg = C / G;
    const long n = P * cpg;
    const float* base = x + (long)g * cpg;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += base[(i / cpg) * C + (i % cpg)];
    const float mean = block_reduce_sum(s, red) / (float)n;
    float v = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = base[(i / cpg) * C + (i % cpg)] - mean;
        v += d * d;
    }
    const float var = block_reduce_sum(v, red) / (float)n;
    if (threadIdx.x == 0) {
        stats[g * 2 + 0] = mean;
        stats[g * 2 + 1] = 1.0f / sqrtf(var + eps);
    }
}
__global__ void groupnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                       const float* __restrict__ w, const float* __restrict__ b,
                                       const float* __restrict__ resid, int act_silu, float* __restrict__ out_f32,
                                       bf16_t* __restrict__ out_bf16, long n, int C, int G) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int cpg = C / G;
    for (; i < n; i += stride) {
        const int c = i % C;
        const int g = c / cpg;
        float y = (x[i] - stats[g * 2]) * stats[g * 2 + 1] * w[c] + b[c];
        if (resid) y += resid[i];
        if (act_silu) y = silu_f(y);
        if (out_f32) out_f32[i] = y;
        if (out_bf16) out_bf16[i] = f32_to_bf16(y);
    }
}
__global__ __launch_bounds__(256) void upscaler_finish_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ std_, float* __restrict__ out,
                                                              long P, int C) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const long p = p0 + k;
        const int c = c0 + tx;
        tile[k][tx] = (p < P && c < C) ? (x[p * C + c] - mean[c]) / std_[c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const long p = p0 + tx;
        if (c < C && p < P) out[(long)c * P + p] = tile[tx][k];
    }
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace

void launch_norm_mod(const float* x, long ldx, const float* scale, const float* shift, long mod_bstride,
                     int rows_per_batch, bf16_t* out, long ldo, int rows, int D, int norm_kind, float eps,
                     int round_norm_bf16, hipStream_t stream, const int32_t* row_map) {
    LTX_REQUIRE(D % 4 == 0 && D <= MAXV * 1024 && ldx % 4 == 0 && ldo % 4 == 0, "norm_mod: D=%d ldx=%ld ldo=%ld", D, ldx, ldo);
    LTX_REQUIRE((scale == nullptr) == (shift == nullptr), "norm_mod: scale/shift must both be set or both null");
    const int rpb = rows_per_batch < 1 ? 1 : rows_per_batch;
#ifdef LTX_EXPERIMENTS  // timing ablation (experiments library only): LTX_ABL_ROWS bit 0 drops this pass after the first 400 calls - the
    // buffers keep realistic stale values, so what the step then gains is the upper bound of any fusion of the pass (profiles/r04_row_ablation.txt)
    { const int abl = ltx_opt(OPT_ABL_ROWS); static long calls = 0; if ((abl & 1) && ++calls > 400) return; }
#endif
    ProfScope prof(PROF_ELEM, (double)rows * D * (4 + 2), stream);  // algorithmic bytes: f32 row in, bf16 row out
    // rows per workgroup: 4 share one fetch of the modulation vectors; at 1536 rows that is 384 workgroups = 1.5 per CU, and two rows per
    // workgroup (768 = 3 per CU) are 10 % faster (9.15 vs 10.2 us alone, 36.39 vs 36.53 ms per forward); no difference from 6144 rows up
    const int r_raw = ltx_opt(OPT_NORM_ROWS);  // A/B option "norm_rows": 2 or 4, anything else is ignored
    const int r_env = (r_raw == 2 || r_raw == 4) ? r_raw : 0;
    const int R_rows = r_env ? r_env : (rows <= 3072 ? 2 : 4);
    if (norm_kind == LTX_NORM_RMS && scale && !row_map && D == 4096 && rows >= 512 && rpb % R_rows == 0) {
#define LTX_ROWS_LAUNCH(R)                                                                                                                      \
    hipLaunchKernelGGL((norm_mod_rows_kernel<4, R>), dim3((rows + R - 1) / R), dim3(256), 0, stream, x, ldx, scale, shift, mod_bstride, rpb, out, ldo, \
                       rows, eps, round_norm_bf16)
        if (R_rows == 2) LTX_ROWS_LAUNCH(2);
        else LTX_ROWS_LAUNCH(4);
#undef LTX_ROWS_LAUNCH
        HIP_CHECK(hipGetLastError());
        return;
    }
#define LTX_NORM_LAUNCH(NV)                                                                                              \
    hipLaunchKernelGGL(norm_mod_kernel<NV>, dim3(rows), dim3(256), 0, stream, x, ldx, scale, shift, mod_bstride, rpb, out, \
                       ldo, D, norm_kind, eps, round_norm_bf16, row_map)
    if (D <= 1024) LTX_NORM_LAUNCH(1);
    else if (D <= 2048) LTX_NORM_LAUNCH(2);
    else if (D <= 4096) LTX_NORM_LAUNCH(4);
    else LTX_NORM_LAUNCH(8);
#undef LTX_NORM_LAUNCH
    HIP_CHECK(hipGetLastError());
}

// R rows per workgroup of qknorm_rope_kernel (D = 4096): the norm weights are fetched once per R rows instead of once per row
// (they are as many bytes as a row), every row is loaded before the first reduction, the R sums share one barrier pair.
template <int R, bool XB = false>
__global__ __launch_bounds__(256) void qknorm_rope_rows_kernel(QkJob j0, QkJob j1, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                               int T, int rows, float eps) {
    constexpr int NP = 2, D = 4096;
    __shared__ float red[4][R];
    const QkJob job = blockIdx.y ? j1 : j0;
    const int row0 = blockIdx.x * R;
    f32x4 va[R][NP], vb[R][NP], wa[NP], wb[NP];
    int col[NP], fc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = threadIdx.x + j * 256;
        col[j] = (p >> 4) * 128 + (p & 15) * 4;
        fc[j] = (p >> 4) * 64 + (p & 15) * 4;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
        const float* xr = XB ? (const float*)((const bf16_t*)job.x + (long)row * job.ldx) : job.x + (long)row * job.ldx;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            va[r][j] = qk_load4<XB>(xr, col[j]);
            vb[r][j] = qk_load4<XB>(xr, col[j] + 64);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        wa[j] = *(const f32x4*)(job.w + col[j]);
        wb[j] = *(const f32x4*)(job.w + col[j] + 64);
    }
    // the table rows too, before the reduction: a dependent load behind the barrier costs a second memory latency per workgroup
    f32x4 c4[R][NP], s4[R][NP];
    if (cosT) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
            const long tb = (long)(row % T) * (D >> 1);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                c4[r][j] = *(const f32x4*)(cosT + tb + fc[j]);
                s4[r][j] = *(const f32x4*)(sinT + tb + fc[j]);
            }
        }
    }
    float ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += va[r][j][e] * va[r][j][e] + vb[r][j][e] * vb[r][j][e];
        ss[r] = wave_reduce_sum(a);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[w][r] = ss[r];
    }
    __syncthreads();
    const float osc = job.out_scale;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        const float rstd = rsqrtf((red[0][r] + red[1][r] + red[2][r] + red[3][r]) / (float)D + eps);
        bf16_t* orow = job.out + (long)row * job.ldo;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = va[r][j][e] * rstd * wa[j][e];
                b[e] = vb[r][j][e] * rstd * wb[j][e];
            }
            if (cosT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = a[e], b0 = b[e];
                    a[e] = a0 * c4[r][j][e] - b0 * s4[r][j][e];
                    b[e] = b0 * c4[r][j][e] + a0 * s4[r][j][e];
                }
            }
            uint2 pa, pb;
            pa.x = pack_bf16x2(a[0] * osc, a[1] * osc);
            pa.y = pack_bf16x2(a[2] * osc, a[3] * osc);
            pb.x = pack_bf16x2(b[0] * osc, b[1] * osc);
            pb.y = pack_bf16x2(b[2] * osc, b[3] * osc);
            *(uint2*)(orow + col[j]) = pa;
            *(uint2*)(orow + col[j] + 64) = pb;
        }
    }
}

// The self-attention pair (q and k of the same tokens) in one workgroup: R rows of BOTH jobs share one fetch of the rotary tables
// (a table row is as many bytes as the f32 row it rotates; with one workgroup per job the second fetch came from L2 / Infinity Cache,
// 768 workgroups later) and one barrier pair.
template <int R, bool XB = false>
__global__ __launch_bounds__(256) void qknorm_rope_pair_kernel(QkJob j0, QkJob j1, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                               int T, int rows, float eps) {
    constexpr int NP = 2, D = 4096, NJ = 2;
    __shared__ float red[4][NJ * R];
    const int row0 = blockIdx.x * R;
    f32x4 va[NJ][R][NP], vb[NJ][R][NP], c4[R][NP], s4[R][NP];
    int col[NP], fc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = threadIdx.x + j * 256;
        col[j] = (p >> 4) * 128 + (p & 15) * 4;
        fc[j] = (p >> 4) * 64 + (p & 15) * 4;
    }
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const QkJob& job = q ? j1 : j0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
            const float* xr = XB ? (const float*)((const bf16_t*)job.x + (long)row * job.ldx) : job.x + (long)row * job.ldx;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                va[q][r][j] = qk_load4<XB>(xr, col[j]);
                vb[q][r][j] = qk_load4<XB>(xr, col[j] + 64);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
        const long tb = (long)(row % T) * (D >> 1);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            c4[r][j] = *(const f32x4*)(cosT + tb + fc[j]);
            s4[r][j] = *(const f32x4*)(sinT + tb + fc[j]);
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NJ; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) a += va[q][r][j][e] * va[q][r][j][e] + vb[q][r][j][e] * vb[q][r][j][e];
            a = wave_reduce_sum(a);
            if (lane == 0) red[w][q * R + r] = a;
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const QkJob& job = q ? j1 : j0;
        const float osc = job.out_scale;
        f32x4 wa[NP], wb[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            wa[j] = *(const f32x4*)(job.w + col[j]);
            wb[j] = *(const f32x4*)(job.w + col[j] + 64);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (row >= rows) break;
            const int k = q * R + r;
            const float rstd = rsqrtf((red[0][k] + red[1][k] + red[2][k] + red[3][k]) / (float)D + eps);
            bf16_t* orow = job.out + (long)row * job.ldo;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                f32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = va[q][r][j][e] * rstd * wa[j][e], b0 = vb[q][r][j][e] * rstd * wb[j][e];
                    a[e] = a0 * c4[r][j][e] - b0 * s4[r][j][e];
                    b[e] = b0 * c4[r][j][e] + a0 * s4[r][j][e];
                }
                uint2 pa, pb;
                pa.x = pack_bf16x2(a[0] * osc, a[1] * osc);
                pa.y = pack_bf16x2(a[2] * osc, a[3] * osc);
                pb.x = pack_bf16x2(b[0] * osc, b[1] * osc);
                pb.y = pack_bf16x2(b[2] * osc, b[3] * osc);
                *(uint2*)(orow + col[j]) = pa;
                *(uint2*)(orow + col[j] + 64) = pb;
            }
        }
    }
}

void launch_qknorm_rope2(const float* x0, const float* w0, bf16_t* out0, const float* x1, const float* w1, bf16_t* out1,
                         long ldx, long ldo, const float* cosT, const float* sinT, int T, int rows, int D, float eps,
                         hipStream_t stream, float out_scale0, bool x_bf16) {
    LTX_REQUIRE(D % 128 == 0 && D <= MAXV * 1024 && ldx % 4 == 0 && ldo % 4 == 0, "qknorm_rope: D=%d", D);

    const QkJob j0{x0, w0, out0, ldx, ldo, out_scale0}, j1{x1, w1, out1, ldx, ldo, 1.0f};
#ifdef LTX_EXPERIMENTS  // timing ablation: LTX_ABL_ROWS bit 1 drops the q|k pass, bit 2 the cross-attention q pass, after two forwards' worth of calls
    { const int abl = ltx_opt(OPT_ABL_ROWS); static long c2 = 0, c1 = 0;
      if (x1 && (abl & 2) && ++c2 > 200) return; if (!x1 && (abl & 4) && ++c1 > 300) return; }
#endif
    // algorithmic bytes: the f32 (or bf16) rows in, the bf16 rows out, cos/sin rows once per job
    ProfScope prof(PROF_ELEM, (double)rows * D * (x1 ? 2 : 1) * ((x_bf16 ? 2 : 4) + 2 + (cosT ? 4 : 0)), stream);
    const dim3 grid(rows, x1 ? 2 : 1);
    const int t = T < 1 ? 1 : T;
    const bool no_pair = ltx_opt(OPT_QKNORM_NO_PAIR) != 0;  // A/B option "qknorm_no_pair"
    if (D == 4096 && rows >= 512 && x1 && cosT && !no_pair) {  // q and k of the same tokens: one fetch of the tables (2.64 -> 2.54 ms of row passes per step)
        constexpr int R = 2;  // 1 and 3 rows per workgroup: within the run-to-run noise of the forward (round 3)
        if (x_bf16) hipLaunchKernelGGL((qknorm_rope_pair_kernel<R, true>), dim3((rows + R - 1) / R), dim3(256), 0, stream, j0, j1, cosT, sinT, t, rows, eps);
        else hipLaunchKernelGGL(qknorm_rope_pair_kernel<R>, dim3((rows + R - 1) / R), dim3(256), 0, stream, j0, j1, cosT, sinT, t, rows, eps);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (D == 4096 && rows >= 512) {   // the DiT at full width: R rows per workgroup share one fetch of the norm weights
        constexpr int R = 2;
        if (x_bf16) hipLaunchKernelGGL((qknorm_rope_rows_kernel<R, true>), dim3((rows + R - 1) / R, x1 ? 2 : 1), dim3(256), 0, stream, j0, j1, cosT, sinT, t, rows, eps);
        else hipLaunchKernelGGL(qknorm_rope_rows_kernel<R>, dim3((rows + R - 1) / R, x1 ? 2 : 1), dim3(256), 0, stream, j0, j1, cosT, sinT, t, rows, eps);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (x_bf16) {
        if (D <= 2048) hipLaunchKernelGGL((qknorm_rope_kernel<1, true>), grid, dim3(256), 0, stream, j0, j1, cosT, sinT, t, D, eps);
        else if (D <= 4096) hipLaunchKernelGGL((qknorm_rope_kernel<2, true>), grid, dim3(256), 0, stream, j0, j1, cosT, sinT, t, D, eps);
        else hipLaunchKernelGGL((qknorm_rope_kernel<4, true>), grid, dim3(256), 0, stream, j0, j1, cosT, sinT, t, D, eps);
    } else if (D <= 2048) hipLaunchKernelGGL(qknorm_rope_kernel<1>, grid, dim3(256), 0, stream, j0, j1, cosT, sinT, t, D, eps);
    else if (D <= 4096) hipLaunchKernelGGL(qknorm_rope_kernel<2>, grid, dim3(256), 0, stream, j0, j1, cosT, sinT, t, D, eps);
    else hipLaunchKernelGGL(qknorm_rope_kernel<4>, grid, dim3(256), 0, stream, j0, j1, cosT, sinT, t, D, eps);
    HIP_CHECK(hipGetLastError());
}

void launch_qknorm_rope(const float* x, long ldx, const float* w, const float* cosT, const float* sinT, int T,
                        bf16_t* out, long ldo, int rows, int D, float eps, hipStream_t stream, float out_scale, bool x_bf16) {
    launch_qknorm_rope2(x, w, out, nullptr, nullptr, nullptr, ldx, ldo, cosT, sinT, T, rows, D, eps, stream, out_scale, x_bf16);
}

void launch_cast_f32_bf16(const float* x, bf16_t* out, long n, hipStream_t stream) {
    const int grid = cdiv(n, 4 * 256) < 2048 ? (cdiv(n, 4 * 256) < 1 ? 1 : cdiv(n, 4 * 256)) : 2048;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, stream, x, out, n);
    HIP_CHECK(hipGetLastError());
}
void launch_cast_bf16_f32(const bf16_t* x, float* out, long n, hipStream_t stream) {
    const int grid = cdiv(n, 256) < 4096 ? (cdiv(n, 256) < 1 ? 1 : cdiv(n, 256)) : 4096;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, stream, x, out, n);
    HIP_CHECK(hipGetLastError());
}
void launch_timestep_embedding(const float* ts, float mult, float* out, int n, int dim, hipStream_t stream) {
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv((long)n * (dim / 2), 256)), dim3(256), 0, stream, ts, mult, out, n, dim);
    HIP_CHECK(hipGetLastError());
}
void launch_make_mod(const float* tables, const float* ada, float* mod, int B, int L, int J, int D, hipStream_t stream) {
    const long n = (long)B * L * J * D;
    const int grid = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
    hipLaunchKernelGGL(make_mod_kernel, dim3(grid), dim3(256), 0, stream, tables, ada, mod, B, L, J, D);
    HIP_CHECK(hipGetLastError());
}
void launch_mask_to_bias(const int32_t* mask, float* bias, long n, hipStream_t stream) {
    hipLaunchKernelGGL(mask_to_bias_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, mask, bias, n);
    HIP_CHECK(hipGetLastError());
}
// [N][D][Tn] -> [D][ld]: one 16-byte piece (8 keys) per thread, rows stay contiguous on both sides
__global__ __launch_bounds__(256) void sp_vt_interleave_kernel(const bf16_t* __restrict__ g, bf16_t* __restrict__ vt, int N, int D,
                                                               int Tn, long ld) {
    const int per_row = Tn >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * D * per_row;
    if (i >= total) return;
    const int c = (int)(i % per_row);
    const long rd = i / per_row;
    const int d = (int)(rd % D);
    const int r = (int)(rd / D);
    const uint4 v = *(const uint4*)(g + ((long)r * D + d) * Tn + c * 8);
    *(uint4*)(vt + (long)d * ld + (long)r * Tn + c * 8) = v;
}

void launch_sp_vt_interleave(const bf16_t* gathered, bf16_t* vt, int N, int D, int Tn, long ld, hipStream_t stream) {
    LTX_REQUIRE(Tn % 8 == 0 && ld % 8 == 0, "sp_vt_interleave: Tn=%d ld=%ld must be multiples of 8", Tn, ld);
    const long total = (long)N * D * (Tn >> 3);
    hipLaunchKernelGGL(sp_vt_interleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, gathered, vt, N, D, Tn, ld);
    HIP_CHECK(hipGetLastError());
}

void launch_patchify_bf16(const float* latent, bf16_t* tokens, int B, int C, int T, hipStream_t stream) {
    hipLaunchKernelGGL(patchify_bf16_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, stream, latent, tokens, C, T);
    HIP_CHECK(hipGetLastError());
}
void launch_unpatchify_f32(const float* tokens, float* latent, int B, int C, int T, hipStream_t stream) {
    hipLaunchKernelGGL(unpatchify_f32_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, stream, tokens, latent, C, T);
    HIP_CHECK(hipGetLastError());
}
void launch_euler_step(float* latent, const float* velocity, float sigma, float sigma_next, long n, hipStream_t stream, int skip_hw,
                       int frames) {
    hipLaunchKernelGGL(euler_step_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, latent, velocity, sigma, sigma_next, n, skip_hw,
                       frames < 1 ? 1 : frames);
    HIP_CHECK(hipGetLastError());
}

namespace {
// latent[c][0][hw] = cond[c][hw] (+ s * noise[c][hw])   (LTXPipeline.swift:2092-2094, :2225-2229)
// the reference evaluates cond + (scale * noise) * sigma^2 as three separate f32 ops: no fused multiply-add here
__global__ void set_frame0_kernel(float* __restrict__ latent, const float* __restrict__ cond, const float* __restrict__ noise,
                                  float scale, float sigma2, int C, int F, int HW) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * HW) return;
    const long c = i / HW, hw = i - c * HW;
    float v = cond[i];
    if (noise) {
        const float a = scale * noise[i];
        const float b = a * sigma2;
        v = v + b;
    }
    latent[(c * F) * HW + hw] = v;
}
// row_map[b*T + t] = b*G + (t < first ? 1 : 0): tokens of frame 0 use timestep group 1 (sigma 0), the rest group 0
__global__ void i2v_rowmap_kernel(int32_t* __restrict__ row_map, int B, int T, int first, int G) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T) return;
    const int b = (int)(i / T), t = (int)(i - (long)b * T);
    row_map[i] = b * G + (t < first ? 1 : 0);
}
}  // namespace
void launch_set_frame0(float* latent, const float* cond, const float* noise, float scale, float sigma2, int C, int F, int HW,
                       hipStream_t stream) {
    hipLaunchKernelGGL(set_frame0_kernel, dim3(cdiv((long)C * HW, 256)), dim3(256), 0, stream, latent, cond, noise, scale, sigma2, C, F, HW);
    HIP_CHECK(hipGetLastError());
}
void launch_i2v_rowmap(int32_t* row_map, int B, int T, int first, int G, hipStream_t stream) {
    hipLaunchKernelGGL(i2v_rowmap_kernel, dim3(cdiv((long)B * T, 256)), dim3(256), 0, stream, row_map, B, T, first, G);
    HIP_CHECK(hipGetLastError());
}
void launch_cfg_combine(const float* uncond, const float* cond, float scale, float* out, long n, hipStream_t stream) {
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, uncond, cond, scale - 1.0f, out, n);
    HIP_CHECK(hipGetLastError());
}
void launch_axpby(const float* a, const float* b, float ca, float cb, float* out, long n, hipStream_t stream) {
    hipLaunchKernelGGL(axpby_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, a, b, ca, cb, out, n);
    HIP_CHECK(hipGetLastError());
}
void launch_ge(const float* a, const float* b, float g, float* out, long n, hipStream_t stream) {
    hipLaunchKernelGGL(ge_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, a, b, g, out, n);
    HIP_CHECK(hipGetLastError());
}
void launch_lincomb(const float* a, const float* b, float ca, float cb, float* out, long n, hipStream_t stream) {
    hipLaunchKernelGGL(lincomb_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, a, b, ca, cb, out, n);
    HIP_CHECK(hipGetLastError());
}
void launch_scale_inplace(float* x, float s, long n, hipStream_t stream) {
    hipLaunchKernelGGL(scale_inplace_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, x, s, n);
    HIP_CHECK(hipGetLastError());
}
void launch_mean_var(const float* x, long n_per_batch, int B, float* stats, hipStream_t stream) {
    hipLaunchKernelGGL(mean_var_kernel, dim3(B), dim3(1024), 0, stream, x, n_per_batch, stats);
    HIP_CHECK(hipGetLastError());
}
void launch_guidance_rescale(float* cfg, const float* stats_cfg, const float* stats_cond, float phi, long n_per_batch,
                             int B, hipStream_t stream) {
    hipLaunchKernelGGL(guidance_rescale_kernel, dim3(cdiv(n_per_batch, 256), B), dim3(256), 0, stream, cfg, stats_cfg, stats_cond, phi, n_per_batch);
    HIP_CHECK(hipGetLastError());
}
void launch_adain(float* x, const float* stats_x, const float* stats_ref, float factor, long n_per_chan, int BC,
                  hipStream_t stream) {
    hipLaunchKernelGGL(adain_kernel, dim3(cdiv(n_per_chan, 256), BC), dim3(256), 0, stream, x, stats_x, stats_ref, factor, n_per_chan);
    HIP_CHECK(hipGetLastError());
}
void launch_vae_prepare(const float* latent, long chan_stride, const float* noise, float noise_scale, const float* mean,
                        const float* std_, bf16_t* out, int C, long P, hipStream_t stream) {
    hipLaunchKernelGGL(vae_prepare_kernel, dim3(cdiv(P, 32), cdiv(C, 32)), dim3(256), 0, stream, latent, chan_stride, noise, noise_scale, mean, std_, out, C, P);
    HIP_CHECK(hipGetLastError());
}
void launch_vae_make_mods_batch(const VaeModsBatch& b, hipStream_t stream) {
    LTX_REQUIRE(b.n >= 1 && b.n <= VaeModsBatch::MAX_JOBS, "vae_make_mods_batch: %d jobs", b.n);
    hipLaunchKernelGGL(vae_make_mods_batch_kernel, dim3(4, b.n), dim3(256), 0, stream, b);
    HIP_CHECK(hipGetLastError());
}
void launch_blend_frames(float* r, const float* nx, int n_frames, long frame_elems, hipStream_t stream) {
    hipLaunchKernelGGL(blend_frames_kernel, dim3(cdiv(frame_elems, 256), n_frames), dim3(256), 0, stream, r, nx, n_frames, frame_elems);
    HIP_CHECK(hipGetLastError());
}
void launch_clip01(float* x, long n, hipStream_t stream) {
    const int grid = cdiv(n, 256) < 65535 * 16 ? cdiv(n, 256) : 65535 * 16;
    hipLaunchKernelGGL(clip01_kernel, dim3(grid), dim3(256), 0, stream, x, n);
    HIP_CHECK(hipGetLastError());
}
void launch_pixelnorm_silu(const float* x, const float* scale, const float* shift, bf16_t* out, long P, int C,
                           hipStream_t stream) {
    LTX_REQUIRE(C % 4 == 0 && C <= 2048, "pixelnorm: C=%d", C);
    LTX_REQUIRE((scale == nullptr) == (shift == nullptr), "pixelnorm: scale/shift must both be set or both null");
    if (C > 1024) {  // the encoder's 2048-channel mid block: 8 chunks per lane
        hipLaunchKernelGGL((pixelnorm_silu_kernel<64, 8>), dim3(cdiv(P, 4)), dim3(256), 0, stream, x, scale, shift, out, P, C);
        HIP_CHECK(hipGetLastError());
        return;
    }
    // lanes per position: enough that each lane holds <= 4 float4 chunks, power of two, <= 64
    const int nchunk = C / 4;
    int lpp = 1;
    while (lpp * 4 < nchunk) lpp <<= 1;
    if (lpp < 8) lpp = 8;
    LTX_REQUIRE(lpp <= 64, "pixelnorm: C=%d too wide", C);
#define PN_LAUNCH(L)                                                                                              \
    hipLaunchKernelGGL((pixelnorm_silu_kernel<L>), dim3(cdiv(P, 256 / L)), dim3(256), 0, stream, x, scale, shift, out, P, C)
    if (lpp == 8) PN_LAUNCH(8);
    else if (lpp == 16) PN_LAUNCH(16);
    else if (lpp == 32) PN_LAUNCH(32);
    else PN_LAUNCH(64);
#undef PN_LAUNCH
    HIP_CHECK(hipGetLastError());
}

void launch_groupnorm_stats(const float* x, long P, int C, int G, float eps, float* stats, hipStream_t stream) {
    LTX_REQUIRE(C % G == 0, "groupnorm: C=%d not divisible by G=%d", C, G);
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(G), dim3(1024), 0, stream, x, P, C, G, eps, stats);
    HIP_CHECK(hipGetLastError());
}
void launch_groupnorm_apply(const float* x, const float* stats, const float* w, const float* b, const float* resid,
                            int act_silu, float* out_f32, bf16_t* out_bf16, long P, int C, int G, hipStream_t stream) {
    const long n = P * C;
    const int grid = cdiv(n, 256) < 4096 ? cdiv(n, 256) : 4096;
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(grid), dim3(256), 0, stream, x, stats, w, b, resid, act_silu, out_f32, out_bf16, n, C, G);
    HIP_CHECK(hipGetLastError());
}
void launch_upscaler_finish(const float* x, const float* mean, const float* std_, float* out, long P, int C, hipStream_t stream) {
    hipLaunchKernelGGL(upscaler_finish_kernel, dim3(cdiv(P, 32), cdiv(C, 32)), dim3(256), 0, stream, x, mean, std_, out, P, C);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// text-embedding connector kernels
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int FE_CHUNK = 16384;  // elements of one (state, batch) plane per workgroup

LTX_DEVFN bool fe_token_valid(const int32_t* mrow, int t, int T, int n_valid, int padding_right) {
    // the reference rebuilds the mask from the sequence length and the padding side (LTXTextEncoder.swift:75-84)
    return padding_right ? (t < n_valid) : (t >= T - n_valid);
}

__global__ __launch_bounds__(256) void fe_stats_partial_kernel(const bf16_t* __restrict__ hidden, const int32_t* __restrict__ mask,
                                                               int B, int T, int D, int padding_right,
                                                               float* __restrict__ partials, int nchunk) {
    __shared__ float red[3][4];
    __shared__ int s_nvalid;
    const int plane = blockIdx.y;  // l*B + b
    const int b = plane % B;
    if (threadIdx.x == 0) s_nvalid = 0;
    __syncthreads();
    int cnt = 0;
    for (int t = threadIdx.x; t < T; t += 256) cnt += mask[(long)b * T + t] != 0;
    atomicAdd(&s_nvalid, cnt);
    __syncthreads();
    const int n_valid = s_nvalid;
    const bf16_t* x = hidden + (long)plane * T * D;
    const long total = (long)T * D;
    const long beg = (long)blockIdx.x * FE_CHUNK;
    const long end = beg + FE_CHUNK < total ? beg + FE_CHUNK : total;
    float s = 0.f, mn = INFINITY, mx = -INFINITY;
    for (long i = beg + threadIdx.x * 8; i < end; i += 256 * 8) {
        const int t = (int)(i / D);  // D % 8 == 0: the 8 elements share a token
        if (!fe_token_valid(nullptr, t, T, n_valid, padding_right)) continue;
        const uint4 raw = *(const uint4*)(x + i);
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = bf16_to_f32((bf16_t)(w[k] & 0xffff)), c = bf16_to_f32((bf16_t)(w[k] >> 16));
            s += a + c;
            mn = fminf(mn, fminf(a, c));
            mx = fmaxf(mx, fmaxf(a, c));
        }
    }
    s = wave_reduce_sum(s);
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    const int lane = threadIdx.x & 63, w_ = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][w_] = s;
        red[1][w_] = mn;
        red[2][w_] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* p = partials + ((long)plane * nchunk + blockIdx.x) * 3;
        p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        p[1] = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
        p[2] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    }
}

__global__ void fe_stats_final_kernel(const float* __restrict__ partials, const int32_t* __restrict__ mask, int B, int T, int D,
                                      float eps, int nchunk, float* __restrict__ stats, int planes) {
    const int plane = blockIdx.x * blockDim.x + threadIdx.x;
    if (plane >= planes) return;
    const int b = plane % B;
    int n_valid = 0;
    for (int t = 0; t < T; ++t) n_valid += mask[(long)b * T + t] != 0;
    double s = 0.0;  // fixed order over the chunks
    float mn = INFINITY, mx = -INFINITY;
    for (int c = 0; c < nchunk; ++c) {
        const float* p = partials + ((long)plane * nchunk + c) * 3;
        s += (double)p[0];
        mn = fminf(mn, p[1]);
        mx = fmaxf(mx, p[2]);
    }
    const float denom = (float)n_valid * (float)D + eps;
    stats[plane * 2 + 0] = (float)s / denom;
    stats[plane * 2 + 1] = (mx - mn) + eps;
}

__global__ __launch_bounds__(256) void fe_norm_concat_kernel(const bf16_t* __restrict__ hidden, const int32_t* __restrict__ mask,
                                                             const float* __restrict__ stats, int states, int B, int T, int D,
                                                             int padding_right, bf16_t* __restrict__ out) {
    __shared__ int s_nvalid;
    const int bt = blockIdx.x;  // b*T + t
    const int b = bt / T, t = bt - b * T;
    if (threadIdx.x == 0) s_nvalid = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < T; i += 256) cnt += mask[(long)b * T + i] != 0;
    atomicAdd(&s_nvalid, cnt);
    __syncthreads();
    const bool valid = fe_token_valid(nullptr, t, T, s_nvalid, padding_right);
    bf16_t* orow = out + (long)bt * D * states;
    const long plane_stride = (long)B * T * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        const bf16_t* x = hidden + ((long)b * T + t) * D + d;
        bf16_t* o = orow + (long)d * states;
        for (int l = 0; l < states; ++l) {
            float v = 0.f;
            if (valid) {
                const float xv = bf16_to_f32(x[(long)l * plane_stride]);
                v = 8.0f * (xv - stats[(l * B + b) * 2 + 0]) / stats[(l * B + b) * 2 + 1];
            }
            o[l] = f32_to_bf16(v);
        }
    }
}

__global__ void register_plan_kernel(const int32_t* __restrict__ mask, int T, int32_t* __restrict__ src) {
    // one thread per batch row: T <= a few thousand, runs once per prompt
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int32_t* m = mask + (long)b * T;
    int32_t* s = src + (long)b * T;
    int n = 0;
    for (int t = 0; t < T; ++t)
        if (m[t] != 0) s[n++] = t;  // valid tokens first, in order
    for (int t = 0; t < T; ++t)
        if (m[t] == 0) s[n++] = t;  // padded tokens after them (never kept unless the padding is on the right)
    for (int p = 0; p < T; ++p)
        if (m[T - 1 - p] == 0) s[p] = -1;  // reverse(valid)[p] == 0 -> learnable register
}

__global__ __launch_bounds__(256) void register_gather_kernel(const bf16_t* __restrict__ enc, const float* __restrict__ registers,
                                                              const int32_t* __restrict__ src, int T, int D, int R,
                                                              float* __restrict__ x) {
    const int bp = blockIdx.x;
    const int b = bp / T, p = bp - b * T;
    const int sidx = src[bp];
    float* o = x + (long)bp * D;
    if (sidx >= 0) {
        const bf16_t* e = enc + ((long)b * T + sidx) * D;
        for (int d = threadIdx.x; d < D; d += 256) o[d] = bf16_to_f32(e[d]);
    } else {
        const float* rg = registers + (long)(p % R) * D;
        for (int d = threadIdx.x; d < D; d += 256) o[d] = rg[d];
    }
}

__global__ void fill_const_i32_kernel(int32_t* p, long n, int32_t v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace

void launch_fe_stats(const bf16_t* hidden, const int32_t* mask, int states, int B, int T, int D, int padding_right, float eps,
                     float* partials, float* stats, hipStream_t stream) {
    LTX_REQUIRE(D % 8 == 0, "connector: hidden dim %d must be a multiple of 8", D);
    const int nchunk = (int)(((long)T * D + FE_CHUNK - 1) / FE_CHUNK);
    hipLaunchKernelGGL(fe_stats_partial_kernel, dim3(nchunk, states * B), dim3(256), 0, stream, hidden, mask, B, T, D, padding_right,
                       partials, nchunk);
    HIP_CHECK(hipGetLastError());
    const int planes = states * B;
    hipLaunchKernelGGL(fe_stats_final_kernel, dim3((planes + 63) / 64), dim3(64), 0, stream, partials, mask, B, T, D, eps, nchunk, stats,
                       planes);
    HIP_CHECK(hipGetLastError());
}
long fe_stats_partials_floats(int states, int B, int T, int D) {
    return (long)states * B * (((long)T * D + FE_CHUNK - 1) / FE_CHUNK) * 3;
}
void launch_fe_norm_concat(const bf16_t* hidden, const int32_t* mask, const float* stats, int states, int B, int T, int D,
                           int padding_right, float eps, bf16_t* out, hipStream_t stream) {
    (void)eps;
    hipLaunchKernelGGL(fe_norm_concat_kernel, dim3(B * T), dim3(256), 0, stream, hidden, mask, stats, states, B, T, D, padding_right, out);
    HIP_CHECK(hipGetLastError());
}
void launch_register_plan(const int32_t* mask, int B, int T, int32_t* src, hipStream_t stream) {
    hipLaunchKernelGGL(register_plan_kernel, dim3(B), dim3(64), 0, stream, mask, T, src);
    HIP_CHECK(hipGetLastError());
}
void launch_register_gather(const bf16_t* enc, const float* registers, const int32_t* src, int B, int T, int D, int R, float* x,
                            hipStream_t stream) {
    hipLaunchKernelGGL(register_gather_kernel, dim3(B * T), dim3(256), 0, stream, enc, registers, src, T, D, R, x);
    HIP_CHECK(hipGetLastError());
}
void launch_fill_const_i32(int32_t* p, long n, int32_t v, hipStream_t stream) {
    hipLaunchKernelGGL(fill_const_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, v);
    HIP_CHECK(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------------------------
// VAE encoder kernels (VideoEncoder.swift)
// ---------------------------------------------------------------------------------------------------------------
namespace {
// pixels f32 [3][T][H][W] -> bf16 channels-last [T][H/4][W/4][64]: channel = c*16 + pw*4 + ph for c < 3 (pW before pH,
// VideoEncoder.swift:25-31), channels 48..63 zero (K-tile padding of conv_in)
__global__ void enc_patchify_kernel(const float* __restrict__ px, bf16_t* __restrict__ out, int T, int H, int W) {
    const long n = (long)T * (H / 4) * (W / 4) * 64;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)(i & 63);
    long pos = i >> 6;
    const int w4 = (int)(pos % (W / 4));
    pos /= (W / 4);
    const int h4 = (int)(pos % (H / 4));
    const int t = (int)(pos / (H / 4));
    float v = 0.f;
    if (ch < 48) {
        const int c = ch >> 4, pw = (ch >> 2) & 3, ph = ch & 3;
        v = px[(((long)c * T + t) * H + (h4 * 4 + ph)) * W + (w4 * 4 + pw)];
    }
    out[i] = f32_to_bf16(v);
}

// out[t2][h2][w2][co] = s2d(conv)[co] + mean_g s2d(x)[co*G + g]  (VAESpaceToDepthDownsample3d, VideoEncoder.swift:146-167)
// s2d channel cs of a tensor with Cs source channels: source channel cs / S, sub = cs % S -> (it, ih, iw); an odd T is padded
// in FRONT with copies of frame 0, i.e. source frame max(0, t2*ft + it - padT)
__global__ void enc_s2d_residual_kernel(const float* __restrict__ conv, int Cc, const float* __restrict__ x, int Cx,
                                        float* __restrict__ out, int Cout, int T, int H, int W, int ft, int fh, int fw) {
    const int S = ft * fh * fw;
    const int padT = (ft - T % ft) % ft;
    const int T2 = (T + padT) / ft, H2 = H / fh, W2 = W / fw;
    const long n = (long)T2 * H2 * W2 * Cout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % Cout);
    long pos = i / Cout;
    const int w2 = (int)(pos % W2);
    pos /= W2;
    const int h2 = (int)(pos % H2);
    const int t2 = (int)(pos / H2);
    auto fetch = [&](const float* src, int C, int cs) {
        const int c = cs / S, sub = cs - c * S;
        const int it = sub / (fh * fw), ih = (sub / fw) % fh, iw = sub % fw;
        int ts = t2 * ft + it - padT;
        ts = ts < 0 ? 0 : ts;
        return src[(((long)ts * H + (h2 * fh + ih)) * W + (w2 * fw + iw)) * C + c];
    };
    const int G = Cx * S / Cout;
    float acc = 0.f;
    for (int g = 0; g < G; ++g) acc += fetch(x, Cx, co * G + g);
    out[i] = fetch(conv, Cc, co) + acc / (float)G;
}

// z f32 [P][ldz] (first C channels) -> out [C][P] f32 with optional (z - mean[c]) / std[c]  (LTXPipeline.swift:1920-1927)
__global__ void enc_finish_kernel(const float* __restrict__ z, long ldz, const float* __restrict__ mean, const float* __restrict__ stdv,
                                  float* __restrict__ out, int C, long P) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * P) return;
    const int c = (int)(i / P);
    const long p = i - (long)c * P;
    float v = z[p * ldz + c];
    if (mean) v = (v - mean[c]) / stdv[c];
    out[i] = v;
}
}  // namespace

void launch_enc_patchify(const float* pixels, bf16_t* out, int T, int H, int W, hipStream_t stream) {
    LTX_REQUIRE(H % 4 == 0 && W % 4 == 0, "encoder patchify: H=%d W=%d must be multiples of 4", H, W);
    const long n = (long)T * (H / 4) * (W / 4) * 64;
    hipLaunchKernelGGL(enc_patchify_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, pixels, out, T, H, W);
    HIP_CHECK(hipGetLastError());
}
void launch_enc_s2d_residual(const float* conv, int Cc, const float* x, int Cx, float* out, int Cout, int T, int H, int W, int ft,
                             int fh, int fw, hipStream_t stream) {
    LTX_REQUIRE(H % fh == 0 && W % fw == 0 && (Cx * ft * fh * fw) % Cout == 0 && Cc * ft * fh * fw == Cout, "s2d: bad shapes");
    const int padT = (ft - T % ft) % ft;
    const long n = (long)((T + padT) / ft) * (H / fh) * (W / fw) * Cout;
    hipLaunchKernelGGL(enc_s2d_residual_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, conv, Cc, x, Cx, out, Cout, T, H, W, ft, fh, fw);
    HIP_CHECK(hipGetLastError());
}
void launch_enc_finish(const float* z, long ldz, const float* mean, const float* stdv, float* out, int C, long P, hipStream_t stream) {
    hipLaunchKernelGGL(enc_finish_kernel, dim3(cdiv((long)C * P, 256)), dim3(256), 0, stream, z, ldz, mean, stdv, out, C, P);
    HIP_CHECK(hipGetLastError());
}
