// common.h - shared device/host helpers for libltxhip (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, workgroups are multiples of 64 threads
//   * bf16 is carried as raw uint16_t/short bits in HBM; arithmetic is f32
//   * "tokens" are rows of the DiT latent sequence (t = (f*H'+h)*W'+w), "text keys" are rows of the
//     projected caption context, "positions" are (f,y,x) voxels of a VAE feature map (channels-last)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LTX_DEVFN __device__ __forceinline__

// f32 -> bf16 round-to-nearest-even (NaN stays NaN: plain cast lowers to v_cvt_pk_bf16_f32 on gfx950).
LTX_DEVFN bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
LTX_DEVFN float bf16_to_f32(bf16_t b) {
    return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
LTX_DEVFN uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// Host-side bf16 conversions (RNE), used by loaders and synthetic weight generation.
static inline bf16_t host_f32_to_bf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static inline float host_bf16_to_f32(bf16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
static inline float host_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else {
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3ffu;
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

LTX_DEVFN float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
LTX_DEVFN float wave_reduce_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// tanh-approximate GELU exactly as the reference's MLXNN.geluApproximate formula
// (LTXFeedForward.swift:13-17): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3))).
LTX_DEVFN float gelu_tanh(float x) {
    // 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3): one v_exp_f32 and one v_rcp_f32 instead of the library
    // tanhf (a branchy ~30-instruction sequence that the FFN epilogue runs 96 times per lane and tile on the SIMDs that also issue the
    // MFMAs). The exponent is formed as x * (c0 + c1 x^2) in base 2 with the constants folded (c0 = -2 sqrt(2/pi) log2(e), c1 = 0.044715
    // c0): three multiplies less per element than the literal form. |error| <= 3e-7 * |x|, far inside the bf16 rounding of the stored
    // value; exp overflow for very negative u gives x * rcp(inf) = -0, the correct limit.
    // (round 4: written as x * v_rcp_f32(...). __fdividef compiles to the full IEEE division here - v_div_scale x2, v_rcp, four FMAs,
    // v_div_fmas, v_div_fixup per element, found in the 192x256 kernel's ISA - and the FFN-up launch spent 13.6 us of 175 in this function)
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c1 = c0 * 0.044715f;
    const float t = x * __builtin_fmaf(c1, x * x, c0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// x * sigmoid(x) with one v_exp_f32 and one v_rcp_f32 (relative error ~2e-7, far inside the bf16 rounding of every stored value); the
// IEEE division it replaces is a ten-instruction sequence that the conv epilogues run per element on the SIMDs that issue the MFMAs
LTX_DEVFN float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four row rotations, no LDS crossbar
LTX_DEVFN float row16_allsum(float v) {
#define LTX_ROW_ROR_ADD(N) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (N), 0xf, 0xf, false))
    LTX_ROW_ROR_ADD(8);
    LTX_ROW_ROR_ADD(4);
    LTX_ROW_ROR_ADD(2);
    LTX_ROW_ROR_ADD(1);
#undef LTX_ROW_ROR_ADD
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id (guide T1): blocks b and b+8 share an XCD under
// round-robin dispatch, so give each XCD a contiguous chunk of the tile grid. Speed only.
LTX_DEVFN int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Status codes mirror LTXError (reference LTXVideo.swift:66-141); keep in sync with include/ltxhip.h.
enum {
    LTXS_OK = 0,
    LTXS_MODEL_NOT_LOADED = 1,
    LTXS_INVALID_CONFIGURATION = 2,
    LTXS_INSUFFICIENT_MEMORY = 3,
    LTXS_WEIGHT_LOADING_FAILED = 4,
    LTXS_GENERATION_FAILED = 5,
    LTXS_GENERATION_CANCELLED = 6,
    LTXS_INVALID_FRAME_COUNT = 7,
    LTXS_INVALID_DIMENSIONS = 8,
    LTXS_FILE_NOT_FOUND = 9,
    LTXS_INVALID_LORA = 10,
    LTXS_HIP_ERROR = 11,
};

struct LtxError {
    int code;
    std::string msg;
};

#define LTX_THROW(code_, ...)                                  \
    do {                                                       \
        char _buf[512];                                        \
        snprintf(_buf, sizeof(_buf), __VA_ARGS__);             \
        throw LtxError{(code_), std::string(_buf)};            \
    } while (0)

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            LTX_THROW(_e == hipErrorOutOfMemory ? LTXS_INSUFFICIENT_MEMORY : LTXS_HIP_ERROR,     \
                      "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
    } while (0)

#define LTX_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) LTX_THROW(LTXS_INVALID_CONFIGURATION, __VA_ARGS__); \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only, and one process may hold contexts on several
// GPUs (ltx_ctx_create(device)): a process-wide "done" flag would leave the second device's first launch of a > 64 KB LDS kernel
// without its attribute. One bit per device, set under a lock so that a second thread cannot launch between check and set.
// CUs of the current device (cached per device index; 256 on MI355X)
inline int device_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cached[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t p;
        v = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        cached[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    std::mutex mu;
    template <class F>
    void run(F&& f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lk(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        f();
        done.fetch_or(bit, std::memory_order_release);
    }
};
