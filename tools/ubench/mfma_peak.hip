// Sustained dense-MFMA rate of the chip: four waves per CU, each issuing independent MFMAs from registers (no memory traffic at all).
// Prints TFLOP/s for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16 at several run lengths, so that the ceiling a GEMM can reach
// at the clocks the part sustains under matrix load is known (the 2.5 PFLOP/s headline assumes 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND> __global__ __launch_bounds__(256) void peak(float* out, int iters, int random) {
    bf16x8 a, b;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = random ? (__bf16)(((int)(h >> 16 & 0xffff) - 32768) * (1.0f / 32768)) : (__bf16)(float)(threadIdx.x & 3);
        h = h * 1664525u + 1013904223u;
        b[i] = random ? (__bf16)(((int)(h >> 16 & 0xffff) - 32768) * (1.0f / 32768)) : (__bf16)1.0f;
    }
    if constexpr (KIND == 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0];
        if (s == 12345.f) out[0] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 12345.f) out[0] = s;
    }
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int random = 0; random < 2; ++random)
    for (int kind = 0; kind < 2; ++kind) {
        for (int iters : {500, 10000, 100000}) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (kind == 0) peak<0><<<256, 256>>>(out, iters, random);
                else peak<1><<<256, 256>>>(out, iters, random);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                // per wave and iteration: kind 0: 16 MFMAs x 16*16*32*2 flops; kind 1: 8 MFMAs x 32*32*16*2 flops
                const double flops = 256.0 * 4 * iters * (kind == 0 ? 16.0 * 16 * 16 * 32 * 2 : 8.0 * 32 * 32 * 16 * 2);
                const double cyc = iters * (kind == 0 ? 16.0 * 16 : 8.0 * 32);
                if (rep == 2)
                    printf("%s %s  %6d iterations  %9.1f us  %7.1f TFLOP/s  (implied clock %.2f GHz if one MFMA pass per 4 cycles)\n",
                           random ? "random data " : "trivial data", kind == 0 ? "16x16x32 bf16" : "32x32x16 bf16", iters, ms * 1e3, flops / (ms * 1e-3) / 1e12, cyc / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
