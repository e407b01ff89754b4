// overlap.hip - does VALU work of one wave overlap MFMA work of ANOTHER wave on the same SIMD (gfx950)?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 (the second wave of each SIMD) run VALU work.
// mode 1: MFMA waves only; 2: VALU waves only; 3: both; 4: both kinds of work in EVERY wave, one after the other (no ping-pong).
// Build: hipcc -O3 --offload-arch=gfx950 -o overlap overlap.hip ; run: ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int VKIND>
__device__ __forceinline__ void valu_block(float (&x)[32], float c) {
    // ~ the softmax cluster of the attention kernel: 32 exp2 + 32 fma + 32 add + 16 cvt_pk
    if (VKIND == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i] * c - 1.0f);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x[i];
        x[0] += s * 1e-30f;
    } else if (VKIND == 1) {  // plain FMAs only, same instruction count (96)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = x[i] * c + 0.5f;
    } else {  // transcendental only
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    }
}

template <int VKIND>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, float c) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = mode == 4 || ((mode & 1) && wave < 4);
    const bool do_v = mode == 4 || ((mode & 2) && wave >= 4);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (do_v) valu_block<VKIND>(x, c);
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// mode 5: ONE wave per SIMD (256 threads) running both kinds of work interleaved in program order, one MFMA then 3 VALU
// instructions (sched_group_barrier) - what a one-wave-per-SIMD attention kernel would have to look like.
template <int MF>
__global__ __launch_bounds__(256, 1) void k1(float* out, int iters, float c) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (MF) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        // softmax-like: 32 fma + 32 exp2 + 32 add
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { x[i] = __builtin_amdgcn_exp2f(x[i] * c - 1.0f); s += x[i]; }
        x[0] += s * 1e-30f;
        if (MF) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);  // VALU | TRANS
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run1(float* out) {
    const int iters = 2000;
    for (int mf = 0; mf < 2; ++mf) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        if (mf) hipLaunchKernelGGL(k1<1>, dim3(256), dim3(256), 0, 0, out, 10, 1.0001f); else hipLaunchKernelGGL(k1<0>, dim3(256), dim3(256), 0, 0, out, 10, 1.0001f);
        (void)hipEventRecord(e0);
        if (mf) hipLaunchKernelGGL(k1<1>, dim3(256), dim3(256), 0, 0, out, iters, 1.0001f); else hipLaunchKernelGGL(k1<0>, dim3(256), dim3(256), 0, 0, out, iters, 1.0001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  one wave per SIMD, %s: %8.1f ns/iter\n", mf ? "32 MFMA interleaved with the softmax-like VALU block" : "softmax-like VALU block only", ms * 1e6f / iters);
    }
}

template <int VKIND>
void run(const char* name, float* out) {
    const int iters = 2000;
    printf("%s\n", name);
    float t[5] = {0};
    for (int mode = 1; mode <= 4; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, out, 10, mode, 1.0001f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, out, iters, mode, 1.0001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        t[mode] = ms * 1e6f / iters;  // ns per iteration
        const char* mn[] = {"", "MFMA waves only (32 MFMA/iter)", "VALU waves only", "both, separate waves", "both, same wave (8 waves)"};
        printf("  mode %d %-34s %8.1f ns/iter\n", mode, mn[mode], t[mode]);
    }
    printf("  separate-wave overlap: both=%.0f vs sum=%.0f max=%.0f\n", t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0>("softmax-like VALU (32 exp2+fma, 32 add)", out);
    run<1>("96 plain FMAs", out);
    run<2>("32 exp2 only", out);
    run1(out);
    hipDeviceSynchronize();
    return 0;
}
