// valu_slots.hip - how many plain VALU instructions fit "for free" between back-to-back MFMAs of ONE wave (one wave per SIMD)?
// For v_mfma_f32_32x32x16_bf16 (8 passes) and v_mfma_f32_16x16x32_bf16 (4 passes), k = 0..8 independent v_fma_f32 after each MFMA,
// written in assembly so the order is exactly as listed. Build: hipcc -O3 --offload-arch=gfx950 -o valu_slots valu_slots.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int BIG, int K>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    float x = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define V1 "v_fma_f32 v40, v40, v41, v42\n\t"
#define V2 "v_fma_f32 v43, v43, v41, v42\n\t"
#define V3 "v_fma_f32 v44, v44, v41, v42\n\t"
#define V4 "v_fma_f32 v45, v45, v41, v42\n\t"
#define V5 "v_fma_f32 v46, v46, v41, v42\n\t"
#define V6 "v_fma_f32 v47, v47, v41, v42\n\t"
#define V7 "v_fma_f32 v48, v48, v41, v42\n\t"
#define V8 "v_fma_f32 v49, v49, v41, v42\n\t"
#define VK ((K > 0 ? V1 : "") )
#define MF32(d) "v_mfma_f32_32x32x16_bf16 a[" #d "], v[32:35], v[36:39], a[" #d "]\n\t"
#define MF16(d) "v_mfma_f32_16x16x32_bf16 a[" #d "], v[32:35], v[36:39], a[" #d "]\n\t"
        if (BIG) {
#define G32(d) asm volatile(MF32(d) ::: "memory"); if (K > 0) asm volatile(V1:::); if (K > 1) asm volatile(V2:::); if (K > 2) asm volatile(V3:::); if (K > 3) asm volatile(V4:::); \
               if (K > 4) asm volatile(V5:::); if (K > 5) asm volatile(V6:::); if (K > 6) asm volatile(V7:::); if (K > 7) asm volatile(V8:::);
            G32(0:15) G32(16:31) G32(32:47) G32(48:63) G32(0:15) G32(16:31) G32(32:47) G32(48:63)
        } else {
#define G16(d) asm volatile(MF16(d) ::: "memory"); if (K > 0) asm volatile(V1:::); if (K > 1) asm volatile(V2:::); if (K > 2) asm volatile(V3:::); if (K > 3) asm volatile(V4:::); \
               if (K > 4) asm volatile(V5:::); if (K > 5) asm volatile(V6:::); if (K > 6) asm volatile(V7:::); if (K > 7) asm volatile(V8:::);
            G16(0:3) G16(4:7) G16(8:11) G16(12:15) G16(16:19) G16(20:23) G16(24:27) G16(28:31)
        }
    }
    asm volatile("" ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49");
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int BIG, int K>
void run(float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BIG, K>), dim3(256), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<BIG, K>), dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %s, %d VALU per MFMA: %6.2f ns per MFMA\n", BIG ? "32x32x16 (8 passes)" : "16x16x32 (4 passes)", K, ms * 1e6f / iters / 8);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    run<1, 0>(out); run<1, 1>(out); run<1, 2>(out); run<1, 3>(out); run<1, 4>(out); run<1, 5>(out); run<1, 6>(out); run<1, 8>(out);
    run<0, 0>(out); run<0, 1>(out); run<0, 2>(out); run<0, 3>(out); run<0, 4>(out); run<0, 6>(out);
    (void)hipDeviceSynchronize();
    return 0;
}
