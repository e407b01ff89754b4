// synth.hip - counter-based N(mean, std^2) fills for synthetic weights / inputs (bench.py and size-independent
// property tests; there are no real checkpoints on the GPU box). Each element i is a pure function of
// (seed, i): splitmix64 -> two uniforms -> Box-Muller, so fills are reproducible and order-independent.
#include "runtime.h"

namespace {

LTX_DEVFN uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
LTX_DEVFN float normal_at(uint64_t seed, uint64_t i) {
    const uint64_t r = splitmix64(seed ^ (i * 0xD6E8FEB86659FD93ull));
    const float u1 = ((float)((r >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777217.0f);  // (0,1]
    const float u2 = (float)((r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);            // [0,1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

__global__ void fill_normal_bf16_kernel(bf16_t* p, long n, uint64_t seed, float mean, float stddev) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = f32_to_bf16(mean + stddev * normal_at(seed, (uint64_t)i));
}
__global__ void fill_normal_f32_kernel(float* p, long n, uint64_t seed, float mean, float stddev, int round_bf16) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = mean + stddev * normal_at(seed, (uint64_t)i);
        if (round_bf16) v = bf16_to_f32(f32_to_bf16(v));
        p[i] = v;
    }
}
__global__ void fill_const_f32_kernel(float* p, long n, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
int grid_for(long n) {
    long g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

}  // namespace

void launch_fill_normal_bf16(bf16_t* p, long n, uint64_t seed, float mean, float stddev, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_normal_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, n, seed, mean, stddev);
    HIP_CHECK(hipGetLastError());
}
void launch_fill_normal_f32(float* p, long n, uint64_t seed, float mean, float stddev, int round_bf16, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_normal_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, n, seed, mean, stddev, round_bf16);
    HIP_CHECK(hipGetLastError());
}
void launch_fill_const_f32(float* p, long n, float v, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_const_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, n, v);
    HIP_CHECK(hipGetLastError());
}
