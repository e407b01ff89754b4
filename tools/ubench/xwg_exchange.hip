// xwg_exchange.hip - what does an exchange of row statistics BETWEEN the workgroups of a one-round launch cost on MI355X?
// (fusing the q / k RMSNorm into the q|k projection needs the sum of squares of a row over 4096 columns, of
// which a tile holds 256; the 16 workgroups that share a row tile would publish partial sums and wait for each other.)
// 256 workgroups x 256 threads, one per CU. Each spins ~WORK_US microseconds (a stand-in for its K loop, with +-JITTER % per workgroup),
// then: publishes 192 partial sums with agent-scope (sc1) stores, s_waitcnt vmcnt(0), barrier, one lane bumps the arrival counter of its
// group of GROUP workgroups and polls it (bounded), barrier, every wave reads the GROUP partials of its rows with sc1 loads and sums them
// in a fixed order. Stamps (100 MHz wall clock): work done -> published -> all arrived -> sums read. Checks the sums.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench/xwg_exchange tools/ubench/xwg_exchange.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256, 1) void xwg_kernel(float* partials, int* counters, float* sums, unsigned long long* stamps, int* err, int group,
                                                     int work_ticks, int jitter_pct, int epoch) {
    __shared__ float lds[1];
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int g = wg / group, member = wg % group;
    // stand-in for the K loop
    const unsigned long long t0 = wall_clock64();
    const int my_ticks = work_ticks + (int)((long)work_ticks * jitter_pct / 100 * ((wg * 2654435761u >> 16) % 201 - 100) / 100);
    while (wall_clock64() - t0 < (unsigned long long)my_ticks) __builtin_amdgcn_s_sleep(4);
    if (tid == 0) stamps[wg * 4 + 0] = wall_clock64();
    // publish 192 partial sums: rows of the group x member slot
    if (tid < 192) {
        const float v = (float)(epoch * 1000 + wg) + 0.25f * tid;
        __hip_atomic_store(partials + ((size_t)g * 192 + tid) * group + member, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        stamps[wg * 4 + 1] = wall_clock64();
        __hip_atomic_fetch_add(counters + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counters + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < group * epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { *err = 1; break; }  // bounded: an error, never a hang
        }
        stamps[wg * 4 + 2] = wall_clock64();
    }
    __syncthreads();
    if (tid < 192) {
        float s = 0.f;
        const float* p = partials + ((size_t)g * 192 + tid) * group;
        for (int m = 0; m < group; ++m) s += __hip_atomic_load(p + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sums[(size_t)wg * 192 + tid] = s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) stamps[wg * 4 + 3] = wall_clock64();
    (void)lds;
}

int main() {
    const int NWG = 256, group = getenv("GROUP") ? atoi(getenv("GROUP")) : 16;
    const int work_us = getenv("WORK_US") ? atoi(getenv("WORK_US")) : 60, jitter = getenv("JITTER") ? atoi(getenv("JITTER")) : 2;
    float *partials, *sums;
    int *counters, *err;
    unsigned long long* stamps;
    (void)hipMalloc(&partials, (size_t)NWG * 192 * 4); (void)hipMalloc(&sums, (size_t)NWG * 192 * 4);
    (void)hipMalloc(&counters, 1024); (void)hipMalloc(&err, 4); (void)hipMalloc(&stamps, NWG * 4 * 8);
    (void)hipMemset(counters, 0, 1024); (void)hipMemset(err, 0, 4);
    std::vector<unsigned long long> st(NWG * 4);
    std::vector<float> hs((size_t)NWG * 192);
    double pub = 0, wait = 0, rd = 0, wmax = 0;
    int bad = 0, herr = 0;
    const int reps = 50;
    for (int e = 1; e <= reps; ++e) {
        hipLaunchKernelGGL(xwg_kernel, dim3(NWG), dim3(256), 0, 0, partials, counters, sums, stamps, err, group, work_us * 100, jitter, e);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(st.data(), stamps, NWG * 4 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs.data(), sums, hs.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        for (int wg = 0; wg < NWG; ++wg) {
            if (e > 5) {
                pub += (st[wg * 4 + 1] - st[wg * 4 + 0]) * 0.01; wait += (st[wg * 4 + 2] - st[wg * 4 + 1]) * 0.01; rd += (st[wg * 4 + 3] - st[wg * 4 + 2]) * 0.01;
                wmax = std::max(wmax, (st[wg * 4 + 2] - st[wg * 4 + 1]) * 0.01);
            }
            const int g = wg / group;
            for (int r = 0; r < 192; r += 37) {
                float ref = 0.f;
                for (int m = 0; m < group; ++m) ref += (float)(e * 1000 + g * group + m) + 0.25f * r;
                if (hs[(size_t)wg * 192 + r] != ref) ++bad;
            }
        }
    }
    const double n = (double)NWG * (reps - 5);
    printf("group of %d workgroups, work %d us +- %d %%: publish (sc1 stores + vmcnt(0) + barrier) %.2f us | wait for the group %.2f us (max %.2f) | read + sum %d partials per row %.2f us | "
           "stale / wrong sums: %d, timeouts: %d\n", group, work_us, jitter, pub / n, wait / n, wmax, group, rd / n, bad, herr);
    return bad || herr;
}
