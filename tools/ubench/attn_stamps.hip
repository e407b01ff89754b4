// attn_stamps.hip - cycle stamps at the phase boundaries of the ping-pong attention kernel (one steady-state tile step of one
// workgroup, wave 0 = group 0 and wave 4 = group 1, both on SIMD 0). Build from the repo root:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPP_STAMPS (ping-pong kernel) or -DW48_STAMPS (48-query assembly kernel) -Iinclude -Iltx-video-swift-mlx_amd/csrc -o tools/ubench/attn_stamps tools/ubench/attn_stamps.hip
#include "../../ltx-video-swift-mlx_amd/csrc/attention.hip"

// the kernel launcher's profiling hooks (runtime.cpp) are not linked into this harness
Profiler* prof_current() { return nullptr; }
ProfRec* Profiler::begin(int, double, hipStream_t) { return nullptr; }
void Profiler::end(ProfRec*, hipStream_t) {}

#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main() {
    const int T = getenv("ATTN_T") ? atoi(getenv("ATTN_T")) : 6144, H = 32, D = 4096;
    std::vector<bf16_t> h((size_t)T * D);
    for (size_t i = 0; i < h.size(); ++i) h[i] = host_f32_to_bf16((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
    bf16_t *q, *k, *vt, *o;
    hipMalloc(&q, h.size() * 2); hipMalloc(&k, h.size() * 2); hipMalloc(&vt, h.size() * 2); hipMalloc(&o, h.size() * 2);
    hipMemcpy(q, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    AttnArgs a;
    a.Q = q; a.ldq = D; a.K = k; a.ldk = D; a.Vt = vt; a.ldvt = T; a.O = o; a.ldo = D;
    a.B = 1; a.H = H; a.Tq = T; a.Tk = T;
    a.q_prescaled = getenv("ATTN_PS") ? 1 : 0;
#ifdef W48_STAMPS
    setenv("LTX_ATTN_IMPL", "4", 1);
    for (int it = 0; it < 3; ++it) launch_attention(a, 0);
    hipDeviceSynchronize();
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) launch_attention(a, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("T = %d, prescaled %d: %.1f us per launch (stamps build)\n", T, a.q_prescaled, ms * 1000 / 20);
    }
    {
        unsigned long long st[8][32];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_w48_stamps), sizeof(st));
        printf("48-query assembly kernel, one tile step (ring slot 0 of the last loop iteration), cycles:\n");
        printf("wave | A: QK+exp/pack  B1: PV k-step 0 + pack  B2: PV k-step 1 + maxima + DMA + check  vmcnt+barrier | step\n");
        for (int w = 0; w < 4; ++w)
            printf("  %d  | %8llu %16llu %14llu %16llu       | %llu\n", w, st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2],
                   st[w][4] - st[w][3], st[w][4] - st[w][0]);
        printf("whole kernel, wave 0 (T = %d, %d key tiles): prologue %llu cycles, tile loop %llu, epilogue %llu\n", T, T / 64,
               st[0][17] - st[0][16], st[0][18] - st[0][17], st[0][19] - st[0][18]);
        return 0;
    }
#endif
#ifdef X32_STAMPS   // -DX32_STAMPS: the 32x32 stream (tools/gen_attn_x32.py --stamps), ATTN_PS is implied
    a.q_prescaled = 1;
    setenv("LTX_ATTN_IMPL", "5", 1);
    for (int it = 0; it < 3; ++it) launch_attention(a, 0);
    hipDeviceSynchronize();
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) launch_attention(a, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("T = %d: %.1f us per launch (stamps build)\n", T, ms * 1000 / 20);
        unsigned long long st[8][16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_x32_stamps), sizeof(st));
        for (int w = 0; w < 4; ++w) {
            printf("wave %d: last full loop iteration (4 tile steps) %llu cycles = %llu per step | prologue %llu, tile loop %llu (%d tiles: %llu per tile), epilogue %llu",
                   w, st[w][0] - st[w][1], (st[w][0] - st[w][1]) / 4, st[w][3] - st[w][2], st[w][4] - st[w][3], T / 64,
                   (st[w][4] - st[w][3]) / (unsigned long long)(T / 64), st[w][5] - st[w][4]);
            if (st[w][7]) printf(" | parts A %llu B1 %llu B2 %llu check+wait+barrier %llu", st[w][7] - st[w][6], st[w][8] - st[w][7], st[w][9] - st[w][8], st[w][10] - st[w][9]);
            printf("\n");
        }
        return 0;
    }
#endif
#ifdef PP_STAMPS
    for (int it = 0; it < 3; ++it) launch_attention(a, 0);
    hipDeviceSynchronize();
    unsigned long long st[8][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pp_stamps), sizeof(st));
    unsigned long long t0 = st[0][0];
    printf("wave simd | t(start) stage softmax waitA  mfma  waitB | step   (cycles; t relative to wave 0)\n");
    for (int w = 0; w < 8; ++w) {
        const unsigned hw = (unsigned)st[w][6];
        printf("  %d   %d   | %7lld %5llu %7llu %5llu %5llu %6llu | %5llu\n", w, (hw >> 4) & 3, (long long)(st[w][0] - t0), st[w][1] - st[w][0],
               st[w][2] - st[w][1], st[w][3] - st[w][2], st[w][4] - st[w][3], st[w][5] - st[w][4], st[w][5] - st[w][0]);
    }
#endif
    return 0;
}
