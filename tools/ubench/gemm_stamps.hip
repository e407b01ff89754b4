// gemm_stamps.hip - cycle stamps inside one K-tile body of the assembly GEMM kernel (all four waves of one workgroup).
// Build from the repo root (after `python3 tools/gen_gemm_asm.py --stamps`):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGEMM_ASM_STAMPS -Iinclude -Iltx-video-swift-mlx_amd/csrc -o tools/ubench/gemm_stamps tools/ubench/gemm_stamps.hip
#include "../../ltx-video-swift-mlx_amd/csrc/gemm.hip"
#include "../../ltx-video-swift-mlx_amd/csrc/options.cpp"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

Profiler* prof_current() { return nullptr; }
ProfRec* Profiler::begin(int, double, hipStream_t) { return nullptr; }
void Profiler::end(ProfRec*, hipStream_t) {}
void launch_norm_mod(const float*, long, const float*, const float*, long, int, bf16_t*, long, int, int, int, float, int, hipStream_t, const int*) { abort(); }

int main() {
    const int M = getenv("GEMM_M") ? atoi(getenv("GEMM_M")) : 1536, N = getenv("GEMM_N") ? atoi(getenv("GEMM_N")) : 8192, K = getenv("GEMM_K") ? atoi(getenv("GEMM_K")) : 4096;
    std::vector<bf16_t> ha((size_t)M * K), hb((size_t)N * K);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = host_f32_to_bf16((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = host_f32_to_bf16((float)((i * 40503u >> 12) & 255) / 256.f - 0.5f);
    bf16_t *a, *b;
    float* c;
    (void)hipMalloc(&a, ha.size() * 2); (void)hipMalloc(&b, hb.size() * 2); (void)hipMalloc(&c, (size_t)M * N * 4);
    (void)hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    GemmArgs g;
    g.A = a; g.lda = K; g.B = b; g.ldb = K; g.M = M; g.N = N; g.K = K;
    if (getenv("GATED")) {  // the DiT's gated-residual epilogue: x += gate[n] * (A.B^T + bias), in place on the f32 stream
        float *gate, *bias;
        (void)hipMalloc(&gate, (size_t)N * 4); (void)hipMalloc(&bias, (size_t)N * 4);
        (void)hipMemset(gate, 0, (size_t)N * 4); (void)hipMemset(bias, 0, (size_t)N * 4);
        (void)hipMemset(c, 0, (size_t)M * N * 4);
        g.ep.resid = 1; g.ep.gate = gate; g.ep.gate_bstride = N; g.ep.rows_per_batch = M; g.ep.bias_n = bias;
    }
    if (getenv("BIAS_M")) {  // the V^T projection: weights as the row operand, bias per output row
        float* bm;
        (void)hipMalloc(&bm, (size_t)M * 4);
        (void)hipMemset(bm, 0, (size_t)M * 4);
        g.ep.bias_m = bm;
    }
    const char* ob = getenv("OUT_BF16");
    if (ob) { g.ep.out_bf16 = (bf16_t*)c; g.ep.ld_bf16 = N; } else { g.ep.out_f32 = c; g.ep.ld_f32 = N; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cfg = getenv("GEMM_CFG") ? atoi(getenv("GEMM_CFG")) : 71;
    for (int it = 0; it < 3; ++it) launch_gemm_bf16_cfg(g, cfg, 0);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) launch_gemm_bf16_cfg(g, cfg, 0);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%dx%dx%d: %.1f us per launch, %.0f TFLOP/s\n", M, N, K, ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    unsigned long long st[5][8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_gemm_stamps), sizeof(st));
#ifdef GEMM_V2_STAMPS
    if (cfg == 21 || cfg == 25) {
        printf("tile_cfg 21 / 25 (ring kernel), wall-clock stamps in ns (100 MHz counter):\nblock wave | prologue (first tile landed)   main loop   epilogue issue   stores drained | total\n");
        for (int w = 0; w < 5; ++w)
            printf("  %3d  %d  | %10llu %20llu %14llu %14llu       | %llu   main loop: %llu cycles = %.0f per K-tile at %.3f GHz\n", w < 4 ? 7 : 200, w < 4 ? w : 0,
                   10 * (st[w][1] - st[w][0]), 10 * (st[w][2] - st[w][1]), 10 * (st[w][3] - st[w][2]), 10 * (st[w][4] - st[w][3]), 10 * (st[w][4] - st[w][0]),
                   st[w][5], (double)st[w][5] / (K / 64), (double)st[w][5] / (10.0 * (st[w][2] - st[w][1])));
        return 0;
    }
#endif
    if (cfg == 73) {
        printf("tile_cfg 73, one K-tile body (last pass), cycles:\nwave | k-step 0 (24 MFMA + 10 LDS-DMA)  k-step 1 to the wait (18 MFMA)  vmcnt wait  lgkmcnt+barrier  tail (6 MFMA + entry reads) | tile\n");
        for (int w = 0; w < 4; ++w)
            printf("  %d  | %10llu %24llu %18llu %14llu %18llu              | %llu\n", w, st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2],
                   st[w][4] - st[w][3], st[w][5] - st[w][4], st[w][5] - st[w][0]);
        printf("whole assembly block: %llu cycles; epilogue: %llu cycles (wave 0)\n", st[0][6], st[0][7]);
        return 0;
    }
    printf("one K-tile body (last pass), cycles:\nwave | k-step 0 (48 MFMA + loads)  vmcnt wait  k-step 1 up to the barrier (36 MFMA + ds_writes)  lgkmcnt+barrier  tail (12 MFMA + entry reads) | tile\n");
    for (int w = 0; w < 4; ++w)
        printf("  %d  | %10llu %20llu %22llu %30llu %20llu          | %llu\n", w, st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2],
               st[w][4] - st[w][3], st[w][5] - st[w][4], st[w][5] - st[w][0]);
    printf("whole assembly block (prologue + %d tiles): %llu cycles; epilogue: %llu cycles (wave 0)\n", K / 64, st[0][6], st[0][7]);
    return 0;
}
