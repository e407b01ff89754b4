// gemm_dtl_stamps.hip - cycle stamps inside one K-tile body of the 192x256 GEMM kernel (tile_cfg 75), all four waves of one
// workgroup, plus launch timing for schedule variants (DTL_KNOBS of tools/gen_gemm_asm_dtl.py). Build from the repo root:
//   python3 tools/gen_gemm_asm_dtl.py --stamps      (or: DTL_KNOBS=bar1=20,dma0=22 python3 tools/gen_gemm_asm_dtl.py)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDTL_STAMPS -Iinclude -Iltx-video-swift-mlx_amd/csrc -o tools/ubench/gemm_dtl_stamps tools/ubench/gemm_dtl_stamps.hip
#include "../../ltx-video-swift-mlx_amd/csrc/gemm.hip"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

Profiler* prof_current() { return nullptr; }
ProfRec* Profiler::begin(int, double, hipStream_t) { return nullptr; }
void Profiler::end(ProfRec*, hipStream_t) {}

int main() {
    const int M = getenv("GEMM_M") ? atoi(getenv("GEMM_M")) : 1536, N = getenv("GEMM_N") ? atoi(getenv("GEMM_N")) : 8192, K = getenv("GEMM_K") ? atoi(getenv("GEMM_K")) : 4096;
    const int NB = 12;  // rotate through distinct weight matrices: HBM-cold B operand, as in the DiT
    std::vector<bf16_t> ha((size_t)M * K), hb((size_t)N * K);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = host_f32_to_bf16((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = host_f32_to_bf16((float)((i * 40503u >> 12) & 255) / 256.f - 0.5f);
    bf16_t *a, *b[NB];
    bf16_t* c;
    (void)hipMalloc(&a, ha.size() * 2); (void)hipMalloc(&c, (size_t)M * N * 2);
    (void)hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    for (int i = 0; i < NB; ++i) {
        (void)hipMalloc(&b[i], hb.size() * 2);
        (void)hipMemcpy(b[i], hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    }
    GemmArgs g;
    g.A = a; g.lda = K; g.ldb = K; g.M = M; g.N = N; g.K = K;
    g.ep.out_bf16 = c; g.ep.ld_bf16 = N;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cfg = getenv("GEMM_CFG") ? atoi(getenv("GEMM_CFG")) : 75;
    for (int it = 0; it < 6; ++it) { g.B = b[it % NB]; launch_gemm_bf16_cfg(g, cfg, 0); }
    (void)hipEventRecord(e0);
    for (int it = 0; it < 24; ++it) { g.B = b[it % NB]; launch_gemm_bf16_cfg(g, cfg, 0); }
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("cfg %d %dx%dx%d: %.1f us per launch, %.0f TFLOP/s\n", cfg, M, N, K, ms * 1000 / 24, 2.0 * M * N * K / (ms / 24 * 1e-3) / 1e12);
    if (cfg != 75 || !getenv("SHOW_STAMPS")) return 0;
    unsigned long long st[5][8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_gemm_stamps), sizeof(st));
    printf("one K-tile body (slot 0, last pass), cycles:\nwave | MFMA 1..26 + k1 reads | lgkmcnt(0) | barrier 1 | MFMA 27..82 + 14 LDS-DMA | vmcnt wait | barrier 2 | MFMA 83..96 + entry reads | tile\n");
    for (int w = 0; w < 4; ++w)
        printf("  %d  | %12llu %16llu %10llu %18llu %18llu %10llu %18llu          | %llu\n", w, st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2],
               st[w][4] - st[w][3], st[w][5] - st[w][4], st[w][6] - st[w][5], st[w][7] - st[w][6], st[w][7] - st[w][0]);
    return 0;
}
