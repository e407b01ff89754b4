// conv_stamps.hip - where a tile of the halo-staged conv kernel spends its time: wall-clock stamps (100 MHz) of EVERY workgroup of one
// launch (entry, first operands landed, K loop done, epilogue stores drained) and the CU it ran on, so that the gap between two
// consecutive workgroups of one CU (dispatch of the next tile) is visible too. The launch is the VAE decoder's 128-channel stage:
// 128 -> 128 channels at 25x128x192 (3200 tiles = 12.5 rounds), outer conv of a res-block (f32 stream read-modify-write + fused
// PixelNorm / SiLU output). Build from the repo root:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCONV_HALO_STAMPS -Iinclude -Iltx-video-swift-mlx_amd/csrc -o tools/ubench/conv_stamps tools/ubench/conv_stamps.hip
#include "../../ltx-video-swift-mlx_amd/csrc/gemm.hip"
#include "../../ltx-video-swift-mlx_amd/csrc/options.cpp"   // the launchers' switch table (defaults; nothing here moves it)

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

Profiler* prof_current() { return nullptr; }
ProfRec* Profiler::begin(int, double, hipStream_t) { return nullptr; }
void Profiler::end(ProfRec*, hipStream_t) {}
// referenced by launch_gemm_bf16's norm-after form, which this harness never calls
void launch_norm_mod(const float*, long, const float*, const float*, long, int, bf16_t*, long, int, int, int, float, int, hipStream_t, const int*) { abort(); }

int main() {
    const int F = getenv("CONV_F") ? atoi(getenv("CONV_F")) : 25, H = getenv("CONV_H") ? atoi(getenv("CONV_H")) : 128, W = getenv("CONV_W") ? atoi(getenv("CONV_W")) : 192, C = getenv("CONV_C") ? atoi(getenv("CONV_C")) : 128;
    const int N = getenv("CONV_N") ? atoi(getenv("CONV_N")) : C;
    const long P = (long)F * H * W;
    std::vector<bf16_t> hx((size_t)P * C), hw((size_t)N * 27 * C);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = host_f32_to_bf16((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = host_f32_to_bf16(((float)((i * 40503u >> 12) & 255) / 256.f - 0.5f) * 0.05f);
    bf16_t *x, *w, *pn;
    float *stream, *vec;
    (void)hipMalloc(&x, hx.size() * 2); (void)hipMalloc(&w, hw.size() * 2); (void)hipMalloc(&pn, (size_t)P * N * 2);
    (void)hipMalloc(&stream, (size_t)P * (N > C ? N : C) * 4); (void)hipMalloc(&vec, 4 * 4096 * 4);
    (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(stream, 0, (size_t)P * (N > C ? N : C) * 4);
    std::vector<float> hv(4 * 4096, 1.0f);
    (void)hipMemcpy(vec, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
    GemmArgs g;
    g.A = x; g.B = w; g.ldb = 27L * C; g.M = (int)P; g.N = N; g.K = 27 * C; g.conv = 1;
    g.geom.F = F; g.geom.H = H; g.geom.W = W; g.geom.C = C; g.geom.pad_mode = 0;
    g.geom.blk_rg = H * W / 192;
    g.ep.bias_n = vec;
    if (N == 128 && !getenv("CONV_PLAIN")) {
        g.ep.out_f32 = stream; g.ep.ld_f32 = N; g.ep.resid = 1; g.ep.gate_scalar = 1.0f;
        g.ep.pn_out = pn; g.ep.ld_pn = N; g.ep.pn_scale = vec + 1024; g.ep.pn_shift = vec + 2048;
    } else {
        g.ep.out_f32 = stream; g.ep.ld_f32 = N;
    }
    float* d2s_out = nullptr;
    if (getenv("CONV_D2S")) {  // the upsampler conv: N = 4 C channels, depth-to-space store with first-frame drop + tiled D2S residual of the input stream
        (void)hipMalloc(&d2s_out, (size_t)(2 * F - 1) * 2 * H * 2 * W * (C / 2) * 4);
        g.ep = GemmEpilogue{};
        g.ep.bias_n = vec;
        g.ep.out_f32 = d2s_out; g.ep.ld_f32 = C / 2; g.ep.d2s = 1; g.ep.resid_src = stream; g.ep.ld_resid = C;
        g.geom.blk_rg = 0;
    }
    if (getenv("CONV_STAGGER")) ltx_opt_set(ltx_opt_find("conv_stagger"), atoi(getenv("CONV_STAGGER")));
    if (getenv("CONV_TALL")) ltx_opt_set(ltx_opt_find("conv_tall"), atoi(getenv("CONV_TALL")));
    const int cfg = N <= 64 ? 27 : 21;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) launch_gemm_bf16_cfg(g, cfg, 0);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int it = 0; it < reps; ++it) launch_gemm_bf16_cfg(g, cfg, 0);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const bool tall = conv_halo2_takes(g) != 0;  // (option conv_tall, CONV_TALL=0 turns it off): 384-row tiles, stamps of the main launch only
    const int tiles = tall ? (int)(P / 384) * (N / 128) / 256 * 256 : (int)((P + 191) / 192) * ((N + 127) / 128);
    if (tiles > 16383) printf("(more tiles than stamp slots: statistics over the first 16383)\n");
    printf("conv %d -> %d at %dx%dx%d: %d tiles, %.1f us per launch, %.0f TFLOP/s\n", C, N, F, H, W, tiles, ms * 1e3 / reps, 2.0 * P * N * 27 * C / (ms / reps * 1e-3) / 1e12);
    static unsigned long long st[16384][10];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_conv_stamps), sizeof(st));
    const int n = tiles < 16384 ? tiles : 16383;
    unsigned long long tmin = ~0ull, tmax = 0;
    double pro = 0, loop = 0, epi = 0, drain = 0, tset = 0, thook0 = 0, thook = 0, cyc = 0;
    std::map<unsigned long long, std::vector<int>> by_cu;
    for (int i = 0; i < n; ++i) {
        tmin = std::min(tmin, st[i][0]); tmax = std::max(tmax, st[i][3]);
        pro += (double)(st[i][1] - st[i][0]); loop += (double)(st[i][2] - st[i][1]); epi += (double)(st[i][3] - st[i][2]); drain += (double)(st[i][3] - st[i][5]); tset += (double)(st[i][6] - st[i][2]); thook0 += (double)(st[i][7] - st[i][6]); thook += (double)(st[i][8] - st[i][7]); cyc += (double)st[i][9];
        by_cu[st[i][4]].push_back(i);
    }
    printf("kernel span by stamps: %.1f us; per tile (mean of %d): prologue %.2f us | K loop %.2f us | epilogue (stores drained) %.2f us, of which the final vmcnt(0) %.2f us | sum %.2f us\n",
           (tmax - tmin) * 0.01, n, pro / n * 0.01, loop / n * 0.01, epi / n * 0.01, drain / n * 0.01, (pro + loop + epi) / n * 0.01);
    printf("  K loop: %.0f shader cycles per tile (s_memtime) over %.2f us wall = %.3f GHz in the loop; %.0f cycles per K-tile of %d\n", cyc / n, loop / n * 0.01,
           cyc / n / (loop / n * 10.0), cyc / n / (27.0 * C / 64.0), 27 * C / 64);
    if (tall) printf("  (tall kernel: 384 x 128 tiles, a K-tile is 48 MFMAs per wave = 1536 cycles of MFMA issue per SIMD; the 192-row tail window of the launch is not in these statistics)\n");
    printf("  inside the epilogue interval: next tile's addresses %.2f us | epilogue up to the request of the next tile %.2f us | issuing that request %.2f us | rest %.2f us\n",
           tset / n * 0.01, thook0 / n * 0.01, thook / n * 0.01, (epi - tset - thook0 - thook) / n * 0.01);
    double gap = 0; long ngap = 0; double gmax = 0;
    std::vector<double> first_start, last_end;
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return st[a][0] < st[b][0]; });
        for (size_t j = 1; j < v.size(); ++j) {
            const double d = ((double)st[v[j]][0] - (double)st[v[j - 1]][3]) * 0.01;
            gap += d; ++ngap; gmax = std::max(gmax, d);
        }
        first_start.push_back((st[v[0]][0] - tmin) * 0.01);
        last_end.push_back((tmax - st[v.back()][3]) * 0.01);
    }
    std::sort(first_start.begin(), first_start.end()); std::sort(last_end.begin(), last_end.end());
    printf("CUs seen: %zu; tiles per CU %.2f; gap between a CU's consecutive tiles (previous stores drained -> next entry): mean %.2f us, max %.2f us\n",
           by_cu.size(), (double)n / by_cu.size(), ngap ? gap / ngap : 0.0, gmax);
    printf("first entry after kernel start: median %.2f us, max %.2f us; idle at the end (last tile of the CU -> kernel end): median %.2f us, max %.2f us\n",
           first_start[first_start.size() / 2], first_start.back(), last_end[last_end.size() / 2], last_end.back());
    // round structure: how synchronised are the CUs? spread of the k-th tile's entry over the CUs
    for (int k : {1, 4, 8, 11}) {
        std::vector<double> e;
        for (auto& kv : by_cu) if ((int)kv.second.size() > k) e.push_back((st[kv.second[k]][0] - tmin) * 0.01);
        if (e.empty()) continue;
        std::sort(e.begin(), e.end());
        printf("  entry of a CU's tile #%d: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us\n", k, e.front(), e[e.size() / 10], e[e.size() / 2], e[e.size() * 9 / 10], e.back());
    }
    return 0;
}
