// lora_quant.hip - load-time weight transforms of the DiT: LoRA fusion (R22) and affine group quantisation (R21).
#include <string.h>

#include <set>

#include "dit.h"
#include "gemm.h"
#include "hostmath.h"

namespace {

// W = bf16( f32(W) + f32( bf16( f32(delta) * s ) ) ): the reference multiplies the bf16 matmul result by the
// effective scale (bf16 array x scalar -> bf16), casts to the weight dtype and adds in bf16
// (LoRALoader.swift:175; LoRAAdapter.swift:145-147).
__global__ void lora_add_kernel(bf16_t* __restrict__ W, const bf16_t* __restrict__ delta, float s, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float d = bf16_to_f32(f32_to_bf16(bf16_to_f32(delta[i]) * s));
        W[i] = f32_to_bf16(bf16_to_f32(W[i]) + d);
    }
}

// One wave per 64-element group (lane = element). MLX affine quantisation (mlx `quantize`, mode affine), as
// documented upstream - the exact rounding of the third-party library cannot be verified in this environment
// (DESIGN.md section 2):
//   w_max, w_min over the group; scale = max((w_max-w_min)/(2^b-1), 1e-7); mask = |w_min| > |w_max|
//   scale = mask ? scale : -scale; edge = mask ? w_min : w_max; q0 = round(edge/scale)
//   scale = q0 != 0 ? edge/q0 : scale; bias = q0 == 0 ? 0 : edge
//   q = clip(round((w - bias)/scale), 0, 2^b-1);  w' = q*scale + bias   (scale, bias stored in bf16)
__global__ __launch_bounds__(256) void fake_quant_kernel(bf16_t* __restrict__ W, long n_groups, int bits) {
    const int lane = threadIdx.x & 63;
    long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long gstride = (long)gridDim.x * 4;
    const float n_bins = (float)((1 << bits) - 1);
    for (; g < n_groups; g += gstride) {
        const float w = bf16_to_f32(W[g * 64 + lane]);
        const float w_max = wave_reduce_max(w);
        const float w_min = -wave_reduce_max(-w);
        const bool mask = fabsf(w_min) > fabsf(w_max);
        float scale = fmaxf((w_max - w_min) / n_bins, 1e-7f);
        scale = mask ? scale : -scale;
        const float edge = mask ? w_min : w_max;
        const float q0 = rintf(edge / scale);
        scale = (q0 != 0.f) ? edge / q0 : scale;
        float bias = (q0 == 0.f) ? 0.f : edge;
        scale = bf16_to_f32(f32_to_bf16(scale));
        bias = bf16_to_f32(f32_to_bf16(bias));
        float q = rintf((w - bias) / scale);
        q = fminf(fmaxf(q, 0.f), n_bins);
        W[g * 64 + lane] = f32_to_bf16(q * scale + bias);
    }
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
void replace_first(std::string& s, const char* from, const char* to) {
    const size_t p = s.find(from);
    if (p != std::string::npos) s.replace(p, strlen(from), to);
}

}  // namespace

void dit_quantize(ltx_ctx* ctx, DiTModel* m, int bits, int group) {
    LTX_REQUIRE(bits == 8 || bits == 4, "transformer quantization: %d bits not supported (bf16, qint8, int4)", bits);
    LTX_REQUIRE(group == 64, "transformer quantization: group size %d not supported (64)", group);
    // every Linear of the model (MLXNN.quantize(model:groupSize:bits:), LTXPipeline.swift:329)
    std::set<const void*> done;
    auto q = [&](const LinearW& l) {
        if (!l.w || done.count(l.w)) return;
        done.insert(l.w);
        LTX_REQUIRE(l.in % 64 == 0, "quantize: in-features %d not a multiple of 64", l.in);
        const long groups = (long)l.out * l.in / 64;
        long grid = (groups + 3) / 4;
        if (grid > 65535 * 4) grid = 65535 * 4;
        hipLaunchKernelGGL(fake_quant_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, l.w, groups, bits);
        HIP_CHECK(hipGetLastError());
    };
    q(m->patchify); q(m->ada_l1); q(m->ada_l2); q(m->ada_lin); q(m->cap_l1); q(m->cap_l2); q(m->proj_out);
    for (auto& b : m->blocks) {
        q(b.qk1); q(b.v1); q(b.o1); q(b.q2); q(b.k2); q(b.v2); q(b.o2); q(b.ff1); q(b.ff2);
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    m->quant_bits = bits;
    for (auto* c : m->ctx_cache) c->version = 0;
}

int dit_fuse_lora(ltx_ctx* ctx, DiTModel* m, const std::string& path, float scale) {
    SafeTensors st;
    try {
        st.open(path);
    } catch (const LtxError& e) {
        if (e.code == LTXS_FILE_NOT_FOUND) throw;
        throw LtxError{LTXS_INVALID_LORA, "Invalid LoRA: " + e.msg};
    }
    hipStream_t stm = ctx->stream;
    int fused = 0;
    DevBuf d_up, d_down, d_delta;
    std::vector<bf16_t> h_up, h_down, h_downT;
    for (auto& kv : st.tensors) {
        const std::string& key = kv.first;
        // parseLoRALayers (LoRALoader.swift:63-111): the down / A matrix identifies a layer
        const bool is_down = key.find("lora_down") != std::string::npos;
        const bool is_a = key.find("lora_A") != std::string::npos;
        if (!is_down && !is_a) continue;
        std::string up_key = key, base = key;
        if (is_down) {
            replace_first(up_key, "lora_down", "lora_up");
            replace_first(base, ".lora_down.weight", "");
            replace_first(base, ".lora_down", "");
        } else {
            replace_first(up_key, "lora_A", "lora_B");
            replace_first(base, ".lora_A.weight", "");
            replace_first(base, ".lora_A", "");
        }
        auto upit = st.tensors.find(up_key);
        if (upit == st.tensors.end()) continue;
        const StTensor& down = kv.second;
        const StTensor& up = upit->second;
        if (down.shape.size() != 2 || up.shape.size() != 2)
            LTX_THROW(LTXS_INVALID_LORA, "Invalid LoRA: %s is not a matrix", key.c_str());
        const int rank = (int)down.shape[0], in = (int)down.shape[1], out = (int)up.shape[0];
        if ((int)up.shape[1] != rank) LTX_THROW(LTXS_INVALID_LORA, "Invalid LoRA: rank mismatch for %s", base.c_str());
        float eff = scale;  // scale * (alpha / rank) when alpha is present (LoRAConfig.swift:84-89)
        auto ait = st.tensors.find(base + ".alpha");
        if (ait != st.tensors.end() && ait->second.numel() == 1) {
            float alpha = 0.f;
            st_to_f32(st, ait->second, &alpha);
            eff = scale * (alpha / (float)rank);
        }
        std::string mk;
        map_lora_key(base, &mk);
        auto sit = m->slots.find(mk);
        if (sit == m->slots.end()) continue;  // "no model weight for ..., skipping" (LoRAAdapter.swift:136-139)
        ParamSlot& slot = sit->second;
        if (slot.kind != SLOT_BF16 || slot.rows != out || slot.cols != in)
            LTX_THROW(LTXS_INVALID_LORA, "Invalid LoRA: %s is [%d,%d] but the layer is [%ld,%ld]", base.c_str(), out, in, slot.rows, slot.cols);
        // delta = up @ down as a K-contiguous GEMM: A = up [out][rank_pad], B = down^T [in][rank_pad]
        const int rp = ((rank + 63) / 64) * 64;
        h_up.assign((size_t)out * rank, 0);
        h_down.assign((size_t)rank * in, 0);
        st_to_bf16(st, up, h_up.data());
        st_to_bf16(st, down, h_down.data());
        std::vector<bf16_t> h_upP((size_t)out * rp, 0);
        for (int o = 0; o < out; ++o) memcpy(&h_upP[(size_t)o * rp], &h_up[(size_t)o * rank], (size_t)rank * 2);
        h_downT.assign((size_t)in * rp, 0);
        for (int r = 0; r < rank; ++r)
            for (int i = 0; i < in; ++i) h_downT[(size_t)i * rp + r] = h_down[(size_t)r * in + i];
        d_up.ensure(h_upP.size() * 2);
        d_down.ensure(h_downT.size() * 2);
        d_delta.ensure((size_t)out * in * 2);
        HIP_CHECK(hipMemcpyAsync(d_up.p, h_upP.data(), h_upP.size() * 2, hipMemcpyHostToDevice, stm));
        HIP_CHECK(hipMemcpyAsync(d_down.p, h_downT.data(), h_downT.size() * 2, hipMemcpyHostToDevice, stm));
        GemmArgs g;
        g.A = d_up.as<bf16_t>(); g.lda = rp;
        g.B = d_down.as<bf16_t>(); g.ldb = rp;
        g.M = out; g.N = in; g.K = rp;
        g.ep.out_bf16 = d_delta.as<bf16_t>(); g.ep.ld_bf16 = in;
        launch_gemm_bf16(g, stm);
        const long n = (long)out * in;
        long grid = (n + 255) / 256;
        if (grid > 65535) grid = 65535;
        hipLaunchKernelGGL(lora_add_kernel, dim3((unsigned)grid), dim3(256), 0, stm, (bf16_t*)slot.dst, d_delta.as<bf16_t>(), eff, n);
        HIP_CHECK(hipGetLastError());
        if (m->quant_bits == 8 || m->quant_bits == 4) {
            // dequant -> merge -> requant (LoRAAdapter.swift:104-131)
            const long groups = n / 64;
            long qg = (groups + 3) / 4;
            if (qg > 65535 * 4) qg = 65535 * 4;
            hipLaunchKernelGGL(fake_quant_kernel, dim3((unsigned)qg), dim3(256), 0, stm, (bf16_t*)slot.dst, groups, m->quant_bits);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(stm));  // host staging vectors are reused by the next layer
        ++fused;
    }
    for (auto* c : m->ctx_cache) c->version = 0;
    return fused;
}
