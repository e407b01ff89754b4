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
// Output: the codes (8 bits: one per byte; 4 bits: two per byte, even element in the low nibble) and the group's scale / bias.
__global__ __launch_bounds__(256) void quant_encode_kernel(const bf16_t* __restrict__ W, long n_groups, int bits, uint8_t* __restrict__ codes,
                                                           bf16_t* __restrict__ scales, bf16_t* __restrict__ biases) {
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
        const bf16_t sb = f32_to_bf16(scale), bb = f32_to_bf16(bias);
        scale = bf16_to_f32(sb);
        bias = bf16_to_f32(bb);
        float q = rintf((w - bias) / scale);
        q = fminf(fmaxf(q, 0.f), n_bins);
        const unsigned qi = (unsigned)q;
        if (bits == 8) {
            codes[g * 64 + lane] = (uint8_t)qi;
        } else {
            const unsigned hi = __shfl_down(qi, 1, 64);
            if ((lane & 1) == 0) codes[g * 32 + (lane >> 1)] = (uint8_t)(qi | (hi << 4));
        }
        if (lane == 0) {
            scales[g] = sb;
            biases[g] = bb;
        }
    }
}

// codes -> bf16 weights, w' = bf16(q * scale + bias): 16 codes (8 bits) or 32 codes (4 bits) = 16 bytes per thread, whole groups only
// (64 elements = 4 or 2 threads). The same arithmetic as the encoder's own reconstruction.
template <int BITS>
__global__ __launch_bounds__(256) void quant_decode_kernel(const uint8_t* __restrict__ codes, const bf16_t* __restrict__ scales,
                                                           const bf16_t* __restrict__ biases, long n_chunks, bf16_t* __restrict__ W) {
    constexpr int PER = BITS == 8 ? 16 : 32;  // elements per 16-byte chunk
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += (long)gridDim.x * 256) {
        const uint4 raw = *(const uint4*)(codes + c * 16);
        const long g = (c * PER) >> 6;
        const float scale = bf16_to_f32(scales[g]), bias = bf16_to_f32(biases[g]);
        const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t out[PER / 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (BITS == 8) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float q0 = (float)((wd[i] >> (16 * j)) & 0xffu), q1 = (float)((wd[i] >> (16 * j + 8)) & 0xffu);
                    out[i * 2 + j] = pack_bf16x2(q0 * scale + bias, q1 * scale + bias);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float q0 = (float)((wd[i] >> (8 * j)) & 0xfu), q1 = (float)((wd[i] >> (8 * j + 4)) & 0xfu);
                    out[i * 4 + j] = pack_bf16x2(q0 * scale + bias, q1 * scale + bias);
                }
            }
        }
        uint4* dst = (uint4*)(W + c * PER);
#pragma unroll
        for (int i = 0; i < PER / 8; ++i) dst[i] = uint4{out[i * 4], out[i * 4 + 1], out[i * 4 + 2], out[i * 4 + 3]};
    }
}

void launch_encode(const bf16_t* W, long out, long in, int bits, uint8_t* codes, bf16_t* scales, bf16_t* biases, hipStream_t stream) {
    const long groups = out * in / 64;
    long grid = (groups + 3) / 4;
    if (grid > 65535 * 4) grid = 65535 * 4;
    hipLaunchKernelGGL(quant_encode_kernel, dim3((unsigned)grid), dim3(256), 0, stream, W, groups, bits, codes, scales, biases);
    HIP_CHECK(hipGetLastError());
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

void launch_dequant(const uint8_t* q, const bf16_t* qs, const bf16_t* qb, long out, long in, int bits, bf16_t* dst, hipStream_t stream) {
    const long n_chunks = out * in * bits / 8 / 16;
    long grid = (n_chunks + 255) / 256;
    if (grid > 16384) grid = 16384;
    if (bits == 8)
        hipLaunchKernelGGL(quant_decode_kernel<8>, dim3((unsigned)grid), dim3(256), 0, stream, q, qs, qb, n_chunks, dst);
    else
        hipLaunchKernelGGL(quant_decode_kernel<4>, dim3((unsigned)grid), dim3(256), 0, stream, q, qs, qb, n_chunks, dst);
    HIP_CHECK(hipGetLastError());
}

const bf16_t* dit_linear_weights(const LinearW& w, hipStream_t stream) {
    if (w.w) return w.w;
    if (!w.q || !w.dq) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: Linear weights (%d x %d)", w.out, w.in);
    // the GEMM that consumes the scratch runs behind this pass on the same stream, and the next pass behind that GEMM
    w.dq->buf.ensure((size_t)w.out * w.in * 2);
    launch_dequant(w.q, w.qs, w.qb, w.out, w.in, w.qbits, w.dq->buf.as<bf16_t>(), stream);
    return w.dq->buf.as<bf16_t>();
}

void dit_quantize(ltx_ctx* ctx, DiTModel* m, int bits, int group) {
    LTX_REQUIRE(bits == 8 || bits == 4, "transformer quantization: %d bits not supported (bf16, qint8, int4)", bits);
    LTX_REQUIRE(group == 64, "transformer quantization: group size %d not supported (64)", group);
    LTX_REQUIRE(m->quant_bits == 16, "the transformer is already quantised (%d bits)", m->quant_bits);
    // every Linear of the model (MLXNN.quantize(model:groupSize:bits:), LTXPipeline.swift:329)
    std::vector<LinearW*> lins = {&m->patchify, &m->ada_l1, &m->ada_l2, &m->ada_lin, &m->cap_l1, &m->cap_l2, &m->proj_out};
    for (auto& b : m->blocks)
        for (LinearW* l : {&b.qk1, &b.v1, &b.o1, &b.q2, &b.k2, &b.v2, &b.o2, &b.ff1, &b.ff2}) lins.push_back(l);
    size_t total = 0, largest = 0;
    for (LinearW* l : lins) {
        LTX_REQUIRE(l->w && l->in % 64 == 0, "quantize: in-features %d not a multiple of 64", l->in);
        const size_t n = (size_t)l->out * l->in;
        total += DeviceArena::padded(n * bits / 8) + 2 * DeviceArena::padded(n / 64 * 2);
        largest = std::max(largest, n);
    }
    m->qarena.reserve(total + 256);
    m->dq.buf.ensure(largest * 2);
    std::map<const bf16_t*, LinearW*> by_w;
    for (LinearW* l : lins) {
        const size_t n = (size_t)l->out * l->in;
        uint8_t* codes = (uint8_t*)m->qarena.take(n * bits / 8);
        bf16_t* qs = (bf16_t*)m->qarena.take(n / 64 * 2);
        bf16_t* qb = (bf16_t*)m->qarena.take(n / 64 * 2);
        launch_encode(l->w, l->out, l->in, bits, codes, qs, qb, ctx->stream);
        l->q = codes;
        l->qs = qs;
        l->qb = qb;
        l->qbits = bits;
        l->dq = &m->dq;
        by_w[l->w] = l;
    }
    // parameter slots of the weights (incl. the to_q / to_k row views of the fused matrix) now name codes / scales / biases
    for (auto& kv : m->slots) {
        ParamSlot& sl = kv.second;
        if (sl.kind != SLOT_BF16) continue;
        auto it = by_w.upper_bound((const bf16_t*)sl.dst);
        LTX_REQUIRE(it != by_w.begin(), "quantize: parameter %s has no Linear", kv.first.c_str());
        --it;
        LinearW* l = it->second;
        const long off = (const bf16_t*)sl.dst - l->w;  // elements from the start of the (fused) matrix: whole rows
        LTX_REQUIRE(off >= 0 && off + sl.numel <= (long)l->out * l->in && off % l->in == 0, "quantize: parameter %s is not a row range", kv.first.c_str());
        sl.q = (uint8_t*)l->q + off * bits / 8;
        sl.qs = (bf16_t*)l->qs + off / 64;
        sl.qb = (bf16_t*)l->qb + off / 64;
        sl.dst = nullptr;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (LinearW* l : lins) l->w = nullptr;
    m->warena.buf.release();   // the bf16 weights are gone: 26 GB -> 13.9 GB (8 bits) / 7.3 GB (4 bits) resident
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
    DevBuf d_up, d_down, d_delta, d_w;
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
        if (slot.q) {
            // quantised model: dequant -> merge -> requant (LoRAAdapter.swift:104-131) through one scratch matrix
            d_w.ensure((size_t)n * 2);
            launch_dequant(slot.q, slot.qs, slot.qb, out, in, m->quant_bits, d_w.as<bf16_t>(), stm);
            hipLaunchKernelGGL(lora_add_kernel, dim3((unsigned)grid), dim3(256), 0, stm, d_w.as<bf16_t>(), d_delta.as<bf16_t>(), eff, n);
            HIP_CHECK(hipGetLastError());
            launch_encode(d_w.as<bf16_t>(), out, in, m->quant_bits, slot.q, slot.qs, slot.qb, stm);
        } else {
            hipLaunchKernelGGL(lora_add_kernel, dim3((unsigned)grid), dim3(256), 0, stm, (bf16_t*)slot.dst, d_delta.as<bf16_t>(), eff, n);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(stm));  // host staging vectors are reused by the next layer
        ++fused;
    }
    for (auto* c : m->ctx_cache) c->version = 0;
    return fused;
}
