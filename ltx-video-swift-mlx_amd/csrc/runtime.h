// runtime.h - context, device memory and safetensors plumbing shared by the DiT / VAE host graphs.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

// RAII device allocation that only grows (activation workspaces are sized on first use and then reused: no
// hipMalloc/hipFree inside a denoise step).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // returns true when (re)allocated
    bool ensure(size_t n, bool zero = false) {
        if (n <= bytes && p) return false;
        release();
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
        if (zero) HIP_CHECK(hipMemset(p, 0, n));
        return true;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

// One big allocation carved into 256-B aligned tensors (weights live here for the lifetime of a model).
struct DeviceArena {
    DevBuf buf;
    size_t used = 0;
    std::vector<size_t> pending;  // sizes requested before commit
    void reserve(size_t total) {
        buf.ensure(total);
        used = 0;
    }
    void* take(size_t n) {
        const size_t off = (used + 255) & ~(size_t)255;
        if (off + n > buf.bytes) LTX_THROW(LTXS_INSUFFICIENT_MEMORY, "arena overflow: need %zu, have %zu", off + n, buf.bytes);
        used = off + n;
        return (char*)buf.p + off;
    }
    static size_t padded(size_t n) { return (n + 255) & ~(size_t)255; }
};

// ---- safetensors (header JSON + mmap) ----
struct StTensor {
    std::string dtype;  // "BF16", "F32", "F16", "I32", ...
    std::vector<long> shape;
    size_t begin = 0, end = 0;  // byte offsets into the data section
    long numel() const {
        long n = 1;
        for (long s : shape) n *= s;
        return n;
    }
};
struct SafeTensors {
    std::map<std::string, StTensor> tensors;
    const uint8_t* data = nullptr;  // start of the data section inside the mapping
    void* map_base = nullptr;
    size_t map_len = 0;
    int fd = -1;
    SafeTensors() = default;
    SafeTensors(const SafeTensors&) = delete;
    ~SafeTensors() { close(); }
    void open(const std::string& path);  // throws LtxError (fileNotFound / weightLoadingFailed)
    void close();
    const uint8_t* ptr(const StTensor& t) const { return data + t.begin; }
};
// Convert `n` elements of a safetensors tensor to bf16 bits / f32 on the host.
void st_to_bf16(const SafeTensors& st, const StTensor& t, bf16_t* out);
void st_to_f32(const SafeTensors& st, const StTensor& t, float* out);

struct DiTModel;
struct VaeModel;
struct ConnectorModel;
struct VaeEncoderModel;
struct UpscalerModel;
struct DistState;

// ---- live per-kernel timing (HIP events recorded on the launch stream around each launch of a kernel family) ----
// Used by bench.py for the roofline numbers: achieved = algorithmic work of the launches / their summed duration.
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_CONV = 2, PROF_ELEM = 3, PROF_NKINDS = 4 };
struct ProfRec {
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;
    double work = 0;  // algorithmic FLOPs (GEMM/attention/conv) or bytes (elementwise)
};
struct Profiler {
    bool on = false;
    std::vector<ProfRec> pool;
    size_t used = 0;
    double total_ms[PROF_NKINDS] = {0, 0, 0, 0};
    double total_work[PROF_NKINDS] = {0, 0, 0, 0};
    long launches[PROF_NKINDS] = {0, 0, 0, 0};
    ProfRec* begin(int kind, double work, hipStream_t s);
    void end(ProfRec* r, hipStream_t s);
    void collect();  // call after the stream is idle
    void reset();
    ~Profiler();
};
Profiler* prof_current();
void prof_set_current(Profiler* p);
struct ProfScope {
    ProfRec* r = nullptr;
    hipStream_t s;
    ProfScope(int kind, double work, hipStream_t stream) : s(stream) {
        Profiler* p = prof_current();
        if (p && p->on) r = p->begin(kind, work, stream);
    }
    ~ProfScope() {
        if (r) prof_current()->end(r, s);
    }
};

struct ltx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    DiTModel* dit = nullptr;
    VaeModel* vae = nullptr;
    UpscalerModel* upscaler = nullptr;
    ConnectorModel* connector = nullptr;
    VaeEncoderModel* vae_encoder = nullptr;
    DistState* dist = nullptr;  // multi-GPU state (dist.h); null = single GPU
    // staging buffers for host-pointer entry points
    DevBuf h2d[8];
    // load report of the last *_load call (mirrors the reference's "unmatched/missing" debug logs)
    int n_loaded = 0, n_missing = 0, n_unmatched = 0;
    Profiler prof;
    // denoise-loop scratch (device)
    DevBuf dn_tokens, dn_vel_tok, dn_vel, dn_vel2, dn_vel3, dn_prev, dn_ts, dn_stats, dn_lat2, dn_step_stats;
    DevBuf dn_rowmap;  // I2V token -> timestep-group map
    DevBuf dn_vel_slice;  // sequence-sharded loop: this rank's [T/N][C] velocity rows before the all-gather
    DevBuf i2v_cond, i2v_noise;  // staged image latent / re-noising draws of the host-pointer denoise entry
    DevBuf op_ws;  // split-K workspace of the kernel-level test hook
};

// synthetic weights (bench / property tests): counter-based normal fill on device
void launch_fill_normal_bf16(bf16_t* p, long n, uint64_t seed, float mean, float stddev, hipStream_t stream);
void launch_fill_normal_f32(float* p, long n, uint64_t seed, float mean, float stddev, int round_bf16, hipStream_t stream);
void launch_fill_const_f32(float* p, long n, float v, hipStream_t stream);
