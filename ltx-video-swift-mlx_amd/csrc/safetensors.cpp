// safetensors.cpp - minimal safetensors reader: 8-byte little-endian header length, JSON header, raw data section.
// The file is mmap'ed (the reference loads through MLX `loadArrays`, also mmap-backed: ModelDownloader.swift:609)
// and tensors are converted / uploaded straight from the mapping, so a 40 GB unified file never needs 40 GB of
// anonymous host memory.
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "runtime.h"

namespace {

struct JsonCursor {
    const char* p;
    const char* end;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    void expect(char c) {
        if (!eat(c)) LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "safetensors header: expected '%c'", c);
    }
    std::string str() {
        ws();
        if (p >= end || *p != '"') LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "safetensors header: expected string");
        ++p;
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': s.push_back('\n'); break;
                    case 't': s.push_back('\t'); break;
                    case 'r': s.push_back('\r'); break;
                    case 'b': s.push_back('\b'); break;
                    case 'f': s.push_back('\f'); break;
                    case 'u':
                        // keep the escape verbatim; tensor names are ASCII in practice
                        s += "\\u";
                        break;
                    default: s.push_back(*p); break;
                }
                ++p;
            } else {
                s.push_back(*p++);
            }
        }
        if (p >= end) LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "safetensors header: unterminated string");
        ++p;
        return s;
    }
    long num() {
        ws();
        char* e = nullptr;
        const double v = strtod(p, &e);
        if (e == p) LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "safetensors header: expected number");
        p = e;
        return (long)v;
    }
    void skip_value() {
        ws();
        if (p >= end) return;
        if (*p == '"') {
            (void)str();
        } else if (*p == '{') {
            ++p;
            if (eat('}')) return;
            do {
                (void)str();
                expect(':');
                skip_value();
            } while (eat(','));
            expect('}');
        } else if (*p == '[') {
            ++p;
            if (eat(']')) return;
            do {
                skip_value();
            } while (eat(','));
            expect(']');
        } else {
            while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
        }
    }
};

}  // namespace

void SafeTensors::open(const std::string& path) {
    close();
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) LTX_THROW(LTXS_FILE_NOT_FOUND, "File not found: %s", path.c_str());
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 8) {
        close();
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: %s is not a safetensors file", path.c_str());
    }
    map_len = (size_t)sb.st_size;
    map_base = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map_base == MAP_FAILED) {
        map_base = nullptr;
        close();
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: mmap(%s) failed", path.c_str());
    }
    const uint8_t* base = (const uint8_t*)map_base;
    uint64_t hlen = 0;
    memcpy(&hlen, base, 8);
    if (hlen > map_len - 8) {
        close();
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: bad safetensors header length in %s", path.c_str());
    }
    data = base + 8 + hlen;
    const size_t data_len = map_len - 8 - hlen;
    JsonCursor c{(const char*)base + 8, (const char*)base + 8 + hlen};
    c.expect('{');
    if (!c.eat('}')) {
        do {
            const std::string name = c.str();
            c.expect(':');
            if (name == "__metadata__") {
                c.skip_value();
                continue;
            }
            StTensor t;
            c.expect('{');
            do {
                const std::string field = c.str();
                c.expect(':');
                if (field == "dtype") {
                    t.dtype = c.str();
                } else if (field == "shape") {
                    c.expect('[');
                    if (!c.eat(']')) {
                        do {
                            t.shape.push_back(c.num());
                        } while (c.eat(','));
                        c.expect(']');
                    }
                } else if (field == "data_offsets") {
                    c.expect('[');
                    t.begin = (size_t)c.num();
                    c.expect(',');
                    t.end = (size_t)c.num();
                    c.expect(']');
                } else {
                    c.skip_value();
                }
            } while (c.eat(','));
            c.expect('}');
            if (t.end < t.begin || t.end > data_len)
                LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: tensor %s has offsets outside %s", name.c_str(), path.c_str());
            tensors.emplace(name, std::move(t));
        } while (c.eat(','));
        c.expect('}');
    }
}

void SafeTensors::close() {
    if (map_base) munmap(map_base, map_len);
    map_base = nullptr;
    map_len = 0;
    data = nullptr;
    if (fd >= 0) ::close(fd);
    fd = -1;
    tensors.clear();
}

static size_t dtype_size(const std::string& d) {
    if (d == "BF16" || d == "F16" || d == "I16" || d == "U16") return 2;
    if (d == "F32" || d == "I32" || d == "U32") return 4;
    if (d == "F64" || d == "I64" || d == "U64") return 8;
    if (d == "I8" || d == "U8" || d == "BOOL" || d == "F8_E4M3" || d == "F8_E5M2") return 1;
    return 0;
}

void st_to_bf16(const SafeTensors& st, const StTensor& t, bf16_t* out) {
    const long n = t.numel();
    const uint8_t* src = st.ptr(t);
    if ((size_t)n * dtype_size(t.dtype) != t.end - t.begin)
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: tensor byte size does not match shape (dtype %s)", t.dtype.c_str());
    if (t.dtype == "BF16") {
        memcpy(out, src, (size_t)n * 2);
    } else if (t.dtype == "F32") {
        const float* f = (const float*)src;
        for (long i = 0; i < n; ++i) out[i] = host_f32_to_bf16(f[i]);
    } else if (t.dtype == "F16") {
        const uint16_t* h = (const uint16_t*)src;
        for (long i = 0; i < n; ++i) out[i] = host_f32_to_bf16(host_f16_to_f32(h[i]));
    } else if (t.dtype == "F64") {
        const double* f = (const double*)src;
        for (long i = 0; i < n; ++i) out[i] = host_f32_to_bf16((float)f[i]);
    } else {
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: unsupported dtype %s", t.dtype.c_str());
    }
}

void st_to_f32(const SafeTensors& st, const StTensor& t, float* out) {
    const long n = t.numel();
    const uint8_t* src = st.ptr(t);
    if ((size_t)n * dtype_size(t.dtype) != t.end - t.begin)
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: tensor byte size does not match shape (dtype %s)", t.dtype.c_str());
    if (t.dtype == "F32") {
        memcpy(out, src, (size_t)n * 4);
    } else if (t.dtype == "BF16") {
        const bf16_t* b = (const bf16_t*)src;
        for (long i = 0; i < n; ++i) out[i] = host_bf16_to_f32(b[i]);
    } else if (t.dtype == "F16") {
        const uint16_t* h = (const uint16_t*)src;
        for (long i = 0; i < n; ++i) out[i] = host_f16_to_f32(h[i]);
    } else if (t.dtype == "F64") {
        const double* f = (const double*)src;
        for (long i = 0; i < n; ++i) out[i] = (float)f[i];
    } else {
        LTX_THROW(LTXS_WEIGHT_LOADING_FAILED, "Failed to load weights: unsupported dtype %s", t.dtype.c_str());
    }
}
