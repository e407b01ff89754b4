// profiler.cpp - HIP-event timing of kernel families on the launch stream (see runtime.h).
#include "runtime.h"

namespace {
thread_local Profiler* g_prof = nullptr;
}

Profiler* prof_current() { return g_prof; }
void prof_set_current(Profiler* p) { g_prof = p; }

ProfRec* Profiler::begin(int kind, double work, hipStream_t s) {
    if (used == pool.size()) {
        ProfRec r;
        HIP_CHECK(hipEventCreate(&r.a));
        HIP_CHECK(hipEventCreate(&r.b));
        pool.push_back(r);
    }
    ProfRec* r = &pool[used++];
    r->kind = kind;
    r->work = work;
    HIP_CHECK(hipEventRecord(r->a, s));
    return r;
}

void Profiler::end(ProfRec* r, hipStream_t s) { (void)hipEventRecord(r->b, s); }

void Profiler::collect() {
    for (size_t i = 0; i < used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pool[i].a, pool[i].b) == hipSuccess) {
            total_ms[pool[i].kind] += ms;
            total_work[pool[i].kind] += pool[i].work;
            launches[pool[i].kind] += 1;
        }
    }
    used = 0;
}

void Profiler::reset() {
    used = 0;
    for (int k = 0; k < PROF_NKINDS; ++k) {
        total_ms[k] = 0;
        total_work[k] = 0;
        launches[k] = 0;
    }
}

Profiler::~Profiler() {
    for (auto& r : pool) {
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
}
