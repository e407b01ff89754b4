// dist.cpp - see dist.h.
#include "dist.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

namespace {

// librccl is resolved on first use: a single-GPU caller never loads it (it is a large library with its own device code),
// and libltxhip.so carries no load-time dependency on it.
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("cannot load librccl: ") + (e ? e : "unknown error");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl has no symbol ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!r.error.empty()) LTX_THROW(LTXS_HIP_ERROR, "%s", r.error.c_str());
    return r;
}

#define RCCL_CHECK(expr)                                                                                       \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) LTX_THROW(LTXS_HIP_ERROR, "RCCL error %s at %s:%d (%s)", rccl().GetErrorString(_r), __FILE__, __LINE__, #expr); \
    } while (0)

void reset(ltx_ctx* ctx) {
    if (!ctx->dist) return;
    DistState* d = ctx->dist;
    if (d->side) (void)hipStreamSynchronize(d->side);
    if (d->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)rccl().CommDestroy((ncclComm_t)d->comm);
    }
    for (hipEvent_t e : d->ev)
        if (e) (void)hipEventDestroy(e);
    if (d->side) (void)hipStreamDestroy(d->side);
    delete d;
    ctx->dist = nullptr;
}

}  // namespace

void dist_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ltx_dist_unique_id hands out 128 bytes");
    ncclUniqueId id;
    RCCL_CHECK(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
}

void dist_init_native(ltx_ctx* ctx, int rank, int world, const void* id128) {
    LTX_REQUIRE(world >= 1 && rank >= 0 && rank < world && id128, "ltx_dist_init: bad rank %d of %d", rank, world);
    reset(ctx);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCL_CHECK(rccl().CommInitRank(&comm, world, id, rank));  // the device is the context's (hipSetDevice done by the ABI guard)
    ctx->dist = new DistState();
    ctx->dist->rank = rank;
    ctx->dist->world = world;
    ctx->dist->comm = comm;
}

void dist_set_transport(ltx_ctx* ctx, int rank, int world, ltx_dist_gather_fn gather, void* user) {
    LTX_REQUIRE(world >= 1 && rank >= 0 && rank < world && (world == 1 || gather), "ltx_dist_set_transport: bad rank %d of %d", rank, world);
    reset(ctx);
    ctx->dist = new DistState();
    ctx->dist->rank = rank;
    ctx->dist->world = world;
    ctx->dist->cb = gather;
    ctx->dist->cb_user = user;
}

void dist_shutdown(ltx_ctx* ctx) { reset(ctx); }

bool dist_can_overlap(const ltx_ctx* ctx) { return ctx->dist && ctx->dist->comm; }

hipStream_t dist_side_stream(ltx_ctx* ctx) {
    DistState* d = ctx->dist;
    LTX_REQUIRE(d, "multi-GPU call without ltx_dist_init / ltx_dist_set_transport on this context");
    if (!d->side) {
        HIP_CHECK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
        for (hipEvent_t& e : d->ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return d->side;
}

void dist_fork(ltx_ctx* ctx, int ev) {
    hipStream_t side = dist_side_stream(ctx);
    HIP_CHECK(hipEventRecord(ctx->dist->ev[ev], ctx->stream));
    HIP_CHECK(hipStreamWaitEvent(side, ctx->dist->ev[ev], 0));
}

void dist_join(ltx_ctx* ctx, int ev) {
    hipStream_t side = dist_side_stream(ctx);
    HIP_CHECK(hipEventRecord(ctx->dist->ev[ev], side));
    HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->dist->ev[ev], 0));
}

void dist_allgather(ltx_ctx* ctx, const void* send, void* recv, long bytes) { dist_allgather_on(ctx, send, recv, bytes, ctx->stream); }

void dist_allgather_on(ltx_ctx* ctx, const void* send, void* recv, long bytes, hipStream_t stream) {
    DistState* d = ctx->dist;
    LTX_REQUIRE(d, "multi-GPU call without ltx_dist_init / ltx_dist_set_transport on this context");
    LTX_REQUIRE(send && recv && bytes > 0, "dist_allgather: bad arguments");
    LTX_REQUIRE(stream == ctx->stream || d->comm, "dist_allgather: only the native transport can run beside the context's stream");
    d->n_collectives++;
    if (d->comm) {
        RCCL_CHECK(rccl().AllGather(send, recv, (size_t)bytes, ncclUint8, (ncclComm_t)d->comm, stream));
    } else if (d->cb) {
        // a failing host transport must stop the call: carrying on would hand unfilled buffers to the next kernel while the peer
        // rank waits inside its collective
        const int rc = d->cb(d->cb_user, send, recv, bytes);
        if (rc != 0) LTX_THROW(LTXS_GENERATION_FAILED, "all-gather transport failed on rank %d of %d (status %d)", d->rank, d->world, rc);
    } else {
        LTX_REQUIRE(d->world == 1, "dist_allgather: no transport");
        if (recv != send) HIP_CHECK(hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
}

void dist_broadcast(ltx_ctx* ctx, void* buf, long bytes, int root) {
    DistState* d = ctx->dist;
    LTX_REQUIRE(d, "multi-GPU call without ltx_dist_init / ltx_dist_set_transport on this context");
    LTX_REQUIRE(buf && bytes > 0 && root >= 0 && root < d->world, "dist_broadcast: bad arguments (root %d of %d)", root, d->world);
    if (d->world == 1) return;
    if (d->comm) {
        d->n_collectives++;
        RCCL_CHECK(rccl().Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, (ncclComm_t)d->comm, ctx->stream));
        return;
    }
    d->stage.ensure((size_t)bytes * d->world);
    dist_allgather(ctx, buf, d->stage.p, bytes);
    if (d->rank != root)
        HIP_CHECK(hipMemcpyAsync(buf, (const char*)d->stage.p + (size_t)root * bytes, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
}

void dist_send_to_root(ltx_ctx* ctx, void* buf, long bytes, int owner, int root) {
    DistState* d = ctx->dist;
    LTX_REQUIRE(d, "multi-GPU call without ltx_dist_init / ltx_dist_set_transport on this context");
    LTX_REQUIRE(buf && bytes > 0 && owner >= 0 && owner < d->world && root >= 0 && root < d->world, "dist_send_to_root: bad arguments");
    if (d->world == 1 || owner == root) return;
    if (d->comm) {
        // point to point over xGMI: the bytes cross one link once (a broadcast would put them on every rank)
        if (d->rank == owner) {
            d->n_collectives++;
            RCCL_CHECK(rccl().Send(buf, (size_t)bytes, ncclUint8, root, (ncclComm_t)d->comm, ctx->stream));
        } else if (d->rank == root) {
            d->n_collectives++;
            RCCL_CHECK(rccl().Recv(buf, (size_t)bytes, ncclUint8, owner, (ncclComm_t)d->comm, ctx->stream));
        }
        return;
    }
    // host transport (tests): it only has an all-gather, in which every rank must take part; the root keeps the owner's slot
    d->stage.ensure((size_t)bytes * d->world);
    dist_allgather(ctx, buf, d->stage.p, bytes);
    if (d->rank == root)
        HIP_CHECK(hipMemcpyAsync(buf, (const char*)d->stage.p + (size_t)owner * bytes, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
}
