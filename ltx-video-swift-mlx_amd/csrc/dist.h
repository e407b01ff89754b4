// dist.h - the multi-GPU layer of the library: one process per GPU, one ltx_ctx per process, RCCL over xGMI reached
// directly (librccl is dlopen'ed on the first ltx_dist_* call; no PyTorch anywhere on this path). The reference is a
// single-device program; what is sharded here is what SURVEY 8(e) lists: the CFG cond/uncond pair
// (LTXPipeline.swift:820-865, :2235-2283), one sample's tokens (sequence parallelism) and the VAE's temporal tiles
// (VideoDecoder.swift:517-592).
//
// Two transports sit behind the same two primitives (all-gather, broadcast), both enqueued on the context's stream:
//   * native: an RCCL communicator created from a unique id the host distributes (ltx_dist_unique_id -> every rank's
//     ltx_dist_init), ncclAllGather / ncclBroadcast on the context's stream, no host synchronisation;
//   * host transport: an all-gather callback supplied by the caller (ltx_dist_set_transport). It exists so that the
//     sharded paths can be exercised by world-size-2 tests over gloo where RCCL cannot run (two processes on one GPU,
//     or no GPU at all for the host logic).
#pragma once
#include "runtime.h"

typedef int (*ltx_dist_gather_fn)(void* user, const void* send, void* recv, long bytes);

struct DistState {
    int rank = 0, world = 1;
    void* comm = nullptr;  // ncclComm_t
    ltx_dist_gather_fn cb = nullptr;
    void* cb_user = nullptr;
    DevBuf stage;  // broadcast over the host transport = all-gather into this, then copy the root's slot
    // second stream + events of the native transport: a collective runs beside the kernels that do not need its result
    // (dist_fork / dist_join order it against the context's stream; created on first use)
    hipStream_t side = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    long n_collectives = 0;  // collectives enqueued so far (reported by ltx_dist_info; lets tests see the exchange happen)
};

void dist_unique_id(void* id128);
void dist_init_native(ltx_ctx* ctx, int rank, int world, const void* id128);
void dist_set_transport(ltx_ctx* ctx, int rank, int world, ltx_dist_gather_fn gather, void* user);
void dist_shutdown(ltx_ctx* ctx);
inline int dist_world(const ltx_ctx* ctx) { return ctx->dist ? ctx->dist->world : 1; }
inline int dist_rank(const ltx_ctx* ctx) { return ctx->dist ? ctx->dist->rank : 0; }

// recv = [world][bytes] in rank order. DEVICE pointers; enqueued on ctx->stream (the host transport may synchronise).
void dist_allgather(ltx_ctx* ctx, const void* send, void* recv, long bytes);
// in place on every rank; root's bytes win
void dist_broadcast(ltx_ctx* ctx, void* buf, long bytes, int root);
// `buf` of rank `owner` -> `buf` of rank `root` (same address range on both; every other rank's buffer is untouched). Native
// transport: one ncclSend / ncclRecv pair, ranks that are neither owner nor root do nothing. Every rank must call it.
void dist_send_to_root(ltx_ctx* ctx, void* buf, long bytes, int owner, int root);
// Overlap (native transport only; dist_can_overlap): dist_fork(ev) makes the side stream wait for everything enqueued on the
// context's stream so far, dist_allgather_on(..., dist_side_stream()) enqueues the collective there, dist_join(ev) makes the
// context's stream wait for the side stream. ev = 0..3 (one event object per fork / join site; re-recorded every use).
bool dist_can_overlap(const ltx_ctx* ctx);
hipStream_t dist_side_stream(ltx_ctx* ctx);
void dist_fork(ltx_ctx* ctx, int ev);
void dist_join(ltx_ctx* ctx, int ev);
void dist_allgather_on(ltx_ctx* ctx, const void* send, void* recv, long bytes, hipStream_t stream);
