// wtp_comm.hip — the context owns an RCCL communicator (multi-GPU behind the boundary).
//
// One process per GPU, one context per process.  The block decomposition of DESIGN.md §7b needs three things from
// the transport: a point-to-point round with the two neighbours along an axis (variable row counts), a reduction
// of the step's statistics, and somebody to carry the communicator id to the other ranks.  The first two live here,
// on the context's stream, so a caller without torch (the Julia side: INTEGRATION.md) drives the whole exchange
// through the C ABI; the id (128 bytes) travels by whatever the caller has (MPI.bcast, a file, an environment
// variable).  The Python host mirror keeps torch.distributed as its default transport and can be switched to this one
// (WTP_COMM=abi), which is how the entry points are exercised.
//
// librccl is opened at the first wtp_comm_* call (dlopen), not linked: a single-GPU user of libwtp never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "wtp_internal.hpp"

namespace wtp {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static Rccl g_rccl;

static int rccl_load(wtp_ctx* ctx) {
    if (g_rccl.lib) return WTP_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ctx, WTP_ERR_STATE, "wtp_comm: librccl.so not found (dlopen)");
    Rccl r;
    r.lib = h;
#define WTP_SYM(name)                                                               \
    r.name = (decltype(r.name))dlsym(h, "nccl" #name);                              \
    if (!r.name) return fail(ctx, WTP_ERR_STATE, "wtp_comm: librccl lacks nccl" #name);
    WTP_SYM(GetUniqueId)
    WTP_SYM(CommInitRank)
    WTP_SYM(CommDestroy)
    WTP_SYM(GroupStart)
    WTP_SYM(GroupEnd)
    WTP_SYM(Send)
    WTP_SYM(Recv)
    WTP_SYM(AllReduce)
    WTP_SYM(AllGather)
    WTP_SYM(GetErrorString)
#undef WTP_SYM
    g_rccl = r;
    return WTP_OK;
}

#define WTP_NCCL(ctx, call)                                                     \
    do {                                                                        \
        ncclResult_t e_ = (call);                                               \
        if (e_ != ncclSuccess) return fail(ctx, WTP_ERR_HIP, g_rccl.GetErrorString(e_)); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == WTP_COMM_ID_BYTES, "wtp.h promises 128 bytes");

} // namespace wtp

using namespace wtp;
#define WTP_API extern "C"

WTP_API int wtp_comm_unique_id(wtp_ctx* ctx, void* id_out) {
    if (!ctx || !id_out) return WTP_ERR_ARG;
    int rc;
    if ((rc = rccl_load(ctx))) return rc;
    ncclUniqueId id;
    WTP_NCCL(ctx, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return WTP_OK;
}

WTP_API int wtp_comm_init(wtp_ctx* ctx, const void* id_in, int rank, int nranks) {
    if (!ctx || !id_in) return WTP_ERR_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, WTP_ERR_ARG, "wtp_comm_init: 0 <= rank < nranks");
    if (ctx->comm) return fail(ctx, WTP_ERR_STATE, "wtp_comm_init: the context has a communicator already");
    int rc;
    if ((rc = rccl_load(ctx))) return rc;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclComm_t comm = nullptr;
    WTP_NCCL(ctx, g_rccl.CommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    if ((rc = ensure(ctx, ctx->comm_scratch, 256))) return rc; // counts out [0..1], counts in [2..3], statistics [8..]
    return WTP_OK;
}

WTP_API int wtp_comm_finalize(wtp_ctx* ctx) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->comm) return WTP_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_size = 0;
    return WTP_OK;
}

// One round with the low and the high neighbour along an axis.  What this rank sends to its high neighbour is what
// that rank receives from its low one, so both sides post (send lo, recv lo, send hi, recv hi) in the same order and
// the pairs match; a rank that is its own neighbour (one rank, or a periodic axis of extent 1) gets its own rows back.
WTP_API int wtp_comm_exchange_rows(wtp_ctx* ctx, int peer_lo, int peer_hi, const void* d_send_lo, int64_t n_send_lo,
                                   const void* d_send_hi, int64_t n_send_hi, void* d_recv_lo, void* d_recv_hi, int64_t cap,
                                   int64_t* n_recv_lo, int64_t* n_recv_hi) {
    if (!ctx || !n_recv_lo || !n_recv_hi) return WTP_ERR_ARG;
    if (!ctx->comm) return fail(ctx, WTP_ERR_STATE, "wtp_comm_exchange_rows before wtp_comm_init");
    if (n_send_lo < 0 || n_send_hi < 0 || cap < 0) return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_rows: negative count");
    if (peer_lo >= ctx->comm_size || peer_hi >= ctx->comm_size) return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_rows: peer out of range");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    int rc;
    if ((rc = ensure_pinned(ctx, 64))) return rc;
    int64_t* h = (int64_t*)ctx->host_pinned;
    int64_t* d = (int64_t*)ctx->comm_scratch.p;
    // 1. the counts
    h[0] = peer_lo >= 0 ? n_send_lo : 0;
    h[1] = peer_hi >= 0 ? n_send_hi : 0;
    h[2] = h[3] = 0;
    WTP_HIP(ctx, hipMemcpyAsync(d, h, 4 * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    WTP_NCCL(ctx, g_rccl.GroupStart());
    if (peer_lo >= 0) {
        WTP_NCCL(ctx, g_rccl.Send(d + 0, 1, ncclInt64, peer_lo, comm, ctx->stream));
        WTP_NCCL(ctx, g_rccl.Recv(d + 2, 1, ncclInt64, peer_lo, comm, ctx->stream));
    }
    if (peer_hi >= 0) {
        WTP_NCCL(ctx, g_rccl.Send(d + 1, 1, ncclInt64, peer_hi, comm, ctx->stream));
        WTP_NCCL(ctx, g_rccl.Recv(d + 3, 1, ncclInt64, peer_hi, comm, ctx->stream));
    }
    WTP_NCCL(ctx, g_rccl.GroupEnd());
    WTP_HIP(ctx, hipMemcpyAsync(h + 4, d + 2, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t in_lo = peer_lo >= 0 ? h[4] : 0, in_hi = peer_hi >= 0 ? h[5] : 0;
    *n_recv_lo = in_lo;
    *n_recv_hi = in_hi;
    if ((h[0] > 0 && !d_send_lo) || (h[1] > 0 && !d_send_hi))
        return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_rows: NULL send buffer with a non-zero count");
    // a receive buffer that is too small (or missing) must not leave the peer hanging in its send: the rows are
    // drained into scratch memory and the call reports the error afterwards
    const bool over_lo = in_lo > 0 && (in_lo > cap || !d_recv_lo), over_hi = in_hi > 0 && (in_hi > cap || !d_recv_hi);
    if (over_lo || over_hi) {
        if ((rc = ensure(ctx, ctx->scratch, (size_t)(in_lo + in_hi) * 16))) return rc;
        if (over_lo) d_recv_lo = ctx->scratch.p;
        if (over_hi) d_recv_hi = (char*)ctx->scratch.p + (size_t)in_lo * 16;
    }
    // 2. the rows (16 bytes each), stream-ordered: the caller's next launch on this context sees them
    WTP_NCCL(ctx, g_rccl.GroupStart());
    if (peer_lo >= 0) {
        if (h[0] > 0) WTP_NCCL(ctx, g_rccl.Send(d_send_lo, (size_t)h[0] * 16, ncclUint8, peer_lo, comm, ctx->stream));
        if (in_lo > 0) WTP_NCCL(ctx, g_rccl.Recv(d_recv_lo, (size_t)in_lo * 16, ncclUint8, peer_lo, comm, ctx->stream));
    }
    if (peer_hi >= 0) {
        if (h[1] > 0) WTP_NCCL(ctx, g_rccl.Send(d_send_hi, (size_t)h[1] * 16, ncclUint8, peer_hi, comm, ctx->stream));
        if (in_hi > 0) WTP_NCCL(ctx, g_rccl.Recv(d_recv_hi, (size_t)in_hi * 16, ncclUint8, peer_hi, comm, ctx->stream));
    }
    WTP_NCCL(ctx, g_rccl.GroupEnd());
    if (over_lo || over_hi)
        return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_rows: a peer sent more rows than the receive buffer holds (counts returned; rows dropped)");
    return WTP_OK;
}

// The global view of a sweep (what `_relax!`'s stop rules read, src/repel.jl:293,305-334): maximum of max_force; sums
// of sum_u, sum_u2, n_move, n_fallback, n_uncovered, n_escaped.  The closest pair is the one of the rank that holds
// the smallest argmin_r; its indices stay that rank's local ones and are set to -1 on the others.
WTP_API int wtp_comm_allreduce_stats(wtp_ctx* ctx, wtp_step_stats* st) {
    if (!ctx || !st) return WTP_ERR_ARG;
    if (!ctx->comm) return fail(ctx, WTP_ERR_STATE, "wtp_comm_allreduce_stats before wtp_comm_init");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    int rc;
    if ((rc = ensure_pinned(ctx, 256))) return rc;
    double* h = (double*)ctx->host_pinned;
    double* d = (double*)ctx->comm_scratch.p + 8;
    h[0] = st->max_force;
    h[1] = -st->argmin_r; // minimum through the same max reduction
    h[2] = st->sum_u;
    h[3] = st->sum_u2;
    h[4] = (double)st->n_move;
    h[5] = (double)st->n_fallback;
    h[6] = (double)st->n_uncovered;
    h[7] = (double)st->n_escaped;
    WTP_HIP(ctx, hipMemcpyAsync(d, h, 8 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    WTP_NCCL(ctx, g_rccl.GroupStart());
    WTP_NCCL(ctx, g_rccl.AllReduce(d, d, 2, ncclFloat64, ncclMax, comm, ctx->stream));
    WTP_NCCL(ctx, g_rccl.AllReduce(d + 2, d + 2, 6, ncclFloat64, ncclSum, comm, ctx->stream));
    WTP_NCCL(ctx, g_rccl.GroupEnd());
    WTP_HIP(ctx, hipMemcpyAsync(h + 8, d, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double* g = h + 8;
    const bool mine = st->argmin_r == -g[1];
    st->max_force = g[0];
    st->argmin_r = -g[1];
    if (!mine) st->argmin_i = st->argmin_j = -1;
    st->sum_u = g[2];
    st->sum_u2 = g[3];
    st->n_move = (int64_t)g[4];
    st->n_fallback = (int64_t)g[5];
    st->n_uncovered = (int64_t)g[6];
    st->n_escaped = (int64_t)g[7];
    return WTP_OK;
}

// ---- primitives of the block iteration (wtp_block.hip; also exported) --------------------------------------------------
// One grouped round: every message is a send to and a receive from peers[j].  Both sides walk their peer lists in
// ascending rank order and post (send, recv) per peer inside ONE group, so the pairs match without any ordering between
// different peers; the counts were agreed on beforehand (they ride with the previous iteration's all-gather).
WTP_API int wtp_comm_exchange_peers(wtp_ctx* ctx, int n_msgs, const int* peers, const void* const* d_send,
                                    const int64_t* n_send, void* const* d_recv, const int64_t* n_recv) {
    if (!ctx) return WTP_ERR_ARG;
    return wtp::comm_exchange_peers_on(ctx, ctx->stream, n_msgs, peers, d_send, n_send, d_recv, n_recv);
}

// (the same round on a stream of the caller's choosing: the block driver posts it on the context's second stream and lets
// the first one rank the owned points meanwhile)
int wtp::comm_exchange_peers_on(wtp_ctx* ctx, hipStream_t stream, int n_msgs, const int* peers, const void* const* d_send,
                                const int64_t* n_send, void* const* d_recv, const int64_t* n_recv) {
    if (!ctx->comm) return fail(ctx, WTP_ERR_STATE, "wtp_comm_exchange_peers before wtp_comm_init");
    if (n_msgs < 0 || (n_msgs > 0 && (!peers || !d_send || !n_send || !d_recv || !n_recv)))
        return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_peers: NULL argument");
    for (int j = 0; j < n_msgs; ++j) {
        if (peers[j] < 0 || peers[j] >= ctx->comm_size) return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_peers: peer out of range");
        if (n_send[j] < 0 || n_recv[j] < 0) return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_peers: negative count");
        if ((n_send[j] > 0 && !d_send[j]) || (n_recv[j] > 0 && !d_recv[j]))
            return fail(ctx, WTP_ERR_ARG, "wtp_comm_exchange_peers: NULL buffer with a non-zero count");
    }
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    bool any = false;
    for (int j = 0; j < n_msgs; ++j) any = any || n_send[j] > 0 || n_recv[j] > 0;
    if (!any) return WTP_OK;
    WTP_NCCL(ctx, g_rccl.GroupStart());
    for (int j = 0; j < n_msgs; ++j) {
        if (n_send[j] > 0) WTP_NCCL(ctx, g_rccl.Send(d_send[j], (size_t)n_send[j] * 16, ncclUint8, peers[j], comm, stream));
        if (n_recv[j] > 0) WTP_NCCL(ctx, g_rccl.Recv(d_recv[j], (size_t)n_recv[j] * 16, ncclUint8, peers[j], comm, stream));
    }
    WTP_NCCL(ctx, g_rccl.GroupEnd());
    return WTP_OK;
}

WTP_API int wtp_comm_allgather_dev(wtp_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes) {
    if (!ctx || !d_send || !d_recv) return WTP_ERR_ARG;
    if (!ctx->comm) return fail(ctx, WTP_ERR_STATE, "wtp_comm_allgather_dev before wtp_comm_init");
    if (bytes <= 0 || (bytes & 7)) return fail(ctx, WTP_ERR_ARG, "wtp_comm_allgather_dev: bytes must be a positive multiple of 8");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    WTP_NCCL(ctx, g_rccl.AllGather(d_send, d_recv, (size_t)bytes / 8, ncclUint64, (ncclComm_t)ctx->comm, ctx->stream));
    return WTP_OK;
}
