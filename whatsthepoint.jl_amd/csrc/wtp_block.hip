// wtp_block.hip — one rank's share of a block-decomposed repel, the whole iteration behind the C ABI
// (include/wtp.h: wtp_block_*).
//
// The reference has one address space and one kd-tree (src/repel.jl:243-334); this is the build's own decomposition
// of the same Jacobi sweep.  The cloud is cut into the cells of an orthtree partition (axis-aligned boxes, one per
// rank, any shape: equidistant octants for the uniform benchmark cloud, count medians for a graded one).  Per
// iteration and rank:
//
//   1. exchange: ONE grouped ncclSend/ncclRecv round with every rank whose box lies within w + margin of this one
//      (faces, edges and corners directly — xGMI is point-to-point, every peer has its own link).  The row counts
//      are already known on both sides: they rode with the previous iteration's all-gather.
//   2. migration: arrivals are appended to the owned set, departures leave it (ordered compaction; 64-bit global ids
//      follow); only when somebody actually crossed a box face by more than `margin`.
//   3. ghosts -> fixed head of the snapshot (wtp_relax_set_fixed_dev: appended, no pass over the cloud).
//   4. hash + sweep of [ghosts ; owned] (relax_step_enqueue).
//   5. from the new positions: classify every owned point against the peers' boxes (which ranks need it as a ghost,
//      does it change owner), count per peer, fill the send rows of the NEXT exchange — three launches.
//   6. one all-gather of {step statistics, row counts per destination rank}, read back with the iteration's only host
//      synchronisation; every rank reduces the statistics in rank order (fixed order: reproducible sums).
//   7. if any rank counted a query whose support reaches past its covered box: undo, widen, repeat.
//
// Order of the rows a peer receives = slot order of the sender's sorted state = a pure function of the sender's
// points, so the ghost ids on the receiving side (and with them every sum) do not depend on scheduling.
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "wtp_device.hpp"

namespace wtp {

constexpr int kBlkMaxPeers = 26;
constexpr int kBlkPasses = 16;                 // a wave owns 64 * 16 consecutive slots ("span")
constexpr int kBlkSpan = 64 * kBlkPasses;
constexpr int kBlkWaves = 4;                   // waves per workgroup
constexpr uint32_t kBlkMigrate = 0x80000000u;  // flag bit 31: the point changes owner; bits 26..30: peer index of the new owner
constexpr int kBlkStatWords = 10;              // wtp_step_stats as 8-byte words

struct BlkGeom {
    int np;
    float in_lo[kBlkMaxPeers][3], in_hi[kBlkMaxPeers][3]; // peer box widened by w + margin: who needs a point as a ghost
    float bx_lo[kBlkMaxPeers][3], bx_hi[kBlkMaxPeers][3]; // peer box itself: who owns a point
    float out_lo[3], out_hi[3];                           // own box widened by margin: beyond it a point changes owner
    float deep_lo[3], deep_hi[3];                         // own box shrunk by w + margin: a point in there is nobody's ghost
};

// columns of the per-span count table: [0, np) ghost rows per peer, [np, 2 np) migrants per peer, 2 np points that stay
__host__ __device__ inline int blk_cols(int np) { return 2 * np + 1; }

struct BlockState {
    bool active = false;
    int rank = 0, nranks = 1;
    std::vector<double> boxes; // nranks x 6
    double w = 0, margin = 0;
    std::vector<int> peers;    // ascending ranks
    BlkGeom geom{};
    int widened = 0;
    // device
    DevBuf flags, span_counts, totals, send, send_mig, gid[2], pool, recv_mig, gsend, grecv, lost;
    int gid_cur = 0;
    int64_t n_owned = 0, n_ghost = 0;
    int64_t send_cap = 0, mig_cap = 0;
    // the plan of the next exchange (host): rows to / from each peer
    bool plan_ready = false;
    std::vector<int64_t> send_cnt, mig_cnt, recv_cnt, rmig_cnt;
    // transport
    bool host_transport = false;
    bool overlap = true; // the owned points are ranked into the cells while the ghost rows travel (WTP_BLOCK_OVERLAP=0: after them)
    wtp_transport tr{};
    std::vector<unsigned char> hbuf_a, hbuf_b;
    // last info
    wtp_block_info info{};
    // stop rules (wtp_block_run_until)
};

static BlockState* bs_of(wtp_ctx* ctx) { return (BlockState*)ctx->block; }

void block_destroy(wtp_ctx* ctx) {
    BlockState* b = bs_of(ctx);
    if (!b) return;
    hipSetDevice(ctx->device);
    DevBuf* bufs[] = {&b->flags, &b->span_counts, &b->totals, &b->send, &b->send_mig, &b->gid[0], &b->gid[1], &b->pool,
                      &b->recv_mig, &b->gsend, &b->grecv, &b->lost};
    for (DevBuf* d : bufs)
        if (d->p) hipFree(d->p);
    delete b;
    ctx->block = nullptr;
}

// ---- kernels --------------------------------------------------------------------------------------------------------

__device__ inline bool blk_inside(const float4& p, const float (&lo)[3], const float (&hi)[3]) {
    return p.x >= lo[0] && p.x < hi[0] && p.y >= lo[1] && p.y < hi[1] && p.z >= lo[2] && p.z < hi[2];
}

// Pass 1: flag of every slot (which peers need the point as a ghost, does it change owner and to whom) and the span's
// counts per column.  97 % of the waves see no flagged point and only pay the load.
__global__ __launch_bounds__(64 * kBlkWaves) void blk_classify_kernel(const float4* __restrict__ P, int64_t n, int32_t n_fixed,
                                                                     BlkGeom g, uint32_t* __restrict__ flags,
                                                                     int32_t* __restrict__ span_counts, int32_t* __restrict__ lost) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t span = (int64_t)blockIdx.x * kBlkWaves + wave;
    const int64_t first = span * kBlkSpan;
    if (first >= n) return;
    const int np = g.np, ncol = blk_cols(np);
    int cnt = 0; // lane c holds the span's count of column c
    int n_lost = 0;
    // the span's sixteen loads in flight together (in a loop with the ballots between them they were sixteen round trips)
    float4 pp[kBlkPasses];
#pragma unroll
    for (int pass = 0; pass < kBlkPasses; ++pass) {
        const int64_t i = first + (int64_t)pass * 64 + lane;
        pp[pass] = P[i < n ? i : n - 1];
    }
#pragma unroll
    for (int pass = 0; pass < kBlkPasses; ++pass) {
        const int64_t i = first + (int64_t)pass * 64 + lane;
        uint32_t flag = 0;
        bool movable = false;
        if (i < n) {
            const float4 p = pp[pass];
            movable = w_to_id(p.w) >= n_fixed;
            // (most points lie deeper inside the box than any peer's layer reaches: six compares settle them)
            if (movable && !blk_inside(p, g.deep_lo, g.deep_hi)) {
                const bool leaves = !blk_inside(p, g.out_lo, g.out_hi);
                int owner = -1;
                for (int q = 0; q < np; ++q) {
                    if (blk_inside(p, g.in_lo[q], g.in_hi[q])) flag |= 1u << q;
                    if (leaves && owner < 0 && blk_inside(p, g.bx_lo[q], g.bx_hi[q])) owner = q;
                }
                if (leaves) {
                    if (owner >= 0) flag = (flag & ~(1u << owner)) | kBlkMigrate | ((uint32_t)owner << 26);
                    else n_lost += 1; // outside every neighbouring box: stays (reported)
                }
            }
            flags[i] = flag;
        }
        const unsigned long long stay = __ballot(movable && !(flag & kBlkMigrate));
        if (lane == 2 * np) cnt += __popcll(stay);
        if (__ballot(flag != 0) == 0ull) continue;
        for (int q = 0; q < np; ++q) {
            const unsigned long long m = __ballot((flag >> q) & 1u);
            if (lane == q) cnt += __popcll(m);
            const unsigned long long mm = __ballot((flag & kBlkMigrate) && (int)((flag >> 26) & 31u) == q);
            if (lane == np + q) cnt += __popcll(mm);
        }
    }
    if (lane < ncol) span_counts[span * ncol + lane] = cnt;
    if (n_lost) atomicAdd(lost, n_lost);
}

// Pass 2: exclusive scan of every column over the spans (one workgroup per column), the column totals behind them.
__global__ __launch_bounds__(256) void blk_scan_kernel(int32_t* __restrict__ span_counts, int64_t nspans, int ncol,
                                                      int32_t* __restrict__ totals) {
    __shared__ int part[256];
    const int col = blockIdx.x, t = threadIdx.x;
    const int64_t per = (nspans + 255) / 256, a = t * per, b = a + per < nspans ? a + per : nspans;
    int s = 0;
    for (int64_t i = a; i < b; ++i) s += span_counts[i * ncol + col];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int j = 0; j < 256; ++j) {
            const int v = part[j];
            part[j] = run;
            run += v;
        }
        totals[col] = run;
    }
    __syncthreads();
    int run = part[t];
    for (int64_t i = a; i < b; ++i) {
        const int v = span_counts[i * ncol + col];
        span_counts[i * ncol + col] = run;
        run += v;
    }
}

// The all-gather payload: [0, 10) the step's statistics as they are, then per DESTINATION rank the ghost rows and the
// migrants this rank will send, then {points lost, owned points}.  One thread: a few dozen words.
__global__ void blk_pack_kernel(const wtp_step_stats* __restrict__ st, const int32_t* __restrict__ totals, int np,
                                const int* __restrict__ peer_rank, int nranks, const int32_t* __restrict__ lost,
                                int64_t n_owned, int64_t* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t* sw = (const int64_t*)st;
    for (int j = 0; j < kBlkStatWords; ++j) out[j] = st ? sw[j] : 0;
    for (int r = 0; r < 2 * nranks; ++r) out[kBlkStatWords + r] = 0;
    for (int q = 0; q < np; ++q) {
        out[kBlkStatWords + peer_rank[q]] = totals[q];
        out[kBlkStatWords + nranks + peer_rank[q]] = totals[np + q];
    }
    out[kBlkStatWords + 2 * nranks] = lost ? *lost : 0;
    out[kBlkStatWords + 2 * nranks + 1] = n_owned;
}

// Pass 3: the send rows.  Ghost rows {x, y, z, 0} of peer q start at row sum_{j<q} totals[j]; a migrant is two rows
// {x, y, z, bits(gid low)}, {bits(gid high), 0, 0, 0}, peer q's start at 2 * sum_{j<q} totals[np + j].  Order inside a
// peer's rows: slot order.
__global__ __launch_bounds__(64 * kBlkWaves) void blk_fill_kernel(const float4* __restrict__ P, int64_t n, int32_t n_fixed, int np,
                                                                 const uint32_t* __restrict__ flags,
                                                                 const int32_t* __restrict__ span_off,
                                                                 const int32_t* __restrict__ totals, const int64_t* __restrict__ gid,
                                                                 float4* __restrict__ send, int64_t send_cap,
                                                                 float4* __restrict__ send_mig, int64_t mig_cap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t span = (int64_t)blockIdx.x * kBlkWaves + wave;
    const int64_t first = span * kBlkSpan;
    if (first >= n) return;
    const int ncol = blk_cols(np);
    // lane c: where column c's next row of this span goes
    int run = 0;
    if (lane < 2 * np) {
        const int c0 = lane < np ? 0 : np;
        int base = 0;
        for (int j = c0; j < lane; ++j) base += totals[j];
        run = base + span_off[span * ncol + lane];
    }
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t ff[kBlkPasses]; // the span's flags, all loads in flight together
#pragma unroll
    for (int pass = 0; pass < kBlkPasses; ++pass) {
        const int64_t i = first + (int64_t)pass * 64 + lane;
        ff[pass] = i < n ? flags[i] : 0u;
    }
#pragma unroll
    for (int pass = 0; pass < kBlkPasses; ++pass) {
        const int64_t i = first + (int64_t)pass * 64 + lane;
        const uint32_t flag = ff[pass];
        if (__ballot(flag != 0) == 0ull) continue;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (flag) p = P[i];
        for (int q = 0; q < np; ++q) {
            const bool g = (flag >> q) & 1u;
            const unsigned long long m = __ballot(g);
            if (m) {
                const int off = __builtin_amdgcn_readlane(run, q);
                if (g) {
                    const int64_t o = (int64_t)off + __popcll(m & below);
                    if (o < send_cap) send[o] = make_float4(p.x, p.y, p.z, 0.f);
                }
                if (lane == q) run += __popcll(m);
            }
            const bool mg = (flag & kBlkMigrate) && (int)((flag >> 26) & 31u) == q;
            const unsigned long long mm = __ballot(mg);
            if (mm) {
                const int off = __builtin_amdgcn_readlane(run, np + q);
                if (mg) {
                    const int64_t o = (int64_t)off + __popcll(mm & below);
                    const int64_t gd = gid[w_to_id(p.w) - n_fixed];
                    if (o < mig_cap) {
                        send_mig[2 * o] = make_float4(p.x, p.y, p.z, __builtin_bit_cast(float, (uint32_t)(gd & 0xFFFFFFFFll)));
                        send_mig[2 * o + 1] = make_float4(__builtin_bit_cast(float, (uint32_t)((uint64_t)gd >> 32)), 0.f, 0.f, 0.f);
                    }
                }
                if (lane == np + q) run += __popcll(mm);
            }
        }
    }
}

// Migration: the points that stay, in slot order, renumbered 0 .. n_stay; their global ids follow.
__global__ __launch_bounds__(64 * kBlkWaves) void blk_compact_kernel(const float4* __restrict__ P, int64_t n, int32_t n_fixed, int np,
                                                                    const uint32_t* __restrict__ flags,
                                                                    const int32_t* __restrict__ span_off,
                                                                    const int64_t* __restrict__ gid_old, float4* __restrict__ Pn,
                                                                    int64_t* __restrict__ gid_new) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t span = (int64_t)blockIdx.x * kBlkWaves + wave;
    const int64_t first = span * kBlkSpan;
    if (first >= n) return;
    const int ncol = blk_cols(np);
    int64_t pos = span_off[span * ncol + 2 * np];
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < kBlkPasses; ++pass) {
        const int64_t i = first + (int64_t)pass * 64 + lane;
        bool keep = false;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            p = P[i];
            keep = w_to_id(p.w) >= n_fixed && !(flags[i] & kBlkMigrate);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int64_t o = pos + __popcll(m & below);
            gid_new[o] = gid_old[w_to_id(p.w) - n_fixed];
            p.w = id_to_w(0.f, (int32_t)o);
            Pn[o] = p;
        }
        pos += __popcll(m);
    }
}

// arrivals behind the points that stayed
__global__ void blk_arrivals_kernel(const float4* __restrict__ rows, int64_t n_arr, int64_t n_stay, float4* __restrict__ Pn,
                                    int64_t* __restrict__ gid_new) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_arr; k += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = rows[2 * k], b = rows[2 * k + 1];
        const uint64_t lo = __builtin_bit_cast(uint32_t, a.w), hi = __builtin_bit_cast(uint32_t, b.x);
        gid_new[n_stay + k] = (int64_t)(lo | (hi << 32));
        Pn[n_stay + k] = make_float4(a.x, a.y, a.z, id_to_w(0.f, (int32_t)(n_stay + k)));
    }
}

// this rank's own emigrants stay around as ghosts for the iteration (their new owner cut its layers before they arrived)
__global__ void blk_emigrant_ghosts_kernel(const float4* __restrict__ mig_rows, int64_t n_mig, float4* __restrict__ pool_tail) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_mig; k += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = mig_rows[2 * k];
        pool_tail[k] = make_float4(a.x, a.y, a.z, 0.f);
    }
}

__global__ void blk_iota_gid_kernel(int64_t* gid, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) gid[k] = k;
}

__global__ void blk_gather_gid_kernel(const float4* __restrict__ P, int64_t n, int32_t n_fixed, const int64_t* __restrict__ gid,
                                      int64_t* __restrict__ out) {
    // (movable order is the order wtp_relax_get_dev writes: by movable index — gid is stored that way already)
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n - n_fixed; k += (int64_t)gridDim.x * blockDim.x)
        out[k] = gid[k];
    (void)P;
}

// ---- host side --------------------------------------------------------------------------------------------------------

static int blk_sync(wtp_ctx* ctx, BlockState* b) {
    ctx->ev_last_end = -1;
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_syncs += 1;
    (void)b;
    return WTP_OK;
}

static bool boxes_near(const double* a, const double* c, double reach) {
    for (int ax = 0; ax < 3; ++ax)
        if (a[ax] - reach >= c[3 + ax] || c[ax] - reach >= a[3 + ax]) return false;
    return true;
}

// peers and the kernel's geometry block for the current ghost width
static int blk_geometry(wtp_ctx* ctx, BlockState* b) {
    const double w_eff = b->w + b->margin;
    const double* mine = &b->boxes[(size_t)b->rank * 6];
    for (int ax = 0; ax < 3; ++ax)
        if (mine[3 + ax] - mine[ax] < b->w + 2.0 * b->margin)
            return fail(ctx, WTP_ERR_ARG, "wtp_block: this rank's box is thinner than a ghost layer (w + 2 margin): use fewer ranks");
    b->peers.clear();
    for (int r = 0; r < b->nranks; ++r) {
        if (r == b->rank) continue;
        // anybody whose box my points can come within w_eff of (my points stray at most ~margin beyond my box)
        if (boxes_near(mine, &b->boxes[(size_t)r * 6], w_eff + b->margin + 1e-12)) b->peers.push_back(r);
    }
    if ((int)b->peers.size() > kBlkMaxPeers)
        return fail(ctx, WTP_ERR_ARG, "wtp_block: more than 26 neighbouring boxes (boxes too thin for this ghost width)");
    BlkGeom& g = b->geom;
    g.np = (int)b->peers.size();
    for (int q = 0; q < g.np; ++q) {
        const double* bx = &b->boxes[(size_t)b->peers[q] * 6];
        for (int ax = 0; ax < 3; ++ax) {
            g.bx_lo[q][ax] = (float)bx[ax];
            g.bx_hi[q][ax] = (float)bx[3 + ax];
            g.in_lo[q][ax] = (float)(bx[ax] - w_eff);
            g.in_hi[q][ax] = (float)(bx[3 + ax] + w_eff);
        }
    }
    for (int ax = 0; ax < 3; ++ax) {
        g.out_lo[ax] = (float)(mine[ax] - b->margin);
        g.out_hi[ax] = (float)(mine[3 + ax] + b->margin);
        // every peer's box lies outside mine, so its layer reaches at most w_eff into my box (one float ulp of slack)
        g.deep_lo[ax] = std::nextafter((float)(mine[ax] + w_eff), std::numeric_limits<float>::infinity());
        g.deep_hi[ax] = std::nextafter((float)(mine[3 + ax] - w_eff), -std::numeric_limits<float>::infinity());
    }
    const size_t np = b->peers.size();
    b->send_cnt.assign(np, 0);
    b->mig_cnt.assign(np, 0);
    b->recv_cnt.assign(np, 0);
    b->rmig_cnt.assign(np, 0);
    // what the snapshot is complete for: the box plus the layers the peers send
    double lo[3], hi[3];
    for (int ax = 0; ax < 3; ++ax) {
        lo[ax] = (double)(float)(mine[ax] - w_eff);
        hi[ax] = (double)(float)(mine[3 + ax] + w_eff);
    }
    return wtp_relax_set_coverage_box(ctx, lo, hi);
}

static int64_t blk_nspans(int64_t n) { return (n + kBlkSpan - 1) / kBlkSpan; }

// launches the fill pass with the current send buffers (again after they grew)
static int blk_fill(wtp_ctx* ctx, BlockState* b) {
    RelaxState& r = ctx->relax;
    const int np = b->geom.np;
    if (np == 0) return WTP_OK;
    int rc;
    if ((rc = ensure(ctx, b->send, 16 * (size_t)b->send_cap))) return rc;
    if ((rc = ensure(ctx, b->send_mig, 32 * (size_t)b->mig_cap))) return rc;
    const int64_t n = r.n, nsp = blk_nspans(n);
    const int grid = (int)((nsp + kBlkWaves - 1) / kBlkWaves);
    hipLaunchKernelGGL(blk_fill_kernel, dim3(grid), dim3(64 * kBlkWaves), 0, ctx->stream, (const float4*)ctx->pts[r.bufP].p, n,
                       (int32_t)r.n_fixed, np, (const uint32_t*)b->flags.p, (const int32_t*)b->span_counts.p,
                       (const int32_t*)b->totals.p, (const int64_t*)b->gid[b->gid_cur].p, (float4*)b->send.p, b->send_cap,
                       (float4*)b->send_mig.p, b->mig_cap);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// classify + count + fill from the session's current P, then the all-gather payload (d_stats may be NULL)
static int blk_classify(wtp_ctx* ctx, BlockState* b, const wtp_step_stats* d_stats) {
    RelaxState& r = ctx->relax;
    if (r.pending.active) return fail(ctx, WTP_ERR_STATE, "wtp_block: classification with a pending fixed head");
    const int np = b->geom.np, ncol = blk_cols(np);
    const int64_t n = r.n, nsp = blk_nspans(n);
    int rc;
    if ((rc = ensure(ctx, b->flags, sizeof(uint32_t) * (size_t)(n + n / 8)))) return rc;
    if ((rc = ensure(ctx, b->span_counts, sizeof(int32_t) * (size_t)(nsp + nsp / 8 + 1) * ncol))) return rc;
    if ((rc = ensure(ctx, b->totals, sizeof(int32_t) * (size_t)(ncol + kBlkMaxPeers + 8)))) return rc;
    if ((rc = ensure(ctx, b->lost, 64))) return rc;
    if (b->send_cap == 0) {
        b->send_cap = 65536 + b->n_owned / 3;
        b->mig_cap = 16384 + b->n_owned / 16;
    }
    const size_t gwords = (size_t)kBlkStatWords + 2 * (size_t)b->nranks + 2;
    if ((rc = ensure(ctx, b->gsend, 8 * gwords))) return rc;
    if ((rc = ensure(ctx, b->grecv, 8 * gwords * (size_t)b->nranks))) return rc;
    const float4* P = (const float4*)ctx->pts[r.bufP].p;
    int32_t* totals = (int32_t*)b->totals.p;
    int* d_peer_rank = (int*)(totals + ncol + 4);
    WTP_HIP(ctx, hipMemsetAsync(b->lost.p, 0, 4, ctx->stream));
    if (np > 0) {
        WTP_HIP(ctx, hipMemcpyAsync(d_peer_rank, b->peers.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice, ctx->stream));
        const int grid = (int)((nsp + kBlkWaves - 1) / kBlkWaves);
        hipLaunchKernelGGL(blk_classify_kernel, dim3(grid), dim3(64 * kBlkWaves), 0, ctx->stream, P, n, (int32_t)r.n_fixed,
                           b->geom, (uint32_t*)b->flags.p, (int32_t*)b->span_counts.p, (int32_t*)b->lost.p);
        hipLaunchKernelGGL(blk_scan_kernel, dim3(ncol), dim3(256), 0, ctx->stream, (int32_t*)b->span_counts.p, nsp, ncol, totals);
        if ((rc = blk_fill(ctx, b))) return rc;
    } else {
        WTP_HIP(ctx, hipMemsetAsync(totals, 0, sizeof(int32_t) * (size_t)ncol, ctx->stream));
    }
    hipLaunchKernelGGL(blk_pack_kernel, dim3(1), dim3(64), 0, ctx->stream, d_stats, (const int32_t*)totals, np,
                       (const int*)d_peer_rank, b->nranks, (const int32_t*)b->lost.p, b->n_owned, (int64_t*)b->gsend.p);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// all-gather of the payload + read-back: the iteration's host synchronisation.  Fills the plan; returns the global
// statistics in *g (if d_stats was given to blk_classify) and whether a send buffer overflowed anywhere.
static int blk_gather(wtp_ctx* ctx, BlockState* b, wtp_step_stats* g, bool* overflow, int64_t* lost_total) {
    const size_t gwords = (size_t)kBlkStatWords + 2 * (size_t)b->nranks + 2;
    const size_t bytes = 8 * gwords;
    int rc;
    if ((rc = ensure_pinned(ctx, bytes * (size_t)b->nranks + 64))) return rc;
    int64_t* h = (int64_t*)ctx->host_pinned;
    if (b->host_transport) {
        b->hbuf_a.resize(bytes);
        WTP_HIP(ctx, hipMemcpyAsync(b->hbuf_a.data(), b->gsend.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = blk_sync(ctx, b))) return rc;
        if (b->tr.allgather(b->tr.user, b->hbuf_a.data(), h, (int64_t)bytes) != 0)
            return fail(ctx, WTP_ERR_STATE, "wtp_block: the caller's allgather callback failed");
    } else if (b->nranks == 1) {
        WTP_HIP(ctx, hipMemcpyAsync(h, b->gsend.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = blk_sync(ctx, b))) return rc;
    } else {
        if ((rc = wtp_comm_allgather_dev(ctx, b->gsend.p, b->grecv.p, (int64_t)bytes))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(h, b->grecv.p, bytes * (size_t)b->nranks, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = blk_sync(ctx, b))) return rc;
    }
    // the plan: what I send (my own row) and what every peer sends me
    const int np = b->geom.np;
    const int64_t* mine = h + (size_t)b->rank * gwords;
    int64_t tot_s = 0, tot_m = 0;
    for (int q = 0; q < np; ++q) {
        b->send_cnt[q] = mine[kBlkStatWords + b->peers[q]];
        b->mig_cnt[q] = mine[kBlkStatWords + b->nranks + b->peers[q]];
        const int64_t* theirs = h + (size_t)b->peers[q] * gwords;
        b->recv_cnt[q] = theirs[kBlkStatWords + b->rank];
        b->rmig_cnt[q] = theirs[kBlkStatWords + b->nranks + b->rank];
        tot_s += b->send_cnt[q];
        tot_m += b->mig_cnt[q];
    }
    // a rank that names me although I do not list it as a peer would be left hanging: boxes must be identical everywhere
    for (int rnk = 0; rnk < b->nranks; ++rnk) {
        if (rnk == b->rank) continue;
        const int64_t* theirs = h + (size_t)rnk * gwords;
        if (theirs[kBlkStatWords + b->rank] || theirs[kBlkStatWords + b->nranks + b->rank]) {
            bool listed = false;
            for (int q = 0; q < np; ++q) listed = listed || b->peers[q] == rnk;
            if (!listed) return fail(ctx, WTP_ERR_STATE, "wtp_block: a rank that is not a neighbour announces rows (boxes differ between ranks?)");
        }
    }
    int64_t lost = 0;
    for (int rnk = 0; rnk < b->nranks; ++rnk) lost += (h + (size_t)rnk * gwords)[kBlkStatWords + 2 * b->nranks];
    // the announced counts are the true ones (the count pass does not look at capacities); rows beyond the capacity
    // of MY send buffers were not written: a local matter, fixed by filling again into larger buffers
    *overflow = tot_s > b->send_cap || tot_m > b->mig_cap;
    *lost_total = lost;
    if (g) {
        // global statistics, reduced in rank order (a fixed order: the sums are reproducible)
        wtp_step_stats out{};
        out.argmin_r = std::numeric_limits<double>::infinity();
        out.argmin_i = out.argmin_j = -1;
        for (int rnk = 0; rnk < b->nranks; ++rnk) {
            wtp_step_stats s;
            memcpy(&s, h + (size_t)rnk * gwords, sizeof(s));
            out.max_force = s.max_force > out.max_force ? s.max_force : out.max_force;
            out.sum_u += s.sum_u;
            out.sum_u2 += s.sum_u2;
            out.n_move += s.n_move;
            out.n_fallback += s.n_fallback;
            out.n_uncovered += s.n_uncovered;
            out.n_escaped += s.n_escaped;
            if (s.n_move > 0 && s.argmin_r < out.argmin_r) {
                out.argmin_r = s.argmin_r;
                out.argmin_i = s.argmin_i; // (local indices of the rank that holds the pair; see wtp_block_step)
                out.argmin_j = s.argmin_j;
            }
        }
        *g = out;
    }
    return WTP_OK;
}

// The plan for the next exchange from the session's current positions: rows filled, counts known on every rank.
static int blk_plan(wtp_ctx* ctx, BlockState* b, const wtp_step_stats* d_stats, wtp_step_stats* g) {
    int rc;
    if ((rc = blk_classify(ctx, b, d_stats))) return rc;
    bool over = false;
    int64_t lost = 0;
    if ((rc = blk_gather(ctx, b, g, &over, &lost))) return rc;
    if (lost > 0)
        return fail(ctx, WTP_ERR_STATE, "wtp_block: " + std::to_string(lost) + " points left their box for a region no neighbouring rank owns");
    if (over) { // my send buffers were too small: the counts stand, the rows are written again into larger ones
        int64_t tot_s = 0, tot_m = 0;
        for (size_t q = 0; q < b->send_cnt.size(); ++q) {
            tot_s += b->send_cnt[q];
            tot_m += b->mig_cnt[q];
        }
        if (tot_s > b->send_cap) b->send_cap = tot_s + tot_s / 2 + 65536;
        if (tot_m > b->mig_cap) b->mig_cap = tot_m + tot_m / 2 + 16384;
        if ((rc = blk_fill(ctx, b))) return rc;
    }
    b->plan_ready = true;
    return WTP_OK;
}

} // namespace wtp

using namespace wtp;
#define WTP_API extern "C"

WTP_API int wtp_block_grid(int nranks, int p_out[3]) {
    if (nranks < 1 || !p_out) return WTP_ERR_ARG;
    long best = -1;
    for (int px = 1; px <= nranks; ++px) {
        if (nranks % px) continue;
        for (int py = px; py <= nranks / px; ++py) {
            if ((nranks / px) % py) continue;
            const int pz = nranks / (px * py);
            if (pz < py) continue;
            const long score = pz - px;
            if (best < 0 || score < best) {
                best = score;
                p_out[0] = px;
                p_out[1] = py;
                p_out[2] = pz;
            }
        }
    }
    return WTP_OK;
}

WTP_API int wtp_block_morton_rank(int ix, int iy, int iz, const int p[3]) {
    if (!p) return -1;
    bool pow2 = true;
    for (int a = 0; a < 3; ++a) pow2 = pow2 && p[a] >= 1 && (p[a] & (p[a] - 1)) == 0;
    if (!pow2) return (iz * p[1] + iy) * p[0] + ix;
    int idx[3] = {ix, iy, iz}, sizes[3] = {p[0], p[1], p[2]};
    int out = 0, bit = 0;
    while (sizes[0] > 1 || sizes[1] > 1 || sizes[2] > 1) {
        for (int a = 0; a < 3; ++a) {
            if (sizes[a] > 1) {
                out |= (idx[a] & 1) << bit;
                idx[a] >>= 1;
                sizes[a] >>= 1;
                ++bit;
            }
        }
    }
    return out;
}

WTP_API int wtp_block_set_transport(wtp_ctx* ctx, const wtp_transport* t) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->block) ctx->block = new BlockState();
    BlockState* b = bs_of(ctx);
    if (b->active) return fail(ctx, WTP_ERR_STATE, "wtp_block_set_transport while a block session is open");
    if (t) {
        if (!t->allgather || !t->exchange) return fail(ctx, WTP_ERR_ARG, "wtp_block_set_transport: both callbacks are needed");
        b->tr = *t;
        b->host_transport = true;
    } else {
        b->host_transport = false;
    }
    return WTP_OK;
}

WTP_API int wtp_block_close(wtp_ctx* ctx) {
    if (!ctx) return WTP_ERR_ARG;
    BlockState* b = bs_of(ctx);
    if (!b || !b->active) return WTP_OK;
    b->active = false;
    b->plan_ready = false;
    return wtp_relax_end(ctx);
}

WTP_API int wtp_block_open(wtp_ctx* ctx, const wtp_block_desc* desc, const void* d_owned_xyz, const int64_t* d_gid,
                           int64_t n_owned, const wtp_spacing_desc* spacing, const wtp_force_desc* force, int k,
                           double alpha_lo, double alpha_max) {
    if (!ctx) return WTP_ERR_ARG;
    if (!desc || !desc->boxes) return fail(ctx, WTP_ERR_ARG, "wtp_block_open: descriptor / boxes is NULL");
    if (desc->nranks < 1 || desc->rank < 0 || desc->rank >= desc->nranks) return fail(ctx, WTP_ERR_ARG, "wtp_block_open: 0 <= rank < nranks");
    if (!(desc->ghost_width > 0)) return fail(ctx, WTP_ERR_ARG, "wtp_block_open: ghost_width must be > 0");
    if (n_owned < 1 || !d_owned_xyz) return fail(ctx, WTP_ERR_ARG, "wtp_block_open: every rank needs at least one owned point");
    if (spacing && spacing->kind == WTP_SPACING_PER_POINT)
        return fail(ctx, WTP_ERR_ARG, "wtp_block_open: constant spacing or a device-evaluated law (not a PER_POINT array)");
    if (!ctx->block) ctx->block = new BlockState();
    BlockState* b = bs_of(ctx);
    if (b->active) return fail(ctx, WTP_ERR_STATE, "wtp_block_open: a block session is open already");
    if (const char* e = getenv("WTP_BLOCK_OVERLAP")) b->overlap = atoi(e) != 0; // (A/B switch)
    if (!b->host_transport && desc->nranks > 1 && (!ctx->comm || ctx->comm_size != desc->nranks || ctx->comm_rank != desc->rank))
        return fail(ctx, WTP_ERR_STATE, "wtp_block_open: wtp_comm_init (same rank / nranks) or wtp_block_set_transport first");
    for (int r = 0; r < desc->nranks; ++r)
        for (int ax = 0; ax < 3; ++ax)
            if (!(desc->boxes[(size_t)r * 6 + ax] < desc->boxes[(size_t)r * 6 + 3 + ax]))
                return fail(ctx, WTP_ERR_ARG, "wtp_block_open: every box needs lo < hi on every axis");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = wtp_relax_init_dev(ctx, d_owned_xyz, n_owned, 0, 3, WTP_F32, spacing, force, k, alpha_lo, alpha_max))) return rc;
    ctx->relax.shard_grid_reuse = true;
    b->rank = desc->rank;
    b->nranks = desc->nranks;
    b->boxes.assign(desc->boxes, desc->boxes + (size_t)desc->nranks * 6);
    b->w = desc->ghost_width;
    b->margin = desc->margin < 0 ? 0.25 * desc->ghost_width : desc->margin;
    b->widened = 0;
    b->n_owned = n_owned;
    b->n_ghost = 0;
    b->send_cap = b->mig_cap = 0;
    b->plan_ready = false;
    b->gid_cur = 0;
    b->info = wtp_block_info{};
    if ((rc = ensure(ctx, b->gid[0], sizeof(int64_t) * (size_t)n_owned))) return rc;
    if (d_gid) {
        WTP_HIP(ctx, hipMemcpyAsync(b->gid[0].p, d_gid, sizeof(int64_t) * (size_t)n_owned, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        hipLaunchKernelGGL(blk_iota_gid_kernel, dim3(1024), dim3(256), 0, ctx->stream, (int64_t*)b->gid[0].p, n_owned);
    }
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the caller's arrays are free again
    if ((rc = blk_geometry(ctx, b))) {
        wtp_relax_end(ctx);
        return rc;
    }
    b->active = true;
    return WTP_OK;
}

// second stream and the two events that tie it to the first (created on first use)
static int blk_streams(wtp_ctx* ctx) {
    if (!ctx->comm_stream) WTP_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    if (!ctx->ev_comm_a) WTP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_a, hipEventDisableTiming));
    if (!ctx->ev_comm_b) WTP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_b, hipEventDisableTiming));
    return WTP_OK;
}

// the exchange the plan describes: ghost rows into the pool, migrants into recv_mig; then the owned set and the ghosts
static int blk_exchange_and_apply(wtp_ctx* ctx, BlockState* b) {
    RelaxState& r = ctx->relax;
    const int np = b->geom.np;
    int64_t n_recv = 0, n_rmig = 0, n_send = 0, n_mig = 0;
    for (int q = 0; q < np; ++q) {
        n_recv += b->recv_cnt[q];
        n_rmig += b->rmig_cnt[q];
        n_send += b->send_cnt[q];
        n_mig += b->mig_cnt[q];
    }
    int rc;
    const int64_t n_pool = n_recv + n_mig; // peers' rows + my own emigrants
    if ((rc = ensure(ctx, b->pool, 16 * (size_t)(n_pool + 16)))) return rc;
    if ((rc = ensure(ctx, b->recv_mig, 32 * (size_t)(n_rmig + 16)))) return rc;
    // nobody crosses in this iteration (the usual case): the owned set stays, so its part of the rebuild can start early
    const bool early = b->overlap && np > 0 && n_mig == 0 && n_rmig == 0;
    // messages: per peer (ascending rank) ghosts, then migrants
    std::vector<int> peers;
    std::vector<const void*> sp;
    std::vector<void*> rp;
    std::vector<int64_t> sn, rn;
    int64_t so = 0, mo = 0, ro = 0, rmo = 0;
    for (int q = 0; q < np; ++q) {
        peers.push_back(b->peers[q]);
        sp.push_back((const char*)b->send.p + 16 * (size_t)so);
        sn.push_back(b->send_cnt[q]);
        rp.push_back((char*)b->pool.p + 16 * (size_t)ro);
        rn.push_back(b->recv_cnt[q]);
        peers.push_back(b->peers[q]);
        sp.push_back((const char*)b->send_mig.p + 32 * (size_t)mo);
        sn.push_back(2 * b->mig_cnt[q]);
        rp.push_back((char*)b->recv_mig.p + 32 * (size_t)rmo);
        rn.push_back(2 * b->rmig_cnt[q]);
        so += b->send_cnt[q];
        mo += b->mig_cnt[q];
        ro += b->recv_cnt[q];
        rmo += b->rmig_cnt[q];
    }
    if (np > 0) {
        if (b->host_transport) {
            // stage through the host: rows out, callback, rows in
            const size_t out_bytes = 16 * (size_t)n_send + 32 * (size_t)n_mig, in_bytes = 16 * (size_t)n_recv + 32 * (size_t)n_rmig;
            b->hbuf_a.resize(out_bytes + 16);
            b->hbuf_b.resize(in_bytes + 16);
            unsigned char* ha = b->hbuf_a.data();
            unsigned char* hb = b->hbuf_b.data();
            if (n_send) WTP_HIP(ctx, hipMemcpyAsync(ha, b->send.p, 16 * (size_t)n_send, hipMemcpyDeviceToHost, ctx->stream));
            if (n_mig) WTP_HIP(ctx, hipMemcpyAsync(ha + 16 * (size_t)n_send, b->send_mig.p, 32 * (size_t)n_mig, hipMemcpyDeviceToHost, ctx->stream));
            if (early) {
                // the host waits for the rows only; the stream goes on ranking the owned points under the callback
                if ((rc = blk_streams(ctx))) return rc;
                WTP_HIP(ctx, hipEventRecord(ctx->ev_comm_a, ctx->stream));
                if ((rc = relax_prerank(ctx, n_pool))) return rc;
                ctx->ev_last_end = -1;
                WTP_HIP(ctx, hipEventSynchronize(ctx->ev_comm_a));
                ctx->n_syncs += 1;
            } else if ((rc = blk_sync(ctx, b)))
                return rc;
            std::vector<const void*> hs(sp.size());
            std::vector<void*> hr(rp.size());
            std::vector<int64_t> sb(sp.size()), rb(rp.size());
            for (size_t j = 0; j < sp.size(); ++j) {
                const bool mig = j & 1;
                hs[j] = mig ? ha + 16 * (size_t)n_send + ((const char*)sp[j] - (const char*)b->send_mig.p)
                            : ha + ((const char*)sp[j] - (const char*)b->send.p);
                hr[j] = mig ? hb + 16 * (size_t)n_recv + ((char*)rp[j] - (char*)b->recv_mig.p) : hb + ((char*)rp[j] - (char*)b->pool.p);
                sb[j] = 16 * sn[j];
                rb[j] = 16 * rn[j];
            }
            if (b->tr.exchange(b->tr.user, (int)peers.size(), peers.data(), hs.data(), sb.data(), hr.data(), rb.data()) != 0)
                return fail(ctx, WTP_ERR_STATE, "wtp_block: the caller's exchange callback failed");
            if (n_recv) WTP_HIP(ctx, hipMemcpyAsync(b->pool.p, hb, 16 * (size_t)n_recv, hipMemcpyHostToDevice, ctx->stream));
            if (n_rmig) WTP_HIP(ctx, hipMemcpyAsync(b->recv_mig.p, hb + 16 * (size_t)n_recv, 32 * (size_t)n_rmig, hipMemcpyHostToDevice, ctx->stream));
            if ((rc = blk_sync(ctx, b))) return rc; // (the host buffers are reused)
        } else if (early) {
            // Exchange and compute overlap: the grouped round runs on the context's second stream, every peer on
            // its own link; the first stream ranks the owned points into the cells meanwhile (the first pass of the rebuild,
            // which does not need the ghosts) and waits for the rows only before it appends them.
            if ((rc = blk_streams(ctx))) return rc;
            WTP_HIP(ctx, hipEventRecord(ctx->ev_comm_a, ctx->stream)); // (send rows and the pool's last readers are in stream order before it)
            WTP_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_a, 0));
            if ((rc = comm_exchange_peers_on(ctx, ctx->comm_stream, (int)peers.size(), peers.data(), sp.data(), sn.data(), rp.data(),
                                             rn.data())))
                return rc;
            WTP_HIP(ctx, hipEventRecord(ctx->ev_comm_b, ctx->comm_stream));
            rc = relax_prerank(ctx, n_pool);
            WTP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_comm_b, 0)); // (also when the ranking failed: the streams join again)
            if (rc) return rc;
        } else {
            if ((rc = wtp_comm_exchange_peers(ctx, (int)peers.size(), peers.data(), sp.data(), sn.data(), rp.data(), rn.data()))) return rc;
        }
    }
    // my emigrants: ghosts for this iteration
    if (n_mig)
        hipLaunchKernelGGL(blk_emigrant_ghosts_kernel, dim3(64), dim3(256), 0, ctx->stream, (const float4*)b->send_mig.p, n_mig,
                           (float4*)b->pool.p + n_recv);
    // the owned set changes only when somebody crossed
    if (n_mig || n_rmig) {
        const int64_t n_stay = b->n_owned - n_mig, n_new = n_stay + n_rmig;
        if (n_new < 1) return fail(ctx, WTP_ERR_STATE, "wtp_block: every owned point of this rank migrated away");
        const int nxt = 1 - b->gid_cur;
        if ((rc = ensure(ctx, b->gid[nxt], sizeof(int64_t) * (size_t)(n_new + n_new / 8)))) return rc;
        const float4* P = (const float4*)ctx->pts[r.bufP].p;
        const int64_t n = r.n;
        const int32_t n_fixed = (int32_t)r.n_fixed;
        void* buf = nullptr;
        if ((rc = relax_swap_begin(ctx, n_new, &buf))) return rc;
        const int64_t nsp = blk_nspans(n);
        const int grid = (int)((nsp + kBlkWaves - 1) / kBlkWaves);
        hipLaunchKernelGGL(blk_compact_kernel, dim3(grid), dim3(64 * kBlkWaves), 0, ctx->stream, P, n, n_fixed, np,
                           (const uint32_t*)b->flags.p, (const int32_t*)b->span_counts.p, (const int64_t*)b->gid[b->gid_cur].p,
                           (float4*)buf, (int64_t*)b->gid[nxt].p);
        if (n_rmig)
            hipLaunchKernelGGL(blk_arrivals_kernel, dim3(64), dim3(256), 0, ctx->stream, (const float4*)b->recv_mig.p, n_rmig, n_stay,
                               (float4*)buf, (int64_t*)b->gid[nxt].p);
        WTP_HIP(ctx, hipGetLastError());
        if ((rc = relax_swap_commit(ctx, n_new))) return rc;
        b->gid_cur = nxt;
        b->n_owned = n_new;
    }
    b->info.n_emigrated = n_mig;
    b->info.n_immigrated = n_rmig;
    b->info.n_sent_rows = n_send;
    b->info.n_recv_rows = n_recv;
    b->n_ghost = n_pool;
    WTP_HIP(ctx, hipGetLastError());
    if (n_pool == 0 && r.n_fixed == 0) return WTP_OK; // no ghosts before, none now: the session is untouched (one rank: it IS the plain session)
    return relax_set_fixed_dev_impl(ctx, n_pool ? b->pool.p : nullptr, n_pool, true); // (the pool is ours: no wait for the copy)
}

WTP_API int wtp_block_step(wtp_ctx* ctx, wtp_step_stats* stats, wtp_block_info* info) {
    if (!ctx) return WTP_ERR_ARG;
    BlockState* b = bs_of(ctx);
    if (!b || !b->active) return fail(ctx, WTP_ERR_STATE, "wtp_block_step before wtp_block_open");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    const int64_t syncs0 = ctx->n_syncs, pre0 = ctx->preranked_builds;
    b->info.redone = 0;
    if ((rc = ensure(ctx, ctx->stats, sizeof(wtp_step_stats)))) return rc;
    wtp_step_stats g{};
    for (int attempt = 0;; ++attempt) {
        if (!b->plan_ready)
            if ((rc = blk_plan(ctx, b, nullptr, nullptr))) return rc; // first iteration / after a widening: one extra gather
        b->plan_ready = false;
        if ((rc = blk_exchange_and_apply(ctx, b))) return rc;
        if ((rc = relax_step_enqueue(ctx, 1, (wtp_step_stats*)ctx->stats.p))) return rc;
        // the next exchange's rows and counts, from the positions this sweep produced; they ride with its statistics
        if ((rc = blk_plan(ctx, b, (const wtp_step_stats*)ctx->stats.p, &g))) return rc;
        if (g.n_uncovered == 0) break;
        // some query's support reaches past the covered box somewhere: everybody undoes the step, widens and repeats it
        if (attempt >= 4) return fail(ctx, WTP_ERR_STATE, "wtp_block: queries still reach past the ghost layer after 4 widenings");
        if ((rc = wtp_relax_revert(ctx))) return rc;
        b->w *= 1.5;
        b->widened += 1;
        b->info.redone = 1;
        b->plan_ready = false;
        if ((rc = blk_geometry(ctx, b))) return rc;
    }
    // the closest pair's indices: global ids (the rank that holds it looks them up; the others cannot, so the pair is
    // reported as ids only where that is free: on the holder.  -1 elsewhere.)
    if (g.argmin_i >= 0) {
        g.argmin_i = -1;
        g.argmin_j = -1;
    }
    b->info.host_syncs = (int32_t)(ctx->n_syncs - syncs0);
    b->info.overlapped = (int32_t)(ctx->preranked_builds - pre0);
    b->info.reserved = 0;
    b->info.n_owned = b->n_owned;
    b->info.n_ghost = b->n_ghost;
    b->info.n_peers = b->geom.np;
    b->info.widened = b->widened;
    b->info.ghost_width = b->w;
    if (stats) *stats = g;
    if (info) *info = b->info;
    return WTP_OK;
}

WTP_API int wtp_block_run(wtp_ctx* ctx, int n_iters, double* conv_out, wtp_step_stats* last, wtp_block_info* info) {
    if (!ctx) return WTP_ERR_ARG;
    if (n_iters < 0) return fail(ctx, WTP_ERR_ARG, "n_iters must be >= 0");
    wtp_step_stats st{};
    wtp_block_info inf{};
    int syncs = 0;
    for (int i = 0; i < n_iters; ++i) {
        int rc = wtp_block_step(ctx, &st, &inf);
        if (rc) return rc;
        syncs += inf.host_syncs;
        if (conv_out) conv_out[i] = st.max_force;
    }
    inf.host_syncs = syncs;
    if (last) *last = st;
    if (info) *info = inf;
    return WTP_OK;
}

// The stop rules of `_relax!` (src/repel.jl:305-334) on the global statistics, in the reference's order: cv_target
// (positions reverted), stall_after on the CV of d_NN / s, tol on max |F| s.  Every rank evaluates the same numbers.
WTP_API int wtp_block_run_until(wtp_ctx* ctx, int max_iters, double tol, int stall_after, double cv_target, double* conv_out,
                                int* n_done, int* reason, wtp_step_stats* last) {
    if (!ctx) return WTP_ERR_ARG;
    if (max_iters < 0) return fail(ctx, WTP_ERR_ARG, "max_iters must be >= 0");
    BlockState* b = bs_of(ctx);
    if (!b || !b->active) return fail(ctx, WTP_ERR_STATE, "wtp_block_run_until before wtp_block_open");
    wtp_step_stats st{};
    int done = 0, why = 0, last_impr = 0;
    double best = std::numeric_limits<double>::infinity();
    for (int i = 1; i <= max_iters; ++i) {
        int rc = wtp_block_step(ctx, &st, nullptr);
        if (rc) return rc;
        done = i;
        if (conv_out) conv_out[i - 1] = st.max_force;
        const double nm = (double)st.n_move;
        if ((cv_target > 0 || stall_after > 0) && st.n_move > 0) {
            const double mu = st.sum_u / nm;
            double var = st.sum_u2 / nm - mu * mu;
            var = var > 0 ? var : 0;
            const double cv = std::sqrt(var) / mu;
            if (cv_target > 0 && cv <= cv_target) {
                // p .= p_old (src/repel.jl:314): the sweep is undone; the plan made from its positions is void
                if ((rc = wtp_relax_revert(ctx))) return rc;
                b->plan_ready = false;
                why = 2;
                break;
            }
            if (stall_after > 0) {
                if (cv < best * (1.0 - 1.0e-3)) {
                    best = cv;
                    last_impr = i;
                } else if (i - last_impr >= stall_after) {
                    why = 3;
                    break;
                }
            }
        }
        if (st.max_force < tol) {
            why = 1;
            break;
        }
    }
    if (n_done) *n_done = done;
    if (reason) *reason = why;
    if (last) *last = st;
    return WTP_OK;
}

WTP_API int wtp_block_get(wtp_ctx* ctx, void* d_xyz_out, int64_t* d_gid_out, int64_t cap, int64_t* n_owned) {
    if (!ctx) return WTP_ERR_ARG;
    BlockState* b = bs_of(ctx);
    if (!b || !b->active) return fail(ctx, WTP_ERR_STATE, "wtp_block_get before wtp_block_open");
    if (n_owned) *n_owned = b->n_owned;
    if (!d_xyz_out && !d_gid_out) return WTP_OK;
    if (cap < b->n_owned) return fail(ctx, WTP_ERR_ARG, "wtp_block_get: buffers too small");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (d_gid_out)
        WTP_HIP(ctx, hipMemcpyAsync(d_gid_out, b->gid[b->gid_cur].p, sizeof(int64_t) * (size_t)b->n_owned, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_xyz_out) {
        if ((rc = wtp_relax_get_dev(ctx, d_xyz_out))) return rc; // (synchronises)
    } else {
        WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return WTP_OK;
}
