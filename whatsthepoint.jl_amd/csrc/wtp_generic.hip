// wtp_generic.hip — exact thread-per-query search over the cell grid with ring expansion.
//
// This is the always-correct path: it serves (a) queries the 27-cell brick kernels hand back
// (k-th neighbour beyond the provable radius, LDS halo overflow, k too large for the register
// network), (b) sweeps against a stale snapshot (rebuild_every > 1, src/repel.jl:245,262-266)
// and (c) RadiusTopology (src/topology.jl:91-97), whose cell edge >= r makes 27 cells exact.
// It follows the same canonical (d2, index) order as the oracle, and sums forces in ascending
// order exactly like src/repel.jl:270-280.
#include "wtp_device.hpp"

namespace wtp {

static constexpr int kThreads = 128;

template <typename T, int MODE> // MODE 0: topology rows, 1: relax sweep
__global__ __launch_bounds__(kThreads) void generic_kernel(SearchArgs<T> a, const int32_t* __restrict__ list,
                                                           const int32_t* __restrict__ list_count,
                                                           int all, int part_base) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    __shared__ Acc sm_acc[kThreads / 64];
    const Grid<T> g = *a.grid;
    const int nq = all ? a.n : *list_count;
    const int K = a.k;
    const bool skip_self = (MODE == 0) && !a.include_self;
    T bd[kGenericKMax];
    int32_t bi[kGenericKMax];
    int32_t bp[kGenericKMax];
    Acc acc = acc_empty();

    for (int qi = blockIdx.x * blockDim.x + threadIdx.x; qi < nq; qi += gridDim.x * blockDim.x) {
        const int slot = all ? qi : list[qi];
        const Pt<T> q = a.query[slot];
        const int32_t id = w_to_id(q.w);
        if (MODE == 1 && id < a.n_fixed) {
            a.out[slot] = q;
            a.forces[slot] = (T)0;
            a.nn_dist[slot] = Lim<T>::inf();
            a.nn_id[slot] = -1;
            continue;
        }
        const int cx = cell_coord(g, q.x, 0), cy = cell_coord(g, q.y, 1), cz = cell_coord(g, q.z, 2);
        int m = 0;
        for (int r = 1;; r *= 2) {
            m = 0;
            const int z0 = cz - r < 0 ? 0 : cz - r, z1 = cz + r > g.n[2] - 1 ? g.n[2] - 1 : cz + r;
            const int y0 = cy - r < 0 ? 0 : cy - r, y1 = cy + r > g.n[1] - 1 ? g.n[1] - 1 : cy + r;
            const int x0 = cx - r < 0 ? 0 : cx - r, x1 = cx + r > g.n[0] - 1 ? g.n[0] - 1 : cx + r;
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y) {
                    const int row = (z * g.n[1] + y) * g.n[0];
                    const int ps = a.cell_start[row + x0], pe = a.cell_start[row + x1 + 1];
                    for (int p = ps; p < pe; ++p) {
                        const Pt<T> c = a.snap[p];
                        const int32_t cid = w_to_id(c.w);
                        if (skip_self && cid == id) continue;
                        const T d = dist2<T>(q.x, q.y, q.z, c.x, c.y, c.z);
                        if (m == K && !lex_lt(d, cid, bd[K - 1], bi[K - 1])) continue;
                        int pos = (m < K) ? m++ : K - 1;
                        while (pos > 0 && lex_lt(d, cid, bd[pos - 1], bi[pos - 1])) {
                            bd[pos] = bd[pos - 1];
                            bi[pos] = bi[pos - 1];
                            bp[pos] = bp[pos - 1];
                            --pos;
                        }
                        bd[pos] = d;
                        bi[pos] = cid;
                        bp[pos] = p;
                    }
                }
            const T g2 = safe_radius2(g, q.x, q.y, q.z, cx, cy, cz, r);
            if (g2 == Lim<T>::inf()) break;          // searched block covers the grid
            if (m == K && bd[K - 1] <= g2) break;    // k-th hit provably final
        }

        if (MODE == 0) {
            int32_t* row = a.idx_out + (int64_t)id * K;
            for (int j = 0; j < K; ++j) row[j] = j < m ? bi[j] : -1;
            if (a.dist_out) {
                T* drow = a.dist_out + (int64_t)id * K;
                for (int j = 0; j < K; ++j) drow[j] = j < m ? wsqrt(bd[j]) : Lim<T>::inf();
            }
        } else {
            const T s = a.spacing_pp ? a.spacing_pp[id] : a.spacing_const;
            T Fx = 0, Fy = 0, Fz = 0;
            int32_t nid = -1;
            T nd = Lim<T>::inf();
            for (int j = 0; j < m; ++j) { // ascending (d2, id): src/repel.jl:270-280
                if (bi[j] == id) continue;
                if (nid < 0) {
                    nid = bi[j];
                    nd = wsqrt(bd[j]);
                }
                const Pt<T> c = a.snap[bp[j]];
                add_force<T>(a, g.dim, s, q.x, q.y, q.z, id, c.x, c.y, c.z, bi[j], bd[j], Fx, Fy, Fz);
            }
            Pt<T> o;
            const T f = step_point<T>(a, s, q.x, q.y, q.z, Fx, Fy, Fz, o.x, o.y, o.z);
            o.w = q.w;
            a.out[slot] = o;
            a.forces[slot] = f;
            a.nn_dist[slot] = nd;
            a.nn_id[slot] = nid;
            acc_point<T>(acc, f, nd, s, id, nid);
            const T need = m == K ? bd[K - 1] : Lim<T>::inf(); // fewer than k local points: nothing is certain
            if (reaches_past_cover<T>(a, q.x, q.y, q.z, need)) atomicAdd(a.uncovered, 1);
        }
    }
    if (MODE == 1) {
        acc_block_reduce(acc, sm_acc);
        if (threadIdx.x == 0) acc_store(&a.partials[part_base + blockIdx.x], acc);
    }
}

static inline int blocks_for(int64_t n, int cap) {
    int64_t b = (n + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

// ---- arbitrary query positions against the current snapshot (wtp_relax_query_knn) -----------------
template <typename T>
__global__ void pack_queries_kernel(const T* __restrict__ xyz, int64_t nq, int dim, Pt<T>* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    Pt<T> p;
    p.x = xyz[i * dim];
    p.y = xyz[i * dim + 1];
    p.z = dim == 3 ? xyz[i * dim + 2] : (T)0;
    p.w = id_to_w((T)0, (int32_t)i); // row of the answer
    out[i] = p;
}

// a.n = number of queries, a.query = packed queries, a.snap / grid / cell_start = the snapshot searched
template <typename T> int launch_query_knn(wtp_ctx* ctx, SearchArgs<T>& a, const T* d_xyz, int dim, Pt<T>* d_packed) {
    hipLaunchKernelGGL(pack_queries_kernel<T>, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream, d_xyz,
                       (int64_t)a.n, dim, d_packed);
    a.query = d_packed;
    a.include_self = 1; // a query is not a snapshot point: nothing to drop
    hipLaunchKernelGGL((generic_kernel<T, 0>), dim3(blocks_for(a.n, 4096)), dim3(kThreads), 0, ctx->stream, a,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, 1, 0);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// ---- RadiusTopology: count, then fill rows sorted by (d2, id) ---------------------------------
template <typename T, bool FILL>
__global__ __launch_bounds__(kThreads) void radius_kernel(SearchArgs<T> a, T r, int32_t* __restrict__ counts,
                                                          const int64_t* __restrict__ offsets,
                                                          int32_t* __restrict__ idx_out, T* __restrict__ d2_tmp,
                                                          const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ list_count) {
    const Grid<T> g = *a.grid;
    const T r2 = r * r; // compared as d2 <= r*r, inclusive (inrange, src/topology.jl:93-94)
    const int nq = list ? *list_count : a.n;
    for (int qi = blockIdx.x * blockDim.x + threadIdx.x; qi < nq; qi += gridDim.x * blockDim.x) {
        const int slot = list ? list[qi] : qi;
        const Pt<T> q = a.snap[slot];
        const int32_t id = w_to_id(q.w);
        const int cx = cell_coord(g, q.x, 0), cy = cell_coord(g, q.y, 1), cz = cell_coord(g, q.z, 2);
        const int z0 = cz - 1 < 0 ? 0 : cz - 1, z1 = cz + 1 > g.n[2] - 1 ? g.n[2] - 1 : cz + 1;
        const int y0 = cy - 1 < 0 ? 0 : cy - 1, y1 = cy + 1 > g.n[1] - 1 ? g.n[1] - 1 : cy + 1;
        const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 > g.n[0] - 1 ? g.n[0] - 1 : cx + 1;
        int64_t base = FILL ? offsets[id] : 0;
        int64_t cap = FILL ? offsets[id + 1] - base : 0;
        int32_t m = 0;
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const int row = (z * g.n[1] + y) * g.n[0];
                const int ps = a.cell_start[row + x0], pe = a.cell_start[row + x1 + 1];
                for (int p = ps; p < pe; ++p) {
                    const Pt<T> c = a.snap[p];
                    const int32_t cid = w_to_id(c.w);
                    if (cid == id) continue; // filter(!=(i), n), src/topology.jl:96
                    const T d = dist2<T>(q.x, q.y, q.z, c.x, c.y, c.z);
                    if (!(d <= r2)) continue;
                    if (FILL) {
                        if (m >= cap) continue;
                        int64_t pos = m;
                        while (pos > 0 && lex_lt(d, cid, d2_tmp[base + pos - 1], idx_out[base + pos - 1])) {
                            d2_tmp[base + pos] = d2_tmp[base + pos - 1];
                            idx_out[base + pos] = idx_out[base + pos - 1];
                            --pos;
                        }
                        d2_tmp[base + pos] = d;
                        idx_out[base + pos] = cid;
                    }
                    ++m;
                }
            }
        if (!FILL) counts[id] = m;
    }
}


// Exact path = wave-per-query kernel over the list (or all points), then this serial kernel over
// whatever the wave kernel could not buffer (fb2 list).  WTP_FORCE_GENERIC=2 runs everything
// through the serial kernel (debug).
template <typename T> int launch_generic_topology(wtp_ctx* ctx, SearchArgs<T>& a, bool all) {
    if (!a.counters_cleared) WTP_HIP(ctx, hipMemsetAsync(a.fb2_count, 0, sizeof(int32_t), ctx->stream));
    if (ctx->force_generic == 2) {
        hipLaunchKernelGGL((generic_kernel<T, 0>), dim3(blocks_for(a.n, all ? 65536 : 1024)), dim3(kThreads), 0,
                           ctx->stream, a, a.fb_list, a.fb_count, all ? 1 : 0, 0);
        WTP_HIP(ctx, hipGetLastError());
        return WTP_OK;
    }
    int rc = launch_wave_topology<T>(ctx, a, all);
    if (rc) return rc;
    hipLaunchKernelGGL((generic_kernel<T, 0>), dim3(256), dim3(kThreads), 0, ctx->stream, a, a.fb2_list,
                       a.fb2_count, 0, 0);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// Partials layout: [0, n_partials - kGenericPartials) belong to the brick kernel, the tail to
// this one.

template <typename T> int launch_generic_sweep(wtp_ctx* ctx, SearchArgs<T>& a, bool all) {
    const int part_base = a.n_partials - kGenericPartials;
    // counters were cleared by the caller (one block for fb_count / fb2_count / uncovered)
    if (ctx->force_generic == 2) {
        a.used_generic = blocks_for(a.n, kGenericPartials);
        hipLaunchKernelGGL((generic_kernel<T, 1>), dim3(a.used_generic), dim3(kThreads), 0, ctx->stream, a, a.fb_list,
                           a.fb_count, all ? 1 : 0, part_base);
        WTP_HIP(ctx, hipGetLastError());
        return WTP_OK;
    }
    int rc = launch_wave_sweep<T>(ctx, a, all);
    if (rc) return rc;
    // serial last resort over what the wave kernel could not buffer: a handful of queries at most
    a.used_generic = blocks_for(a.n / 16, 256);
    hipLaunchKernelGGL((generic_kernel<T, 1>), dim3(a.used_generic), dim3(kThreads), 0, ctx->stream, a, a.fb2_list,
                       a.fb2_count, 0, part_base);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// The brick kernel serves fp32 clouds (WTP_FORCE_GENERIC switches it off); the caller provides the hand-back list.
template <typename T> static bool brick_radius_usable(wtp_ctx*, SearchArgs<T>&) { return false; }
template <> bool brick_radius_usable<float>(wtp_ctx* ctx, SearchArgs<float>& a) {
    return !ctx->force_generic && a.fb_list != nullptr && a.fb_count != nullptr;
}
template <typename T> static int brick_radius(wtp_ctx*, SearchArgs<T>&) { return WTP_OK; }
template <> int brick_radius<float>(wtp_ctx* ctx, SearchArgs<float>& a) { return launch_brick_radius(ctx, a); }

// Fill phase when the count phase parked the brick kernel's rows: one thread per query copies its row (at most 32 ids,
// already in canonical order) to its place in the CSR; queries the brick kernel handed back are filled by the wave kernel.
__global__ void radius_copy_rows_kernel(int64_t n, const int32_t* __restrict__ tmp, const uint8_t* __restrict__ done,
                                        const int32_t* __restrict__ arena, const int64_t* __restrict__ arena_off,
                                        const int64_t* __restrict__ offsets, int32_t* __restrict__ idx) {
    // sixteen lanes per row: rows of the wave kernel's arena run to hundreds of ids.  A row is a chain (mark -> offsets and
    // place in the arena -> ids), and the kernel's time was the sum of those chains: the marks and offsets of the group's
    // NEXT row are asked for before this row's ids are copied.
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int l = threadIdx.x & 15;
    auto meta = [&](int64_t q, uint8_t& d, int64_t& o, int64_t& len, int64_t& ao) {
        d = 0;
        o = len = ao = 0;
        if (q < n) {
            d = done[q];
            o = offsets[q];
            len = offsets[q + 1] - o;
            ao = arena_off ? arena_off[q] : 0; // (read for every row: waiting for the mark first is the chain this avoids)
        }
    };
    uint8_t d, dn;
    int64_t o, len, ao, on, lenn, aon;
    meta(grp, d, o, len, ao);
    for (int64_t q = grp; q < n; q += ngrp) {
        meta(q + ngrp, dn, on, lenn, aon);
        if (d) {
            const int32_t* row;
            int64_t cnt = len;
            if (d == 1) {
                row = tmp + q * 32;
                cnt = cnt > 32 ? 32 : cnt;
            } else {
                row = arena + ao;
            }
            for (int64_t j0 = l; j0 < cnt; j0 += 64) { // four loads of the row in flight, then the four stores
                int32_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = j0 + 16 * u < cnt ? row[j0 + 16 * u] : 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + 16 * u < cnt) idx[o + j0 + 16 * u] = v[u];
            }
        }
        d = dn;
        o = on;
        len = lenn;
        ao = aon;
    }
}

template <typename T>
int launch_radius_count(wtp_ctx* ctx, SearchArgs<T>& a, T r, int32_t* d_counts) {
    if (ctx->force_generic == 2) {
        hipLaunchKernelGGL((radius_kernel<T, false>), dim3(blocks_for(a.n, 65536)), dim3(kThreads), 0, ctx->stream,
                           a, r, d_counts, (const int64_t*)nullptr, (int32_t*)nullptr, (T*)nullptr,
                           (const int32_t*)nullptr, (const int32_t*)nullptr);
        WTP_HIP(ctx, hipGetLastError());
        return WTP_OK;
    }
    // the brick-staged wave-per-query kernel (wtp_radb.hip) needs somewhere to park its rows and a list to hand back to
    const bool dense = !ctx->force_generic && a.rad_done && a.rad_arena && a.rad_arena_pos && a.rad_bricks && a.fb_list && a.fb_count &&
                       !(getenv("WTP_RADIUS_DENSE") && atoi(getenv("WTP_RADIUS_DENSE")) == 0); // (A/B switch)
    a.rad_dense = dense ? radius_dense_hcap<T>() : 0;
    if (brick_radius_usable<T>(ctx, a) || dense) {
        // fp32: LDS-staged brick kernel first (lane per query, rows up to 32 entries); then the dense bricks (fp64: all bricks),
        // wave per query from LDS; the wave kernel takes what they hand back
        WTP_HIP(ctx, hipMemsetAsync(a.fb_count, 0, sizeof(int32_t), ctx->stream));
        int rc;
        if (brick_radius_usable<T>(ctx, a)) {
            a.radius2 = r * r;
            a.rad_counts = d_counts;
            a.rad_offsets = nullptr;
            a.rad_fill = 0;
            if ((rc = brick_radius<T>(ctx, a))) return rc;
        }
        if (dense && (rc = launch_radius_dense<T>(ctx, a, r, d_counts))) return rc;
        return launch_wave_radius_count<T>(ctx, a, r, d_counts, a.fb_list, a.fb_count);
    }
    return launch_wave_radius_count<T>(ctx, a, r, d_counts, nullptr, nullptr);
}

// Rows are ranked by the wave kernel; rows longer than its LDS list (fb2 list) are insertion-
// sorted in place by the serial kernel, which needs a d2 scratch row (ctx->scratch).
template <typename T>
int launch_radius_fill(wtp_ctx* ctx, SearchArgs<T>& a, T r, const int64_t* d_offsets, int32_t* d_idx) {
    if (ctx->force_generic == 2) {
        hipLaunchKernelGGL((radius_kernel<T, true>), dim3(blocks_for(a.n, 65536)), dim3(kThreads), 0, ctx->stream,
                           a, r, (int32_t*)nullptr, d_offsets, d_idx, (T*)ctx->scratch.p, (const int32_t*)nullptr,
                           (const int32_t*)nullptr);
        WTP_HIP(ctx, hipGetLastError());
        return WTP_OK;
    }
    WTP_HIP(ctx, hipMemsetAsync(a.fb2_count, 0, sizeof(int32_t), ctx->stream));
    int rc;
    if (a.rad_done && (a.rad_arena || (brick_radius_usable<T>(ctx, a) && a.rad_tmp))) {
        // the count phase parked the rows it found: copy them.  What it could not park (arena full, rows beyond the wave
        // kernel's list) is searched again: by the wave kernel over the brick kernel's hand-back list, which still stands
        // (fp32), or over all points (it skips the parked ones), then by the serial kernel
        hipLaunchKernelGGL(radius_copy_rows_kernel, dim3(blocks_for((int64_t)a.n * 16, 16384)), dim3(kThreads), 0, ctx->stream,
                           (int64_t)a.n, (const int32_t*)a.rad_tmp, (const uint8_t*)a.rad_done, (const int32_t*)a.rad_arena,
                           (const int64_t*)a.rad_arena_off, d_offsets, d_idx);
        if ((brick_radius_usable<T>(ctx, a) && a.rad_tmp) || a.rad_dense)
            rc = launch_wave_radius_fill<T>(ctx, a, r, d_offsets, d_idx, a.fb_list, a.fb_count);
        else
            rc = launch_wave_radius_fill<T>(ctx, a, r, d_offsets, d_idx, nullptr, nullptr);
    } else if (brick_radius_usable<T>(ctx, a)) {
        WTP_HIP(ctx, hipMemsetAsync(a.fb_count, 0, sizeof(int32_t), ctx->stream));
        a.radius2 = r * r;
        a.rad_counts = nullptr;
        a.rad_offsets = d_offsets;
        a.rad_fill = 1;
        a.idx_out = d_idx;
        if ((rc = brick_radius<T>(ctx, a))) return rc;
        rc = launch_wave_radius_fill<T>(ctx, a, r, d_offsets, d_idx, a.fb_list, a.fb_count);
    } else {
        rc = launch_wave_radius_fill<T>(ctx, a, r, d_offsets, d_idx, nullptr, nullptr);
    }
    if (rc) return rc;
    hipLaunchKernelGGL((radius_kernel<T, true>), dim3(256), dim3(kThreads), 0, ctx->stream, a, r, (int32_t*)nullptr,
                       d_offsets, d_idx, (T*)ctx->scratch.p, (const int32_t*)a.fb2_list, (const int32_t*)a.fb2_count);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

#define INST(T)                                                                              \
    template int launch_generic_topology<T>(wtp_ctx*, SearchArgs<T>&, bool);                 \
    template int launch_query_knn<T>(wtp_ctx*, SearchArgs<T>&, const T*, int, Pt<T>*);         \
    template int launch_generic_sweep<T>(wtp_ctx*, SearchArgs<T>&, bool);                    \
    template int launch_radius_count<T>(wtp_ctx*, SearchArgs<T>&, T, int32_t*);              \
    template int launch_radius_fill<T>(wtp_ctx*, SearchArgs<T>&, T, const int64_t*, int32_t*);
INST(float)
INST(double)
#undef INST

} // namespace wtp
