// wtp_brick64.hip — compact-support repel sweep for fp64 clouds (gfx950).
//
// Float64 is the reference's default machine type, so its sweeps deserve more than the
// wave-per-query exact path (10 ns per query).  This is the fp64 sibling of brick_kernel<1,0,1>
// (wtp_brick.hip): same brick / halo / per-lane-query structure, same certificate —
//   ClippedSpacingForce is exactly 0 beyond u0*s (src/repel_forces.jl:96-100); if the ball of that
//   radius holds n_lim <= k points (self included) they are among the k nearest, so the sum over
//   them is the reference's sum over its k-list (src/repel.jl:270-280) —
// but it keeps the reference's summation ORDER as well: the few in-support neighbours (4-5 on
// average, capacity 16) are sorted by the canonical (d2, index) in registers and their IEEE force
// terms are added in that order, so fp64 results stay bit-identical to the oracle's sequential
// evaluation, like the wave kernel's.  Per candidate the scan filters with a fused-multiply-add d2
// against a threshold 4 ulp wider; every survivor's canonical d2 is recomputed before use.
// Queries it cannot certify (support not covered by the 27 cells, n_lim > k, no neighbour in
// reach, more than 16 in-support neighbours, ring or LDS overflow) go to the wave kernel's list.
#include "wtp_device.hpp"
#include "wtp_sortnet.hpp"

namespace wtp {

constexpr int kB64Threads = 256;
constexpr int kB64Ring = 32;   // ring rows per lane (+1 dump row)
constexpr int kB64SU = 2;      // candidates per scan step
constexpr int kB64In = 16;     // in-support neighbours sorted in registers
constexpr int kB64OwnRows = BY * BZ;

struct Brick64Smem {
    int hstart[HCELLS + 1];
    int hglobal[HCELLS];
    int own_pref[kB64OwnRows + 1];
    int scan_tmp[kB64Threads / 64 + 1];
    Acc acc[kB64Threads / 64];
};

// compare-exchange on the (d2, id, lds index) triples: ascending canonical (d2, id)
#define WTP_CE(k_unused, I, J)                                  \
    {                                                           \
        const bool sw_ = lex_lt(kd[J], ki[J], kd[I], ki[I]);    \
        const T td_ = sw_ ? kd[J] : kd[I];                      \
        const T ud_ = sw_ ? kd[I] : kd[J];                      \
        const int32_t ti_ = sw_ ? ki[J] : ki[I];                \
        const int32_t ui_ = sw_ ? ki[I] : ki[J];                \
        const int32_t tx_ = sw_ ? kx[J] : kx[I];                \
        const int32_t ux_ = sw_ ? kx[I] : kx[J];                \
        kd[I] = td_;                                            \
        kd[J] = ud_;                                            \
        ki[I] = ti_;                                            \
        ki[J] = ui_;                                            \
        kx[I] = tx_;                                            \
        kx[J] = ux_;                                            \
    }

__device__ inline float fma_t64(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double fma_t64(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> struct WideEps;
template <> struct WideEps<float> { static constexpr float v = 1.f + 0x1p-21f; };
template <> struct WideEps<double> { static constexpr double v = 1.0 + 0x1p-50; };

static __host__ __device__ size_t b64_smem_bytes(int hcap, size_t ptsz) {
    return (size_t)(hcap + kB64SU) * ptsz + (size_t)(kB64Ring + 1) * kB64Threads * sizeof(uint16_t) + sizeof(Brick64Smem);
}

template <typename T>
__global__ __launch_bounds__(kB64Threads, 2) void brick_cs_kernel(SearchArgs<T> a, int hcap) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Pt<T>* pts = reinterpret_cast<Pt<T>*>(smem_raw);
    uint16_t* ring_all = reinterpret_cast<uint16_t*>(smem_raw + (size_t)(hcap + kB64SU) * sizeof(Pt<T>));
    Brick64Smem* sm = reinterpret_cast<Brick64Smem*>(smem_raw + (size_t)(hcap + kB64SU) * sizeof(Pt<T>) +
                                                     (size_t)(kB64Ring + 1) * kB64Threads * sizeof(uint16_t));
    const int tid = threadIdx.x;
    uint16_t* ring = ring_all + tid; // entry j at ring[j * kB64Threads]; row kB64Ring is the dump row
    const Grid<T> g = *a.grid;
    const int K = a.k;
    Acc acc = acc_empty();
    for (int j = 0; j <= kB64Ring; ++j) ring[j * kB64Threads] = 0; // masked reads use row 0: keep it a valid index
    if (tid < kB64SU) { // padding past the staged points: finite coordinates, never taken (masked by index)
        Pt<T> z;
        z.x = z.y = z.z = (T)0;
        z.w = id_to_w((T)0, -1);
        pts[hcap + tid] = z;
    }

    // XCD-aware brick order: blocks sharing blockIdx % 8 share an L2 (same scheme as brick_kernel)
    const int groups = 8;
    const int per = (g.nbricks + groups - 1) / groups;
    const int xcd = blockIdx.x % groups;
    const int lane_blk = blockIdx.x / groups;
    const int blk_per_group = gridDim.x / groups;
    const int b_end = (xcd + 1) * per < g.nbricks ? (xcd + 1) * per : g.nbricks;

    for (int brick = xcd * per + lane_blk; brick < b_end; brick += blk_per_group) {
        const int bx = brick % g.nb[0];
        const int by = (brick / g.nb[0]) % g.nb[1];
        const int bz = brick / (g.nb[0] * g.nb[1]);
        const int ox = bx * BX - 1, oy = by * BY - 1, oz = bz * BZ - 1;

        __syncthreads();
        // ---- 1. halo cell table -------------------------------------------------------------
        int my_cnt = 0;
        if (tid < HCELLS) {
            const int hx = tid % HX, hy = (tid / HX) % HY, hz = tid / (HX * HY);
            const int gx = ox + hx, gy = oy + hy, gz = oz + hz;
            int gs = 0;
            if (gx >= 0 && gx < g.n[0] && gy >= 0 && gy < g.n[1] && gz >= 0 && gz < g.n[2]) {
                const int cell = (gz * g.n[1] + gy) * g.n[0] + gx;
                gs = a.cell_start[cell];
                my_cnt = a.cell_start[cell + 1] - gs;
            }
            sm->hglobal[tid] = gs;
        }
        {
            int incl = my_cnt;
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            if (lane == 63) sm->scan_tmp[wave] = incl;
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int w = 0; w < kB64Threads / 64; ++w) {
                    const int t = sm->scan_tmp[w];
                    sm->scan_tmp[w] = run;
                    run += t;
                }
                sm->scan_tmp[kB64Threads / 64] = run;
            }
            __syncthreads();
            const int ex = incl - my_cnt + sm->scan_tmp[wave];
            if (tid < HCELLS) sm->hstart[tid] = ex;
            if (tid == HCELLS) sm->hstart[HCELLS] = sm->scan_tmp[kB64Threads / 64];
        }
        __syncthreads();
        const int halo_total = sm->hstart[HCELLS];
        const bool overflow = halo_total > hcap;

        // ---- 2. own-row prefix ----------------------------------------------------------------
        if (tid == 0) {
            int run = 0;
            for (int r = 0; r < kB64OwnRows; ++r) {
                const int hy = 1 + r % BY, hz = 1 + r / BY;
                const int base = (hz * HY + hy) * HX;
                sm->own_pref[r] = run;
                run += sm->hstart[base + 1 + BX] - sm->hstart[base + 1];
            }
            sm->own_pref[kB64OwnRows] = run;
        }
        // ---- 3. stage halo points: every x-row of HX cells is one contiguous global run -------
        if (!overflow) {
            const int wave = tid >> 6, lane = tid & 63;
            const int hx_lo = ox < 0 ? -ox : 0;
            for (int row = wave; row < HY * HZ; row += kB64Threads / 64) {
                const int base = row * HX;
                const int ls = sm->hstart[base];
                const int len = sm->hstart[base + HX] - ls;
                if (len <= 0) continue;
                const int gs = sm->hglobal[base + hx_lo];
                for (int i = lane; i < len; i += 64) pts[ls + i] = a.snap[gs + i];
            }
        }
        __syncthreads();

        // ---- 4. queries -------------------------------------------------------------------------
        const int Q = sm->own_pref[kB64OwnRows];
        for (int qb = 0; qb < Q; qb += kB64Threads) {
            const int qn = qb + tid;
            const bool active = qn < Q;
            int r = 0;
            if (active) {
#pragma unroll
                for (int t = 1; t < kB64OwnRows; ++t) r += (sm->own_pref[t] <= qn) ? 1 : 0;
            }
            const int hy0 = 1 + r % BY, hz0 = 1 + r / BY;
            const int rbase = (hz0 * HY + hy0) * HX + 1;
            const int off = qn - sm->own_pref[r];
            const int gslot = sm->hglobal[rbase] + off;
            if (!active) continue;
            if (overflow) {
                const int pos = atomicAdd(a.fb_count, 1);
                a.fb_list[pos] = gslot;
                continue;
            }
            const Pt<T> qp = pts[sm->hstart[rbase] + off];
            const int32_t qid = w_to_id(qp.w);
            if (qid < a.n_fixed) { // the wall: never moves (src/repel.jl:80,256)
                a.out[gslot] = qp;
                a.forces[gslot] = (T)0;
                a.nn_dist[gslot] = Lim<T>::inf();
                a.nn_id[gslot] = -1;
                continue;
            }
            const int cx = cell_coord(g, qp.x, 0), cy = cell_coord(g, qp.y, 1), cz = cell_coord(g, qp.z, 2);
            const int hx = cx - ox, hy = cy - oy, hz = cz - oz;
            const T g2 = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, 1);
            const T s = a.spacing_pp ? a.spacing_pp[qid] : a.spacing_const;
            const T lim = (a.u0 * a.u0) * (s * s);
            T tau = g2;
            const bool cs_fail = !(lim <= g2);            // the 27 cells do not certify the support
            const T tnn = (a.tnn_frac * g.c) * (a.tnn_frac * g.c); // margin that (almost) always holds the nearest neighbour
            const T tcs = lim > tnn ? lim : tnn;
            tau = tcs < tau ? tcs : tau;
            T tau_s = tau * WideEps<T>::v;
            bool giveup = false;
            int ra = 0; // ring entries

            // ---- scan: trimmed per-lane rows (same scheme as brick_kernel<1,0,1>, WTP_LANE_ROWS=2) -----------
            // A row (dy, dz) lies at least gy, gz from the query (distance to its own cell's faces less the
            // cell map's rounding margin): only candidates within rem = tau - gy^2 - gz^2 along x can pass the
            // filter, so the row is dropped when rem < 0 and its end cells when the query is farther than
            // sqrt(rem) from the own cell's x faces.  The surviving runs are queued as start | end << 16
            // (LDS point indices) in nine registers; a lane takes its next run by a register shift.
            uint32_t rq[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            int rn = 0;
            {
                const T cxl = g.org[0] + (T)cx * g.c, cyl = g.org[1] + (T)cy * g.c, czl = g.org[2] + (T)cz * g.c;
                auto gap2 = [&](T d) {
                    const T t = d - g.margin;
                    return t > (T)0 ? t * t : (T)0;
                };
                const T lx2 = gap2(qp.x - cxl), ux2 = gap2(cxl + g.c - qp.x);
                const T ly2 = gap2(qp.y - cyl), uy2 = gap2(cyl + g.c - qp.y);
                const T lz2 = gap2(qp.z - czl), uz2 = gap2(czl + g.c - qp.z);
                const int base0 = (hz * HY + hy) * HX + (hx - 1);
                uint32_t st[9], en[9];
#pragma unroll
                for (int r9 = 0; r9 < 9; ++r9) {
                    const int dzr = r9 / 3, dyr = r9 % 3;
                    const T gy2 = dyr == 0 ? ly2 : (dyr == 2 ? uy2 : (T)0);
                    const T gz2 = dzr == 0 ? lz2 : (dzr == 2 ? uz2 : (T)0);
                    const T rem = tau_s - gy2 - gz2;
                    const int rb = base0 + (dzr - 1) * (HY * HX) + (dyr - 1) * HX;
                    const int i0 = rb + (lx2 <= rem ? 0 : 1), i1 = rb + (ux2 <= rem ? 3 : 2);
                    st[r9] = (uint32_t)sm->hstart[i0];
                    en[r9] = rem < (T)0 ? st[r9] : (uint32_t)sm->hstart[i1];
                }
#pragma unroll
                for (int r9 = 0; r9 < 9; ++r9) {
                    const bool put = en[r9] > st[r9];
                    const uint32_t pk = st[r9] | (en[r9] << 16);
#pragma unroll
                    for (int u9 = 0; u9 < 9; ++u9) rq[u9] = (put && rn == u9) ? pk : rq[u9];
                    rn += put ? 1 : 0;
                }
            }
            int pa = 0, ea = 0;
            auto advance = [&]() {
                if (pa >= ea && rn > 0) {
                    const uint32_t pk = rq[0];
#pragma unroll
                    for (int u9 = 0; u9 < 8; ++u9) rq[u9] = rq[u9 + 1];
                    --rn;
                    pa = (int)(pk & 0xFFFFu);
                    ea = (int)(pk >> 16);
                }
            };
            advance();
            while (__any(pa < ea)) {
                if (__any(ra > kB64Ring - kB64SU)) { // some lane's ring is full (dense cluster): it gives up
                    if (ra > kB64Ring - kB64SU) {
                        giveup = true;
                        ra = 0;
                        tau_s = (T)-1;
                    }
                }
#pragma unroll
                for (int u = 0; u < kB64SU; ++u) {
                    const Pt<T> c = pts[pa + u]; // past the run end: the next cell or the padding, masked below
                    const T ex = qp.x - c.x, ey = qp.y - c.y, ez = qp.z - c.z;
                    const T d = fma_t64(ez, ez, fma_t64(ey, ey, ex * ex));
                    const bool take = (pa + u < ea) && (d <= tau_s);
                    ring[(take ? ra : kB64Ring) * kB64Threads] = (uint16_t)(pa + u);
                    ra += take ? 1 : 0;
                }
                pa += kB64SU;
                advance();
            }
            const int cnt = ra;
            bool fallback = giveup || cs_fail;

            // ---- pass over the ring: canonical d2, nearest neighbour, the in-support set -----------------
            T kd[kB64In];
            int32_t ki[kB64In], kx[kB64In];
#pragma unroll
            for (int u = 0; u < kB64In; ++u) {
                kd[u] = Lim<T>::inf();
                ki[u] = 0x7FFFFFFF;
                kx[u] = 0;
            }
            int n_lim = 0, n_in = 0;
            int32_t nid = 0x7FFFFFFF;
            T nd2 = Lim<T>::inf();
            for (int j = 0; __any(!fallback && j < cnt); ++j) {
                const int idx = ring[(j < cnt ? j : 0) * kB64Threads];
                const Pt<T> c = pts[idx];
                const T d = dist2<T>(qp.x, qp.y, qp.z, c.x, c.y, c.z);
                const int32_t cid = w_to_id(c.w);
                const bool valid = !fallback && j < cnt && d <= tau;
                const bool other = valid && cid != qid; // self skipped by index (src/repel.jl:271)
                const bool nearer = other && lex_lt(d, cid, nd2, nid);
                nd2 = nearer ? d : nd2;
                nid = nearer ? cid : nid;
                const bool in_sup = valid && d <= lim;
                n_lim += in_sup ? 1 : 0;
                const bool put = in_sup && other;
#pragma unroll
                for (int u = 0; u < kB64In; ++u) {
                    const bool here = put && n_in == u;
                    kd[u] = here ? d : kd[u];
                    ki[u] = here ? cid : ki[u];
                    kx[u] = here ? idx : kx[u];
                }
                n_in += put ? 1 : 0;
            }
            fallback = fallback || n_lim > K || nid == 0x7FFFFFFF || n_in > kB64In;
            if (fallback) {
                const int pos = atomicAdd(a.fb_count, 1);
                a.fb_list[pos] = gslot;
                continue;
            }
            // ---- canonical order, then the reference's sequential sum (src/repel.jl:270-280) --------------
            WTP_SORTNET_16(k)
            T Fx = 0, Fy = 0, Fz = 0;
#pragma unroll
            for (int u = 0; u < kB64In; ++u) {
                if (__any(u < n_in)) {
                    if (u < n_in) {
                        const Pt<T> c = pts[kx[u]];
                        add_force<T>(a, g.dim, s, qp.x, qp.y, qp.z, qid, c.x, c.y, c.z, ki[u], kd[u], Fx, Fy, Fz);
                    }
                }
            }
            Pt<T> o;
            const T f = step_point<T>(a, s, qp.x, qp.y, qp.z, Fx, Fy, Fz, o.x, o.y, o.z);
            o.w = qp.w;
            const T nd = wsqrt(nd2);
            const T need = nd2 > lim ? nd2 : lim; // sharded sessions: what the answer rests on
            if (reaches_past_cover<T>(a, qp.x, qp.y, qp.z, need)) atomicAdd(a.uncovered, 1);
            a.out[gslot] = o;
            a.forces[gslot] = f;
            a.nn_dist[gslot] = nd;
            a.nn_id[gslot] = nid;
            acc_point<T>(acc, f, nd, s, qid, nid);
        }
    }
    __syncthreads();
    acc_block_reduce(acc, sm->acc);
    if (tid == 0) acc_store(&a.partials[blockIdx.x], acc);
}

template <typename T> int launch_brick_cs(wtp_ctx* ctx, SearchArgs<T>& a) {
    int hcap = a.brick_hcap > 0 ? a.brick_hcap : 1280;
    if (hcap > 2047) hcap = 2047; // ring entries are 16-bit LDS point indices, and two workgroups share a CU's LDS
    const size_t smem = b64_smem_bytes(hcap, sizeof(Pt<T>));
    int occ = launch_occupancy_of(ctx, (const void*)brick_cs_kernel<T>, kB64Threads, smem);
    if (occ > 4) occ = 4;
    int gsz = ctx->sm_count * occ;
    gsz -= gsz % 8;
    if (gsz < 8) gsz = 8;
    if (gsz > brick_partials()) gsz = brick_partials() - brick_partials() % 8;
    a.used_brick = gsz;
    hipLaunchKernelGGL((brick_cs_kernel<T>), dim3(gsz), dim3(kB64Threads), smem, ctx->stream, a, hcap);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template int launch_brick_cs<double>(wtp_ctx*, SearchArgs<double>&);

} // namespace wtp
