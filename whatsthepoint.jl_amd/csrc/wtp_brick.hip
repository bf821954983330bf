// wtp_brick.hip — the hot kernels: 27-cell k-NN with the candidates staged in LDS (gfx950, fp32).
//
// One 256-thread workgroup sweeps a brick of BX x BY x BZ cells.  It stages the brick's
// (BX+2)(BY+2)(BZ+2) halo cells from the sorted Pt array into LDS (one ds_read_b128 per
// candidate afterwards), then every lane owns one query at a time:
//   scan      the 9 x-rows of its 3x3x3 neighbourhood, 4 candidates per step, straight-line:
//             candidates with d2 <= tau are appended (16-bit LDS slot) to a per-lane ring in LDS
//             through a branch-free "dump row" store.  tau starts at
//             min(provable radius, gamma_cap * cell)^2, so whatever is kept is exact.
//   select    ring keys (d2 bit patterns, monotone for d2 >= 0) go through a 64-wide Batcher
//             network held in VGPRs (integer min/max); the k-th key is the cut.  With k known
//             at compile time (k = 21, the reference default) dead comparators are eliminated.
//             The same code instance serves ring-pressure prunes and the final selection.
//   topology  MODE 0: survivors (d2 <= cut) are compacted and a 32-wide network on 64-bit keys
//             (d2 bits << 32 | id) gives the canonical order; the first k are the row of the
//             query's original id (_build_knn_neighbors, src/topology.jl:79-84).
//   sweep     MODE 1: the Miotti force of src/repel.jl:270-291 is accumulated over the cut set,
//             the point is stepped, and max|F|s, sum u, sum u^2 and the closest pair are reduced
//             per wave with shuffles (src/repel.jl:293,374-403).
// Queries the fast path cannot certify (k-th hit beyond the provable radius, a tie exactly at
// the cut, ring overflow, halo larger than LDS) are appended to a work list for wtp_wave.hip;
// results are therefore always the exact canonical lists.
//
// Roofline: the kernel streams 16 B/point in and 28 B/point out (MODE 1) — its HBM floor is
// ~0.07 ms for 10 M points — but executes several thousand VALU lane-ops per query, so it is
// bound by vector-ALU and LDS issue, not by HBM.
#include "wtp_device.hpp"
#include "wtp_sortnet.hpp"

namespace wtp {

#ifndef WTP_DIAG
#define WTP_DIAG 0 // diagnostic build: s_memtime stamps per phase (never quote its run time)
#endif
#define DIAG_STAMP(i)                                              \
    if (WTP_DIAG) {                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        dt[i] += t_ - t_last;                                      \
        t_last = t_;                                               \
    }
#ifndef WTP_ABL
#define WTP_ABL 0 // timing-only ablation builds (results wrong): 1 no force pass, 2 no select, 4 no scan
#endif
constexpr int NB = 64;          // per-lane candidate ring / sorting-network width
constexpr int kFastKMax = 32;   // k >= this goes to the wave kernel
constexpr int kOwnRows = BY * BZ;
#ifndef WTP_LANE_ROWS
#define WTP_LANE_ROWS 2 // CS sweep: 0 = rows in lockstep, 1 = every lane walks its 9 rows at its own pace, 2 = ... its TRIMMED rows (default)
#endif
#ifndef WTP_SCAN_U
#define WTP_SCAN_U 8
#endif
static_assert(WTP_SCAN_U == 8, "lds_read_group is written for 8 points per step");
constexpr int SCAN_U = WTP_SCAN_U;            // candidates per scan step (LDS reads issued together)
constexpr int kPadBytes = SCAN_U * 16;         // reads past a run end stay inside this padding

struct BrickSmem {
    int hstart[HCELLS + 1];   // LDS slot of the first point of each halo cell
    int hglobal[HCELLS];      // global (sorted) index of the first point of each halo cell
    int own_pref[kOwnRows + 1];
    int scan_tmp[kBrickThreads / 64 + 1];
    Acc acc[kBrickThreads / 64];
};

#define WTP_CE(k, i, j)                      \
    {                                        \
        auto lo_ = k[i] < k[j] ? k[i] : k[j];\
        auto hi_ = k[i] < k[j] ? k[j] : k[i];\
        k[i] = lo_;                          \
        k[j] = hi_;                          \
    }

// ring entries are byte offsets into the staged point area (< 64 KiB)
__device__ inline float4 lds_pt(const float4* pts, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(pts) + byte_off);
}

typedef float f4 __attribute__((ext_vector_type(4)));

// SCAN_U consecutive staged points in one go: explicit ds_read_b128 (the compiler would shrink
// the loads to b96 when .w is unused, which costs twice the LDS cycles per instruction) and a
// single wait.  The wait is inside the statement, so the outputs are valid when it returns.
__device__ inline void lds_read_group(f4 (&c)[8], uint32_t addr) {
    asm volatile(
        "ds_read_b128 %0, %8\n\t"
        "ds_read_b128 %1, %8 offset:16\n\t"
        "ds_read_b128 %2, %8 offset:32\n\t"
        "ds_read_b128 %3, %8 offset:48\n\t"
        "ds_read_b128 %4, %8 offset:64\n\t"
        "ds_read_b128 %5, %8 offset:80\n\t"
        "ds_read_b128 %6, %8 offset:96\n\t"
        "ds_read_b128 %7, %8 offset:112\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7])
        : "v"(addr)
        : "memory");
}

__device__ inline void lds_read_group(f4 (&c)[4], uint32_t addr) {
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:16\n\t"
        "ds_read_b128 %2, %4 offset:32\n\t"
        "ds_read_b128 %3, %4 offset:48\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
        : "v"(addr)
        : "memory");
}

__device__ inline uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ inline float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// Sort this lane's ring keys (d2 recomputed from the staged points); return the k-th smallest
// (index K-1) and the next one (index K).  KT > 0: K is the compile-time KT.
struct NoVisit {
    __device__ inline void operator()(const float4&, float, bool) const {}
};

// `visit(point, d2, valid)` sees every ring entry once while its point is in registers: the
// sweep uses it to accumulate whatever does not depend on the cut (see SpecForce).
template <int KT, class Visit>
__device__ inline void ring_select(const float4* __restrict__ pts, const uint16_t* __restrict__ ring, int cnt,
                                   float qx, float qy, float qz, int K, uint32_t& kth, uint32_t& next,
                                   Visit& visit) {
    uint32_t k[NB];
#pragma unroll
    for (int c8 = 0; c8 < NB / 8; ++c8) {
        if (__any(cnt > c8 * 8)) { // wave-uniform: skip chunks no lane has filled
#pragma unroll
            for (int j = c8 * 8; j < c8 * 8 + 8; ++j) {
                const float4 c = lds_pt(pts, ring[(j < cnt ? j : 0) * kBrickThreads]);
                const float df = dist2<float>(qx, qy, qz, c.x, c.y, c.z);
                const uint32_t d = f2u(df);
                k[j] = j < cnt ? d : 0x7F800000u;
                visit(c, df, j < cnt);
            }
        } else {
#pragma unroll
            for (int j = c8 * 8; j < c8 * 8 + 8; ++j) k[j] = 0x7F800000u;
        }
    }
    WTP_SORTNET_64(k)
    if (KT > 0) {
        kth = k[KT - 1];
        next = k[KT];
    } else {
        kth = k[0];
        next = k[1];
#pragma unroll
        for (int j = 1; j < kFastKMax; ++j) {
            kth = (j == K - 1) ? k[j] : kth;
            next = (j == K - 1) ? k[j + 1] : next;
        }
    }
}

// KNNTopology rows (round 2): ONE network instead of two.  The key is the upper 26 bits of d2 with the ring position
// in the low 6: truncation is monotone, so entries whose truncated d2 differ come out in exact order, and the low
// bits lead back to the candidate — no second pass that compacts the survivors and sorts 64-bit (d2, id) keys.
// Entries that share a truncated d2 (relative difference < 2^-17: ~0.5 % of the queries have such a pair among
// their k nearest) are put in exact (d2, id) order afterwards by the caller; a shared bucket AT the cut sends the
// query to the exact path, like an exact tie did before.
constexpr uint32_t kSlotMask = 63u;
static_assert(NB == 64, "ring position rides in 6 bits of the key");
template <int KT>
__device__ inline void ring_sort_packed(const float4* __restrict__ pts, const uint16_t* __restrict__ ring, int cnt,
                                        float qx, float qy, float qz, uint32_t (&k)[NB]) {
#pragma unroll
    for (int c8 = 0; c8 < NB / 8; ++c8) {
        if (__any(cnt > c8 * 8)) { // wave-uniform: skip chunks no lane has filled
#pragma unroll
            for (int j = c8 * 8; j < c8 * 8 + 8; ++j) {
                const float4 c = lds_pt(pts, ring[(j < cnt ? j : 0) * kBrickThreads]);
                const uint32_t d = f2u(dist2<float>(qx, qy, qz, c.x, c.y, c.z));
                k[j] = j < cnt ? ((d & ~kSlotMask) | (uint32_t)j) : (0x7F800000u | (uint32_t)j);
            }
        } else {
#pragma unroll
            for (int j = c8 * 8; j < c8 * 8 + 8; ++j) k[j] = 0x7F800000u | (uint32_t)j;
        }
    }
    WTP_SORTNET_64(k)
}

// Keep ring entries with d2 <= lim (stable, in place).
__device__ inline void ring_compact(const float4* __restrict__ pts, uint16_t* __restrict__ ring, int& cnt,
                                    float lim, float qx, float qy, float qz) {
    int keep = 0;
    for (int j = 0; __any(j < cnt); ++j) {
        const uint16_t s = ring[(j < cnt ? j : 0) * kBrickThreads];
        const float4 c = lds_pt(pts, s);
        const float d = dist2<float>(qx, qy, qz, c.x, c.y, c.z);
        const bool ok = (j < cnt) && (d <= lim);
        ring[(ok ? keep : NB) * kBrickThreads] = s;
        keep += ok ? 1 : 0;
    }
    cnt = keep;
}

// Fast-math force law for the brick path (1-ulp rcp; the wave kernel keeps the IEEE forms).
// Laws 0..2 are one branch-free expression  max((A - B*u2) / (u2+beta)^2, lo)  with
// (A,B,lo) = (1,0,-inf) inverse distance, (1,1,-inf) equilibrium, (u0^2,1,0) clipped; law 3 swaps the
// denominator for (u2+beta)^gamma.  u2 = d2/s^2.
struct ForceCoef {
    float A, B, lo, beta, gamma;
    int strong;
};
__device__ inline ForceCoef force_coef(int kind, float beta, float u0, float gamma) {
    ForceCoef c;
    c.A = kind == WTP_FORCE_CLIPPED_SPACING ? u0 * u0 : 1.f;
    c.B = kind == WTP_FORCE_INVERSE_DISTANCE ? 0.f : 1.f;
    c.lo = kind == WTP_FORCE_CLIPPED_SPACING ? 0.f : -Lim<float>::inf();
    c.beta = beta;
    c.gamma = gamma;
    c.strong = kind == WTP_FORCE_STRONG_SPACING;
    return c;
}
__device__ inline float force_fast(const ForceCoef& c, float u2) {
    const float d = u2 + c.beta;
    float inv = __builtin_amdgcn_rcpf(d * d);
    if (c.strong) inv = __builtin_amdgcn_exp2f(-c.gamma * __builtin_amdgcn_logf(d)); // wave-uniform
    const float f = (c.A - c.B * u2) * inv;
    return f > c.lo ? f : c.lo;
}

// Visitor of the sweep's final key pass.  Two things never depend on the k-th cut: the nearest
// other point (always inside the k set for k >= 2), and — for ClippedSpacingForce, whose support
// ends at u0*s — the force sum, as long as the support radius turns out to lie inside the cut
// (then every point that contributes is one of the k nearest; the usual case: u0*s ~ h, r_k ~ 1.7 h).
// The kernel checks that condition after the selection and otherwise runs the explicit loop.
struct SpecForce {
    float qx, qy, qz, inv_s2, lim;
    int32_t qid;
    ForceCoef fc;
    bool on;           // wave-uniform: law is the clipped one
    float Fx, Fy, Fz, nd2;
    int32_t nid;
    bool coincident;
    __device__ inline void reset() {
        Fx = Fy = Fz = 0.f;
        nd2 = Lim<float>::inf();
        nid = 0x7FFFFFFF;
        coincident = false;
    }
    __device__ inline void operator()(const float4& c, float d, bool valid) {
        if (!on) return;
        const int32_t cid = w_to_id(c.w);
        const bool in = valid && (cid != qid); // self skipped by index (src/repel.jl:271)
        const bool nearer = in && lex_lt(d, cid, nd2, nid);
        nd2 = nearer ? d : nd2;
        nid = nearer ? cid : nid;
        const bool act = in && (d <= lim);
        const float f = force_fast(fc, d * inv_s2);
        const float coef = (act && d > 0.f) ? f * __builtin_amdgcn_rsqf(d) : 0.f;
        Fx += coef * (qx - c.x);
        Fy += coef * (qy - c.y);
        Fz += coef * (qz - c.z);
        coincident = coincident || (act && !(d > 0.f));
    }
};

template <int MODE, int KT, int CS>
__global__ __launch_bounds__(kBrickThreads, CS ? 4 : 2) void brick_kernel(SearchArgs<float> a, int hcap) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    constexpr int nb = CS ? 32 : NB; // ring rows: the compact-support sweep keeps only the support
    // candidates per scan step: the compact-support grid has short runs (3 cells x ~3.5 points), where
    // steps of 8 would spend half their slots past the run end
    constexpr int SU = CS ? 4 : SCAN_U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* pts = reinterpret_cast<float4*>(smem_raw);
    uint16_t* ring_all = reinterpret_cast<uint16_t*>(smem_raw + (size_t)hcap * sizeof(float4) + kPadBytes);
    BrickSmem* sm = reinterpret_cast<BrickSmem*>(smem_raw + (size_t)hcap * sizeof(float4) + kPadBytes +
                                                 (size_t)(nb + 1) * kBrickThreads * sizeof(uint16_t));
    const int tid = threadIdx.x;
    uint16_t* ring = ring_all + tid; // entry j at ring[j * kBrickThreads]; row nb is a dump row

    const Grid<float> g = *a.grid;
    if (MODE == 2 && g.rad_wave_only) return; // rows expected to outgrow this kernel's 32 entries: the wave kernel takes every query (grid_setup_kernel)
    const int K = KT > 0 ? KT : a.k;
    const bool skip_self = (MODE == 0 && !a.include_self) || MODE == 2; // radius rows never hold the point itself (src/topology.jl:96)
    const float cap2 = (a.gamma_cap * g.c) * (a.gamma_cap * g.c);
    Acc acc = acc_empty();
    unsigned long long dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = WTP_DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    // masked-off ring reads use row 0: make every slot ever read from it a valid LDS point index
#pragma unroll 1
    for (int j = 0; j <= nb; ++j) ring[j * kBrickThreads] = 0;

    // XCD-aware brick order: blocks sharing blockIdx % 8 share an L2; give each such group one
    // contiguous slab of bricks so halo re-reads of neighbouring bricks hit that L2.
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
        const int ox = bx * BX - 1, oy = by * BY - 1, oz = bz * BZ - 1; // halo origin (cell coords)

        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // previous brick's LDS no longer in use
        // ---- 1. halo cell table ---------------------------------------------------------
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
                for (int w = 0; w < kBrickThreads / 64; ++w) {
                    const int t = sm->scan_tmp[w];
                    sm->scan_tmp[w] = run;
                    run += t;
                }
                sm->scan_tmp[kBrickThreads / 64] = run;
            }
            __syncthreads();
            const int ex = incl - my_cnt + sm->scan_tmp[wave];
            if (tid < HCELLS) sm->hstart[tid] = ex;
            if (tid == HCELLS) sm->hstart[HCELLS] = sm->scan_tmp[kBrickThreads / 64];
        }
        __syncthreads();
        const int halo_total = sm->hstart[HCELLS];
        // RadiusTopology: a brick whose cells hold more than ~7.4 points has rows beyond this kernel's 32 entries (a row is
        // ~4.06 cells' worth of points); it is handed to the wave kernel up front, like a brick that does not fit LDS,
        // instead of being staged and scanned first (the dense part of a graded cloud)
        const bool overflow = halo_total > hcap || (MODE == 2 && halo_total > kRadDenseMin);
        // (count phase with the dense kernel behind it, wtp_radb.hip: such a brick is that kernel's, unless it outgrows its LDS too)
        if (MODE == 2 && a.rad_dense > 0 && halo_total > kRadDenseMin && halo_total <= a.rad_dense) continue;

        // ---- 2. own-row prefix (queries = points of the BX*BY*BZ own cells) ------------------
        if (tid == 0) {
            int run = 0;
            for (int r = 0; r < kOwnRows; ++r) {
                const int hy = 1 + r % BY, hz = 1 + r / BY;
                const int base = (hz * HY + hy) * HX;
                sm->own_pref[r] = run;
                run += sm->hstart[base + 1 + BX] - sm->hstart[base + 1];
            }
            sm->own_pref[kOwnRows] = run;
        }
        // ---- 3. stage halo points: each x-row of HX cells is one contiguous global run ----------
        if (!overflow) {
            const int wave = tid >> 6, lane = tid & 63;
            const int hx_lo = ox < 0 ? -ox : 0; // first halo column inside the grid
            // A wave owns HY*HZ / 4 = 9 rows.  Their first 64 points are fetched TOGETHER (nine
            // independent global loads in flight, then nine LDS stores) instead of one row per round
            // trip: staging was 36 % of the kernel's wave time with the rows serialised.
            constexpr int kWaves = kBrickThreads / 64;
            constexpr int kRows = (HY * HZ + kWaves - 1) / kWaves;
            int ls[kRows], len[kRows], gs[kRows];
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int row = wave + j * kWaves;
                const bool ok = row < HY * HZ;
                const int base = (ok ? row : 0) * HX;
                ls[j] = sm->hstart[base];
                len[j] = ok ? sm->hstart[base + HX] - ls[j] : 0;
                gs[j] = sm->hglobal[base + hx_lo];
            }
            float4 v[kRows];
#pragma unroll
            for (int j = 0; j < kRows; ++j) { // unconditional, clamped: a partially defined array would live in scratch
                const int src = gs[j] + (lane < len[j] ? lane : 0);
                v[j] = a.snap[src < a.n ? src : a.n - 1];
            }
#pragma unroll
            for (int j = 0; j < kRows; ++j)
                if (lane < len[j]) pts[ls[j] + lane] = v[j];
#pragma unroll
            for (int j = 0; j < kRows; ++j) // rows longer than a wave (dense cells): the rest, row by row
                for (int i = lane + 64; i < len[j]; i += 64) pts[ls[j] + i] = a.snap[gs[j] + i];
        }
        __syncthreads();

        // ---- 4. queries -----------------------------------------------------------------------
        DIAG_STAMP(0) // brick setup + halo staging
        __builtin_amdgcn_s_setprio(1); // waves that run queries issue ahead of waves that stage (measured on the round-2 sweep: -3.5 %)
        const int Q = sm->own_pref[kOwnRows];
        const int rot = ((brick / blk_per_group + blockIdx.x) & 3) << 6;
        for (int qb = 0; qb < Q; qb += kBrickThreads) {
            // the last, partial round of a brick falls on wave 0 of the round; rotating the wave order from brick to
            // brick spreads those rounds over the four SIMDs instead of loading the first one
            const int q = qb + ((tid + rot) & (kBrickThreads - 1));
            const bool active = q < Q;
            int r = 0;
            if (active) {
#pragma unroll
                for (int t = 1; t < kOwnRows; ++t) r += (sm->own_pref[t] <= q) ? 1 : 0;
            }
            const int hy0 = 1 + r % BY, hz0 = 1 + r / BY;
            const int rbase = (hz0 * HY + hy0) * HX + 1;
            const int off = q - sm->own_pref[r];
            const int gslot = sm->hglobal[rbase] + off; // index in the sorted arrays
            if (!active) continue;
            if (overflow) { // halo does not fit LDS: the wave kernel takes the whole brick
                const int pos = atomicAdd(a.fb_count, 1);
                a.fb_list[pos] = gslot;
                continue;
            }
            const float4 qp = pts[sm->hstart[rbase] + off];
            const int32_t qid = w_to_id(qp.w);
            if (MODE == 1 && qid < a.n_fixed) { // the wall: never moves (src/repel.jl:80,256)
                a.out[gslot] = qp;
                a.forces[gslot] = 0.f;
                a.nn_dist[gslot] = Lim<float>::inf();
                a.nn_id[gslot] = -1;
                continue;
            }
            const int cx = cell_coord(g, qp.x, 0), cy = cell_coord(g, qp.y, 1), cz = cell_coord(g, qp.z, 2);
            const int hx = cx - ox, hy = cy - oy, hz = cz - oz;
            const float g2 = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, 1);
            // First filter radius of the k-selection: the ball that is EXPECTED to hold a.cap_count points at the density
            // this query sees — its 27 cells hold n27, so r^3 = cap_count / n27 * 27 / (4 pi / 3) cell volumes.  A fixed
            // gamma_cap * c served the interior and failed at the faces of the cloud, where half the block is empty and
            // the k-th neighbour lies 1.26 x further: 12 % of a 1 M-point cloud sits in face cells, a tenth of those
            // came up short and went to the exact path (1.24 % of all queries, 0.14 of the call's 0.74 ms).
            float capq = cap2;
            if (!CS && MODE != 2 && a.cap_count > 0.f) {
                int n27 = 0;
#pragma unroll
                for (int r9 = 0; r9 < 9; ++r9) {
                    const int b = ((hz + r9 / 3 - 1) * HY + (hy + r9 % 3 - 1)) * HX + (hx - 1);
                    n27 += sm->hstart[b + 3] - sm->hstart[b];
                }
                const float x = a.cap_count * 6.4458f / (float)(n27 > 1 ? n27 : 1);
                capq = (g.c * g.c) * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * 0.66666667f);
            }
            float tau = g2 < capq ? g2 : capq;
            // MODE 2 (RadiusTopology): the threshold IS the answer — everything with d2 <= r^2 (inclusive,
            // src/topology.jl:93); the 27 cells are complete when r lies inside the provable radius
            bool rad_fail = false;
            if (MODE == 2) {
                rad_fail = !(a.radius2 <= g2);
                tau = rad_fail ? -1.f : a.radius2;
            }
            const int32_t skip_id = skip_self ? qid : -1; // ids are >= 0
            int cnt = 0;
            bool giveup = false;
            // sweep: spacing at the query and the visitor that rides along the key pass
            const float s = (MODE == 1) ? (a.spacing_pp ? a.spacing_pp[qid] : a.spacing_const) : 1.f;
            SpecForce spec;
            if (MODE == 1) {
                spec.qx = qp.x;
                spec.qy = qp.y;
                spec.qz = qp.z;
                spec.qid = qid;
                spec.inv_s2 = 1.f / (s * s);
                spec.lim = (a.u0 * a.u0) * (s * s);
                spec.fc = force_coef(a.force_kind, a.beta, a.u0, a.gamma);
                spec.on = (a.force_kind == WTP_FORCE_CLIPPED_SPACING) && (K >= 2) && !(WTP_ABL & 1);
                spec.reset();
            }
            // CS (compact support) sweep, ClippedSpacingForce only: the force sums over the k nearest
            // points, but points beyond u0*s contribute exactly 0, so once
            //   n_lim = #{points with d2 <= (u0 s)^2}  <=  k
            // holds, those n_lim points ARE among the k nearest (everything else is farther) and the
            // sum over them equals the reference's sum over its k-list — no selection needed.  The
            // ring then only collects the support (plus a margin that always contains the nearest
            // neighbour); queries where the count exceeds k, or whose support is not certified by the
            // searched cells, go to the exact path.
            bool cs_fail = false;
            if (CS) {
                const float tnn = (a.tnn_frac * g.c) * (a.tnn_frac * g.c); // < provable radius (>= c - c/256); holds the nearest neighbour
                const float tcs = spec.lim > tnn ? spec.lim : tnn;
                // the support only has to lie inside the provable radius of THIS query (c .. 1.5 c depending
                // on where it sits in its cell); the gamma cap that bounds the k-selection ring does not apply
                cs_fail = !(spec.lim <= g2);
                tau = tcs < g2 ? tcs : g2;
            }
            // CS scan filter: d2 with fused multiply-adds (3 VALU instead of 5) against a threshold
            // 4 ulp wider.  The ring pass below recomputes the canonical d2 of every survivor and
            // applies the exact cut, so the filter only has to be conservative.
            float tau_s = CS ? tau * (1.f + 0x1p-21f) : tau;

            // ---- scan / select loop: one instance of the network serves prunes and the final cut
            // LDS byte addresses: points at [0, hcap*16), this lane's ring row j at ring_b + j*512
            const uint32_t ring_b = (uint32_t)hcap * 16u + (uint32_t)kPadBytes + (uint32_t)tid * 2u;
            // (row nb of the ring is a spare row: an unconditional append of a rejected candidate may land there)
            const uint32_t full_b = ring_b + (uint32_t)(nb - SU) * (kBrickThreads * 2u);
            uint32_t ra = ring_b; // next free ring entry
            const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw; // LDS byte address of the point area
            DIAG_STAMP(1) // query setup
            int row = 0;
            uint32_t pa, ea;      // current candidate run [pa, ea) as byte offsets into pts
            {
                const int base = ((hz - 1) * HY + (hy - 1)) * HX + (hx - 1);
                pa = (uint32_t)sm->hstart[base] * 16u;
                ea = (uint32_t)sm->hstart[base + 3] * 16u;
            }
            uint32_t kth = 0, next = 0;
            uint32_t keys[NB]; // MODE 0: the ring sorted by packed key (ring_sort_packed); unused otherwise
            constexpr int kPre = (KT > 0 ? KT : kFastKMax - 1) + 3; // sorted prefix the topology path looks at (K <= kFastKMax - 1)
            int rbase_l = ((hz - 1) * HY + (hy - 1)) * HX + (hx - 1); // CS lane rows: hstart index of my row
            bool lr_started = false; // CS trimmed lane rows: first entry vs resume after ring pressure
            uint32_t lr_q[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}; // queued runs: start | end << 16 (byte offsets)
            int lr_n = 0;
            for (;;) {
                bool pressure = false;
                if (WTP_LANE_ROWS == 2) {
                    // Per-lane rows WITH trimming: a row (dy, dz) of the neighbourhood lies at least
                    // gy, gz away from the query along y and z (distance to its own cell's faces, less the
                    // cell map's rounding margin), so only candidates within rem = tau - gy^2 - gz^2 along
                    // x can pass the filter: the row is skipped when rem < 0, its first / last cell when
                    // the query is farther than sqrt(rem) from the own cell's lower / upper x face.  On
                    // average 17 of the 27 cells survive.  Every lane walks its own trimmed rows, so the
                    // wave runs for max over lanes of the trimmed total.
                    const float cxl = g.org[0] + (float)cx * g.c, cyl = g.org[1] + (float)cy * g.c,
                                czl = g.org[2] + (float)cz * g.c;
                    auto gap2 = [&](float d) {
                        const float t = d - g.margin;
                        return t > 0.f ? t * t : 0.f;
                    };
                    const float lx2 = gap2(qp.x - cxl), ux2 = gap2(cxl + g.c - qp.x);
                    const float ly2 = gap2(qp.y - cyl), uy2 = gap2(cyl + g.c - qp.y);
                    const float lz2 = gap2(qp.z - czl), uz2 = gap2(czl + g.c - qp.z);
                    const int base0 = (hz * HY + hy) * HX + (hx - 1);
                    // The trimmed runs of the 9 rows are worked out once, all LDS reads in flight together,
                    // and queued (non-empty ones only) as packed byte offsets start | end << 16 in nine
                    // registers; taking the next run is then a register shift, no memory in the scan loop.
                    if (!lr_started) {
                        lr_started = true;
                        pa = ea = 0u;
                        lr_n = 0;
                        uint32_t st[9], en[9];
#pragma unroll
                        for (int r9 = 0; r9 < 9; ++r9) {
                            const int dzr = r9 / 3, dyr = r9 % 3;
                            const float gy2 = dyr == 0 ? ly2 : (dyr == 2 ? uy2 : 0.f);
                            const float gz2 = dzr == 0 ? lz2 : (dzr == 2 ? uz2 : 0.f);
                            const float rem = (CS ? tau_s : tau) - gy2 - gz2; // tau only shrinks afterwards: still conservative
                            const int rb = base0 + (dzr - 1) * (HY * HX) + (dyr - 1) * HX;
                            const int i0 = rb + (lx2 <= rem ? 0 : 1), i1 = rb + (ux2 <= rem ? 3 : 2);
                            st[r9] = (uint32_t)sm->hstart[i0];
                            en[r9] = rem < 0.f ? st[r9] : (uint32_t)sm->hstart[i1];
                        }
#pragma unroll
                        for (int r9 = 0; r9 < 9; ++r9) {
                            const bool put = en[r9] > st[r9];
                            const uint32_t pk = (st[r9] * 16u) | ((en[r9] * 16u) << 16);
#pragma unroll
                            for (int u9 = 0; u9 < 9; ++u9) lr_q[u9] = (put && lr_n == u9) ? pk : lr_q[u9];
                            lr_n += put ? 1 : 0;
                        }
                    }
                    auto advance = [&]() {
                        if (pa >= ea && lr_n > 0) {
                            const uint32_t pk = lr_q[0];
#pragma unroll
                            for (int u9 = 0; u9 < 8; ++u9) lr_q[u9] = lr_q[u9 + 1];
                            --lr_n;
                            pa = pk & 0xFFFFu;
                            ea = pk >> 16;
                        }
                    };
                    advance();
                    while (__any(pa < ea)) {
                        if (__any(ra > full_b)) {
                            pressure = true;
                            break;
                        }
#if WTP_DIAG == 2 // lane utilisation of the scan: slots 3 / 4 count wave steps x 64 and busy lane steps
                        dt[3] += 64;
                        dt[4] += (unsigned long long)__popcll(__ballot(pa < ea));
#endif
                        f4 c[SU];
                        lds_read_group(c, lds_base + pa);
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            float d;
                            if (CS) {
                                const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                                d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                            } else {
                                d = dist2<float>(qp.x, qp.y, qp.z, c[u].x, c[u].y, c[u].z);
                            }
                            const float tl = (pa + 16u * u < ea) ? (CS ? tau_s : tau) : -1.f;
                            bool take = d <= tl;
                            if (MODE != 1) take = take && (w_to_id(c[u].w) != skip_id);
                            *reinterpret_cast<uint16_t*>(smem_raw + ra) = (uint16_t)(pa + 16u * u); // unconditional: a slot past the end is rewritten or never read
                            ra += take ? (kBrickThreads * 2u) : 0u;
                        }
                        pa += 16u * SU;
                        advance();
                    }
                    if (!pressure) row = 9;
                } else if (CS && WTP_LANE_ROWS == 1) {
                    // Every lane advances to its next row as soon as its current one is exhausted,
                    // so the wave runs for max over lanes of (sum of row lengths) steps instead of
                    // sum over rows of (max over lanes).
                    while (__any(row < 9)) {
                        if (__any(ra > full_b)) {
                            pressure = true;
                            break;
                        }
                        f4 c[SU];
                        lds_read_group(c, lds_base + pa);
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                            const float d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                            const float tl = (pa + 16u * u < ea) ? tau_s : -1.f;
                            const bool take = d <= tl;
                            *reinterpret_cast<uint16_t*>(smem_raw + ra) = (uint16_t)(pa + 16u * u); // unconditional: a slot past the end is rewritten or never read
                            ra += take ? (kBrickThreads * 2u) : 0u;
                        }
                        pa += 16u * SU;
                        while (pa >= ea && row < 9) { // next non-empty row of this lane
                            ++row;
                            if (row < 9) {
                                rbase_l += (row % 3 == 0) ? (HY * HX - 2 * HX) : HX;
                                pa = (uint32_t)sm->hstart[rbase_l] * 16u;
                                ea = (uint32_t)sm->hstart[rbase_l + 3] * 16u;
                            } else {
                                pa = ea = 0u; // done: reads stay at the start of the point area, all masked
                            }
                        }
                    }
                } else
                while (row < 9) {
                    // SCAN_U candidates per step: all reads issued together (one wait), then
                    // branch-free appends.  Reads past the run end stay inside the padded point
                    // area and are masked.  (A ping-pong prefetch of the next step was measured
                    // slower: the LDS pipe, not its latency, is the co-bottleneck.)
                    while (!(WTP_ABL & 4) && __any(pa < ea)) {
                        if (__any(ra > full_b)) {
                            pressure = true;
                            break;
                        }
                        f4 c[SU];
                        lds_read_group(c, lds_base + pa);
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            float d;
                            if (CS) {
                                const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                                d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                            } else {
                                d = dist2<float>(qp.x, qp.y, qp.z, c[u].x, c[u].y, c[u].z);
                            }
                            // run-end masking folded into the threshold (a VALU select) instead of
                            // and-ing lane masks on the scalar unit: no VALU->SALU->VALU round trip
                            const float tl = (pa + 16u * u < ea) ? (CS ? tau_s : tau) : -1.f;
                            bool take = d <= tl;
                            if (MODE != 1) take = take && (w_to_id(c[u].w) != skip_id);
                            *reinterpret_cast<uint16_t*>(smem_raw + ra) = (uint16_t)(pa + 16u * u); // unconditional: a slot past the end is rewritten or never read
                            ra += take ? (kBrickThreads * 2u) : 0u;
                        }
                        pa += 16u * SU;
                    }
                    if (pressure) break;
                    ++row;
                    if (row < 9) {
                        const int dz = row / 3 - 1, dy = row % 3 - 1;
                        const int base = ((hz + dz) * HY + (hy + dy)) * HX + (hx - 1);
                        pa = (uint32_t)sm->hstart[base] * 16u;
                        ea = (uint32_t)sm->hstart[base + 3] * 16u;
                    }
                }
                cnt = (int)((ra - ring_b) / (kBrickThreads * 2u));
                if (CS || MODE == 2) { // no selection: everything within tau is in the ring
                    if (pressure) { // some lane's ring is full (dense cluster): only that lane gives up
                        if (ra > full_b) {
                            giveup = true;
                            ra = ring_b;
                            tau = -1.f;
                            tau_s = -1.f;
                        }
                        continue; // resume the row scan
                    }
                    kth = f2u(tau);
                    next = ~0u;
                    DIAG_STAMP(2) // scan
                    break;
                }
                if (WTP_ABL & 2) {
                    kth = f2u(tau);
                    next = kth + 1;
                } else {
                    DIAG_STAMP(2) // scan
                    if (MODE == 1) {
                        spec.reset();
                        ring_select<KT>(pts, ring, cnt, qp.x, qp.y, qp.z, K, kth, next, spec);
                    } else {
                        ring_sort_packed<KT>(pts, ring, cnt, qp.x, qp.y, qp.z, keys);
                        // the cut as the upper end of the k-th entry's bucket: "d2 <= cut" is then exactly the
                        // sorted prefix, and kth == next says that the (k+1)-th shares the bucket
                        if (KT > 0) {
                            kth = keys[KT - 1];
                            next = keys[KT];
                        } else {
                            kth = keys[0];
                            next = keys[1];
#pragma unroll
                            for (int j = 1; j < kFastKMax; ++j) {
                                kth = (j == K - 1) ? keys[j] : kth;
                                next = (j == K - 1) ? keys[j + 1] : next;
                            }
                        }
                        kth |= kSlotMask;
                        next |= kSlotMask;
                    }
                    DIAG_STAMP(3) // select
                }
                if (!pressure) break;
                if (cnt >= K) {
                    const float t = u2f(kth);
                    tau = t < tau ? t : tau;
                }
                if (MODE == 0) {
                    // survivors = the sorted prefix: the K nearest so far plus up to two more from the k-th's bucket
                    // (a bucket that reaches further is a mass tie: exact path).  Their candidates are looked up
                    // through the low key bits and rewritten to the head of the ring.
                    const int keep = cnt < K + 2 ? cnt : K + 2;
                    uint32_t past = keys[kPre - 1];
#pragma unroll
                    for (int j = 2; j < kPre - 1; ++j) past = (j == K + 2) ? keys[j] : past;
                    if (cnt > K + 2 && (past | kSlotMask) == kth) {
                        giveup = true;
                        tau = -1.f;
                    }
                    uint16_t o[kPre - 1];
#pragma unroll
                    for (int j = 0; j < kPre - 1; ++j) o[j] = ring[(keys[j] & kSlotMask) * kBrickThreads];
#pragma unroll
                    for (int j = 0; j < kPre - 1; ++j)
                        if (j < keep) ring[j * kBrickThreads] = o[j];
                    cnt = giveup ? 0 : keep;
                } else
                ring_compact(pts, ring, cnt, tau, qp.x, qp.y, qp.z);
                if (cnt > NB - SCAN_U) { // ring still full (mass tie): give up, the wave kernel takes it
                    giveup = true;
                    cnt = 0;
                    tau = -1.f;
                }
                ra = ring_b + (uint32_t)cnt * (kBrickThreads * 2u);
                DIAG_STAMP(4) // prune compaction
            }

            bool fallback = !WTP_ABL && ((!CS && cnt < K) || giveup || cs_fail || (kth == next)); // tie at the cut -> exact path
            if (MODE == 2) fallback = giveup || rad_fail || cnt > 32; // longer rows: the wave kernel's LDS list
            if (MODE == 2) {
                if (!fallback) {
                    if (!a.rad_fill) a.rad_counts[qid] = cnt; // first phase of the CSR: row lengths
                    if (a.rad_fill || a.rad_tmp) {
                        // the row in canonical (d2, id) order: written to its place in the CSR (fill phase), or parked in the
                        // count phase already so that the fill phase is a copy and not a second search (a.rad_tmp)
                        uint64_t k[32];
#pragma unroll
                        for (int j = 0; j < 32; ++j) {
                            const float4 c = lds_pt(pts, ring[(j < cnt ? j : 0) * kBrickThreads]);
                            const uint32_t d = f2u(dist2<float>(qp.x, qp.y, qp.z, c.x, c.y, c.z));
                            k[j] = j < cnt ? (((uint64_t)d << 32) | (uint32_t)w_to_id(c.w)) : ~0ull;
                        }
                        WTP_SORTNET_32(k)
                        int32_t* orow = a.rad_fill ? a.idx_out + a.rad_offsets[qid] : a.rad_tmp + (int64_t)qid * 32;
#pragma unroll
                        for (int j = 0; j < 32; ++j)
                            if (j < cnt) orow[j] = (int32_t)(uint32_t)k[j];
                        if (!a.rad_fill) a.rad_done[qid] = 1;
                    }
                }
            } else if (MODE == 0) {
                if (!fallback) {
                    // the row = the first K entries of the sorted prefix; their exact (d2, id) through the low key bits
                    constexpr int KR = KT > 0 ? KT : kFastKMax - 1;
                    float dd[KR];
                    int32_t ii[KR];
                    bool risky = false;
#pragma unroll
                    for (int j = 0; j < KR; ++j) {
                        const float4 c = lds_pt(pts, ring[(keys[j] & kSlotMask) * kBrickThreads]);
                        dd[j] = dist2<float>(qp.x, qp.y, qp.z, c.x, c.y, c.z);
                        ii[j] = w_to_id(c.w);
                        if (j + 1 < KR) risky = risky || ((j + 1 < K) && ((keys[j] ^ keys[j + 1]) <= kSlotMask));
                        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0); // four look-ups in flight, not all of them (registers)
                    }
                    // entries that share a bucket may be out of order: exact (d2, id) exchange passes until none moves
                    bool again = risky;
                    while (__any(again)) {
                        again = false;
#pragma unroll
                        for (int j = 0; j + 1 < KR; ++j) {
                            const bool sw = (j + 1 < K) && lex_lt(dd[j + 1], ii[j + 1], dd[j], ii[j]);
                            const float td = dd[j];
                            const int32_t ti = ii[j];
                            dd[j] = sw ? dd[j + 1] : td;
                            ii[j] = sw ? ii[j + 1] : ti;
                            dd[j + 1] = sw ? td : dd[j + 1];
                            ii[j + 1] = sw ? ti : ii[j + 1];
                            again = again || sw;
                        }
                    }
                    int32_t* orow = a.idx_out + (int64_t)qid * K;
                    float* drow = a.dist_out ? a.dist_out + (int64_t)qid * K : nullptr;
#pragma unroll
                    for (int j = 0; j < KR; ++j) {
                        if (j < K) {
                            orow[j] = ii[j];
                            if (drow) drow[j] = wsqrt(dd[j]);
                        }
                    }
                }
            } else {
                if (!fallback) {
                    const float cut = u2f(kth);
                    const float inv_s2 = spec.inv_s2;
                    // ClippedSpacingForce vanishes for u >= u0: neighbours beyond u0*s add nothing
                    const float lim = (a.force_kind == WTP_FORCE_CLIPPED_SPACING && spec.lim < cut) ? spec.lim : cut;
                    const ForceCoef fc = spec.fc;
                    // the key pass already summed everything when the law's support lies inside the cut
                    const bool spec_ok = !CS && spec.on && (spec.lim <= cut);
                    int n_lim = 0; // CS: points (self included) inside the law's support
                    bool coincident = spec_ok ? spec.coincident : false;
                    float Fx = spec_ok ? spec.Fx : 0.f, Fy = spec_ok ? spec.Fy : 0.f, Fz = spec_ok ? spec.Fz : 0.f;
                    int32_t nid = spec_ok ? spec.nid : 0x7FFFFFFF;
                    float nd2 = spec_ok ? spec.nd2 : Lim<float>::inf();
                    // otherwise: explicit pass over the ring, 4 entries per step (offsets, then points,
                    // then branch-free math; the LDS round trips are issued together)
                    for (int j0 = 0; !(WTP_ABL & 1) && __any(!spec_ok && j0 < cnt); j0 += 4) {
                        uint16_t off[4];
                        float4 c[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) off[u] = ring[((j0 + u) < cnt ? (j0 + u) : 0) * kBrickThreads];
#pragma unroll
                        for (int u = 0; u < 4; ++u) c[u] = lds_pt(pts, off[u]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float dx = qp.x - c[u].x, dy = qp.y - c[u].y, dz = qp.z - c[u].z;
                            const float d = (dx * dx + dy * dy) + dz * dz;
                            const int32_t cid = w_to_id(c[u].w);
                            // the kk nearest, self skipped by index (src/repel.jl:271)
                            const bool in = !spec_ok && ((j0 + u) < cnt) && (d <= cut) && (cid != qid);
                            const bool nearer = in && lex_lt(d, cid, nd2, nid);
                            nd2 = nearer ? d : nd2;
                            nid = nearer ? cid : nid;
                            const bool act = in && (d <= lim);
                            if (CS) n_lim += (((j0 + u) < cnt) && (d <= lim)) ? 1 : 0;
                            const float f = force_fast(fc, d * inv_s2);
                            const float coef = (act && d > 0.f) ? f * __builtin_amdgcn_rsqf(d) : 0.f;
                            Fx += coef * dx;
                            Fy += coef * dy;
                            Fz += coef * dz;
                            coincident = coincident || (act && !(d > 0.f));
                        }
                    }
                    DIAG_STAMP(6) // force loop
                    if (CS && (n_lim > K || nid == 0x7FFFFFFF)) coincident = true; // not provable here: exact path
                    if (coincident) { // r == 0 needs the substitute direction: exact path (rare)
                        const int pos = atomicAdd(a.fb_count, 1);
                        a.fb_list[pos] = gslot;
                        continue;
                    }
                    float4 o;
                    const float f = step_point<float>(a, s, qp.x, qp.y, qp.z, Fx, Fy, Fz, o.x, o.y, o.z);
                    o.w = qp.w;
                    const bool has = nid != 0x7FFFFFFF;
                    const float nd = has ? wsqrt(nd2) : Lim<float>::inf();
                    // sharded sessions: the k-set (CS: the support and the nearest neighbour) must lie
                    // inside the range the ghost layer covers, else the local answer is not the global one
                    const float need = CS ? (nd2 > spec.lim ? nd2 : spec.lim) : cut;
                    if (reaches_past_cover<float>(a, qp.x, qp.y, qp.z, need)) atomicAdd(a.uncovered, 1);
                    a.out[gslot] = o;
                    a.forces[gslot] = f;
                    a.nn_dist[gslot] = nd;
                    a.nn_id[gslot] = has ? nid : -1;
                    acc_point<float>(acc, f, nd, s, qid, has ? nid : -1);
                }
            }
            if (fallback) {
                const int pos = atomicAdd(a.fb_count, 1);
                a.fb_list[pos] = gslot;
            }
            DIAG_STAMP(5) // force / topology sort + outputs
        }
    }
    if (WTP_DIAG && a.diag && (tid & 63) == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&a.diag[i], dt[i]);
        atomicAdd(&a.diag[7], 1ull);
    }
    if (MODE == 1) {
        __syncthreads();
        acc_block_reduce(acc, sm->acc);
        if (tid == 0) acc_store(&a.partials[blockIdx.x], acc);
    }
}

static size_t brick_smem_bytes(int hcap, int nb) {
    return (size_t)hcap * sizeof(float4) + kPadBytes + (size_t)(nb + 1) * kBrickThreads * sizeof(uint16_t) + sizeof(BrickSmem);
}

// LDS budget: 160 KiB per CU.  Default point capacity 2560 (two workgroups per CU at rho ~ 8);
// the caller may pass a smaller capacity when it knows the occupancy (compact-support sweep).
template <int MODE, int KT, int CS> static int brick_launch(wtp_ctx* ctx, SearchArgs<float>& a) {
    const int nb = CS ? 32 : NB;
    const int hcap = a.brick_hcap > 0 ? a.brick_hcap : 2560;
    int occ = launch_occupancy_of(ctx, (const void*)brick_kernel<MODE, KT, CS>, kBrickThreads, brick_smem_bytes(hcap, nb));
    if (occ > 4) occ = 4;
    int gsz = ctx->sm_count * occ;
    gsz -= gsz % 8;
    if (gsz < 8) gsz = 8;
    if (MODE == 1) a.used_brick = gsz;
    hipLaunchKernelGGL((brick_kernel<MODE, KT, CS>), dim3(gsz), dim3(kBrickThreads), brick_smem_bytes(hcap, nb),
                       ctx->stream, a, hcap);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int brick_partials() { return 1024 + 8 + 512 + 3072; } // bricks, then the follow-up and the ball kernel of the round-2 sweep (wtp_cs2.hip)

// RadiusTopology through the brick kernel (fp32): rows of up to 32 entries are counted / sorted and written
// here, the rest (longer rows, queries the 27 cells cannot certify, bricks too large for LDS) is appended
// to a.fb_list for the wave kernel.  The caller cleared a.fb_count.
int launch_brick_radius(wtp_ctx* ctx, SearchArgs<float>& a) {
    a.gamma_cap = (float)ctx->gamma_cap;
    a.k = 1;
    return brick_launch<2, 0, 0>(ctx, a);
}

template <> int launch_topology<float>(wtp_ctx* ctx, SearchArgs<float>& a) {
    if (a.k > kFastKMax - 1 || ctx->force_generic) return launch_generic_topology<float>(ctx, a, true);
    if (a.ksel_bx > 0) { // the caller built the grid for the x-slowest layout (wtp_ksel.hip)
        if (!a.counters_cleared) WTP_HIP(ctx, hipMemsetAsync(a.fb_count, 0, sizeof(int32_t), ctx->stream));
        const int rk = launch_ksel_topology(ctx, a);
        if (rk) return rk;
        a.fb_r0 = 3;
        return launch_generic_topology<float>(ctx, a, false);
    }
    a.gamma_cap = (float)ctx->gamma_cap;
    a.cap_count = (float)(4.18879 * ctx->gamma_cap * ctx->gamma_cap * ctx->gamma_cap * ctx->rho * (a.k + 1) / 22.0);
    if (!a.counters_cleared) WTP_HIP(ctx, hipMemsetAsync(a.fb_count, 0, sizeof(int32_t), ctx->stream));
    int rc = a.k == 21 ? brick_launch<0, 21, 0>(ctx, a) : brick_launch<0, 0, 0>(ctx, a);
    if (rc) return rc;
    return launch_generic_topology<float>(ctx, a, false);
}

template <> int launch_topology<double>(wtp_ctx* ctx, SearchArgs<double>& a) {
    return launch_generic_topology<double>(ctx, a, true); // fp64: exact wave-per-query path
}

template <> int launch_sweep<float>(wtp_ctx* ctx, SearchArgs<float>& a, bool fresh) {
    // the caller cleared the counter block; partial slots need no clearing: the reduction reads only
    // the slots this step's launches write (a.used_*)
    ctx->n_sweep_launches += 1;
    if (!fresh || a.k > kFastKMax - 1 || ctx->force_generic) {
        const int sp = span_begin(ctx, 1);
        int rc;
        if (!fresh && !ctx->force_generic && a.ball_list && a.force_kind == WTP_FORCE_CLIPPED_SPACING && a.k >= 2 && !ctx->full_select) {
            // A stale snapshot (rebuild_every > 1, src/repel.jl:245) and the default law: the query has moved away from its
            // snapshot entry, so the brick kernels (queries = the staged points) do not apply, but the ball kernel's argument
            // does — the support ball around the point where it is NOW, searched in the block that provably holds it, at
            // most k points in it — with eight lanes per query instead of the wave kernel's 64 (10.5 -> see DESIGN.md).
            rc = launch_cs_all_slots(ctx, a.fb_list, a.n, a.fb_count);
            if (!rc) rc = launch_cs_ball(ctx, a, a.ball_list, a.ball_count);
            if (!rc) rc = launch_generic_sweep<float>(ctx, a, false);
        } else {
            rc = launch_generic_sweep<float>(ctx, a, true);
        }
        span_end(ctx, sp);
        return rc;
    }
    if (a.ksel_bx > 0) { // the session built the grid for the x-slowest layout (wtp_ksel.hip); a.cap_count is the caller's
        const int spk = span_begin(ctx, 1);
        int rk = launch_ksel_sweep(ctx, a);
        span_end(ctx, spk);
        if (rk) return rk;
        const int spk2 = span_begin(ctx, 2);
        a.fb_r0 = 3;
        rk = launch_generic_sweep<float>(ctx, a, false);
        span_end(ctx, spk2);
        return rk;
    }
    a.gamma_cap = (float)ctx->gamma_cap_sweep;
    a.cap_count = (float)(4.18879 * ctx->gamma_cap_sweep * ctx->gamma_cap_sweep * ctx->gamma_cap_sweep * ctx->rho * (a.k + 1) / 22.0);
    const int sp = span_begin(ctx, 1);
    // ClippedSpacingForce (the reference default) takes the compact-support sweep unless
    // WTP_FULL_SELECT=1 asks for the explicit k-selection on every query (both give the same output)
    // (the caller sized the grid for it and says so by passing the LDS point capacity)
    const bool cs = a.brick_hcap > 0 && a.force_kind == WTP_FORCE_CLIPPED_SPACING && a.k >= 2 && !ctx->full_select;
    if (cs && a.cs2_bx > 0 && a.spacing_pp && a.ball_list && !getenv("WTP_CS2_DEAD_OFF")) {
        // variable spacing: bricks that would hand every point back are found first and passed over (wtp_cs2.hip)
        const int dead_cap = (int)(ctx->cell_start.cap / sizeof(int32_t) / 4 + 4096);
        int rd = ensure(ctx, ctx->brick_dead, (size_t)dead_cap);
        if (!rd) rd = launch_cs2_dead(ctx, a, (uint8_t*)ctx->brick_dead.p, dead_cap);
        if (rd) {
            span_end(ctx, sp);
            return rd;
        }
        a.brick_dead = (const uint8_t*)ctx->brick_dead.p;
        a.brick_dead_cap = dead_cap;
    }
    int rc = cs ? (a.cs2_bx > 0 ? launch_cs2(ctx, a) : brick_launch<1, 0, 1>(ctx, a))
                : (a.k == 21 ? brick_launch<1, 21, 0>(ctx, a) : brick_launch<1, 0, 0>(ctx, a));
    span_end(ctx, sp);
    if (rc) return rc;
    const int sp2 = span_begin(ctx, 2);
    if (cs && a.cs2_bx > 0) rc = launch_cs2_followup(ctx, a); // nearest neighbour of the queries the bricks left open
    // variable spacing: hand-backs whose support outgrew their cell are finished ball by ball (wtp_cs2.hip)
    if (!rc && cs && a.spacing_pp && a.ball_list) rc = launch_cs_ball(ctx, a, a.ball_list, a.ball_count);
    if (!rc) rc = launch_generic_sweep<float>(ctx, a, false);
    span_end(ctx, sp2);
    return rc;
}

template <> int launch_sweep<double>(wtp_ctx* ctx, SearchArgs<double>& a, bool fresh) {
    ctx->n_sweep_launches += 1;
    // ClippedSpacingForce on a fresh snapshot: compact-support brick sweep (wtp_brick64.hip), the wave
    // kernel takes what it hands back.  Everything else in fp64: the exact wave-per-query path.
    const bool cs = fresh && a.brick_hcap > 0 && a.force_kind == WTP_FORCE_CLIPPED_SPACING && a.k >= 2 &&
                    a.k <= kFastKMax - 1 && !ctx->full_select && !ctx->force_generic;
    if (!cs) {
        const int sp = span_begin(ctx, 1);
        int rc;
        if (!fresh && a.ball_list && a.force_kind == WTP_FORCE_CLIPPED_SPACING && a.k >= 2 && a.k <= kFastKMax - 1 &&
            !ctx->full_select && !ctx->force_generic) {
            // a stale snapshot and the default law: every query through the Float64 ball kernel (as in fp32, launch_sweep<float>)
            rc = launch_cs_all_slots(ctx, a.fb_list, a.n, a.fb_count);
            if (!rc) rc = launch_cs_ball64(ctx, a, a.ball_list, a.ball_count);
            if (!rc) rc = launch_generic_sweep<double>(ctx, a, false);
        } else {
            rc = launch_generic_sweep<double>(ctx, a, true);
        }
        span_end(ctx, sp);
        return rc;
    }
    a.gamma_cap = ctx->gamma_cap;
    const int sp = span_begin(ctx, 1);
    int rc = launch_brick_cs<double>(ctx, a);
    // variable spacing: the hand-backs whose support ball is wider than a cell (wtp_ball64.hip), before the exact path
    if (!rc && a.spacing_pp && a.ball_list) rc = launch_cs_ball64(ctx, a, a.ball_list, a.ball_count);
    span_end(ctx, sp);
    if (rc) return rc;
    const int sp2 = span_begin(ctx, 2);
    rc = launch_generic_sweep<double>(ctx, a, false);
    span_end(ctx, sp2);
    return rc;
}

} // namespace wtp
