// wtp_ksel.hip — k-selection on the x-slowest halo layout (round 3): KNNTopology rows and the repel sweep with the
// explicit k-selection (every force law), fp32 3-D clouds, k + self <= 24.
//
// What it computes is what brick_kernel<0,..> / brick_kernel<1,..,0> (wtp_brick.hip) compute —
// `_build_knn_neighbors` (src/topology.jl:79-84) and the sweep closure of `_relax!` (src/repel.jl:256-292) with the
// k nearest taken explicitly — on the layout of the default sweep (wtp_cs2.hip) instead of 4 x 4 x 4 bricks of
// 9-point cells:
//
//   * cells of ~1.2 points (edge c ~ 0.62 r_k): the k nearest lie inside the 5 x 5 x 5 block around the query's cell
//     (provable radius >= 2c - margin ~ 1.25 r_k), which holds ~150 points instead of the 243 of 27 nine-point cells;
//   * bricks of BX x 2 x 2 own cells, halo (BX+4) x 6 x 6 staged in LDS in (hx, hz, hy) order — x SLOWEST: the
//     5 x 5 x 5 block of a query is then ONE contiguous run of 173 halo cells (its 125 plus 48 cells three rows away in
//     y or z, whose points fail the distance test by construction).  The scan is one software-pipelined loop over
//     ~208 consecutive LDS slots, the same trip count for every lane: no rows, no run queue, no per-lane pace;
//   * hits (d2 <= tau, tau = min(provable radius, density-scaled cap)^2) are a bit mask of the run's slots: one
//     v_cmp + v_addc per candidate into the word being filled, a finished word (32 slots) parked in the lane's ring —
//     one LDS store per 32 candidates, and the loop is the code of one word; the eight words are read back, their set
//     bits written as run positions (one byte each) to the same ring, read back four at a time, and turned into 64
//     sort keys: upper 24 bits of the canonical d2 | position;
//   * one 64-key Batcher network (VGPRs); the first k + 2 entries are looked up again for their exact (d2, id),
//     put in canonical order where the truncated keys tie, and certified against tau.
//
// Whatever the fast path cannot certify (fewer than k hits inside tau, more than 64 hits, a run longer than 256
// slots, a tie that reaches past the window, a coincident neighbour in the sweep, LDS overflow) is appended to
// a.fb_list for the exact wave path, so results are always the exact canonical lists / sums.
//
// Roofline: 16 B/point in, 4 k B/point out (rows) or 28 B/point (sweep); several thousand VALU lane-ops per query:
// bound by vector-ALU issue, not by HBM.
#include "wtp_device.hpp"
#include "wtp_sortnet.hpp"

namespace wtp {

#ifndef WTP_DIAG
#define WTP_DIAG 0 // diagnostic build: s_memtime stamps per phase (never quote its run time)
#endif
#define KS_STAMP(i)                                                 \
    if (WTP_DIAG) {                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        dt[i] += t_ - t_last;                                       \
        t_last = t_;                                                \
    }

#define WTP_CE(k, i, j)                       \
    {                                         \
        auto lo_ = k[i] < k[j] ? k[i] : k[j]; \
        auto hi_ = k[i] < k[j] ? k[j] : k[i]; \
        k[i] = lo_;                           \
        k[j] = hi_;                           \
    }

constexpr int kKsThreads = 256;
constexpr int kKsSlab = 36;                              // halo cells per x index: (2+4) x (2+4)
constexpr int kKsRows = 9;                               // halo x-rows per wave
constexpr int kKsMaxBX = 59;                             // own cells along x; halo + the closing column <= 64 lanes
constexpr int kKsMaxCells = (kKsMaxBX + 4) * kKsSlab;
constexpr int kKsRun = 4 * kKsSlab + 4 * 6 + 4 + 1;      // halo cells in a query's run: 173
constexpr int kKsWords = 8;                              // hit-mask registers: 256 slots of a run
constexpr int kKsSU = 8;                                 // candidates per scan step
constexpr int kKsPadBytes = 2 * kKsSU * 16;              // far sentinels behind the staged points
constexpr int kKsMaxQ = 512;                             // queries per brick the lane table covers
constexpr int kKsRing = 64;                              // hits per query = keys of the network
constexpr int kKsKMax = 24;                              // largest k (self included where it is searched): the reference's 21 + self, + 2 candidates for the fp64 re-ranking
constexpr int kKsRingLane = 68;                          // bytes of a lane's ring: 64 entries, 17 dwords apart (odd: the lanes of a wave start in 64 different banks)
constexpr int kKsWaveBytes = 64 * kKsKMax * 4 + 256;     // per wave: the lanes' rings, later the wave's 64 rows (k ids each) + their row ids, written out together
constexpr int kKsRingBytes = (kKsThreads / 64) * kKsWaveBytes;
static_assert(64 * kKsRingLane <= kKsWaveBytes, "the rows' staging area holds the rings");

struct KsSmem {
    uint16_t ls[kKsMaxCells + 4];        // LDS slot of the first point of each halo cell, (hx, hz, hy) order
    uint32_t hown[kKsMaxBX + 5][4];      // index in the sorted array of the first point of the four own cells of a slab
    uint16_t qpref[kKsMaxBX + 6];        // first query of each own slab (index = hx; entry BX + 2 closes the table)
    uint8_t qslab[kKsMaxQ];              // slab (hx) of each query
    uint32_t wsum[4][64], wcnt[4][64], wown[2][64]; // per wave and column: sums of cell_start / of the counts (see the table phase)
    Acc acc[kKsThreads / 64];
};

typedef float ks_f4 __attribute__((ext_vector_type(4)));

__device__ inline float4 ks_pt(const unsigned char* base, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(base + byte_off);
}
__device__ inline uint32_t ks_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// Fast-math force law (1-ulp rcp), the expression of brick_kernel's force_fast: laws 0..2 are
//   max((A - B u2) / (u2 + beta)^2, lo)  with (A, B, lo) = (1,0,-inf) | (1,1,-inf) | (u0^2,1,0);  law 3: (u2 + beta)^gamma below.
struct KsForce {
    float A, B, lo, beta, gamma;
    int strong;
};
__device__ inline KsForce ks_force_coef(int kind, float beta, float u0, float gamma) {
    KsForce c;
    c.A = kind == WTP_FORCE_CLIPPED_SPACING ? u0 * u0 : 1.f;
    c.B = kind == WTP_FORCE_INVERSE_DISTANCE ? 0.f : 1.f;
    c.lo = kind == WTP_FORCE_CLIPPED_SPACING ? 0.f : -Lim<float>::inf();
    c.beta = beta;
    c.gamma = gamma;
    c.strong = kind == WTP_FORCE_STRONG_SPACING;
    return c;
}
__device__ inline float ks_force(const KsForce& c, float u2) {
    const float d = u2 + c.beta;
    float inv = __builtin_amdgcn_rcpf(d * d);
    if (c.strong) inv = __builtin_amdgcn_exp2f(-c.gamma * __builtin_amdgcn_logf(d)); // wave-uniform
    const float f = (c.A - c.B * u2) * inv;
    return f > c.lo ? f : c.lo;
}

static size_t ksel_smem_bytes(int hcap) {
    return (size_t)hcap * 16 + kKsPadBytes + kKsRingBytes + sizeof(KsSmem);
}

// MODE 0: KNNTopology rows.  MODE 1: repel sweep.  KT > 0: k known at compile time.
template <int MODE, int KT>
__global__ __launch_bounds__(kKsThreads, 2) void ksel_kernel(SearchArgs<float> a, int hcap, int bx_max) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* pts = reinterpret_cast<float4*>(smem_raw);
    const uint32_t ring_off = (uint32_t)hcap * 16u + (uint32_t)kKsPadBytes;
    KsSmem* sm = reinterpret_cast<KsSmem*>(smem_raw + ring_off + kKsRingBytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: row geometry stays in scalar registers

    const Grid<float> g = *a.grid;
    const int K = KT > 0 ? KT : a.k;
    constexpr int KR = KT > 0 ? KT : kKsKMax; // rows / sums look at the first KR sorted entries
    constexpr int KW = KR + 2;                // window that is put in exact order
    const bool skip_self = MODE == 0 && !a.include_self;
    // bricks along x: as few as bx_max allows, of equal length
    const int nbx = (g.n[0] + bx_max - 1) / bx_max;
    const int BX = (g.n[0] + nbx - 1) / nbx;
    const int HX = BX + 4, ncell = HX * kKsSlab;
    const int nby = (g.n[1] + 1) / 2, nbz = (g.n[2] + 1) / 2;
    const int nbricks = nbx * nby * nbz;
    Acc acc = acc_empty();
    unsigned long long dt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = WTP_DIAG ? __builtin_amdgcn_s_memtime() : 0ull;

    // XCD-aware brick order (blocks sharing blockIdx % 8 share an L2): one contiguous slab of bricks each
    const int groups = 8;
    const int per = (nbricks + groups - 1) / groups;
    const int xcd = blockIdx.x % groups;
    const int lane_blk = blockIdx.x / groups;
    const int blk_per_group = gridDim.x / groups;
    const int b_end = (xcd + 1) * per < nbricks ? (xcd + 1) * per : nbricks;

    struct BrickPos {
        int bx, by, bz;
    };
    auto brick_pos = [&](int brick) {
        BrickPos p;
        p.bx = brick % nbx;
        p.by = (brick / nbx) % nby;
        p.bz = brick / (nbx * nby);
        return p;
    };
    // the cell table of a brick is loaded one brick ahead (nine registers), its points right after the table phase:
    // wave w owns the halo x-rows r = 9w .. 9w+8 (r = hz*6 + hy), lane = column hx.  cell_start at the column clamped
    // to the grid: column 0 is the row's first index in the sorted array, column HX its one-past-last.
    auto load_cells = [&](const BrickPos& bp, int (&vv)[kKsRows]) {
        const int ox = bp.bx * BX - 2, oy = bp.by * 2 - 2, oz = bp.bz * 2 - 2;
        const int gx_lo = ox < 0 ? 0 : ox, gx_hi = (ox + HX - 1) < g.n[0] - 1 ? (ox + HX - 1) : g.n[0] - 1;
        int gxc = ox + lane;
        gxc = gxc < gx_lo ? gx_lo : (gxc > gx_hi + 1 ? gx_hi + 1 : gxc);
#pragma unroll
        for (int j = 0; j < kKsRows; ++j) {
            const int r = wave * kKsRows + j;
            const int gy = oy + r % 6, gz = oz + r / 6;
            const bool row_ok = gy >= 0 && gy < g.n[1] && gz >= 0 && gz < g.n[2];
            // (a row outside the grid reads cell_start[0] = 0 in every column: no points)
            vv[j] = a.cell_start[row_ok ? (gz * g.n[1] + gy) * g.n[0] + gxc : 0];
        }
    };

    // the first 64 points of each of the wave's rows (longer rows: the rest at staging time)
    auto load_points = [&](const int (&vv)[kKsRows], float4 (&pp)[kKsRows], int ln) {
        const int last = a.n - 1;
#pragma unroll
        for (int j = 0; j < kKsRows; ++j) {
            const int gs = __builtin_amdgcn_readfirstlane(vv[j]);
            const int len = __builtin_amdgcn_readlane(vv[j], HX) - gs;
            const int i = gs + (ln < len ? ln : 0);
            pp[j] = a.snap[i < last ? i : last];
        }
    };
    int brick = xcd * per + lane_blk;
    int v[kKsRows];
#pragma unroll
    for (int j = 0; j < kKsRows; ++j) v[j] = 0;
    if (brick < b_end) load_cells(brick_pos(brick), v);
    for (; brick < b_end; brick += blk_per_group) {
        // lane / thread index as the table phase sees them: opaque per brick, so the LDS addresses built from them are
        // recomputed here instead of being hoisted out of the brick loop and spilled across the queries (a spill reload in
        // this phase costs a `s_waitcnt vmcnt(0)`, i.e. the full latency of the rows' point loads in flight)
        int ln = lane, tv = tid;
        asm volatile("" : "+v"(ln), "+v"(tv));
        const BrickPos pos = brick_pos(brick);
        const int ox = pos.bx * BX - 2, oy = pos.by * 2 - 2, oz = pos.bz * 2 - 2; // halo origin (cell coordinates)
        const int next = brick + blk_per_group;

        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // previous brick's LDS no longer in use
        KS_STAMP(8) // wait for the brick's slowest wave
        // ---- 1. cell table: the prefix the LDS order (hx, hz, hy) needs — points left of column hx in all rows, plus
        //         the column's cells in rows before r — is a sum of cell_start values: no scan across lanes ----------
        int cn[kKsRows], row_gs[kKsRows], row_len[kKsRows];
        float4 pv[kKsRows];
        load_points(v, pv, ln); // in flight while the tables are built (held across the queries of the previous brick they cost
                            // more in spills than the wait they save: measured)
        {
            int vs = 0, cs = 0;
#pragma unroll
            for (int j = 0; j < kKsRows; ++j) {
                const int nxt = __builtin_amdgcn_ds_bpermute(((ln + 1) & 63) << 2, v[j]); // (the library shuffles rebuild the lane id: one more value kept across the queries)
                cn[j] = ln < HX ? nxt - v[j] : 0;
                row_gs[j] = __builtin_amdgcn_readfirstlane(v[j]);
                row_len[j] = __builtin_amdgcn_readlane(v[j], HX) - row_gs[j];
                vs += v[j];
                cs += cn[j];
            }
            if (ln <= HX) {
                sm->wsum[wave][ln] = (uint32_t)vs;
                sm->wcnt[wave][ln] = (uint32_t)cs;
                // own rows: r = 14, 15 (wave 1, j = 5, 6) and r = 20, 21 (wave 2, j = 2, 3)
                if (wave == 1) sm->wown[0][ln] = (uint32_t)(v[5] + v[6]);
                if (wave == 2) sm->wown[1][ln] = (uint32_t)(v[2] + v[3]);
            }
            if (ln >= 2 && ln < HX - 2) {
                if (wave == 1) {
                    sm->hown[ln][0] = (uint32_t)v[5];
                    sm->hown[ln][1] = (uint32_t)v[6];
                }
                if (wave == 2) {
                    sm->hown[ln][2] = (uint32_t)v[2];
                    sm->hown[ln][3] = (uint32_t)v[3];
                }
            }
        }
        __syncthreads();
        KS_STAMP(0) // cell table
        int halo_total;
        {
            uint32_t left = 0, g0 = 0, before = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                left += sm->wsum[w][ln <= HX ? ln : 0];
                g0 += sm->wsum[w][0];
                tot += sm->wsum[w][HX];
                before += w < wave ? sm->wcnt[w][ln <= HX ? ln : 0] : 0u;
            }
            halo_total = (int)(tot - g0);
            if (ln < HX) {
                uint32_t e = left - g0 + before;
#pragma unroll
                for (int j = 0; j < kKsRows; ++j) {
                    sm->ls[ln * kKsSlab + wave * kKsRows + j] = (uint16_t)(e > 0xFFFFu ? 0xFFFFu : e);
                    e += (uint32_t)cn[j];
                }
            }
            if (tv == 0) sm->ls[ncell] = (uint16_t)(halo_total > 0xFFFF ? 0xFFFF : halo_total);
            // own queries in front of slab hx (slabs 2 .. BX+1 hold own cells; entry BX+2 closes the table)
            if (wave == 3 && ln >= 2 && ln <= BX + 2) {
                const uint32_t q = (sm->wown[0][ln] + sm->wown[1][ln]) - (sm->wown[0][2] + sm->wown[1][2]);
                sm->qpref[ln] = (uint16_t)(q > 0xFFFFu ? 0xFFFFu : q);
            }
        }
        // (no barrier here: what follows reads table entries its own wave wrote — the `ls` of its nine rows — or the column
        // sums that were complete at the barrier above; the other waves' entries are read after the staging barrier)
        KS_STAMP(9) // prefix tables
        const bool overflow = halo_total > hcap;
        if (tv >= 2 && tv <= BX + 1) {
            const uint32_t qb0 = sm->wown[0][2] + sm->wown[1][2];
            const uint32_t qa = (sm->wown[0][tv] + sm->wown[1][tv]) - qb0, qe = (sm->wown[0][tv + 1] + sm->wown[1][tv + 1]) - qb0;
            const int q0 = (int)(qa > 0xFFFFu ? 0xFFFFu : qa), q1 = (int)(qe > 0xFFFFu ? 0xFFFFu : qe);
            for (int q = q0; q < q1 && q < kKsMaxQ; ++q) sm->qslab[q] = (uint8_t)tv;
        }
        // ---- 2. stage the halo: global rows are contiguous, the LDS order is (hx, hz, hy) ---------------
        if (!overflow) {
            // destination = start of the point's cell in LDS + its rank in the cell; the cell's first index in the sorted
            // array is the table value of lane hx (a cross-lane read, every lane takes part)
            auto place = [&](const float4 p, int i, int j, int vj) { // i: index in the row; every lane takes part (cross-lane read)
                int hx = cell_coord(g, p.x, 0) - ox;
                hx = hx < 0 ? 0 : (hx > HX - 1 ? HX - 1 : hx);
                const int cell_gs = __builtin_amdgcn_ds_bpermute(hx << 2, vj);
                int dest = (int)sm->ls[hx * kKsSlab + wave * kKsRows + j] + (row_gs[j] + i - cell_gs);
                dest = dest < 0 ? 0 : (dest > hcap - 1 ? hcap - 1 : dest); // never outside the point area
                if (i < row_len[j]) pts[dest] = p;
            };
#pragma unroll
            for (int j = 0; j < kKsRows; ++j) place(pv[j], ln, j, v[j]);
#pragma unroll
            for (int j = 0; j < kKsRows; ++j) // rows longer than a wave: the rest, row by row
                for (int i0 = 64; i0 < row_len[j]; i0 += 64) {
                    const int i = i0 + ln;
                    place(a.snap[row_gs[j] + (i < row_len[j] ? i : 0)], i, j, v[j]);
                }
            // one scan step past the staged points: far sentinels instead of another brick's leftovers
            if (tv < 2 * kKsSU) pts[halo_total + tv] = make_float4(1e30f, 1e30f, 1e30f, 0.f);
        }
        // the next brick's cell table: in flight while this brick's queries run
        if (next < b_end) load_cells(brick_pos(next), v);
        KS_STAMP(10) // staging (waits for the rows' points)
        __syncthreads();
        KS_STAMP(1) // prefix tables, staging

        // ---- 3. queries ------------------------------------------------------------------------------
        const int Q = sm->qpref[BX + 2];
        if (overflow || Q > kKsMaxQ) {
            // dense brick (halo larger than the LDS point area, or more queries than the lane table holds):
            // every own point goes to the exact path.  Own points = four x-rows of the sorted array.
            for (int r4 = 0; r4 < 4; ++r4) {
                const int gy = oy + 2 + (r4 & 1), gz = oz + 2 + (r4 >> 1);
                if (gy >= g.n[1] || gz >= g.n[2]) continue;
                const int gx0 = pos.bx * BX, gx1 = (gx0 + BX - 1) < g.n[0] - 1 ? (gx0 + BX - 1) : g.n[0] - 1;
                const int base = (gz * g.n[1] + gy) * g.n[0];
                const int s = a.cell_start[base + gx0], e = a.cell_start[base + gx1 + 1];
                for (int i = s + tid; i < e; i += kKsThreads) {
                    const int p = atomicAdd(a.fb_count, 1);
                    a.fb_list[p] = i;
                }
            }
            continue;
        }
        __builtin_amdgcn_s_setprio(1); // waves that run queries issue ahead of waves that stage
        for (int qb = 0; qb < Q; qb += kKsThreads) {
            const int q = qb + tid;
            const bool active = q < Q;
            // lane table -> (slab, own cell, slot)
            int hx = 2, cellr = 14, slot = 0, gslot = 0;
            if (active) {
                hx = sm->qslab[q];
                const int off = q - (int)sm->qpref[hx];
                const int b = hx * kKsSlab;
                const int s14 = sm->ls[b + 14], s15 = sm->ls[b + 15], s16 = sm->ls[b + 16];
                const int s20 = sm->ls[b + 20], s21 = sm->ls[b + 21];
                const int na = s16 - s14;
                int which;
                if (off < na) {
                    slot = s14 + off;
                    which = slot < s15 ? 0 : 1;
                } else {
                    slot = s20 + (off - na);
                    which = slot < s21 ? 2 : 3;
                }
                cellr = which == 0 ? 14 : (which == 1 ? 15 : (which == 2 ? 20 : 21));
                const int cell_ls = which == 0 ? s14 : (which == 1 ? s15 : (which == 2 ? s20 : s21));
                gslot = (int)sm->hown[hx][which] + (slot - cell_ls);
            }
            // rows are written out by the whole wave together: a lane without a query runs along as one that gave up
            if (MODE == 1 ? !active : !__any(active)) continue;
            const uint32_t qoff = (uint32_t)slot * 16u;
            const float4 qp = pts[slot];
            const int32_t qid = w_to_id(qp.w);
            if (MODE == 1 && qid < a.n_fixed) { // the wall: never moves (src/repel.jl:80,256)
                a.out[gslot] = qp;
                a.forces[gslot] = 0.f;
                a.nn_dist[gslot] = Lim<float>::inf();
                a.nn_id[gslot] = -1;
                continue;
            }
            const int hy = cellr % 6, hz = cellr / 6;
            const int cx = ox + hx, cy = oy + hy, cz = oz + hz;
            const float g2 = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, 2);
            // candidate run: halo cells P(hx-2, hz-2, hy-2) .. P(hx+2, hz+2, hy+2), contiguous in LDS
            const int Pf = (hx - 2) * kKsSlab + (hz - 2) * 6 + (hy - 2);
            const uint32_t pa0 = (uint32_t)sm->ls[Pf] * 16u, ea = (uint32_t)sm->ls[Pf + kKsRun] * 16u;
            const uint32_t len = (ea - pa0) >> 4;
            // first filter radius: the ball that is EXPECTED to hold a.cap_count points at the density this query sees
            // (its run of 173 cells holds len): r^3 = cap_count / len * 173 / (4 pi / 3) cell volumes
            float tau = g2;
            if (a.cap_count > 0.f) {
                const float x = a.cap_count * 41.3007f / (float)(len > 1u ? len : 1u);
                const float capq = (g.c * g.c) * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * 0.66666667f);
                tau = capq < tau ? capq : tau;
            }
            bool giveup = len > (uint32_t)(kKsWords * 32) || !active;
            const uint32_t ea_s = giveup ? pa0 : ea; // (a lane that gave up must not prolong the wave's loop)
            // scan filter: d2 with fused multiply-adds against a threshold 4 ulp wider; the exact canonical d2 of every
            // hit is recomputed for its key, so the filter only has to be conservative
            const float tau_s = giveup ? -1.f : tau * (1.f + 0x1p-21f);
            const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
            uint32_t m[kKsWords]; // slot i of the run = bit (31 - i % 32) of word i / 32
            const uint32_t ring_l = ring_off + (uint32_t)wave * kKsWaveBytes + (uint32_t)lane * kKsRingLane; // this lane's ring
            KS_STAMP(2) // query setup
            int nw = 0; // mask words the wave's runs needed (wave-uniform)
            {
                // Software pipeline over half steps of four candidates (wtp_cs2.hip): a register group is refilled as soon
                // as its three subtractions have consumed it, a use waits only for ITS read (`lgkmcnt(3)`).  One trip of
                // the loop fills one mask word (four steps of eight); finished words are parked in the lane's ring, so the
                // loop is code of one word, not of eight (the instruction cache is shared by two compute units).
                ks_f4 c[4];
                uint32_t pa = pa0;
                uint32_t addr = lds_base + pa0;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                             : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                             : "v"(addr)
                             : "memory");
#define KS_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg)::"memory")
#define KS_REFILL(reg, a, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(reg) : "v"(a), "n"(off) : "memory")
#define KS_TEST(dist)                                                             \
    asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" \
                 : "+v"(hb)                                                        \
                 : "v"(dist), "v"(thr)                                             \
                 : "vcc")
                for (;;) {
                    uint32_t hb = 0;
                    bool half = false;
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        // a lane whose run has ended keeps reading its first slots with an impossible threshold.  Inside a
                        // run no end mask is needed: the slots a last step reads past the run's end belong to cells three
                        // away from the query (or are the sentinels behind the last cell) and fail the distance test.
                        const float thr = pa < ea_s ? tau_s : -1.f;
                        const uint32_t pn = pa + 16u * kKsSU;
                        const uint32_t addr_n = lds_base + (pn < ea_s ? pn : pa0); // what the NEXT step reads
#pragma unroll
                        for (int u = 0; u < 4; ++u) { // slots 0..3; the group is refilled with slots 4..7 of this step
                            KS_WAIT3(c[u]);
                            const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                            switch (u) {
                                case 0: KS_REFILL(c[0], addr, 64); break;
                                case 1: KS_REFILL(c[1], addr, 80); break;
                                case 2: KS_REFILL(c[2], addr, 96); break;
                                default: KS_REFILL(c[3], addr, 112); break;
                            }
                            const float d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                            KS_TEST(d);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { // slots 4..7; refilled with slots 0..3 of the next step
                            KS_WAIT3(c[u]);
                            const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                            switch (u) {
                                case 0: KS_REFILL(c[0], addr_n, 0); break;
                                case 1: KS_REFILL(c[1], addr_n, 16); break;
                                case 2: KS_REFILL(c[2], addr_n, 32); break;
                                default: KS_REFILL(c[3], addr_n, 48); break;
                            }
                            const float d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                            KS_TEST(d);
                        }
                        pa = pn;
                        addr = addr_n;
                        if (st == 1 && !__any(pa < ea_s)) { // the wave's runs end in the first half of this word
                            hb <<= 16;
                            half = true;
                            break;
                        }
                    }
                    *reinterpret_cast<uint32_t*>(smem_raw + ring_l + 4u * (uint32_t)nw) = hb;
                    ++nw;
                    if (half || nw == kKsWords || !__any(pa < ea_s)) break;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])::"memory"); // the reads issued for a step that does not come
#undef KS_WAIT3
#undef KS_REFILL
#undef KS_TEST
            }
#pragma unroll
            for (int w = 0; w < kKsWords; ++w) {
                const uint32_t mw = *reinterpret_cast<const uint32_t*>(smem_raw + ring_l + 4u * (uint32_t)w);
                m[w] = w < nw ? mw : 0u;
            }
            // the query's own bit (always set: d = 0): rows without the point itself drop it here
            if (skip_self) {
                const uint32_t ps = (qoff - pa0) >> 4;
                const uint32_t sb = 0x80000000u >> (ps & 31u);
#pragma unroll
                for (int w = 0; w < kKsWords; ++w) m[w] &= (ps >> 5) == (uint32_t)w ? ~sb : ~0u;
            }
            int cnt = 0;
#pragma unroll
            for (int w = 0; w < kKsWords; ++w) cnt += __builtin_popcount(m[w]);
            if (cnt > kKsRing) { // a dense cluster inside the filter ball: exact path
                giveup = true;
                cnt = 0;
#pragma unroll
                for (int w = 0; w < kKsWords; ++w) m[w] = 0u;
            }
            KS_STAMP(3) // scan
            // ---- hits -> run positions in the lane's ring (bytes), taken off the mask words from the top (v_ffbh) ----
            {
                uint32_t ja = ring_l; // next free entry (the mask words parked there are in registers by now)
#pragma unroll
                for (int w = 0; w < kKsWords; ++w) {
                    uint32_t mw = m[w];
                    while (__any(mw != 0u)) {
#pragma unroll
                        for (int rep = 0; rep < 2; ++rep) {
                            const bool on = mw != 0u;
                            uint32_t l;
                            asm("v_ffbh_u32 %0, %1" : "=v"(l) : "v"(mw));
                            mw &= ~(0x80000000u >> (l & 31u));
                            if (on) {
                                smem_raw[ja] = (unsigned char)((uint32_t)(w * 32) + l);
                                ++ja;
                            }
                        }
                    }
                }
            }
            KS_STAMP(4) // extraction
            // ---- keys: upper 24 bits of the canonical d2 | run position (truncation is monotone: entries whose
            //      truncated d2 differ come out of the network in exact order) ------------------------------------
            uint32_t k[kKsRing];
            float dd[KW];
            int32_t ii[KW];
            bool risky = false;
            {
                // Lookups are issued a chunk of four ahead of their use (a scheduling barrier keeps them there): with two waves
                // per SIMD the LDS round trip of a chunk is otherwise exposed once per chunk; the compiler's counted waits (LDS
                // returns in order) let a chunk's use wait for that chunk only.  KS_USE makes the whole register tuple live at
                // the point of use, so the read stays a ds_read_b128 (a b96 costs twice the LDS cycles) without being waited
                // for where it is issued; KS_KEEP keeps a value out of the branches the compiler would otherwise build around
                // every `valid ? f(x) : sentinel`.  (Explicit asm reads + hand-placed waits, as in the scan, are not safe
                // here: at this register pressure the allocator may copy a register between the asm that issues its load and
                // the asm that waits for it.)
#define KS_USE(v4) asm volatile("" : "+v"(v4))
#define KS_KEEP(x) asm volatile("" : "+v"(x))
                auto ld = [&](uint32_t off) { return *reinterpret_cast<const ks_f4*>(smem_raw + off); };
                ks_f4 cA[4], cB[4];
                uint32_t pA[4], pB[4];
                // ---- keys from the ring: positions four to a word ----
                uint32_t rp[kKsRing / 4];
#pragma unroll
                for (int gq = 0; gq < kKsRing / 4; ++gq) rp[gq] = *reinterpret_cast<const uint32_t*>(smem_raw + ring_l + 4u * (uint32_t)gq);
                auto chunk_issue = [&](int c4, uint32_t (&ps)[4], ks_f4 (&cc)[4]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ps[u] = (c4 * 4 + u) < cnt ? ((rp[c4] >> (8 * u)) & 255u) : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) cc[u] = ld(pa0 + ps[u] * 16u);
                };
                chunk_issue(0, pA, cA);
#pragma unroll
                for (int c4 = 0; c4 < kKsRing / 4; ++c4) {
                    ks_f4(&cc)[4] = (c4 & 1) ? cB : cA;
                    ks_f4(&cn)[4] = (c4 & 1) ? cA : cB;
                    uint32_t(&pc)[4] = (c4 & 1) ? pB : pA;
                    uint32_t(&pn)[4] = (c4 & 1) ? pA : pB;
                    if (__any(cnt > c4 * 4)) { // wave-uniform: skip chunks no lane has filled
                        if (c4 + 1 < kKsRing / 4 && __any(cnt > (c4 + 1) * 4)) chunk_issue(c4 + 1, pn, cn);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            KS_USE(cc[u]);
                            uint32_t key = (ks_f2u(dist2<float>(qp.x, qp.y, qp.z, cc[u].x, cc[u].y, cc[u].z)) & ~255u) | pc[u];
                            KS_KEEP(key);
                            k[c4 * 4 + u] = (c4 * 4 + u) < cnt ? key : 0x7F800000u;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) k[c4 * 4 + u] = 0x7F800000u;
                    }
                }
                KS_STAMP(5) // keys
                WTP_SORTNET_64(k)
                // ---- window: exact (d2, id) of the first K + 2 sorted entries, chunks of four ----
                constexpr int NCH = (KW + 3) / 4;
                auto win_issue = [&](int ch, ks_f4 (&cc)[4]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) cc[u] = ld(pa0 + (k[ch * 4 + u] & 255u) * 16u);
                };
                win_issue(0, cA);
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    ks_f4(&cc)[4] = (ch & 1) ? cB : cA;
                    ks_f4(&cn)[4] = (ch & 1) ? cA : cB;
                    if (ch + 1 < NCH) win_issue(ch + 1, cn);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = ch * 4 + u;
                        if (j < KW) {
                            const bool on = j < cnt && j < K + 2;
                            KS_USE(cc[u]);
                            float d = dist2<float>(qp.x, qp.y, qp.z, cc[u].x, cc[u].y, cc[u].z);
                            KS_KEEP(d);
                            dd[j] = on ? d : Lim<float>::inf();
                            const float wf = cc[u].w; // (by value: __builtin_bit_cast applied to the element expression itself reads element 0)
                            ii[j] = on ? w_to_id(wf) : 0x7FFFFFFF;
                            if (j + 1 < KW) risky = risky || (j + 1 < cnt && j + 1 < K + 2 && ((k[j] ^ k[j + 1]) < 256u));
                        }
                    }
                }
#undef KS_USE
#undef KS_KEEP
            }
            // a bucket that straddles the window's end: its members outside the window might belong inside (a mass tie)
            bool tie_out = false;
            if (KT > 0) {
                tie_out = cnt > KT + 2 && ((k[KT + 2] ^ k[KT + 1]) < 256u) && ((k[KT + 1] ^ k[KT - 1]) < 256u);
            } else {
                uint32_t kin = k[0], kend = k[1], kout = k[2]; // entries K - 1, K + 1, K + 2
#pragma unroll
                for (int j = 1; j < kKsKMax; ++j) {
                    kin = (j == K - 1) ? k[j] : kin;
                    kend = (j == K - 1) ? k[j + 2] : kend;
                    kout = (j == K - 1) ? k[j + 3] : kout;
                }
                tie_out = cnt > K + 2 && ((kout ^ kend) < 256u) && ((kend ^ kin) < 256u);
            }
            // entries that share a bucket may be out of order: exact (d2, id) exchange passes until none moves
            bool again = risky;
            while (__any(again)) {
                again = false;
#pragma unroll
                for (int j = 0; j + 1 < KW; ++j) {
                    const bool sw = lex_lt(dd[j + 1], ii[j + 1], dd[j], ii[j]);
                    const float td = dd[j];
                    const int32_t ti = ii[j];
                    dd[j] = sw ? dd[j + 1] : td;
                    ii[j] = sw ? ii[j + 1] : ti;
                    dd[j + 1] = sw ? td : dd[j + 1];
                    ii[j + 1] = sw ? ti : ii[j + 1];
                    again = again || sw;
                }
            }
            float cutd = dd[KR - 1];
            int32_t cuti = ii[KR - 1];
            if (KT == 0) {
#pragma unroll
                for (int j = 0; j < KR - 1; ++j) {
                    cutd = (j == K - 1) ? dd[j] : cutd;
                    cuti = (j == K - 1) ? ii[j] : cuti;
                }
            }
            // certified: k hits, the k-th inside the radius the filter (and the searched block) is complete for
            const bool fallback = giveup || tie_out || cnt < K || !(cutd <= tau);
            KS_STAMP(6) // network, window
            if (MODE == 0) {
                // Rows go out through LDS: a lane's k ids are 84 bytes at a random place of the row array, so 64 lanes
                // writing their own rows touch 64 cache lines per store instruction (measured: 18 % of the kernel's wave
                // time).  Each lane parks its row in the wave's staging area (the rings are dead by now), then the wave
                // writes the 64 rows as one stream: consecutive lanes = consecutive ids of a row.
                uint32_t* ost = reinterpret_cast<uint32_t*>(smem_raw + ring_off + (uint32_t)wave * kKsWaveBytes);
                int32_t* oq = reinterpret_cast<int32_t*>(ost + 64 * kKsKMax);
                const bool row_ok = active && !fallback;
                __builtin_amdgcn_wave_barrier(); // (every ring read of the wave is done: program order)
                oq[lane] = row_ok ? qid : -1;
                const uint32_t mK = (65536u + (uint32_t)K - 1u) / (uint32_t)K; // e / K = (e * mK) >> 16 for e < 64 K
                for (int pass = 0; pass < (a.dist_out ? 2 : 1); ++pass) {
                    if (row_ok) {
#pragma unroll
                        for (int j = 0; j < KR; ++j)
                            if (j < K) ost[lane * K + j] = pass == 0 ? (uint32_t)ii[j] : ks_f2u(wsqrt(dd[j]));
                    }
                    __builtin_amdgcn_wave_barrier();
                    uint32_t* dst = pass == 0 ? reinterpret_cast<uint32_t*>(a.idx_out) : reinterpret_cast<uint32_t*>(a.dist_out);
                    // element e = lane + 64 j of the wave's 64 x K block; all reads first, then the stores
                    int32_t rq[KR];
                    uint32_t val[KR];
                    int col[KR];
#pragma unroll
                    for (int j = 0; j < KR; ++j) {
                        const int e = (j < K) ? lane + 64 * j : lane;
                        const int row = (int)(((uint32_t)e * mK) >> 16);
                        col[j] = e - row * K;
                        rq[j] = oq[row];
                        val[j] = ost[e];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < KR; ++j)
                        if (j < K && rq[j] >= 0) dst[(int64_t)rq[j] * K + col[j]] = val[j];
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                if (!fallback) {
                    // the force over the k nearest, self skipped by index (src/repel.jl:270-280): membership = not after
                    // the k-th pair in canonical order; the window entries are read again for their coordinates
                    const float s = a.spacing_pp ? a.spacing_pp[qid] : a.spacing_const;
                    const float inv_s2 = 1.f / (s * s);
                    const KsForce fc = ks_force_coef(a.force_kind, a.beta, a.u0, a.gamma);
                    float Fx = 0.f, Fy = 0.f, Fz = 0.f, nd2 = Lim<float>::inf();
                    int32_t nid = 0x7FFFFFFF;
                    bool coincident = false;
#pragma unroll
                    for (int j0 = 0; j0 < KW; j0 += 4) {
                        float4 c[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (j0 + u < KW) c[u] = ks_pt(smem_raw, pa0 + (k[j0 + u] & 255u) * 16u);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int j = j0 + u;
                            if (j < KW) {
                                const float dx = qp.x - c[u].x, dy = qp.y - c[u].y, dz = qp.z - c[u].z;
                                const float d = (dx * dx + dy * dy) + dz * dz;
                                const int32_t cid = w_to_id(c[u].w);
                                const bool in = (j < cnt) && (j < K + 2) && !lex_lt(cutd, cuti, d, cid) && (cid != qid);
                                const bool nearer = in && lex_lt(d, cid, nd2, nid);
                                nd2 = nearer ? d : nd2;
                                nid = nearer ? cid : nid;
                                const float f = ks_force(fc, d * inv_s2);
                                const float coef = (in && d > 0.f) ? f * __builtin_amdgcn_rsqf(d) : 0.f;
                                Fx += coef * dx;
                                Fy += coef * dy;
                                Fz += coef * dz;
                                coincident = coincident || (in && !(d > 0.f));
                            }
                        }
                    }
                    if (coincident) { // r == 0 needs the substitute direction: exact path (rare)
                        const int p = atomicAdd(a.fb_count, 1);
                        a.fb_list[p] = gslot;
                        continue;
                    }
                    float4 o;
                    const float f = step_point<float>(a, s, qp.x, qp.y, qp.z, Fx, Fy, Fz, o.x, o.y, o.z);
                    o.w = qp.w;
                    const bool has = nid != 0x7FFFFFFF;
                    const float nd = has ? wsqrt(nd2) : Lim<float>::inf();
                    // sharded sessions: the k-set must lie inside the range the ghost layer covers
                    if (reaches_past_cover<float>(a, qp.x, qp.y, qp.z, cutd)) atomicAdd(a.uncovered, 1);
                    a.out[gslot] = o;
                    a.forces[gslot] = f;
                    a.nn_dist[gslot] = nd;
                    a.nn_id[gslot] = has ? nid : -1;
                    acc_point<float>(acc, f, nd, s, qid, has ? nid : -1);
                }
            }
            if (fallback && active) {
                const int p = atomicAdd(a.fb_count, 1);
                a.fb_list[p] = gslot;
            }
            KS_STAMP(7) // rows / force, outputs
        }
    }
    if (WTP_DIAG && a.diag && lane == 0) {
        for (int i = 0; i < 11; ++i) atomicAdd(&a.diag[i], dt[i]);
        atomicAdd(&a.diag[15], 1ull);
    }
    if (MODE == 1) {
        __syncthreads();
        acc_block_reduce(acc, sm->acc);
        if (tid == 0) acc_store(&a.partials[blockIdx.x], acc);
    }
}

int ksel_max_bx() { return kKsMaxBX; }
int ksel_kmax() { return kKsKMax; }

template <int MODE, int KT> static int ksel_launch(wtp_ctx* ctx, SearchArgs<float>& a) {
    const int hcap = a.brick_hcap > 0 ? a.brick_hcap : 2432;
    const int bx = a.ksel_bx < 1 ? 1 : (a.ksel_bx > kKsMaxBX ? kKsMaxBX : a.ksel_bx);
    int occ = launch_occupancy_of(ctx, (const void*)ksel_kernel<MODE, KT>, kKsThreads, ksel_smem_bytes(hcap));
    if (occ > 2) occ = 2;
    if (occ < 1) return fail(ctx, WTP_ERR_HIP, "ksel_kernel: the LDS point area does not fit a compute unit");
    int gsz = ctx->sm_count * occ;
    gsz -= gsz % 8;
    if (gsz < 8) gsz = 8;
    if (MODE == 1) a.used_brick = gsz;
    hipLaunchKernelGGL((ksel_kernel<MODE, KT>), dim3(gsz), dim3(kKsThreads), ksel_smem_bytes(hcap), ctx->stream, a, hcap, bx);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// KNNTopology rows (a.ksel_bx > 0: the caller built the grid for this layout); the caller cleared a.fb_count
int launch_ksel_topology(wtp_ctx* ctx, SearchArgs<float>& a) {
    if (a.k == 24) return ksel_launch<0, 24>(ctx, a); // fp64 KNNTopology at k = 21: k + self + 2 candidates for the exact re-ranking
    return a.k == 21 ? ksel_launch<0, 21>(ctx, a) : ksel_launch<0, 0>(ctx, a);
}

// repel sweep with the explicit k-selection, fresh snapshot
int launch_ksel_sweep(wtp_ctx* ctx, SearchArgs<float>& a) {
    return a.k == 21 ? ksel_launch<1, 21>(ctx, a) : ksel_launch<1, 0>(ctx, a);
}

} // namespace wtp
