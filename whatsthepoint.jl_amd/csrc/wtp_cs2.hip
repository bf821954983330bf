// wtp_cs2.hip — the default repel sweep (round 2): ClippedSpacingForce on fp32 3-D clouds.
//
// What it computes is the body of `_relax!`'s sweep closure (src/repel.jl:256-292) with the force
// law of src/repel_forces.jl:96-100, exactly like brick_kernel<1,0,1> (wtp_brick.hip), whose
// compact-support argument it keeps: the clipped law is 0 for r >= u0*s, so when the support ball
// holds n_lim <= k points they ARE among the k nearest and the sum over them equals the reference's
// sum over its k-list, term for term.  What is new is everything around that argument:
//
//   * the nearest neighbour no longer rides on a margin of the ring.  If the support holds another
//     point, the nearest of them is the global nearest (everything outside is farther).  The ~1.5 % of
//     the queries whose support is empty are finished by their own wave after the brick's queries: 64
//     lanes scan the query's 27 cells cooperatively (staged in LDS already), a lexicographic
//     (d2, id) minimum over the wave, certified against the provable radius; the few that even the 27
//     cells cannot certify go to the exact path.  So the cells only have to cover the support:
//     c = 1.01 u0 s instead of 1.52 s — rho ~ 1 point per cell instead of 3.5, 28 candidates in the 27
//     cells instead of 95;
//   * bricks are BX x 2 x 2 cells (BX ~ 56) and the halo is staged in LDS in (hx, hz, hy) order —
//     x SLOWEST.  A query's 3x3x3 neighbourhood is then ONE contiguous run of 43 halo cells (its 27
//     plus 16 cells two rows away in y or z, whose points fail the distance test by construction), so
//     the scan is a single loop over ~44 consecutive LDS slots: no rows, no run table, no queue;
//   * global loads stay coalesced: a halo x-row is one contiguous run of the sorted array, the
//     permutation happens in the LDS store (destination = start of the point's cell + its rank).
//
// Everything the fast path cannot certify — support wider than the provable radius (variable
// spacing), more than k points inside the support, a coincident neighbour (r = 0 needs the
// substitute direction), ring or LDS overflow — is appended to a.fb_list for the exact wave path,
// so results are always the exact sums.
//
// Roofline: 16 B/point in, 28 B/point out; ~1.3 k VALU lane-ops per query.
#include "wtp_device.hpp"

namespace wtp {

#ifndef WTP_DIAG
#define WTP_DIAG 0 // diagnostic build: s_memtime stamps per phase + loop-trip counts (never quote its run time)
#endif
#define CS_STAMP(i)                                                 \
    if (WTP_DIAG) {                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        dt[i] += t_ - t_last;                                       \
        t_last = t_;                                                \
    }

constexpr int kCsThreads = 256;
constexpr int kCsSlab = 16;                         // halo cells per x index: (2+2) x (2+2)
constexpr int kCsMaxBX = 61;                        // own cells along x; halo + the closing column <= 64 lanes
constexpr int kCsMaxCells = (kCsMaxBX + 2) * kCsSlab;
constexpr int kCsMaskSteps = 12;                    // scan steps the three hit-mask registers hold (96 slots of a run)
constexpr int kCsSU = 8;                            // candidates per scan step
constexpr int kCsPadBytes = 2 * kCsSU * 16;          // far sentinels behind the staged points: a run's last step and the half step read ahead of it
constexpr int kCsMaxQ = 512;                        // queries per brick the lane table covers
constexpr int kCsMiss = 16;                         // empty-support queries per wave per brick
constexpr int kCsRun = 2 * kCsSlab + 2 * 4 + 3;     // halo cells in a query's run: 43

struct CsMiss {
    uint32_t pa, ea, qoff; // candidate run and the query's own slot (byte offsets into the point area)
    int32_t gslot, qid;
    float f, s, g2;
};

struct Cs2Smem {
    uint32_t hglobal[kCsMaxCells];   // index in the sorted array of the first point of each halo cell
    uint16_t ls[kCsMaxCells + 4];    // LDS slot of the first point of each halo cell, (hx, hz, hy) order
    uint16_t qpref[kCsMaxBX + 2];    // first query of each own slab
    uint8_t qslab[kCsMaxQ];          // slab (hx) of each query
    // per wave (= four halo x-rows) and per column hx: sum of cell_start, sum of the cells' counts, and
    // (waves 1, 2) the sum of cell_start over the two own rows: every prefix the brick needs is a difference of these
    uint32_t wsum[4][64], wcnt[4][64], wown[2][64];
    int scan_tmp[kCsThreads / 64 + 1];
    int miss_n[kCsThreads / 64];
    CsMiss miss[kCsThreads / 64][kCsMiss];
    Acc acc[kCsThreads / 64];
};

typedef float cs_f4 __attribute__((ext_vector_type(4)));

// Per-lane statistics in 9 registers instead of Acc's 14 (same values: a float compared as float or as the
// double it converts to orders the same; the sums stay in double like acc_point's)
struct CsAcc {
    float max_force, argmin_r;
    double sum_u, sum_u2;
    int32_t argmin_i, argmin_j, n_move;
};
// Adaptive step and displacement cap (src/repel.jl:282-291) with the hardware's 1-ulp sqrt / reciprocal instead of
// the IEEE sequences step_point uses (~45 VALU less per query; the fast path's coordinates agree with the reference
// to rounding anyway: its force sum runs in scan order).  Returns forces[id] = |F| s.
__device__ inline float cs_step_point(const SearchArgs<float>& a, float s, float xi, float yi, float zi, float Fx, float Fy,
                                      float Fz, float& xo, float& yo, float& zo) {
    const float Fn = __builtin_amdgcn_sqrtf((Fx * Fx + Fy * Fy) + Fz * Fz);
    float al = __builtin_amdgcn_rcpf(Fn + 1.0e-30f);
    al = al < a.alpha_lo ? a.alpha_lo : al;
    al = al > a.alpha_max ? a.alpha_max : al;
    const float sa = s * al;
    float dx = sa * Fx, dy = sa * Fy, dz = sa * Fz;
    const float dn = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);
    const float sc = dn > s ? s * __builtin_amdgcn_rcpf(dn) : 1.f;
    xo = xi + dx * sc;
    yo = yi + dy * sc;
    zo = zi + dz * sc;
    return Fn * s;
}

__device__ inline void cs_acc_point(CsAcc& c, float force, float nd, float s, int32_t id, int32_t nn) {
    c.max_force = force > c.max_force ? force : c.max_force;
    // d_nn / s in T as the reference's CV monitor divides (src/repel.jl:374-384): reciprocal + one exact-residual
    // correction = the correctly rounded quotient (but for rare near-halfway cases, 1 ulp, invisible in the sums);
    // the IEEE sequence costs registers this kernel does not have
    const float rs = __builtin_amdgcn_rcpf(s), q0 = nd * rs;
    const double u = (double)__builtin_fmaf(__builtin_fmaf(-s, q0, nd), rs, q0);
    c.sum_u += u;
    c.sum_u2 += u * u;
    c.n_move += 1;
    if (nd < c.argmin_r || (nd == c.argmin_r && id < c.argmin_i)) {
        c.argmin_r = nd;
        c.argmin_i = id;
        c.argmin_j = nn;
    }
}

// ClippedSpacingForce (src/repel_forces.jl:96-100) on u2 = d2 / s^2, fast reciprocal (1 ulp), the same
// expression as brick_kernel's force_fast with (A, B, lo) = (u0^2, 1, 0)
struct ForceCoefCs {
    float A, beta;
};
__device__ inline ForceCoefCs force_coef_cs(float beta, float u0) { return ForceCoefCs{u0 * u0, beta}; }
__device__ inline float force_fast_cs(const ForceCoefCs& c, float u2) {
    const float d = u2 + c.beta;
    const float inv = __builtin_amdgcn_rcpf(d * d);
    const float f = (c.A - u2) * inv;
    return f > 0.f ? f : 0.f;
}

// four consecutive staged points: explicit ds_read_b128 (the compiler would shrink the loads to b96 when .w
// is unused, which costs twice the LDS cycles per instruction) and a single wait
template <int OFF> __device__ inline void cs_read_group(cs_f4 (&c)[4], uint32_t addr) {
    asm volatile(
        "ds_read_b128 %0, %4 offset:%5\n\t"
        "ds_read_b128 %1, %4 offset:%6\n\t"
        "ds_read_b128 %2, %4 offset:%7\n\t"
        "ds_read_b128 %3, %4 offset:%8\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
        : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48)
        : "memory");
}

__device__ inline float4 cs_pt(const unsigned char* base, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(base + byte_off);
}

static size_t cs2_smem_bytes(int hcap) {
    return (size_t)hcap * 16 + kCsPadBytes + sizeof(Cs2Smem);
}

// CH: runs longer than the hit masks are taken in chunks (variable spacing, cells that hold several points).  The plain
// variant gives such queries up, as round 3's first version did: it is the one the uniform headline runs, and the chunk loop
// costs it two registers it does not have (128 VGPRs + 28 B of scratch: 0.572 -> 0.593 ms).
template <bool CH>
__global__ __launch_bounds__(kCsThreads, 4) void cs2_kernel(SearchArgs<float> a, int hcap, int BX) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* pts = reinterpret_cast<float4*>(smem_raw);
    Cs2Smem* sm = reinterpret_cast<Cs2Smem*>(smem_raw + (size_t)hcap * 16 + kCsPadBytes);
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction; saying so keeps the per-row geometry and the row bounds in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const Grid<float> g = *a.grid;
    const int K = a.k;
    const int HX = BX + 2, ncell = HX * kCsSlab;
    const int nbx = (g.n[0] + BX - 1) / BX, nby = (g.n[1] + 1) / 2, nbz = (g.n[2] + 1) / 2;
    const int nbricks = nbx * nby * nbz;
    CsAcc acc;
    acc.max_force = 0.f;
    acc.argmin_r = Lim<float>::inf();
    acc.sum_u = acc.sum_u2 = 0.0;
    acc.argmin_i = acc.argmin_j = -1;
    acc.n_move = 0;
    unsigned long long dt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = WTP_DIAG ? __builtin_amdgcn_s_memtime() : 0ull;

    // constant spacing: everything that depends on s only is wave-uniform
    const float s_c = a.spacing_const, inv_s2_c = 1.f / (s_c * s_c), lim_c = (a.u0 * a.u0) * (s_c * s_c);
    const float gmin = g.c - g.margin;
    const bool const_ok = !a.spacing_pp && gmin > 0.f && lim_c <= gmin * gmin; // smallest provable radius covers the support

    // XCD-aware brick order (blocks sharing blockIdx % 8 share an L2): one contiguous slab of bricks each
    // (variable spacing: the live bricks of a graded cloud sit in a shell, so contiguous slabs per XCD would leave the XCDs
    // that own the two z faces with twice the work of the others: the bricks are dealt out round-robin instead)
    const int groups = a.brick_dead ? 1 : 8;
    const int per = (nbricks + groups - 1) / groups;
    const int xcd = blockIdx.x % groups;
    const int lane_blk = blockIdx.x / groups;
    const int blk_per_group = gridDim.x / groups;
    const int b_end = (xcd + 1) * per < nbricks ? (xcd + 1) * per : nbricks;

    // The loads of a brick are issued one brick ahead and sit in registers while the previous brick's queries
    // run (waited for in place they are two dependent global round trips, ~40 % of a wave's time when measured):
    //   top of brick i    the cell-table loads of brick i+1 (4 registers), in flight during the stage of brick i
    //   end of stage i    row bounds of brick i+1 from them, then its point loads (16 registers)
    // Brick coordinates advance with the brick index by a fixed stride: kept as three counters with carries
    // (the divisions by the runtime brick counts cost ~60 VALU each, several per brick and wave, when written as % and /)
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
    const BrickPos stride = brick_pos(blk_per_group); // blk_per_group < nbricks: a valid position
    auto advance = [&](BrickPos& p) {
        p.bx += stride.bx;
        int c = p.bx >= nbx ? 1 : 0;
        p.bx -= c ? nbx : 0;
        p.by += stride.by + c;
        c = p.by >= nby ? 1 : 0;
        p.by -= c ? nby : 0;
        p.bz += stride.bz + c;
    };
    auto origin = [&](const BrickPos& p, int& bx, int& ox, int& oy, int& oz) {
        bx = p.bx;
        ox = p.bx * BX - 1;
        oy = p.by * 2 - 1;
        oz = p.bz * 2 - 1;
    };
    // cell_start at the column clamped to the grid: column 0 is the row's first index in the sorted array, column HX
    // its one-past-last, and neighbours differ by a cell's count
    auto load_cells = [&](const BrickPos& bp, int (&vv)[4]) {
        int bx, ox, oy, oz;
        origin(bp, bx, ox, oy, oz);
        const int gx_lo = ox < 0 ? 0 : ox, gx_hi = (ox + HX - 1) < g.n[0] - 1 ? (ox + HX - 1) : g.n[0] - 1;
        int gxc = ox + lane;
        gxc = gxc < gx_lo ? gx_lo : (gxc > gx_hi + 1 ? gx_hi + 1 : gxc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 4 + j;
            const int gy = oy + (r & 3), gz = oz + (r >> 2);
            const bool row_ok = gy >= 0 && gy < g.n[1] && gz >= 0 && gz < g.n[2];
            // (a row outside the grid reads cell_start[0] = 0 in every column: no points)
            vv[j] = a.cell_start[row_ok ? (gz * g.n[1] + gy) * g.n[0] + gxc : 0];
        }
    };
    auto load_points = [&](const int (&vv)[4], float4 (&pp)[4]) {
        const int last = a.n - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gs = __builtin_amdgcn_readfirstlane(vv[j]);
            const int len = __builtin_amdgcn_readlane(vv[j], HX < 63 ? HX : 63) - gs;
            const int i = gs + (lane < len ? lane : 0);
            pp[j] = a.snap[i < last ? i : last];
        }
    };

    // variable spacing: bricks whose own points ALL ask for a support wider than any cell block certifies were listed for
    // the ball kernel by cs2_dead_kernel; they are passed over here (no tables, no staging, no barriers)
    auto skip_dead = [&](int& b, BrickPos& p) {
        while (a.brick_dead && b < b_end && b < a.brick_dead_cap && a.brick_dead[b]) {
            b += blk_per_group;
            advance(p);
        }
    };
    int brick = xcd * per + lane_blk;
    BrickPos pos = brick_pos(brick < nbricks ? brick : 0);
    skip_dead(brick, pos);
    BrickPos pos_next = pos;
    int next = brick;
    int v[4] = {0, 0, 0, 0};
    float4 pv[4];
    pv[0] = pv[1] = pv[2] = pv[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (brick < b_end) {
        load_cells(pos, v);
        load_points(v, pv);
    }
    for (; brick < b_end; brick = next, pos = pos_next) {
        int bx, ox, oy, oz; // halo origin (cell coordinates)
        origin(pos, bx, ox, oy, oz);
        next = brick + blk_per_group;
        advance(pos_next);
        skip_dead(next, pos_next);

        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // previous brick's LDS no longer in use
        int vn[4] = {0, 0, 0, 0};
        if (next < b_end) load_cells(pos_next, vn);
        // ---- 1. cell table.  Wave w owns the halo x-rows r = 4w .. 4w+3 (r = hz*4 + hy), lane = column hx.
        //         The prefix the LDS order (hx, hz, hy) needs — points left of column hx in all rows, plus the
        //         column's cells in rows before r — is a sum of cell_start values: no scan across lanes. -----
        int cn[4];
        int row_gs[4], row_len[4];
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nxt = __shfl_down(v[j], 1, 64);
                cn[j] = lane < HX ? nxt - v[j] : 0;
                row_gs[j] = __builtin_amdgcn_readfirstlane(v[j]);
                row_len[j] = __builtin_amdgcn_readlane(v[j], HX < 63 ? HX : 63) - row_gs[j];
            }
            if (lane <= HX) {
                sm->wsum[wave][lane] = (uint32_t)(v[0] + v[1] + v[2] + v[3]);
                sm->wcnt[wave][lane] = (uint32_t)(cn[0] + cn[1] + cn[2] + cn[3]);
                if (wave == 1 || wave == 2) sm->wown[wave - 1][lane] = (uint32_t)(v[1] + v[2]); // rows 5, 6 / 9, 10
            }
            if (lane < HX) {
                uint4 hv;
                hv.x = (uint32_t)v[0];
                hv.y = (uint32_t)v[1];
                hv.z = (uint32_t)v[2];
                hv.w = (uint32_t)v[3];
                *reinterpret_cast<uint4*>(&sm->hglobal[lane * kCsSlab + wave * 4]) = hv;
            }
        }
        __syncthreads();
        CS_STAMP(0) // cell table
        int halo_total;
        {
            uint32_t left = 0, g0 = 0, before = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                left += sm->wsum[w][lane <= HX ? lane : 0];
                g0 += sm->wsum[w][0];
                tot += sm->wsum[w][HX];
                before += w < wave ? sm->wcnt[w][lane <= HX ? lane : 0] : 0u;
            }
            halo_total = (int)(tot - g0);
            if (lane < HX) {
                uint32_t e = left - g0 + before, o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = e > 0xFFFFu ? 0xFFFFu : e;
                    e += (uint32_t)cn[j];
                }
                uint2 w2;
                w2.x = o[0] | (o[1] << 16);
                w2.y = o[2] | (o[3] << 16);
                *reinterpret_cast<uint2*>(&sm->ls[lane * kCsSlab + wave * 4]) = w2;
            }
            if (tid == 0) sm->ls[ncell] = (uint16_t)(halo_total > 0xFFFF ? 0xFFFF : halo_total);
            if (tid < kCsThreads / 64) sm->miss_n[tid] = 0;
            // own queries in front of slab hx (slabs 1 .. BX hold own cells; entry BX closes the table)
            if (wave == 3 && lane >= 1 && lane <= BX + 1) {
                const uint32_t q = (sm->wown[0][lane] + sm->wown[1][lane]) - (sm->wown[0][1] + sm->wown[1][1]);
                sm->qpref[lane - 1] = (uint16_t)(q > 0xFFFFu ? 0xFFFFu : q);
            }
        }
        __syncthreads();
        CS_STAMP(6) // prefix tables
        const bool overflow = halo_total > hcap;
        if (tid >= 1 && tid <= BX) {
            const int q0 = sm->qpref[tid - 1], q1 = sm->qpref[tid];
            for (int q = q0; q < q1 && q < kCsMaxQ; ++q) sm->qslab[q] = (uint8_t)tid;
        }
        // ---- 2. stage the halo: global rows are contiguous, the LDS order is (hx, hz, hy) ---------------
        if (!overflow) {
            auto place = [&](const float4& p, int r, int gidx) {
                int hx = cell_coord(g, p.x, 0) - ox;
                hx = hx < 0 ? 0 : (hx > HX - 1 ? HX - 1 : hx);
                const int P = hx * kCsSlab + r;
                int dest = (int)sm->ls[P] + (gidx - (int)sm->hglobal[P]);
                dest = dest < 0 ? 0 : (dest > hcap - 1 ? hcap - 1 : dest); // never outside the point area
                pts[dest] = p;
            };
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane < row_len[j]) place(pv[j], wave * 4 + j, row_gs[j] + lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) // rows longer than a wave: the rest, row by row
                for (int i = lane + 64; i < row_len[j]; i += 64) place(a.snap[row_gs[j] + i], wave * 4 + j, row_gs[j] + i);
            // one scan step past the staged points: far sentinels instead of another brick's leftovers
            // (a run's last step reads up to 7 slots past its end; every CELL past a run is two cells
            // away from the query, so only the slots past the last cell need this)
            if (tid < 2 * kCsSU) pts[halo_total + tid] = make_float4(1e30f, 1e30f, 1e30f, 0.f);
        }
        // the next brick: its cell table has arrived by now, its points fly while this brick's queries run
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vn[j];
        if (next < b_end) load_points(v, pv);
        __syncthreads();

        CS_STAMP(7) // own table, staging
        if (WTP_DIAG) dt[8] += 1; // bricks (per wave)
        // ---- 4. queries ------------------------------------------------------------------------------
#ifdef CS_ABL_NOQUERY // timing-only ablation (results wrong): the stage alone
        const int Q = 0;
#else
        const int Q = sm->qpref[BX];
#endif
        if (overflow || Q > kCsMaxQ) {
            // dense brick (halo larger than the LDS point area, or more queries than the lane table holds):
            // every own point goes to the exact path.  Own points = four x-rows of the sorted array.
            for (int r4 = 0; r4 < 4; ++r4) {
                const int gy = oy + 1 + (r4 & 1), gz = oz + 1 + (r4 >> 1);
                if (gy >= g.n[1] || gz >= g.n[2]) continue;
                const int gx0 = bx * BX, gx1 = (gx0 + BX - 1) < g.n[0] - 1 ? (gx0 + BX - 1) : g.n[0] - 1;
                const int base = (gz * g.n[1] + gy) * g.n[0];
                const int s = a.cell_start[base + gx0], e = a.cell_start[base + gx1 + 1];
                for (int i = s + tid; i < e; i += kCsThreads) {
                    const int pos = atomicAdd(a.fb_count, 1);
                    a.fb_list[pos] = i;
                }
            }
            continue;
        }
        // waves that run queries issue ahead of waves that stage (those mostly wait for loads and barriers anyway):
        // 653 -> 630 us per launch at 10 M points
        __builtin_amdgcn_s_setprio(1);
        for (int qb = 0; qb < Q; qb += kCsThreads) {
            const int q = qb + tid;
            const bool active = q < Q;
            // lane table -> (slab, own cell, slot)
            int hx = 1, cellp = 5, slot = 0, gslot = 0;
            if (active) {
                hx = sm->qslab[q];
                const int off = q - (int)sm->qpref[hx - 1];
                const int b = hx * kCsSlab;
                const int s5 = sm->ls[b + 5], s6 = sm->ls[b + 6], s7 = sm->ls[b + 7];
                const int s9 = sm->ls[b + 9], s10 = sm->ls[b + 10];
                const int n56 = s7 - s5;
                if (off < n56) {
                    slot = s5 + off;
                    cellp = slot < s6 ? 5 : 6;
                } else {
                    slot = s9 + (off - n56);
                    cellp = slot < s10 ? 9 : 10;
                }
                gslot = (int)sm->hglobal[b + cellp] + (slot - (int)sm->ls[b + cellp]);
            }
            if (!active) continue;
            const uint32_t qoff = (uint32_t)slot * 16u;
            const float4 qp = pts[slot];
            const int32_t qid = w_to_id(qp.w);
            if (qid < a.n_fixed) { // the wall: never moves (src/repel.jl:80,256)
                a.out[gslot] = qp;
                a.forces[gslot] = 0.f;
                a.nn_dist[gslot] = Lim<float>::inf();
                a.nn_id[gslot] = -1;
                continue;
            }
            const int hy = cellp & 3, hz = cellp >> 2;
            const int cx = ox + hx, cy = oy + hy, cz = oz + hz;
            float s = s_c, inv_s2 = inv_s2_c, lim = lim_c;
            // the support must lie inside the radius up to which the 27 cells are provably complete.  With a
            // constant spacing and cells sized for it that holds for every query (the radius is at least
            // c - margin): checked once per kernel, and the radius is worked out for the few misses only.
            bool cs_fail = false;
            if (!const_ok) { // wave-uniform
                if (a.spacing_pp) {
                    s = a.spacing_pp[qid];
                    inv_s2 = 1.f / (s * s);
                    lim = (a.u0 * a.u0) * (s * s);
                }
                cs_fail = !(lim <= safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, 1));
            }
            // candidate run: halo cells P(hx-1, hz-1, hy-1) .. P(hx+1, hz+1, hy+1), contiguous in LDS
            const int Pf = (hx - 1) * kCsSlab + (hz - 1) * 4 + (hy - 1);
            const uint32_t pa_first = (uint32_t)sm->ls[Pf] * 16u, ea_all = (uint32_t)sm->ls[Pf + kCsRun] * 16u;
            // Hits are kept as a bit mask of the run's slots in three registers (96 slots = 12 steps; longer runs, a dense
            // cluster, give up): the scan loop then holds no LDS store at all.  Round 2 appended a byte per hit to a ring in
            // LDS — one ds_write_b8 (4 LDS cycles) next to every ds_read_b128 (4 + bank conflicts), which made the scan
            // LDS-bound: 4 waves x ~11 LDS cycles per candidate against 4 x ~10 VALU cycles on four SIMDs in parallel.
            // Runs longer than the masks (cells that hold several points each: coarse grids of graded clouds, clusters) are taken
            // in chunks of 96 slots — scan, then force pass, then the next chunk; the sums simply carry on.  Four chunks at most.
            constexpr uint32_t kChunk = (uint32_t)(kCsMaskSteps * kCsSU) * 16u;
            bool giveup = (ea_all - pa_first) > (CH ? 4u : 1u) * kChunk;
            const uint32_t ea_all_s = giveup ? pa_first : ea_all; // (a lane that gave up must not prolong the wave's loops)
            const float tau_s = (cs_fail || giveup) ? -1.f : lim * (1.f + 0x1p-21f); // FMA filter, 4 ulp wide; the force pass is exact
            const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
            // ---- force pass state: canonical d2 of every hit, exact cut, force sum (src/repel.jl:270-280) ----
            const ForceCoefCs fc = force_coef_cs(a.beta, a.u0);
            int n_oth = 0; // points other than the query inside the support (the query itself is always there)
            bool coincident = false;
            float Fx = 0.f, Fy = 0.f, Fz = 0.f;
            int32_t nid = 0x7FFFFFFF;
            float nd2 = Lim<float>::inf();
            auto visit = [&](const float4& c) {
                const float dx = qp.x - c.x, dy = qp.y - c.y, dz = qp.z - c.z;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const int32_t cid = w_to_id(c.w);
                const bool act = (d <= lim) & (cid != qid);
                n_oth += act ? 1 : 0;
                const bool nearer = act & ((d < nd2) | ((d == nd2) & (cid < nid)));
                nd2 = nearer ? d : nd2;
                nid = nearer ? cid : nid;
                // f(u) / r with one reciprocal square root: (A - u2) / ((u2 + beta)^2 sqrt(d)) = (A - u2) rsq(d (u2 + beta)^4)
                const float u2 = d * inv_s2, t = u2 + fc.beta, t2 = t * t;
                const float r = __builtin_amdgcn_rsqf((d * t2) * t2);
                const float f = fmaxf(fc.A - u2, 0.f) * r;
                const bool pos = d > 0.f;
                const float coef = (act & pos) ? f : 0.f;
                Fx = __builtin_fmaf(coef, dx, Fx); // (this path's sum order differs from the reference's anyway)
                Fy = __builtin_fmaf(coef, dy, Fy);
                Fz = __builtin_fmaf(coef, dz, Fz);
                coincident = coincident | (act & !pos);
            };
            for (uint32_t cbase = 0;; cbase += kChunk) { // wave-uniform trip count: one chunk for runs of up to 96 slots
            const bool live = pa_first + cbase < ea_all_s;
            const uint32_t pa0 = live ? pa_first + cbase : pa_first; // (a lane without slots in this chunk rereads its first ones, masked)
            const uint32_t ea_s = live ? (ea_all_s < pa0 + kChunk ? ea_all_s : pa0 + kChunk) : pa0;
            uint32_t m0 = 0, m1 = 0, m2 = 0; // slot i of the chunk = bit (8 S - 1 - i) of m2:m1:m0 after S steps
            CS_STAMP(1) // query setup
            if (WTP_DIAG) {
                dt[9] += 1;                                           // query rounds (per wave)
                dt[10] += (unsigned long long)__popcll(__ballot(true)); // queries
            }
            int S = 0;
            {
                // Software pipeline over half steps of four candidates: a register group is refilled (ds_read_b128 into the
                // same registers) as soon as its three subtractions have consumed it, and a use waits only for ITS read
                // (LDS returns in order: `lgkmcnt(3)` = everything but the three youngest reads has landed).  The wave no
                // longer drains the LDS queue before every group of four.
                cs_f4 c[4];
                uint32_t pa = pa0;
                uint32_t addr = lds_base + pa0;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                             : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                             : "v"(addr)
                             : "memory");
#define CS_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg)::"memory")
#define CS_REFILL(reg, a, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(reg) : "v"(a), "n"(off) : "memory")
                // compare, then mask = 2 mask + hit: one add-with-carry per candidate
#define CS_TEST(dist)                                                                     \
    asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" \
                 : "+v"(hb)                                                                \
                 : "v"(dist), "v"(thr)                                                     \
                 : "vcc")
                for (;;) {
                    if (WTP_DIAG) {
                        dt[11] += 1; // scan steps (per wave)
                        dt[12] += (unsigned long long)__popcll(__ballot(pa < ea_s)); // busy lanes
                    }
                    // a lane whose run has ended keeps reading its first slots with an impossible threshold.  Inside a
                    // run no end mask is needed: the slots a last step reads past the run's end belong to cells two
                    // away from the query (or are the sentinels behind the last cell) and fail the distance test.
                    const float thr = pa < ea_s ? tau_s : -1.f;
                    const uint32_t pn = pa + 16u * kCsSU;
                    const uint32_t addr_n = lds_base + (pn < ea_s ? pn : pa0); // what the NEXT step reads
                    uint32_t hb = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { // slots 0..3; the group is refilled with slots 4..7 of this step
                        CS_WAIT3(c[u]);
                        const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                        switch (u) {
                            case 0: CS_REFILL(c[0], addr, 64); break;
                            case 1: CS_REFILL(c[1], addr, 80); break;
                            case 2: CS_REFILL(c[2], addr, 96); break;
                            default: CS_REFILL(c[3], addr, 112); break;
                        }
                        const float d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                        CS_TEST(d);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { // slots 4..7; refilled with slots 0..3 of the next step
                        CS_WAIT3(c[u]);
                        const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                        switch (u) {
                            case 0: CS_REFILL(c[0], addr_n, 0); break;
                            case 1: CS_REFILL(c[1], addr_n, 16); break;
                            case 2: CS_REFILL(c[2], addr_n, 32); break;
                            default: CS_REFILL(c[3], addr_n, 48); break;
                        }
                        const float d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                        CS_TEST(d);
                    }
                    m2 = __builtin_amdgcn_alignbit(m2, m1, 24);
                    m1 = __builtin_amdgcn_alignbit(m1, m0, 24);
                    m0 = (m0 << 8) | hb;
                    ++S;
                    pa = pn;
                    addr = addr_n;
                    if (!__any(pa < ea_s)) break;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])::"memory"); // the reads issued for a step that does not come
#undef CS_WAIT3
#undef CS_REFILL
#undef CS_TEST
            }
            S = __builtin_amdgcn_readfirstlane(S);
            // the query's own bit: it is always set (d = 0), and skipping it here (src/repel.jl:271 skips self by index)
            // saves a trip of the force pass for the lanes with the most hits
            if (!(cs_fail || giveup) && qoff >= pa0 && ((qoff - pa0) >> 4) < (uint32_t)(8 * S)) { // (the query lies in this chunk)
                const uint32_t bit = (uint32_t)(8 * S - 1) - ((qoff - pa0) >> 4);
                const uint32_t one = 1u << (bit & 31u);
                m0 &= bit < 32u ? ~one : ~0u;
                m1 &= (bit >= 32u && bit < 64u) ? ~one : ~0u;
                m2 &= bit >= 64u ? ~one : ~0u;
            }
            CS_STAMP(2) // scan

            // ---- force pass over this chunk's hits: taken off the mask words from the top (v_ffbh), two per trip; a lane whose
            // word is empty reads its own point instead, which the self-exclusion by index (src/repel.jl:271) drops ----
            auto word_pass = [&](uint32_t mw, int top_slot) { // top_slot: the run slot of the word's bit 31
                const uint32_t base_w = pa0 + (uint32_t)(top_slot * 16); // byte offset of that slot in the point area
                while (__any(mw != 0u)) {
                    if (WTP_DIAG) dt[13] += 1; // force-pass trips (per wave)
                    // v_ffbh_u32 gives -1 for an empty word: the shift below then clears bit 0 (clear already) and the
                    // address is replaced by the query's own slot
                    uint32_t l1, l2;
                    const bool v1 = mw != 0u;
                    asm("v_ffbh_u32 %0, %1" : "=v"(l1) : "v"(mw));
                    const uint32_t o1 = v1 ? base_w + (l1 << 4) : qoff;
                    mw &= ~(0x80000000u >> (l1 & 31u));
                    const bool v2 = mw != 0u;
                    asm("v_ffbh_u32 %0, %1" : "=v"(l2) : "v"(mw));
                    const uint32_t o2 = v2 ? base_w + (l2 << 4) : qoff;
                    mw &= ~(0x80000000u >> (l2 & 31u));
                    const float4 c1 = cs_pt(smem_raw, o1), c2 = cs_pt(smem_raw, o2);
                    visit(c1);
                    visit(c2);
                }
            };
            if (8 * S > 64) word_pass(m2, 8 * S - 96); // runs beyond 64 slots (one wave-round in ~15): their first slots
            {
                // the two low words in ONE loop: a lane takes its hits from m1 until that is empty, then from m0 — the trip count
                // is the largest hit count of the wave, not the sum of the two words' largest counts (6.8 -> ~5 trips per round)
                const uint32_t base1 = pa0 + (uint32_t)((8 * S - 64) * 16), base0 = pa0 + (uint32_t)((8 * S - 32) * 16);
                auto take = [&](uint32_t& off) {
                    const bool hi = m1 != 0u;
                    const uint32_t cur = hi ? m1 : m0;
                    uint32_t l;
                    asm("v_ffbh_u32 %0, %1" : "=v"(l) : "v"(cur));
                    off = cur != 0u ? (hi ? base1 : base0) + (l << 4) : qoff;
                    const uint32_t rest = cur & ~(0x80000000u >> (l & 31u));
                    m1 = hi ? rest : m1;
                    m0 = hi ? m0 : rest;
                };
                while (__any((m1 | m0) != 0u)) {
                    if (WTP_DIAG) dt[13] += 1; // force-pass trips (per wave)
                    uint32_t o1, o2;
                    take(o1);
                    take(o2);
                    const float4 c1 = cs_pt(smem_raw, o1), c2 = cs_pt(smem_raw, o2);
                    visit(c1);
                    visit(c2);
                }
            }
            if (!CH || !__any(pa_first + cbase + kChunk < ea_all_s)) break;
            } // chunks
            const int n_lim = n_oth + 1;
            CS_STAMP(3) // ring pass
            // not provable here: support wider than the certified radius, ring overflow, more than k points
            // inside the support (then some of them are NOT among the k nearest), r = 0 (substitute direction)
            if (cs_fail || giveup || n_lim > K || coincident) {
                const int pos = atomicAdd(a.fb_count, 1);
                a.fb_list[pos] = gslot;
                continue;
            }
            float4 o;
            const float f = cs_step_point(a, s, qp.x, qp.y, qp.z, Fx, Fy, Fz, o.x, o.y, o.z);
            o.w = qp.w;
            a.out[gslot] = o;
            a.forces[gslot] = f;
            const bool has = nid != 0x7FFFFFFF;
            if (has) {
                // the nearest point of the support is the nearest overall: everything outside is farther
                const float nd = wsqrt(nd2);
                if (reaches_past_cover<float>(a, qp.x, qp.y, qp.z, lim)) atomicAdd(a.uncovered, 1);
                a.nn_dist[gslot] = nd;
                a.nn_id[gslot] = nid;
                cs_acc_point(acc, f, nd, s, qid, nid);
            }
            // empty support: the wave looks for the nearest neighbour afterwards (deterministic order:
            // position = rank of the lane among this round's misses)
            const unsigned long long mm = __ballot(!has);
            if (mm) {
                const int base_m = sm->miss_n[wave];
                if (!has) {
                    const int pos = base_m + (int)__popcll(mm & ((1ull << lane) - 1ull));
                    if (pos < kCsMiss) {
                        CsMiss e;
                        e.pa = pa_first;
                        e.ea = ea_all;
                        e.qoff = qoff;
                        e.gslot = gslot;
                        e.qid = qid;
                        e.f = f;
                        e.s = s;
                        e.g2 = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, 1);
                        sm->miss[wave][pos] = e;
                    } else { // list full: exact path (it recomputes the whole query)
                        const int p2 = atomicAdd(a.fb_count, 1);
                        a.fb_list[p2] = gslot;
                    }
                }
                // one lane updates the count after every lane has read it
                const int first = __ffsll((long long)mm) - 1;
                const int tot = base_m + (int)__popcll(mm);
                __builtin_amdgcn_wave_barrier();
                if (lane == first) sm->miss_n[wave] = tot < kCsMiss ? tot : kCsMiss;
                __builtin_amdgcn_wave_barrier();
            }
            CS_STAMP(4) // step, outputs, statistics
        }

        // ---- 5. empty-support queries of this wave: cooperative nearest-neighbour search in the run ----------
        {
            const int nm = __builtin_amdgcn_readfirstlane(sm->miss_n[wave]);
            for (int m = 0; m < nm; ++m) {
                const CsMiss e = sm->miss[wave][m];
                const float4 qp = cs_pt(smem_raw, e.qoff);
                float bd = Lim<float>::inf();
                int32_t bi = 0x7FFFFFFF;
                for (uint32_t p = e.pa + (uint32_t)lane * 16u; p < e.ea; p += 64u * 16u) {
                    const float4 c = cs_pt(smem_raw, p);
                    const int32_t cid = w_to_id(c.w);
                    const float d = dist2<float>(qp.x, qp.y, qp.z, c.x, c.y, c.z);
                    if (cid != e.qid && lex_lt(d, cid, bd, bi)) {
                        bd = d;
                        bi = cid;
                    }
                }
#pragma unroll
                for (int dlt = 32; dlt >= 1; dlt >>= 1) {
                    const float od = __shfl_xor(bd, dlt, 64);
                    const int32_t oi = __shfl_xor(bi, dlt, 64);
                    if (lex_lt(od, oi, bd, bi)) {
                        bd = od;
                        bi = oi;
                    }
                }
                if (lane == 0) {
                    // certified when the winner lies inside the radius up to which the 27 cells are complete
                    // (the run's extra cells only add points beyond it)
                    if (bi != 0x7FFFFFFF && bd <= e.g2) {
                        const float nd = wsqrt(bd);
                        if (reaches_past_cover<float>(a, qp.x, qp.y, qp.z, bd)) atomicAdd(a.uncovered, 1);
                        a.nn_dist[e.gslot] = nd;
                        a.nn_id[e.gslot] = bi;
                        cs_acc_point(acc, e.f, nd, e.s, e.qid, bi);
                    } else { // not certified by the 27 cells: the follow-up kernel searches the shell around them.
                        // It starts from this search's winner: the SQUARED distance (exact bits) waits in nn_dist.
                        a.nn_dist[e.gslot] = bd;
                        a.nn_id[e.gslot] = bi == 0x7FFFFFFF ? -1 : bi;
                        const int pos = atomicAdd(a.nn_count, 1);
                        a.nn_list[pos] = e.gslot;
                    }
                }
            }
            if (WTP_DIAG) dt[14] += (unsigned long long)nm;
        }
        CS_STAMP(5) // empty-support follow-up
    }
    if (WTP_DIAG && a.diag && lane == 0) {
        for (int i = 0; i < 15; ++i) atomicAdd(&a.diag[i], dt[i]);
        atomicAdd(&a.diag[15], 1ull);
    }
    __syncthreads();
    Acc full = acc_empty();
    full.max_force = (double)acc.max_force;
    full.sum_u = acc.sum_u;
    full.sum_u2 = acc.sum_u2;
    full.n_move = acc.n_move;
    if (acc.argmin_i >= 0) {
        full.argmin_r = (double)acc.argmin_r;
        full.argmin_i = acc.argmin_i;
        full.argmin_j = acc.argmin_j;
    }
    acc_block_reduce(full, sm->acc);
    if (tid == 0) acc_store(&a.partials[blockIdx.x], full);
}

// Brick census (once per session): own points Q and halo points H of every brick for a candidate BX,
// as histograms (Q in bins of 2, H in bins of 8), so the host can size BX and the LDS point area
// from what the bricks of THIS cloud hold instead of from a density model.
__global__ void cs2_census_kernel(const int32_t* __restrict__ cell_start, const Grid<float>* __restrict__ gp, int BX,
                                  unsigned int* __restrict__ out /* [256 Q-bins | 256 H-bins | non-empty bricks] */) {
    const Grid<float> g = *gp;
    const int nbx = (g.n[0] + BX - 1) / BX, nby = (g.n[1] + 1) / 2, nbz = (g.n[2] + 1) / 2;
    const int nbricks = nbx * nby * nbz;
    for (int brick = blockIdx.x * blockDim.x + threadIdx.x; brick < nbricks; brick += gridDim.x * blockDim.x) {
        const int bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
        const int gx0 = bx * BX, gx1 = (gx0 + BX - 1) < g.n[0] - 1 ? (gx0 + BX - 1) : g.n[0] - 1;
        const int hx0 = gx0 > 0 ? gx0 - 1 : 0, hx1 = gx1 + 1 < g.n[0] - 1 ? gx1 + 1 : g.n[0] - 1;
        int Q = 0, H = 0;
        for (int r = 0; r < 16; ++r) {
            const int gy = by * 2 - 1 + (r & 3), gz = bz * 2 - 1 + (r >> 2);
            if (gy < 0 || gy >= g.n[1] || gz < 0 || gz >= g.n[2]) continue;
            const int base = (gz * g.n[1] + gy) * g.n[0];
            H += cell_start[base + hx1 + 1] - cell_start[base + hx0];
            const bool own = ((r & 3) == 1 || (r & 3) == 2) && ((r >> 2) == 1 || (r >> 2) == 2);
            if (own) Q += cell_start[base + gx1 + 1] - cell_start[base + gx0];
        }
        if (Q == 0) continue;
        atomicAdd(&out[Q / 2 < 255 ? Q / 2 : 255], 1u);
        atomicAdd(&out[256 + (H / 8 < 255 ? H / 8 : 255)], 1u);
        atomicAdd(&out[512], 1u);
    }
}

int launch_cs2_census(wtp_ctx* ctx, int BX, unsigned int* d_out513) {
    WTP_HIP(ctx, hipMemsetAsync(d_out513, 0, 513 * sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(cs2_census_kernel, dim3(512), dim3(256), 0, ctx->stream, (const int32_t*)ctx->cell_start.p,
                       (const Grid<float>*)ctx->grid.p, BX, d_out513);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int cs2_max_bx() { return kCsMaxBX; }

// Follow-up of cs2_kernel: the queries whose nearest neighbour the 27 staged cells could not certify
// (empty support AND the winner beyond the provable radius: ~0.5 % of a uniform cloud).  Force, step and
// new position are final already; what is missing is nn_dist / nn_id and the point's share of the
// statistics.  One lane per query, 5 x 5 x 5 cells straight from the sorted array (25 contiguous runs),
// canonical (d2, id) minimum, certified against the radius the 125 cells cover; whatever is left (isolated
// points) goes to the exact path, which recomputes the whole query.
constexpr int kNnFixBlocks = 512;
constexpr int kNnFixThreads = 1024; // 16 waves per block, one query per wave at a time: the search is a chain of
                                    // dependent global loads, so it wants many waves in flight, not many lanes per query
__global__ __launch_bounds__(kNnFixThreads, 8) void cs2_nnfix_kernel(SearchArgs<float> a, int part_base) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    __shared__ Acc sacc[kNnFixThreads / 64];
    const Grid<float> g = *a.grid;
    const int n = *a.nn_count;
    const int lane = threadIdx.x & 63;
    const int wave_g = (blockIdx.x * kNnFixThreads + threadIdx.x) >> 6, nwaves = (gridDim.x * kNnFixThreads) >> 6;
    Acc acc = acc_empty();
    // four queries per wave, 16 lanes each.  The 27 cells around the query were searched by its brick; what is open
    // is the shell of the 5 x 5 x 5 block: 25 x-rows, each cut into {cell cx-2 | cells cx-1..cx+1 | cell cx+2}, minus
    // the middle of the 9 inner rows.  A piece is read only if its box lies closer than the current winner — usually
    // one to three of the 66 (reading all 25 rows cost 3 KB of scattered lines per query: bandwidth, 86 us per step).
    const int grp = lane >> 4, l16 = lane & 15;
    for (int i0 = wave_g * 4; i0 < n; i0 += nwaves * 4) {
        const int i = i0 + grp;
        const bool on = i < n;
        const int gslot = a.nn_list[on ? i : i0];
        const float4 qp = a.snap[gslot];
        const int32_t qid = w_to_id(qp.w);
        const int cx = cell_coord(g, qp.x, 0), cy = cell_coord(g, qp.y, 1), cz = cell_coord(g, qp.z, 2);
        float bd = a.nn_dist[gslot]; // squared distance of the brick's winner (+inf when it found nothing)
        int32_t bi = a.nn_id[gslot];
        bi = bi < 0 ? 0x7FFFFFFF : bi;
        // distance from the query to the slab of cells [c + lo, c + hi] along one axis, less the cell map's rounding
        auto gap = [&](float v, int axis, int c, int lo, int hi) {
            const float a0 = g.org[axis] + (float)(c + lo) * g.c, a1 = g.org[axis] + (float)(c + hi + 1) * g.c;
            float d = v < a0 ? a0 - v : (v > a1 ? v - a1 : 0.f);
            d -= g.margin;
            return d > 0.f ? d : 0.f;
        };
        for (int piece = l16; piece < 75; piece += 16) {
            const int r = piece / 3, seg = piece % 3;
            const int dz = r / 5 - 2, dy = r % 5 - 2;
            const bool inner = dz >= -1 && dz <= 1 && dy >= -1 && dy <= 1;
            if (inner && seg == 1) continue; // searched by the brick
            const int gz = cz + dz, gy = cy + dy;
            if (gz < 0 || gz >= g.n[2] || gy < 0 || gy >= g.n[1]) continue;
            int xa = seg == 0 ? cx - 2 : (seg == 1 ? cx - 1 : cx + 2), xb = seg == 1 ? cx + 1 : xa;
            xa = xa < 0 ? 0 : xa;
            xb = xb > g.n[0] - 1 ? g.n[0] - 1 : xb;
            if (xa > xb) continue;
            // edge cells are unbounded outward (clamped map): no pruning against a box that includes one
            const bool edge = xa == 0 || xb == g.n[0] - 1 || gy == 0 || gy == g.n[1] - 1 || gz == 0 || gz == g.n[2] - 1;
            if (!edge) {
                const float gx2 = gap(qp.x, 0, 0, xa, xb), gy2 = gap(qp.y, 1, gy, 0, 0), gz2 = gap(qp.z, 2, gz, 0, 0);
                if ((gx2 * gx2 + gy2 * gy2) + gz2 * gz2 > bd) continue; // nothing in there can beat the winner
            }
            const int base = (gz * g.n[1] + gy) * g.n[0];
            const int p1 = a.cell_start[base + xb + 1];
            for (int p = a.cell_start[base + xa]; p < p1; p += 4) {
                float4 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = a.snap[p + u < p1 ? p + u : p];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t cid = w_to_id(c[u].w);
                    const float d = dist2<float>(qp.x, qp.y, qp.z, c[u].x, c[u].y, c[u].z);
                    if (p + u < p1 && cid != qid && lex_lt(d, cid, bd, bi)) {
                        bd = d;
                        bi = cid;
                    }
                }
            }
        }
#pragma unroll
        for (int dlt = 8; dlt >= 1; dlt >>= 1) { // minimum over the group's 16 lanes
            const float od = __shfl_xor(bd, dlt, 64);
            const int32_t oi = __shfl_xor(bi, dlt, 64);
            if (lex_lt(od, oi, bd, bi)) {
                bd = od;
                bi = oi;
            }
        }
        if (l16 == 0 && on) {
            const float g2 = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, 2);
            if (bi != 0x7FFFFFFF && bd <= g2) {
                const float s = a.spacing_pp ? a.spacing_pp[qid] : a.spacing_const;
                const float nd = wsqrt(bd);
                if (reaches_past_cover<float>(a, qp.x, qp.y, qp.z, bd)) atomicAdd(a.uncovered, 1);
                a.nn_dist[gslot] = nd;
                a.nn_id[gslot] = bi;
                acc_point<float>(acc, a.forces[gslot], nd, s, qid, bi);
            } else {
                const int pos = atomicAdd(a.fb_count, 1);
                a.fb_list[pos] = gslot;
            }
        }
    }
    acc_block_reduce(acc, sacc);
    if (threadIdx.x == 0) acc_store(&a.partials[part_base + blockIdx.x], acc);
}

// ---- hand-backs whose support is wider than their cell (variable spacing: the cells follow the spacing a TYPICAL
// point asks for, the coarse part of a graded cloud asks for more) ------------------------------------------
// The compact-support argument needs no brick: the ball of radius u0*s around the query, searched in the block of
// (2R+1)^3 cells that provably contains it (R <= 4), holds n_lim <= k points -> they are the k-set's contributing
// part and the sum over them is the reference's sum (src/repel.jl:270-280, src/repel_forces.jl:96-100).  Eight
// lanes per query, eight queries per wave (round 3; sixteen / four before), one x-row of the block per lane at a time; no LDS, so many waves are in
// flight and the chain of dependent loads (spacing, row bounds, points) overlaps across queries.  The exact
// wave-per-query path took 7 ns for each of these (16 % of a 64x-graded cloud: 1.15 of 1.58 ms per iteration).
// What this kernel cannot certify either — more than k points in the ball, a coincident neighbour, a support wider
// than four cells, an empty ball whose nearest neighbour the block does not certify — moves on to that path.
constexpr int kBallThreads = 256;
constexpr int kBallBlocksMax = 3072; // (512 left two thirds of the wave slots empty: the kernel is a chain of dependent loads per query, its rate is the number of queries in flight)
constexpr int kBallRMax = 4;
#ifndef WTP_BALL_LANES
#define WTP_BALL_LANES 8 // (measured on the graded 10 M cloud, ms per iteration: 16 lanes 9.68, 8 lanes 9.10, 4 lanes 9.61)
#endif
constexpr int kBallLanes = WTP_BALL_LANES;          // lanes per query (a power of two)
constexpr int kBallPerWave = 64 / kBallLanes;
__global__ __launch_bounds__(kBallThreads, 6) void cs_ball_kernel(SearchArgs<float> a, const int32_t* __restrict__ list,
                                                               const int32_t* __restrict__ list_count,
                                                               int32_t* __restrict__ out_list, int32_t* __restrict__ out_count,
                                                               int part_base) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    __shared__ Acc sacc[kBallThreads / 64];
    const Grid<float> g = *a.grid;
    const int n = *list_count;
    const int K = a.k;
    const int lane = threadIdx.x & 63;
    const int grp = lane / kBallLanes, l16 = lane % kBallLanes; // (l16: the lane's index inside its query's group)
    const int wave_g = (blockIdx.x * kBallThreads + threadIdx.x) >> 6, nwaves = (gridDim.x * kBallThreads) >> 6;
    const ForceCoefCs fc = force_coef_cs(a.beta, a.u0);
    Acc acc = acc_empty();
    for (int i0 = wave_g * kBallPerWave; i0 < n; i0 += nwaves * kBallPerWave) {
        const int i = i0 + grp;
        const bool on = i < n;
        const int gslot = list[on ? i : i0];
        const float4 qp = a.query[gslot]; // (the point where it is now: the snapshot's own entry on a fresh one, its moved self on a stale one)
        const int32_t qid = w_to_id(qp.w);
        const float s = a.spacing_pp ? a.spacing_pp[qid] : a.spacing_const;
        const float inv_s2 = 1.f / (s * s);
        const float lim = (a.u0 * a.u0) * (s * s);
        const int cx = cell_coord(g, qp.x, 0), cy = cell_coord(g, qp.y, 1), cz = cell_coord(g, qp.z, 2);
        // smallest block that provably holds the ball
        int R = 0;
        float g2 = 0.f;
#pragma unroll
        for (int r = 1; r <= kBallRMax; ++r) {
            const float t = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, r);
            const bool take = R == 0 && lim <= t;
            g2 = take ? t : g2;
            R = take ? r : R;
        }
        const bool ok = on && R > 0;
        const int side = 2 * R + 1, nrows = ok ? side * side : 0;
        const int x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R > g.n[0] - 1 ? g.n[0] - 1 : cx + R;
        int n_lim = 0;
        bool coincident = false;
        float Fx = 0.f, Fy = 0.f, Fz = 0.f;
        int32_t nid = 0x7FFFFFFF;
        float nd2 = Lim<float>::inf();
        // a lane's rows (at most eleven of the 81): all bounds first, then the points four loads at a time — the loop is a
        // chain of global round trips otherwise (one per point: 3.7 ns per query measured, 161 k queries 0.6 ms)
        constexpr int kRowsPerLane = ((2 * kBallRMax + 1) * (2 * kBallRMax + 1) + kBallLanes - 1) / kBallLanes;
        int ps[kRowsPerLane], pe[kRowsPerLane];
#pragma unroll
        for (int j = 0; j < kRowsPerLane; ++j) {
            const int row = l16 + kBallLanes * j;
            const int y = cy + row % side - R, z = cz + row / side - R;
            const bool in = row < nrows && y >= 0 && y < g.n[1] && z >= 0 && z < g.n[2];
            const int base = in ? (z * g.n[1] + y) * g.n[0] : 0;
            ps[j] = a.cell_start[base + (in ? x0 : 0)];
            pe[j] = in ? a.cell_start[base + x1 + 1] : ps[j]; // (an absent row: empty)
        }
        auto visit = [&](const float4& c, bool valid) {
            const float dx = qp.x - c.x, dy = qp.y - c.y, dz = qp.z - c.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            const int32_t cid = w_to_id(c.w);
            const bool inl = valid && d <= lim;
            n_lim += inl ? 1 : 0;
            const bool other = valid && cid != qid; // self skipped by index (src/repel.jl:271)
            const bool nearer = other && lex_lt(d, cid, nd2, nid);
            nd2 = nearer ? d : nd2;
            nid = nearer ? cid : nid;
            const bool act = inl && other;
            const float f = force_fast_cs(fc, d * inv_s2);
            const float coef = (act && d > 0.f) ? f * __builtin_amdgcn_rsqf(d) : 0.f;
            Fx = __builtin_fmaf(coef, dx, Fx);
            Fy = __builtin_fmaf(coef, dy, Fy);
            Fz = __builtin_fmaf(coef, dz, Fz);
            coincident = coincident || (act && !(d > 0.f));
        };
#pragma unroll
        for (int j = 0; j < kRowsPerLane; ++j) {
            for (int p = ps[j]; p < pe[j]; p += 4) {
                float4 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = a.snap[p + u < pe[j] ? p + u : p];
#pragma unroll
                for (int u = 0; u < 4; ++u) visit(c[u], p + u < pe[j]);
            }
        }
        // the sixteen lanes of the query: sums in a fixed order, lexicographic minimum
#pragma unroll
        for (int dlt = kBallLanes / 2; dlt >= 1; dlt >>= 1) {
            Fx += __shfl_xor(Fx, dlt, 64);
            Fy += __shfl_xor(Fy, dlt, 64);
            Fz += __shfl_xor(Fz, dlt, 64);
            n_lim += __shfl_xor(n_lim, dlt, 64);
            coincident = coincident || (__shfl_xor((int)coincident, dlt, 64) != 0);
            const float od = __shfl_xor(nd2, dlt, 64);
            const int32_t oi = __shfl_xor(nid, dlt, 64);
            if (lex_lt(od, oi, nd2, nid)) {
                nd2 = od;
                nid = oi;
            }
        }
        if (l16 == 0 && on) {
            // nearest neighbour: inside the ball it is the global one; outside it must lie within the radius the
            // block is complete for
            const bool nn_ok = nid != 0x7FFFFFFF && (nd2 <= lim || nd2 <= g2);
            if (!ok || n_lim > K || coincident || !nn_ok || qid < a.n_fixed) {
                const int pos = atomicAdd(out_count, 1);
                out_list[pos] = gslot;
            } else {
                float4 o;
                const float f = cs_step_point(a, s, qp.x, qp.y, qp.z, Fx, Fy, Fz, o.x, o.y, o.z);
                o.w = qp.w;
                a.out[gslot] = o;
                a.forces[gslot] = f;
                const float nd = wsqrt(nd2);
                if (reaches_past_cover<float>(a, qp.x, qp.y, qp.z, nd2 > lim ? nd2 : lim)) atomicAdd(a.uncovered, 1);
                a.nn_dist[gslot] = nd;
                a.nn_id[gslot] = nid;
                acc_point<float>(acc, f, nd, s, qid, nid);
            }
        }
    }
    acc_block_reduce(acc, sacc);
    if (threadIdx.x == 0) acc_store(&a.partials[part_base + blockIdx.x], acc);
}

// every slot of the snapshot as a hand-back list (sweeps against a stale snapshot: the ball kernel takes all queries)
__global__ void cs_all_slots_kernel(int32_t* __restrict__ list, int32_t n, int32_t* __restrict__ count) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) list[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = n;
}

int launch_cs_all_slots(wtp_ctx* ctx, int32_t* list, int32_t n, int32_t* count) {
    hipLaunchKernelGGL(cs_all_slots_kernel, dim3(1024), dim3(256), 0, ctx->stream, list, n, count);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// Runs between the brick kernels' follow-up and the exact path; on return a.fb_list / a.fb_count name what is left.
int launch_cs_ball(wtp_ctx* ctx, SearchArgs<float>& a, int32_t* rest_list, int32_t* rest_count) {
    int blocks = (int)((a.n + 1023) / 1024);
    blocks = blocks < 8 ? 8 : (blocks > kBallBlocksMax ? kBallBlocksMax : blocks);
    if (const char* e = getenv("WTP_BALL_BLOCKS")) blocks = atoi(e) > 0 && atoi(e) <= kBallBlocksMax ? atoi(e) : blocks;
    hipLaunchKernelGGL(cs_ball_kernel, dim3(blocks), dim3(kBallThreads), 0, ctx->stream, a, (const int32_t*)a.fb_list,
                       (const int32_t*)a.fb_count, rest_list, rest_count, a.used_brick);
    a.used_brick += blocks;
    a.fb_list = rest_list;
    a.fb_count = rest_count;
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// Variable spacing (graded clouds): the cells follow the spacing of the DENSE part, so in the coarse bulk every point's
// support is wider than the radius any 27-cell block certifies and the brick kernel would stage a brick (tables, halo,
// four barriers) only to hand each of its two or three points back.  One thread per brick looks at the brick's own points
// first: if every one is movable and asks for a support beyond (1.5 c)^2 — the largest radius a block certifies — the brick
// is marked dead and its points go straight to the hand-back list (the ball kernel's input).  A dense brick is left at its
// first point.  Graded 10 M points: 240 k of 300 k bricks are dead, cs2_kernel 3.4 -> see DESIGN.md.
__global__ void cs2_dead_kernel(SearchArgs<float> a, int BX, uint8_t* __restrict__ dead, int dead_cap) {
    if (a.stop && *a.stop) return;
    const Grid<float> g = *a.grid;
    const int nbx = (g.n[0] + BX - 1) / BX, nby = (g.n[1] + 1) / 2, nbz = (g.n[2] + 1) / 2;
    const int nbricks = nbx * nby * nbz;
    const float g2max = (1.5f * g.c) * (1.5f * g.c);
    for (int brick = blockIdx.x * blockDim.x + threadIdx.x; brick < nbricks && brick < dead_cap; brick += gridDim.x * blockDim.x) {
        const int bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
        const int gx0 = bx * BX, gx1 = (gx0 + BX - 1) < g.n[0] - 1 ? (gx0 + BX - 1) : g.n[0] - 1;
        int rs[4], re[4];
        bool is_dead = true;
        for (int r4 = 0; r4 < 4; ++r4) {
            const int gy = by * 2 + (r4 & 1), gz = bz * 2 + (r4 >> 1);
            rs[r4] = re[r4] = 0;
            if (gy >= g.n[1] || gz >= g.n[2]) continue;
            const int base = (gz * g.n[1] + gy) * g.n[0];
            rs[r4] = a.cell_start[base + gx0];
            re[r4] = a.cell_start[base + gx1 + 1];
        }
        for (int r4 = 0; r4 < 4 && is_dead; ++r4)
            for (int i = rs[r4]; i < re[r4] && is_dead; ++i) {
                const int32_t id = w_to_id(a.snap[i].w);
                if (id < a.n_fixed) {
                    is_dead = false;
                } else {
                    const float s = a.spacing_pp[id];
                    is_dead = (a.u0 * a.u0) * (s * s) > g2max;
                }
            }
        dead[brick] = is_dead ? 1 : 0;
        if (is_dead) {
            const int q = (re[0] - rs[0]) + (re[1] - rs[1]) + (re[2] - rs[2]) + (re[3] - rs[3]);
            if (q > 0) {
                int pos = atomicAdd(a.fb_count, q);
                for (int r4 = 0; r4 < 4; ++r4)
                    for (int i = rs[r4]; i < re[r4]; ++i) a.fb_list[pos++] = i;
            }
        }
    }
}

int launch_cs2_dead(wtp_ctx* ctx, SearchArgs<float>& a, uint8_t* d_dead, int dead_cap) {
    hipLaunchKernelGGL(cs2_dead_kernel, dim3(1024), dim3(256), 0, ctx->stream, a, a.cs2_bx, d_dead, dead_cap);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int launch_cs2(wtp_ctx* ctx, SearchArgs<float>& a) {
    const int hcap = a.brick_hcap, BX = a.cs2_bx;
    if (hcap < 64 || hcap > 4096 || BX < 1 || BX > kCsMaxBX)
        return fail(ctx, WTP_ERR_STATE, "compact-support sweep launched without a brick geometry");
    const size_t smem = cs2_smem_bytes(hcap);
    const bool chunked = a.cs2_chunked != 0;
    const void* fn = chunked ? (const void*)cs2_kernel<true> : (const void*)cs2_kernel<false>;
    if (ctx->cs2_smem != smem || ctx->cs2_fn != fn) { // per context: several contexts (devices) may coexist in one process
        WTP_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        int occ = 0;
        hipError_t e = chunked ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cs2_kernel<true>, kCsThreads, smem)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cs2_kernel<false>, kCsThreads, smem);
        if (e != hipSuccess || occ < 1) occ = 1;
        ctx->cs2_occ = occ > 4 ? 4 : occ;
        ctx->cs2_smem = smem;
        ctx->cs2_fn = fn;
    }
    int gsz = ctx->sm_count * ctx->cs2_occ;
    gsz -= gsz % 8;
    if (gsz < 8) gsz = 8;
    if (chunked)
        hipLaunchKernelGGL(cs2_kernel<true>, dim3(gsz), dim3(kCsThreads), smem, ctx->stream, a, hcap, BX);
    else
        hipLaunchKernelGGL(cs2_kernel<false>, dim3(gsz), dim3(kCsThreads), smem, ctx->stream, a, hcap, BX);
    a.used_brick = gsz;
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// the follow-up kernel; its partial slots follow the bricks' (brick_partials() leaves room for them)
int launch_cs2_followup(wtp_ctx* ctx, SearchArgs<float>& a) {
    // ~0.5 % of the queries end up here: a block per 2048 points keeps small clouds from paying for 512 idle blocks
    // (18 us at 47 k points)
    int blocks = (int)((a.n + 2047) / 2048);
    blocks = blocks < 8 ? 8 : (blocks > kNnFixBlocks ? kNnFixBlocks : blocks);
    hipLaunchKernelGGL(cs2_nnfix_kernel, dim3(blocks), dim3(kNnFixThreads), 0, ctx->stream, a, a.used_brick);
    a.used_brick += blocks;
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

} // namespace wtp
