// wtp_wave.hip — exact wave-per-query search (gfx950): one 64-lane wavefront owns one query.
//
// Serves every query the 27-cell brick kernels hand back, and whole sweeps the brick path does
// not cover (fp64, k > 48, stale snapshots of rebuild_every > 1, src/repel.jl:245,262-266).
//   gather   the wave walks the x-rows of the (2r+1)^dim cell block around the query's cell;
//            each row is one contiguous run of the sorted Pt array; the runs of up to 64 rows are
//            flattened (scan of their lengths) and read 64 candidates per step whatever rows they
//            come from.  Candidates inside the provable radius are appended to a per-wave LDS list
//            with ballot + mbcnt prefix sums.
//   select   the k-th smallest d2 is found by quickselect on the d2 bit patterns (monotone for
//            d2 >= 0), pivoting on candidates: count(d2 <= pivot) is a ballot + s_bcnt per 64
//            candidates, keys held in VGPRs.  Survivors (d2 <= cut) are ranked by the canonical (d2, index) order with a
//            broadcast compare loop, giving the sorted first k.
//   emit     MODE 0 writes the row (coalesced); MODE 1 evaluates the k contributions in
//            parallel, then lane 0 adds them in ascending order exactly like
//            src/repel.jl:270-280, steps the point (:282-291) and accumulates the reductions.
// Rings grow (r = 2, 4, ...) until the k-th hit is provably final.  A candidate list larger
// than the LDS buffer sends the query to the serial kernel of wtp_generic.hip.
#include "wtp_device.hpp"

namespace wtp {

static constexpr int kWaves = 4;             // waves per workgroup
static constexpr int kThreads = kWaves * 64;
static constexpr int kCap = 512;             // candidates buffered per wave (1024 until round 3: the list is what limits the waves per CU, and the kernel is latency-bound)
static constexpr int kKeyRegs = kCap / 64;
static constexpr int kSurv = 256;            // survivors (d2 <= cut) ranked per wave

template <typename T> struct Bits;
template <> struct Bits<float> {
    using U = uint32_t;
    static __device__ U of(float v) { return __builtin_bit_cast(uint32_t, v); }
    static __device__ float back(U u) { return __builtin_bit_cast(float, u); }
    static constexpr U kInf = 0x7F800000u;
};
template <> struct Bits<double> {
    using U = uint64_t;
    static __device__ U of(double v) { return __builtin_bit_cast(uint64_t, v); }
    static __device__ double back(U u) { return __builtin_bit_cast(double, u); }
    static constexpr U kInf = 0x7FF0000000000000ull;
};

// SWEEP: the slot of every candidate (the force terms fetch the point again) and the k force terms; topology rows need
// neither — 10.5 KB per wave instead of 17.6 KB (fp32), i.e. 15 instead of 9 waves per CU: the kernel serves hand-backs one
// query per wave, so its run time is (queries / resident waves) x the latency of one query
template <typename T, bool SWEEP> struct WaveSmem;
template <typename T> struct WaveSmem<T, true> {
    T d2[kCap];
    int32_t id[kCap];
    int32_t slot[kCap];
    T sd2[kSurv];       // survivors
    int32_t sid[kSurv];
    int32_t sslot[kSurv];
    T od2[kGenericKMax]; // sorted first k
    int32_t oid[kGenericKMax];
    int32_t oslot[kGenericKMax];
    // (the k force terms live in d2[], which is dead once the first k are ranked: 3 x kGenericKMax <= kCap)
};
static_assert(3 * kGenericKMax <= kCap, "the force terms reuse the candidate list");
template <typename T> struct WaveSmem<T, false> {
    T d2[kCap];
    int32_t id[kCap];
    T sd2[kSurv];
    int32_t sid[kSurv];
    T od2[kGenericKMax];
    int32_t oid[kGenericKMax];
};

__device__ inline int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }

template <typename T, int MODE>
__global__ __launch_bounds__(kThreads) void wave_kernel(SearchArgs<T> a, const int32_t* __restrict__ list,
                                                        const int32_t* __restrict__ list_count, int all,
                                                        int part_base) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using U = typename Bits<T>::U;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    using Smem = WaveSmem<T, MODE == 1>;
    Smem* sm = reinterpret_cast<Smem*>(smem_raw) + wave;
    Acc* sm_acc = reinterpret_cast<Acc*>(smem_raw + sizeof(Smem) * kWaves);
    const Grid<T> g = *a.grid;
    const int nq = all ? a.n : *list_count;
    const int K = a.k;
    const bool skip_self = (MODE == 0) && !a.include_self;
    Acc acc = acc_empty();
    const int wave_global = blockIdx.x * kWaves + wave;
    const int wave_stride = gridDim.x * kWaves;

    for (int qi = wave_global; qi < nq; qi += wave_stride) {
        const int slot = all ? qi : list[qi];
        const Pt<T> q = a.query[slot];
        const int32_t id = w_to_id(q.w);
        if (MODE == 1 && id < a.n_fixed) {
            if (lane == 0) {
                a.out[slot] = q;
                a.forces[slot] = (T)0;
                a.nn_dist[slot] = Lim<T>::inf();
                a.nn_id[slot] = -1;
            }
            continue;
        }
        const int cx = cell_coord(g, q.x, 0), cy = cell_coord(g, q.y, 1), cz = cell_coord(g, q.z, 2);
        int m = 0;
        bool overflow = false;
        // Compact-support shortcut (ClippedSpacingForce, the default law): points beyond u0*s contribute
        // exactly 0 to the sum over the k nearest (src/repel_forces.jl:96-100), so if the ball of radius
        // u0*s holds 2..k points (self included) they ARE the nearest ones, the nearest neighbour is among
        // them and the sum over them is the reference's sum (adding zeros changes nothing).  That ball
        // needs a much smaller block than the k-th neighbour and no selection at all.  Anything else
        // (more than k points, list overflow) takes the general path below.
        int Kq = K;            // neighbours emitted: K, or the ball's population on the shortcut
        T cs_lim = (T)0;
        int cs_r = 0;
        if (MODE == 1 && a.force_kind == WTP_FORCE_CLIPPED_SPACING && K >= 2) {
            const T s = a.spacing_pp ? a.spacing_pp[id] : a.spacing_const;
            cs_lim = (a.u0 * a.u0) * (s * s);
            for (int r = 1; r <= 6 && !cs_r; ++r)
                if (safe_radius2(g, q.x, q.y, q.z, cx, cy, cz, r) >= cs_lim) cs_r = r;
        }
        bool cs_try = cs_r > 0, cs_done = false;
        // hand-backs of the brick kernels already failed at 27 cells: start at 5^3; whole-cloud runs
        // (fp64, stale snapshots) start at the 27 cells, which certify most queries
        // (hand-backs of wtp_ksel.hip failed at 5^3 of its small cells: they start at 7^3, a.fb_r0 = 3)
        for (int r2 = all ? 1 : (a.fb_r0 > 0 ? a.fb_r0 : 2);;) {
            const int r = cs_try ? cs_r : r2;
            m = 0;
            // the support-ball attempt gathers everything the block certifies (>= the ball), so that a query
            // alone in its ball finds its nearest neighbour in the same list
            const T g2 = safe_radius2(g, q.x, q.y, q.z, cx, cy, cz, r);
            const int z0 = cz - r < 0 ? 0 : cz - r, z1 = cz + r > g.n[2] - 1 ? g.n[2] - 1 : cz + r;
            const int y0 = cy - r < 0 ? 0 : cy - r, y1 = cy + r > g.n[1] - 1 ? g.n[1] - 1 : cy + r;
            const int x0 = cx - r < 0 ? 0 : cx - r, x1 = cx + r > g.n[0] - 1 ? g.n[0] - 1 : cx + r;
            // Rows of the block are independent contiguous runs of the sorted array.  Lanes fetch
            // the run bounds of up to 64 rows in one go.
            auto consume = [&](int p, bool valid, const Pt<T>& c) {
                bool take = false;
                T d = 0;
                int32_t cid = 0;
                if (valid) {
                    cid = w_to_id(c.w);
                    d = dist2<T>(q.x, q.y, q.z, c.x, c.y, c.z);
                    take = (d <= g2) && !(skip_self && cid == id);
                }
                const unsigned long long mask = __ballot(take);
                const int pos = m + __popcll(mask & ((1ull << lane) - 1ull));
                if (take && pos < kCap) {
                    sm->d2[pos] = d;
                    sm->id[pos] = cid;
                    if constexpr (MODE == 1) sm->slot[pos] = p;
                }
                m += __popcll(mask);
                if (m > kCap) overflow = true;
            };
            const int ny_rows = y1 - y0 + 1;
            const int nrows = ny_rows * (z1 - z0 + 1);
            for (int rb = 0; rb < nrows && !overflow; rb += 64) {
                int my_ps = 0, my_pe = 0;
                if (rb + lane < nrows) {
                    const int i = rb + lane;
                    const int row = ((z0 + i / ny_rows) * g.n[1] + (y0 + i % ny_rows)) * g.n[0];
                    my_ps = a.cell_start[row + x0];
                    my_pe = a.cell_start[row + x1 + 1];
                }
                // The rows' runs are flattened into one candidate stream: an exclusive scan of the run
                // lengths across the lanes, then every step takes the next 64 candidates whatever rows
                // they come from (a 6-step search over the scanned offsets finds each lane's row).  A
                // sparse block of 25 rows x 5 points costs 2 steps instead of 25; two steps' loads are
                // in flight together.
                const int my_len = my_pe - my_ps;
                int incl = my_len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                const int total = __shfl(incl, 63, 64);
                const int excl = incl - my_len;
                for (int base = 0; base < total && !overflow; base += 2 * 64) {
                    Pt<T> c[2];
                    int pp[2];
                    bool vv[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int f = base + u * 64 + lane;
                        vv[u] = f < total;
                        int lo = 0, hi = 63;
#pragma unroll
                        for (int it = 0; it < 6; ++it) { // largest row whose first candidate is <= f
                            const int mid = (lo + hi + 1) >> 1;
                            const bool ge = __shfl(excl, mid, 64) <= f;
                            lo = ge ? mid : lo;
                            hi = ge ? hi : mid - 1;
                        }
                        pp[u] = __shfl(my_ps, lo, 64) + (f - __shfl(excl, lo, 64));
                        c[u] = a.snap[vv[u] ? pp[u] : 0];
                    }
                    consume(pp[0], vv[0], c[0]);
                    if (base + 64 < total && !overflow) consume(pp[1], vv[1], c[1]);
                }
            }
            if (cs_try) {
                cs_try = false;
                int m_lim = 0; // population of the support ball (self included)
                if (!overflow) {
                    for (int i0 = 0; i0 < m; i0 += 64) {
                        const int i = i0 + lane;
                        m_lim += __popcll(__ballot(i < m && sm->d2[i] <= cs_lim));
                    }
                }
                if (!overflow && m_lim >= 2 && m_lim <= K) {
                    Kq = m_lim;
                    cs_done = true; // the answer is everything within the ball: the cut is its radius
                    break;
                }
                // Alone in its support ball: every term of the sum is exactly 0 and only the nearest
                // neighbour (nn_dist / nn_id) is still unknown — a 2-nearest search (self + one) gives
                // the same step as the k-list would.  The list already holds everything the block
                // certifies, so usually it is right there.
                if (!overflow && m_lim == 1) {
                    Kq = 2;
                    if (m >= 2) break; // the 2nd smallest lies inside the certified radius: done
                    r2 = r + 1;        // nobody else in the block: grow it
                }
                overflow = false;
                continue; // general path
            }
            if (overflow) break;
            if (m >= Kq) break;                      // everything inside g2 is known: the Kq-th is final
            if (g2 == Lim<T>::inf()) break;          // block covers the grid
            r2 *= 2;
        }
        if (overflow || m < Kq) { // m < K only when fewer than k points exist in reach (validated upstream)
            if (lane == 0) {
                const int pos = atomicAdd(a.fb2_count, 1);
                a.fb2_list[pos] = slot;
            }
            continue;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- select: k-th smallest d2 over the bit patterns, keys in registers -----------------------
        U key[kKeyRegs];
#pragma unroll
        for (int j = 0; j < kKeyRegs; ++j) {
            const int i = j * 64 + lane;
            key[j] = i < m ? Bits<T>::of(sm->d2[i]) : ~(U)0;
        }
        // Quickselect with ballots.  Invariant: count(key <= lo) < Kq <= count(key <= hi); a pivot is
        // any candidate strictly inside (lo, hi).  ~2 ln m rounds expected,
        // against the 32 (fp32) / 64 (fp64) rounds a bisection over the bit pattern needs.
        U lo = 0, hi = cs_done ? Bits<T>::of(cs_lim) : Bits<T>::kInf;
        bool lo_valid = false;
        while (!cs_done) { // cs_done: everything within the support ball is the answer, no selection
            bool found = false;
            U pivot = 0;
#pragma unroll
            for (int j = 0; j < kKeyRegs; ++j) {
                if (!found && j * 64 < m) {
                    const unsigned long long mask = __ballot((!lo_valid || key[j] > lo) && key[j] < hi);
                    if (mask) {
                        const unsigned long long upper = mask & (~0ull << 31);
                        const int src = __builtin_ctzll(upper ? upper : mask);
                        pivot = __shfl(key[j], src, 64);
                        found = true;
                    }
                }
            }
            if (!found) break; // nothing strictly between: the Kq-th smallest is hi
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < kKeyRegs; ++j)
                if (j * 64 < m) cnt += __popcll(__ballot(key[j] <= pivot));
            if (cnt >= Kq) {
                hi = pivot;
            } else {
                lo = pivot;
                lo_valid = true;
            }
        }
        lo = hi;
        const U cut = lo;
        // ---- survivors: d2 <= cut (k of them, more only on ties at the cut) --------------------------
        int ns = 0;
#pragma unroll
        for (int j = 0; j < kKeyRegs; ++j) {
            if (j * 64 < m) {
                const int i = j * 64 + lane;
                const bool take = key[j] <= cut;
                const unsigned long long mask = __ballot(take);
                const int pos = ns + __popcll(mask & ((1ull << lane) - 1ull));
                if (take && pos < kSurv) {
                    sm->sd2[pos] = sm->d2[i];
                    sm->sid[pos] = sm->id[i];
                    if constexpr (MODE == 1) sm->sslot[pos] = sm->slot[i];
                }
                ns += __popcll(mask);
            }
        }
        if (ns > kSurv) { // a mass tie at the cut: serial kernel
            if (lane == 0) {
                const int pos = atomicAdd(a.fb2_count, 1);
                a.fb2_list[pos] = slot;
            }
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- canonical rank among survivors -> sorted first k ----------------------------------------
        for (int i = lane; i < ns; i += 64) {
            const T md = sm->sd2[i];
            const int32_t mi = sm->sid[i];
            int rank = 0;
            for (int j = 0; j < ns; ++j) rank += lex_lt(sm->sd2[j], sm->sid[j], md, mi) ? 1 : 0;
            if (rank < Kq) {
                sm->od2[rank] = md;
                sm->oid[rank] = mi;
                if constexpr (MODE == 1) sm->oslot[rank] = sm->sslot[i];
            }
        }
        __builtin_amdgcn_wave_barrier();

        if constexpr (MODE == 0) {
            for (int j = lane; j < K; j += 64) {
                a.idx_out[(int64_t)id * K + j] = sm->oid[j];
                if (a.dist_out) a.dist_out[(int64_t)id * K + j] = wsqrt(sm->od2[j]);
            }
        } else {
            const T s = a.spacing_pp ? a.spacing_pp[id] : a.spacing_const;
            T* const sfx = sm->d2;
            T* const sfy = sm->d2 + kGenericKMax;
            T* const sfz = sm->d2 + 2 * kGenericKMax;
            for (int j = lane; j < Kq; j += 64) {
                T fx = 0, fy = 0, fz = 0;
                if (sm->oid[j] != id) {
                    const Pt<T> c = a.snap[sm->oslot[j]];
                    add_force<T>(a, g.dim, s, q.x, q.y, q.z, id, c.x, c.y, c.z, sm->oid[j], sm->od2[j], fx, fy, fz);
                }
                sfx[j] = fx;
                sfy[j] = fy;
                sfz[j] = fz;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                T Fx = 0, Fy = 0, Fz = 0;
                int32_t nid = -1;
                T nd = Lim<T>::inf();
                for (int j = 0; j < Kq; ++j) { // ascending (d2, id), self skipped by index (:271)
                    if (sm->oid[j] == id) continue;
                    if (nid < 0) {
                        nid = sm->oid[j];
                        nd = wsqrt(sm->od2[j]);
                    }
                    Fx = Fx + sfx[j];
                    Fy = Fy + sfy[j];
                    Fz = Fz + sfz[j];
                }
                Pt<T> o;
                const T f = step_point<T>(a, s, q.x, q.y, q.z, Fx, Fy, Fz, o.x, o.y, o.z);
                o.w = q.w;
                a.out[slot] = o;
                a.forces[slot] = f;
                a.nn_dist[slot] = nd;
                a.nn_id[slot] = nid;
                acc_point<T>(acc, f, nd, s, id, nid);
                // sharded sessions: what the answer rests on — the k-th neighbour, or the support ball
                const T last = sm->od2[Kq - 1];
                const T need = cs_done ? cs_lim : (Kq < K ? (last > cs_lim ? last : cs_lim) : last);
                if (reaches_past_cover<T>(a, q.x, q.y, q.z, need)) atomicAdd(a.uncovered, 1);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (MODE == 1) {
        __syncthreads();
        acc_block_reduce(acc, sm_acc);
        if (threadIdx.x == 0) acc_store(&a.partials[part_base + blockIdx.x], acc);
    }
}

// ---- RadiusTopology, wave per point (src/topology.jl:91-97) --------------------------------------
// The hash was built with cell edge >= r, so the 3^dim cell block is complete.  COUNT: ballot +
// popcount of d2 <= r*r (self removed by index).  FILL: hits go to the per-wave LDS list, each
// lane ranks its entry by the canonical (d2, index) order with a broadcast compare loop and
// writes it to its final place in the CSR row.  Rows longer than the LDS list are handed to the
// serial kernel (wtp_generic.hip) through the fb2 list.
// Round 3: (a) the nine x-rows of the block are fetched TOGETHER — bounds by lanes 0..8 in one round trip, then one load
// per row in flight at once — instead of nine dependent (bounds -> points) chains per query; (b) a lean per-wave list
// (8 or 12 bytes per entry, 256 entries: eight workgroups per CU instead of two); (c) fp32 rows are ranked on ONE 64-bit
// key per entry, (d2 bits << 32 | id) — monotone in (d2, id) because d2 >= 0 — two keys per LDS read, every lane ranking
// all the entries it owns in the same pass.  Graded 1 M-point cloud, rows of ~65: fill 3.0 -> see DESIGN.md §4.
constexpr int kRadCap = 512;  // entries a wave ranks in LDS; longer rows: the serial kernel
template <typename T> struct RadSmem;
template <> struct RadSmem<float> {
    unsigned long long key[kRadCap + 2];
};
template <> struct RadSmem<double> {
    double d2[kRadCap + 2];
    int32_t id[kRadCap + 2];
};

template <typename T, bool FILL>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 8 : 5) void wave_radius_kernel(SearchArgs<T> a, T r, int32_t* __restrict__ counts,
                                                               const int64_t* __restrict__ offsets,
                                                               int32_t* __restrict__ idx_out,
                                                               const int32_t* __restrict__ list,
                                                               const int32_t* __restrict__ list_count) {
    __shared__ RadSmem<T> sm_all[FILL ? kWaves : 1];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (uniform, and the compiler knows: the queries' chain of loads goes through the scalar unit)
    const int lane = threadIdx.x & 63;
    RadSmem<T>* sm = &sm_all[FILL ? wave : 0];
    const Grid<T> g = *a.grid;
    const T r2 = r * r; // inclusive, compared as d2 <= r*r
    const int wave_global = blockIdx.x * kWaves + wave;
    const int wave_stride = gridDim.x * kWaves;
    // list: the queries the brick kernel handed back — unless the grid says that kernel stood aside (rad_wave_only)
    const bool all = !list || (g.rad_wave_only && !a.rad_dense); // (with the dense kernel in front the list is what is left in either case)
    const int nq = all ? a.n : *list_count;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int64_t arena_used = 0; // ids this wave has parked in its share of the arena (count phase with ranking)
    // A query is a chain of dependent round trips (list -> point -> row bounds -> candidates [-> row offset]) and the CU is
    // full of such waves, so a wave's time is the sum of its chains.  The chain is software-pipelined across the wave's
    // queries: while query i is scanned, the row bounds of query i + 1, the point of query i + 2 and the list entry of
    // query i + 3 are in flight; what stays exposed per query is the candidates' round trip.
    auto slot_of = [&](int qi) { return qi < nq ? (all ? qi : list[qi]) : 0; };
    // lanes 0 .. 8: the bounds of one x-row each of the 3 x 3 rows around p (absent rows: empty)
    auto row_bounds = [&](const Pt<T>& p, bool on, int& b_ps, int& b_pe) {
        const int cx = cell_coord(g, p.x, 0), cy = cell_coord(g, p.y, 1), cz = cell_coord(g, p.z, 2);
        const int z0 = cz - 1 < 0 ? 0 : cz - 1, z1 = cz + 1 > g.n[2] - 1 ? g.n[2] - 1 : cz + 1;
        const int y0 = cy - 1 < 0 ? 0 : cy - 1, y1 = cy + 1 > g.n[1] - 1 ? g.n[1] - 1 : cy + 1;
        const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 > g.n[0] - 1 ? g.n[0] - 1 : cx + 1;
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        b_ps = 0;
        b_pe = 0;
        if (on && lane < nrows) {
            const int row = ((z0 + lane / ny) * g.n[1] + (y0 + lane % ny)) * g.n[0];
            b_ps = a.cell_start[row + x0];
            b_pe = a.cell_start[row + x1 + 1];
        }
    };
    int qi = wave_global;
    int slot0 = slot_of(qi), slot1 = slot_of(qi + wave_stride), slot2 = slot_of(qi + 2 * wave_stride);
    Pt<T> q0 = a.snap[slot0], q1 = a.snap[slot1];
    int b0_ps, b0_pe;
    row_bounds(q0, qi < nq, b0_ps, b0_pe);
    for (; qi < nq; qi += wave_stride) {
        const int slot = slot0;
        const Pt<T> q = q0;
        asm volatile("" : "+v"(b0_ps), "+v"(b0_pe)); // (what is derived from the bounds stays inside this iteration: nine row addresses carried around the loop are 18 registers)
        const int my_ps = b0_ps, my_pe = b0_pe;
        const int32_t id = w_to_id(q.w);
        // fill phase proper (counts == nullptr): a row the count phase ranked and parked in the arena is copied, not searched
        // again (such rows exist only where this kernel serves every query); the row's place in the CSR array is asked for now
        int done = 0;
        if (FILL && !counts && all && a.rad_done) done = a.rad_done[id];
        int64_t off_lo = 0, off_hi = 0;
        if (FILL && !counts) {
            off_lo = offsets[id];
            off_hi = offsets[id + 1];
        }
        // the stages behind this query move up one step
        row_bounds(q1, qi + wave_stride < nq, b0_ps, b0_pe);
        slot0 = slot1;
        q0 = q1;
        slot1 = slot2;
        q1 = a.snap[slot2];
        slot2 = slot_of(qi + 3 * wave_stride);
        if (done == 2) continue;
        int ps[9], len[9];
        Pt<T> c[9];
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            ps[rr] = __builtin_amdgcn_readlane(my_ps, rr);
            len[rr] = __builtin_amdgcn_readlane(my_pe, rr) - ps[rr];
        }
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) { // the first 64 points of every row: nine loads in flight
            // (every lane loads — those past the row's end its last point again, masked out below: a load under a lane mask
            // leaves the other lanes' registers undefined, and the compiler then carried them around the loop)
            const int last = len[rr] - 1;
            c[rr] = a.snap[ps[rr] + (len[rr] > 0 ? (lane < last ? lane : last) : 0)];
        }
        int m = 0;
        auto visit = [&](const Pt<T>& cc, bool on) {
            bool take = false;
            T d = 0;
            int32_t cid = 0;
            if (on) {
                cid = w_to_id(cc.w);
                d = dist2<T>(q.x, q.y, q.z, cc.x, cc.y, cc.z);
                take = (d <= r2) && (cid != id); // filter(!=(i), n), src/topology.jl:96
            }
            const unsigned long long mask = __ballot(take);
            if (FILL) {
                const int pos = m + __popcll(mask & below);
                if (take && pos < kRadCap) {
                    if constexpr (sizeof(T) == 4) {
                        sm->key[pos] = ((unsigned long long)__builtin_bit_cast(uint32_t, (float)d) << 32) | (uint32_t)cid;
                    } else {
                        reinterpret_cast<RadSmem<double>*>(sm)->d2[pos] = (double)d;
                        reinterpret_cast<RadSmem<double>*>(sm)->id[pos] = cid;
                    }
                }
            }
            m += __popcll(mask);
        };
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            if (len[rr] <= 0) continue; // (wave-uniform)
            visit(c[rr], lane < len[rr]);
            for (int p0 = 64; p0 < len[rr]; p0 += 64) { // rows beyond 64 points (dense clusters)
                const bool on = p0 + lane < len[rr];
                Pt<T> cc{};
                int first = ps[rr] + p0;
                asm volatile("" : "+s"(first)); // (its own address arithmetic: nine row addresses kept alive for this loop cost 18 registers)
                if (on) cc = a.snap[first + lane];
                visit(cc, on);
            }
        }
        if (!FILL) {
            if (lane == 0) counts[id] = m;
            continue;
        }
        // FILL with `counts`: the COUNT phase run with ranking — the row goes to a bump-allocated arena (its length to counts),
        // the fill phase then copies it (radius_copy_rows_kernel) instead of searching and ranking a second time
        // (only where this kernel serves every query: for the brick kernel's hand-backs ranking in the count phase costs what
        // it saves in the fill phase)
        const bool arena = counts != nullptr && all && a.rad_arena != nullptr;
        if (counts && lane == 0) counts[id] = m;
        if (counts && !arena) continue; // count phase, plain
        if (m > kRadCap) { // row longer than the LDS list: serial kernel (from the fill phase proper)
            if (!arena && lane == 0) {
                const int pos = atomicAdd(a.fb2_count, 1);
                a.fb2_list[pos] = slot;
            }
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        int64_t base, cap;
        if (arena) {
            // every wave owns an equal share of the arena and fills it from the front (its queries are strided over the list,
            // so the shares fill evenly): no atomic — one counter for a million rows serialises the whole kernel
            const int64_t share = a.rad_arena_cap / wave_stride;
            if (arena_used + m > share) continue; // share full: the fill phase searches this row again
            base = (int64_t)wave_global * share + arena_used;
            arena_used += m;
            cap = m;
            idx_out = a.rad_arena;
            if (lane == 0) {
                a.rad_arena_off[id] = base;
                a.rad_done[id] = 2;
            }
        } else {
            base = off_lo;
            cap = off_hi - off_lo;
        }
        if constexpr (sizeof(T) == 4) {
            // every lane ranks the (up to four) entries it owns against all m keys, two keys per LDS read
            if (lane == 0) sm->key[m] = ~0ull; // an odd m reads one key past the end
            __builtin_amdgcn_wave_barrier();
            // (two owned entries per pass over the keys: rows of up to 128 entries take one pass, and the registers stay few)
#pragma unroll 1
            for (int o0 = 0; o0 < m; o0 += 128) {
                const int e0 = o0 + lane, e1 = o0 + 64 + lane;
                const unsigned long long mine0 = e0 < m ? sm->key[e0] : 0ull, mine1 = e1 < m ? sm->key[e1] : 0ull;
                int rank0 = 0, rank1 = 0;
                for (int j = 0; j < m; j += 2) {
                    const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&sm->key[j]);
                    rank0 += (kk.x < mine0 ? 1 : 0) + (kk.y < mine0 ? 1 : 0);
                    rank1 += (kk.x < mine1 ? 1 : 0) + (kk.y < mine1 ? 1 : 0);
                }
                if (e0 < m && rank0 < cap) idx_out[base + rank0] = (int32_t)(uint32_t)mine0;
                if (e1 < m && rank1 < cap) idx_out[base + rank1] = (int32_t)(uint32_t)mine1;
            }
        } else {
            const RadSmem<double>* sd = reinterpret_cast<const RadSmem<double>*>(sm);
            for (int i = lane; i < m; i += 64) {
                const T md = (T)sd->d2[i];
                const int32_t mi = sd->id[i];
                int rank = 0;
                for (int j = 0; j < m; ++j) rank += lex_lt((T)sd->d2[j], sd->id[j], md, mi) ? 1 : 0;
                if (rank < cap) idx_out[base + rank] = mi;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int MODE> static size_t wave_smem() { return sizeof(WaveSmem<T, MODE == 1>) * kWaves + sizeof(Acc) * kWaves; }

template <typename T>
int launch_wave_radius_count(wtp_ctx* ctx, SearchArgs<T>& a, T r, int32_t* d_counts, const int32_t* list,
                             const int32_t* list_count) {
    int64_t want = ((int64_t)a.n / (list ? 16 : 1) + kWaves - 1) / kWaves;
    int nb = (int)(want > 16384 ? 16384 : (want < 64 ? 64 : want));
    if (a.rad_arena && a.rad_done && !list) // count WITH ranking where this kernel serves every query by construction: rows parked for the fill phase
        hipLaunchKernelGGL((wave_radius_kernel<T, true>), dim3(nb), dim3(kThreads), 0, ctx->stream, a, r, d_counts,
                           (const int64_t*)nullptr, (int32_t*)nullptr, list, list_count);
    else
        hipLaunchKernelGGL((wave_radius_kernel<T, false>), dim3(nb), dim3(kThreads), 0, ctx->stream, a, r,
                           d_counts, (const int64_t*)nullptr, (int32_t*)nullptr, list, list_count);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T>
int launch_wave_radius_fill(wtp_ctx* ctx, SearchArgs<T>& a, T r, const int64_t* d_offsets, int32_t* d_idx,
                            const int32_t* list, const int32_t* list_count) {
    int64_t want = ((int64_t)a.n / (list ? 16 : 1) + kWaves - 1) / kWaves;
    int nb = (int)(want > 16384 ? 16384 : (want < 64 ? 64 : want));
    hipLaunchKernelGGL((wave_radius_kernel<T, true>), dim3(nb), dim3(kThreads), 0, ctx->stream, a, r,
                       (int32_t*)nullptr, d_offsets, d_idx, list, list_count);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T, int MODE> static int launch_wave(wtp_ctx* ctx, SearchArgs<T>& a, bool all, int part_base) {
    (void)launch_occupancy_of(ctx, (const void*)wave_kernel<T, MODE>, kThreads, (wave_smem<T, MODE>()));
    // hand-back lists are a small fraction of the cloud: size the grid by the cloud, not by the chip
    // (an idle block still pays its reduction and its partial: 27 us per step at 47 k points with 2048)
    int64_t want = all ? ((int64_t)a.n + kWaves - 1) / kWaves : (int64_t)a.n / 256;
    if (!all && want > 2048) want = 2048;
    if (!all && want < 64) want = 64;
    int nb = (int)(want > kWavePartials ? kWavePartials : (want < 1 ? 1 : want));
    if (MODE == 1) a.used_wave = nb;
    const size_t smem_bytes = wave_smem<T, MODE>();
    hipLaunchKernelGGL((wave_kernel<T, MODE>), dim3(nb), dim3(kThreads), smem_bytes, ctx->stream, a, a.fb_list,
                       a.fb_count, all ? 1 : 0, part_base);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T> int launch_wave_topology(wtp_ctx* ctx, SearchArgs<T>& a, bool all) {
    return launch_wave<T, 0>(ctx, a, all, 0);
}

template <typename T> int launch_wave_sweep(wtp_ctx* ctx, SearchArgs<T>& a, bool all) {
    return launch_wave<T, 1>(ctx, a, all, brick_partials());
}

#define INST(T)                                                                                  \
    template int launch_wave_topology<T>(wtp_ctx*, SearchArgs<T>&, bool);                        \
    template int launch_wave_radius_count<T>(wtp_ctx*, SearchArgs<T>&, T, int32_t*, const int32_t*, const int32_t*); \
    template int launch_wave_radius_fill<T>(wtp_ctx*, SearchArgs<T>&, T, const int64_t*, int32_t*, const int32_t*, const int32_t*); \
    template int launch_wave_sweep<T>(wtp_ctx*, SearchArgs<T>&, bool);
INST(float)
INST(double)
#undef INST

} // namespace wtp
