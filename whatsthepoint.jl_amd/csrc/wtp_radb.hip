// RadiusTopology for the dense part of a cloud (src/topology.jl:91-100): brick-staged, one wave per query.
//
// The wave-per-query kernel (wtp_wave.hip) fetches the 27 cells of every query from global memory — at 15 points per
// cell that is 6.7 KB per query, and a 10 M-point graded cloud moves 40 GB through the L2 for a 1.7 GB answer.  The
// lane-per-query brick kernel (wtp_brick.hip, MODE 2) stages a brick once but keeps a row in 32 registers.  This kernel
// is the piece between them: a brick of cells and its one-cell halo are staged in LDS once (3.4 points fetched per
// query instead of 420), then each wave of the workgroup takes the brick's queries one at a time — nine contiguous LDS
// runs scanned 64 candidates per step, hits compacted into the wave's list, ranked by (d2, index) and parked in the
// arena the fill phase copies from (wtp_generic.hip: radius_copy_rows_kernel).  Same cells, same d2 expression, same
// order as the wave kernel: bit-identical rows.
//
// It runs in the COUNT phase only, and only when that phase parks its rows (SearchArgs::rad_done / rad_arena).  fp32:
// bricks whose halo holds more than kRadDenseMin points (the brick kernel stands aside for exactly those, same
// number); fp64 and grids flagged rad_wave_only: every brick.  What does not fit — a brick beyond the LDS area, a row
// beyond the wave's list, an arena that is full — goes to the hand-back list of the wave kernel, as before.
#include "wtp_device.hpp"
#include "wtp_internal.hpp"

namespace wtp {

constexpr int kRdThreads = 1024, kRdWaves = kRdThreads / 64;
constexpr int kRdCap = 128;     // entries a wave ranks (two per lane); longer rows: the wave kernel
constexpr int kRdChunk = 2048;  // ids a wave takes from the arena at a time (one atomic per ~30 rows)

template <typename T> struct RdGeom;
template <> struct RdGeom<float> {
    static constexpr int bx = 4, by = 4, bz = 4;
};
template <> struct RdGeom<double> { // 32-byte points: a smaller brick keeps the dense cells inside LDS
    static constexpr int bx = 4, by = 2, bz = 2;
};

template <typename T> struct RdList;
template <> struct RdList<float> {
    unsigned long long key[kRdCap + 2]; // (d2 bits << 32 | id): monotone in (d2, id) because d2 >= 0
};
template <> struct RdList<double> {
    double d2[kRdCap + 2];
    int32_t id[kRdCap + 2];
};

template <typename T> struct RdTables {
    static constexpr int hx = RdGeom<T>::bx + 2, hy = RdGeom<T>::by + 2, hz = RdGeom<T>::bz + 2;
    static constexpr int hcells = hx * hy * hz, own_rows = RdGeom<T>::by * RdGeom<T>::bz;
    int hstart[hcells + 1]; // LDS slot of the first point of each halo cell
    int hglobal[hcells];    // global (sorted) index of the first point of each halo cell
    int own_pref[own_rows + 1];
    int scan_tmp[kRdWaves + 1];
    int push_base;
};

template <typename T> static size_t rd_smem_bytes(int hcap) {
    return (size_t)hcap * sizeof(Pt<T>) + sizeof(RdTables<T>) + 16 + sizeof(RdList<T>) * kRdWaves;
}

// The row in the wave's list -> canonical order (d2, index), written to out[0 .. m).  Counting, for each entry, the keys
// below it is m^2 / 64 compare steps per lane and was 40 % of the kernel; the entries are first grouped into eight shells
// by a monotone function of d2 (equal steps of d2 / r^2), in place, through the registers.  An entry then ranks itself
// against the keys of its own shell only: what lies in earlier shells is below it by construction (that count is the
// shell's first slot), what lies in later ones — and the two sentinels past the end — is above it and never counts, so
// the loop needs no bounds test.
template <typename T>
__device__ inline void rd_rank_store(RdList<T>* lst, int m, int lane, unsigned long long below, T r2, int32_t* __restrict__ out) {
    const int e0 = lane, e1 = lane + 64;
    const bool two = m > 64; // (wave-uniform) rows beyond 64 entries: two per lane
    unsigned long long kd0 = 0, kd1 = 0; // fp32: (d2 bits << 32 | id); fp64: the bits of d2 (d2 >= 0: monotone as an integer)
    int32_t id0 = 0, id1 = 0;
    float f0 = 0.f, f1 = 0.f;
    if constexpr (sizeof(T) == 4) {
        RdList<float>* l = reinterpret_cast<RdList<float>*>(lst);
        if (e0 < m) kd0 = l->key[e0];
        if (two && e1 < m) kd1 = l->key[e1];
        id0 = (int32_t)(uint32_t)kd0;
        id1 = (int32_t)(uint32_t)kd1;
        f0 = __builtin_bit_cast(float, (uint32_t)(kd0 >> 32));
        f1 = __builtin_bit_cast(float, (uint32_t)(kd1 >> 32));
    } else {
        RdList<double>* l = reinterpret_cast<RdList<double>*>(lst);
        if (e0 < m) {
            kd0 = __builtin_bit_cast(unsigned long long, l->d2[e0]);
            id0 = l->id[e0];
        }
        if (two && e1 < m) {
            kd1 = __builtin_bit_cast(unsigned long long, l->d2[e1]);
            id1 = l->id[e1];
        }
        f0 = (float)__builtin_bit_cast(double, kd0);
        f1 = (float)__builtin_bit_cast(double, kd1);
    }
    auto less = [&](unsigned long long kj, int32_t ij, unsigned long long km, int32_t im) {
        if constexpr (sizeof(T) == 4) return kj < km;
        else return kj < km || (kj == km && ij < im);
    };
    auto read = [&](int i, unsigned long long& k, int32_t& ii) {
        if constexpr (sizeof(T) == 4) {
            k = reinterpret_cast<const RdList<float>*>(lst)->key[i];
            ii = 0;
        } else {
            k = __builtin_bit_cast(unsigned long long, reinterpret_cast<const RdList<double>*>(lst)->d2[i]);
            ii = reinterpret_cast<const RdList<double>*>(lst)->id[i];
        }
    };
    auto write = [&](int i, unsigned long long k, int32_t ii) {
        if constexpr (sizeof(T) == 4) {
            reinterpret_cast<RdList<float>*>(lst)->key[i] = k;
        } else {
            reinterpret_cast<RdList<double>*>(lst)->d2[i] = __builtin_bit_cast(double, k);
            reinterpret_cast<RdList<double>*>(lst)->id[i] = ii;
        }
    };
    int ss0 = 0, ss1 = 0, span = m; // first slot of the entry's shell; the longest shell
    if (m > 16) {
        const float inv8 = (float)r2 > 0.f ? 8.f / (float)r2 : 0.f;
        const int k0 = e0 < m ? (int)fminf(f0 * inv8, 7.f) : 8, k1 = (two && e1 < m) ? (int)fminf(f1 * inv8, 7.f) : 8;
        int first = 0, g0 = 0, g1 = 0;
        span = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long b0 = __ballot(k0 == k), b1 = two ? __ballot(k1 == k) : 0ull;
            const int c0 = __popcll(b0), c = c0 + __popcll(b1);
            if (k0 == k) {
                g0 = first + __popcll(b0 & below);
                ss0 = first;
            }
            if (k1 == k) {
                g1 = first + c0 + __popcll(b1 & below);
                ss1 = first;
            }
            first += c;
            span = c > span ? c : span;
        }
        __builtin_amdgcn_wave_barrier(); // (every entry is in a register: the list is rewritten in place)
        if (e0 < m) write(g0, kd0, id0);
        if (two && e1 < m) write(g1, kd1, id1);
    }
    if (lane < 2) write(m + lane, ~0ull, 0x7fffffff); // two sentinels: the loop reads pairs and runs past the entry's shell
    __builtin_amdgcn_wave_barrier();
    int rank0 = ss0, rank1 = ss1;
    for (int t = 0; t < span; t += 2) {
        unsigned long long ka, kb;
        int32_t ia, ib;
        const int i0 = ss0 + t < m ? ss0 + t : m;
        read(i0, ka, ia);
        read(i0 + 1, kb, ib);
        rank0 += (less(ka, ia, kd0, id0) ? 1 : 0) + (less(kb, ib, kd0, id0) ? 1 : 0);
        if (two) {
            const int i1 = ss1 + t < m ? ss1 + t : m;
            read(i1, ka, ia);
            read(i1 + 1, kb, ib);
            rank1 += (less(ka, ia, kd1, id1) ? 1 : 0) + (less(kb, ib, kd1, id1) ? 1 : 0);
        }
    }
    if (e0 < m) out[rank0] = id0;
    if (two && e1 < m) out[rank1] = id1;
}

template <typename T>
__global__ __launch_bounds__(kRdThreads, 8) void rad_dense_kernel(SearchArgs<T> a, T r, int32_t* __restrict__ counts, int hcap,
                                                                  const int32_t* __restrict__ bricks,
                                                                  const int32_t* __restrict__ n_bricks) {
    using G = RdGeom<T>;
    using Tb = RdTables<T>;
    constexpr int HXc = Tb::hx, HYc = Tb::hy, HZc = Tb::hz, HC = Tb::hcells, OR = Tb::own_rows;
    extern __shared__ __attribute__((aligned(16))) unsigned char rd_smem[];
    Pt<T>* pts = reinterpret_cast<Pt<T>*>(rd_smem);
    Tb* sm = reinterpret_cast<Tb*>(rd_smem + (size_t)hcap * sizeof(Pt<T>));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RdList<T>* lst = reinterpret_cast<RdList<T>*>(rd_smem + (size_t)hcap * sizeof(Pt<T>) + ((sizeof(Tb) + 15) & ~(size_t)15)) + wave;

    const Grid<T> g = *a.grid;
    const int dense_min = (sizeof(T) == 8 || g.rad_wave_only) ? -1 : kRadDenseMin; // fp32: below it the brick kernel has served the brick
    const T r2 = r * r; // inclusive, compared as d2 <= r*r
    const int nbx = (g.n[0] + G::bx - 1) / G::bx, nby = (g.n[1] + G::by - 1) / G::by;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int64_t chunk_base = 0;
    int chunk_left = 0; // ids left in the wave's current piece of the arena

    // The bricks this kernel serves were listed by rad_census_kernel (a graded cloud's grid is mostly empty cells: 5 M bricks
    // of which a few thousand are dense; looking at each from here cost 69 ms).  They are dealt round-robin: dense bricks are
    // neighbours in space and in the list.
    const int nlist = *n_bricks;
    for (int bi = blockIdx.x; bi < nlist; bi += gridDim.x) {
        const int brick = bricks[bi];
        const int bxi = brick % nbx, byi = (brick / nbx) % nby, bzi = brick / (nbx * nby);
        const int ox = bxi * G::bx - 1, oy = byi * G::by - 1, oz = bzi * G::bz - 1; // halo origin (cell coordinates)
        __syncthreads(); // the previous brick's LDS is no longer in use
        // ---- 1. halo cell table: global start and count of every cell, exclusive scan = LDS slots --------------------
        int my_cnt = 0;
        if (tid < HC) {
            const int hx = tid % HXc, hy = (tid / HXc) % HYc, hz = tid / (HXc * HYc);
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
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            if (lane == 63) sm->scan_tmp[wave] = incl;
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int w = 0; w < kRdWaves; ++w) {
                    const int t = sm->scan_tmp[w];
                    sm->scan_tmp[w] = run;
                    run += t;
                }
                sm->scan_tmp[kRdWaves] = run;
            }
            __syncthreads();
            if (tid < HC) sm->hstart[tid] = incl - my_cnt + sm->scan_tmp[wave];
            if (tid == HC) sm->hstart[HC] = sm->scan_tmp[kRdWaves];
        }
        __syncthreads();
        const int halo_total = sm->hstart[HC];
        if (halo_total <= dense_min) continue; // (uniform) the lane-per-query kernel's brick
        // ---- 2. own rows: the queries are the points of the brick's own cells -----------------------------------------
        if (tid == 0) {
            int run = 0;
            for (int rr = 0; rr < OR; ++rr) {
                const int base = ((1 + rr / G::by) * HYc + (1 + rr % G::by)) * HXc;
                sm->own_pref[rr] = run;
                run += sm->hstart[base + 1 + G::bx] - sm->hstart[base + 1];
            }
            sm->own_pref[OR] = run;
        }
        __syncthreads();
        const int Q = sm->own_pref[OR];
        if (Q == 0) continue; // (uniform)
        if (tid == 0 && halo_total <= hcap) atomicAdd((int32_t*)a.rad_arena_pos + 3, Q); // (diagnostics: queries served here, WTP_DEBUG prints it)
        if (halo_total > hcap) { // does not fit the LDS area: the wave kernel takes the brick's queries
            if (dense_min >= 0) continue; // (the brick kernel has handed them back already)
            if (tid == 0) sm->push_base = atomicAdd(a.fb_count, Q);
            __syncthreads();
            const int pb = sm->push_base;
            for (int q = tid; q < Q; q += kRdThreads) {
                int rr = 0;
#pragma unroll
                for (int t = 1; t < OR; ++t) rr += (sm->own_pref[t] <= q) ? 1 : 0;
                const int rbase = ((1 + rr / G::by) * HYc + (1 + rr % G::by)) * HXc + 1;
                a.fb_list[pb + q] = sm->hglobal[rbase] + (q - sm->own_pref[rr]);
            }
            continue;
        }
        // ---- 3. stage the halo: each x-row of cells is one contiguous run, in global memory and in LDS ------------------
        {
            const int hx_lo = ox < 0 ? -ox : 0; // first halo column inside the grid
            constexpr int NR = HYc * HZc, KR = (NR + kRdWaves - 1) / kRdWaves;
            int ls[KR], len[KR], gs[KR];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int row = wave + j * kRdWaves;
                const bool ok = row < NR;
                const int base = (ok ? row : 0) * HXc;
                ls[j] = sm->hstart[base];
                len[j] = ok ? sm->hstart[base + HXc] - ls[j] : 0;
                gs[j] = sm->hglobal[base + hx_lo];
            }
            Pt<T> v[KR];
#pragma unroll
            for (int j = 0; j < KR; ++j) { // the first 64 points of the wave's rows: all loads in flight together (clamped, unconditional)
                const int src = gs[j] + (lane < len[j] ? lane : 0);
                v[j] = a.snap[src < a.n ? src : a.n - 1];
            }
#pragma unroll
            for (int j = 0; j < KR; ++j)
                if (lane < len[j]) pts[ls[j] + lane] = v[j];
#pragma unroll
            for (int j = 0; j < KR; ++j)
                for (int i = lane + 64; i < len[j]; i += 64) pts[ls[j] + i] = a.snap[gs[j] + i];
        }
        __syncthreads();
        // ---- 4. one wave per query ---------------------------------------------------------------------------------------
        // (a wave takes CONSECUTIVE queries: their rows then sit next to each other in its piece of the arena, in the order the
        // copy pass reads them)
        const int per_wave = (Q + kRdWaves - 1) / kRdWaves;
        const int q_end = (wave + 1) * per_wave < Q ? (wave + 1) * per_wave : Q;
        for (int q = wave * per_wave; q < q_end; ++q) {
            int rr;
            {
                const bool le = lane >= 1 && lane < OR && sm->own_pref[lane < OR ? lane : 0] <= q;
                rr = __popcll(__ballot(le));
            }
            const int hy0 = 1 + rr % G::by, hz0 = 1 + rr / G::by;
            const int rbase = (hz0 * HYc + hy0) * HXc + 1;
            const int off = q - sm->own_pref[rr];
            const int gslot = sm->hglobal[rbase] + off;
            const Pt<T> qp = pts[sm->hstart[rbase] + off];
            const int32_t id = w_to_id(qp.w);
            const int hxq = cell_coord(g, qp.x, 0) - ox;
            // lanes 0 .. 8: the LDS run of one row of three cells each
            int my_s = 0, my_e = 0;
            if (lane < 9) {
                const int b = ((hz0 + lane / 3 - 1) * HYc + (hy0 + lane % 3 - 1)) * HXc + hxq - 1;
                my_s = sm->hstart[b];
                my_e = sm->hstart[b + 3];
            }
            int m = 0;
            auto visit = [&](int p, bool on) {
                bool take = false;
                T d = 0;
                int32_t cid = 0;
                if (on) {
                    const Pt<T> cc = pts[p];
                    cid = w_to_id(cc.w);
                    d = dist2<T>(qp.x, qp.y, qp.z, cc.x, cc.y, cc.z);
                    take = (d <= r2) && (cid != id); // filter(!=(i), n), src/topology.jl:96
                }
                const unsigned long long mask = __ballot(take);
                const int pos = m + __popcll(mask & below);
                if (take && pos < kRdCap) {
                    if constexpr (sizeof(T) == 4) {
                        reinterpret_cast<RdList<float>*>(lst)->key[pos] =
                            ((unsigned long long)__builtin_bit_cast(uint32_t, (float)d) << 32) | (uint32_t)cid;
                    } else {
                        reinterpret_cast<RdList<double>*>(lst)->d2[pos] = (double)d;
                        reinterpret_cast<RdList<double>*>(lst)->id[pos] = cid;
                    }
                }
                m += __popcll(mask);
            };
            // candidates of the nine runs together: prefix of the run lengths over lanes 0 .. 8
            int incl = my_e - my_s;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            const int total = __builtin_amdgcn_readlane(incl, 8);
            if (total <= 128) {
                // few candidates (the sparse part of the cloud, where this kernel serves every brick): the nine runs as ONE
                // sequence — one or two steps of 64 instead of nine mostly empty ones; lane -> run by a compare chain
                const int ex = incl - (my_e - my_s), delta = my_s - ex; // slot = position in the sequence + delta of its run
                int pj[9], dj[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    pj[j] = __builtin_amdgcn_readlane(ex, j);
                    dj[j] = __builtin_amdgcn_readlane(delta, j);
                }
                for (int b = 0; b < total; b += 64) {
                    const int idx = b + lane;
                    int p = idx + dj[0];
#pragma unroll
                    for (int j = 1; j < 9; ++j) p = idx >= pj[j] ? idx + dj[j] : p;
                    visit(p, idx < total);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int s0 = __builtin_amdgcn_readlane(my_s, j), s1 = __builtin_amdgcn_readlane(my_e, j);
                    for (int p0 = s0; p0 < s1; p0 += 64) visit(p0 + lane, p0 + lane < s1);
                }
            }
            if (lane == 0) counts[id] = m;
            // a place for the row: from the wave's piece of the arena, a new piece when it is used up
            bool park = m <= kRdCap;
            if (park && m > chunk_left) {
                unsigned long long nb = 0;
                if (lane == 0) nb = atomicAdd(a.rad_arena_pos, (unsigned long long)kRdChunk);
                chunk_base = (int64_t)__shfl(nb, 0, 64);
                chunk_left = kRdChunk;
                if (chunk_base + kRdChunk > a.rad_arena_cap) { // arena used up: this and every later row of the wave is handed back
                    chunk_left = 0;
                    park = false;
                }
            }
            if (!park) {
                if (lane == 0) a.fb_list[atomicAdd(a.fb_count, 1)] = gslot;
                continue;
            }
            const int64_t base = chunk_base;
            chunk_base += m;
            chunk_left -= m;
            __builtin_amdgcn_wave_barrier();
            rd_rank_store<T>(lst, m, lane, below, r2, a.rad_arena + base);
            if (lane == 0) {
                a.rad_arena_off[id] = base;
                a.rad_done[id] = 2;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// One thread per brick: does the dense kernel have anything to do there?  Own points (the queries) and halo points from the
// row ends in cell_start; the bricks that qualify are appended to the list (order does not matter: rows land at their own
// offsets whoever ranks them).
template <typename T>
__global__ void rad_census_kernel(SearchArgs<T> a, int hcap, int32_t* __restrict__ bricks, int32_t* __restrict__ n_bricks) {
    using G = RdGeom<T>;
    const Grid<T> g = *a.grid;
    const int dense_min = (sizeof(T) == 8 || g.rad_wave_only) ? -1 : kRadDenseMin;
    const int nbx = (g.n[0] + G::bx - 1) / G::bx, nby = (g.n[1] + G::by - 1) / G::by, nbz = (g.n[2] + G::bz - 1) / G::bz;
    const int nbricks = nbx * nby * nbz;
    const int lane = threadIdx.x & 63;
    const int span = (nbricks + 63) / 64 * 64; // whole waves
    for (int brick = blockIdx.x * blockDim.x + threadIdx.x; brick < span; brick += gridDim.x * blockDim.x) {
        bool want = false;
        if (brick < nbricks) {
            const int bxi = brick % nbx, byi = (brick / nbx) % nby, bzi = brick / (nbx * nby);
            const int x0 = bxi * G::bx, x1 = x0 + G::bx < g.n[0] ? x0 + G::bx : g.n[0];
            int own = 0;
            for (int rr = 0; rr < G::by * G::bz; ++rr) {
                const int gy = byi * G::by + rr % G::by, gz = bzi * G::bz + rr / G::by;
                if (gy < g.n[1] && gz < g.n[2]) {
                    const int row = (gz * g.n[1] + gy) * g.n[0];
                    own += a.cell_start[row + x1] - a.cell_start[row + x0];
                }
            }
            if (own > 0) {
                const int hx0 = x0 - 1 < 0 ? 0 : x0 - 1, hx1 = x0 + G::bx + 1 < g.n[0] ? x0 + G::bx + 1 : g.n[0];
                int halo = 0;
                for (int rr = 0; rr < (G::by + 2) * (G::bz + 2); ++rr) {
                    const int gy = byi * G::by - 1 + rr % (G::by + 2), gz = bzi * G::bz - 1 + rr / (G::by + 2);
                    if (gy >= 0 && gy < g.n[1] && gz >= 0 && gz < g.n[2]) {
                        const int row = (gz * g.n[1] + gy) * g.n[0];
                        halo += a.cell_start[row + hx1] - a.cell_start[row + hx0];
                    }
                }
                want = halo > dense_min && !(halo > hcap && dense_min >= 0); // (fp32: beyond both kernels' LDS the brick kernel has handed the queries back)
            }
        }
        const unsigned long long m = __ballot(want);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(n_bricks, __popcll(m));
            base = __shfl(base, 0, 64);
            if (want) bricks[base + __popcll(m & ((1ull << lane) - 1ull))] = brick;
        }
    }
}

// LDS points per workgroup: two workgroups per CU
template <typename T> static int rd_hcap() {
    const size_t budget = 80 * 1024 - sizeof(RdTables<T>) - 32 - sizeof(RdList<T>) * kRdWaves;
    return (int)(budget / sizeof(Pt<T>)) / 64 * 64;
}

template <typename T> int launch_radius_dense(wtp_ctx* ctx, SearchArgs<T>& a, T r, int32_t* d_counts) {
    const int hcap = rd_hcap<T>();
    const size_t smem = rd_smem_bytes<T>(hcap);
    if (!ctx->rad_dense_attr[sizeof(T) == 8]) { // (once per context: the call is not free, and the uniform clouds' rows never reach this kernel)
        WTP_HIP(ctx, hipFuncSetAttribute((const void*)rad_dense_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        ctx->rad_dense_attr[sizeof(T) == 8] = true;
    }
    a.rad_dense = hcap;
    int32_t* n_bricks = (int32_t*)a.rad_arena_pos + 2; // (the counter block the caller cleared: [0, 8) the arena's next free id)
    hipLaunchKernelGGL(rad_census_kernel<T>, dim3(2048), dim3(256), 0, ctx->stream, a, hcap, a.rad_bricks, n_bricks);
    hipLaunchKernelGGL(rad_dense_kernel<T>, dim3(2 * 256), dim3(kRdThreads), smem, ctx->stream, a, r, d_counts, hcap,
                       (const int32_t*)a.rad_bricks, (const int32_t*)n_bricks);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}
template int launch_radius_dense<float>(wtp_ctx*, SearchArgs<float>&, float, int32_t*);
template int launch_radius_dense<double>(wtp_ctx*, SearchArgs<double>&, double, int32_t*);

template <typename T> int radius_dense_hcap() { return rd_hcap<T>(); }
template int radius_dense_hcap<float>();
template int radius_dense_hcap<double>();

} // namespace wtp
