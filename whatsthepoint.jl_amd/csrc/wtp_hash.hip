// wtp_hash.hip — counting-sort spatial hash over packed {x,y,z,id} points (gfx950).
//
// Replaces the kd-tree build of the reference (KDTree(coords), src/repel.jl:218,252; the
// KNearestSearch constructor behind src/topology.jl:80).  Passes (algorithmic bytes per point
// in brackets, fp32):
//   bbox        read Pt                                   [16]  (fused into setup on rebuilds)
//   cell_rank   read Pt, atomic count, write cell+rank    [16 + 8]
//   scan        exclusive scan of per-cell counts         [~1]
//   scatter     read Pt + cell/rank, write sorted Pt      [16 + 8 + 16]
//   canon       per-cell order by id (deterministic runs) [in L2]
// All sizes that depend on the data (cells, bricks) live in the device-resident Grid, so a
// rebuild needs no host round trip.
#include "wtp_device.hpp"

namespace wtp {

static constexpr int kThreads = 256;

template <typename T>
__global__ void load_points_kernel(const T* __restrict__ xyz, Pt<T>* __restrict__ out, int64_t n, int dim) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Pt<T> p;
        p.x = xyz[i * dim + 0];
        p.y = xyz[i * dim + 1];
        p.z = dim == 3 ? xyz[i * dim + 2] : (T)0;
        p.w = id_to_w((T)0, (int32_t)i);
        out[i] = p;
    }
}

template <typename T>
__global__ void bbox_kernel(const Pt<T>* __restrict__ pts, int64_t n, T* __restrict__ part, int64_t v_old = 0,
                            int32_t v_fixed_old = 0) {
    __shared__ T sm[6][kThreads / 64];
    T mn[3] = {Lim<T>::inf(), Lim<T>::inf(), Lim<T>::inf()};
    T mx[3] = {-Lim<T>::inf(), -Lim<T>::inf(), -Lim<T>::inf()};
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Pt<T> p = pts[i];
        if (i < v_old && w_to_id(p.w) < v_fixed_old) continue; // stale fixed point of the input view
        // non-finite coordinates (x - x != 0) do not shape the box; such points end up in edge cells
        if (!(p.x - p.x == (T)0)) p.x = mn[0] < mx[0] ? mn[0] : (T)0;
        if (!(p.y - p.y == (T)0)) p.y = mn[1] < mx[1] ? mn[1] : (T)0;
        if (!(p.z - p.z == (T)0)) p.z = mn[2] < mx[2] ? mn[2] : (T)0;
        mn[0] = p.x < mn[0] ? p.x : mn[0];
        mx[0] = p.x > mx[0] ? p.x : mx[0];
        mn[1] = p.y < mn[1] ? p.y : mn[1];
        mx[1] = p.y > mx[1] ? p.y : mx[1];
        mn[2] = p.z < mn[2] ? p.z : mn[2];
        mx[2] = p.z > mx[2] ? p.z : mx[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) {
            T o = __shfl_down(mn[a], d, 64);
            mn[a] = o < mn[a] ? o : mn[a];
            o = __shfl_down(mx[a], d, 64);
            mx[a] = o > mx[a] ? o : mx[a];
        }
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) {
            sm[a][wave] = mn[a];
            sm[3 + a][wave] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            T lo = sm[a][0], hi = sm[3 + a][0];
            for (int w = 1; w < kThreads / 64; ++w) {
                lo = sm[a][w] < lo ? sm[a][w] : lo;
                hi = sm[3 + a][w] > hi ? sm[3 + a][w] : hi;
            }
            part[blockIdx.x * 6 + a] = lo;
            part[blockIdx.x * 6 + 3 + a] = hi;
        }
    }
}

// One thread: final bbox reduce + grid parameters.  rho_k = target points per cell.
template <typename T>
__global__ void grid_setup_kernel(const T* __restrict__ part, int nparts, Grid<T>* __restrict__ g,
                                  int64_t npts, int dim, double rho_k, double radius, double min_cell, int cell_cap,
                                  double cell_scale, const double* __restrict__ box, const int32_t* __restrict__ stop) {
    if (stop && *stop) return; // a stop rule fired earlier in this batch: the state of that iteration stays as it is
    // one wave: lanes stride over the per-block partials, shuffle-reduce, lane 0 does the setup
    double mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        T lo = Lim<T>::inf(), hi = -Lim<T>::inf();
        for (int b = threadIdx.x; b < nparts; b += 64) {
            lo = part[b * 6 + a] < lo ? part[b * 6 + a] : lo;
            hi = part[b * 6 + 3 + a] > hi ? part[b * 6 + 3 + a] : hi;
        }
        for (int d = 32; d >= 1; d >>= 1) {
            T o = __shfl_down(lo, d, 64);
            lo = o < lo ? o : lo;
            o = __shfl_down(hi, d, 64);
            hi = o > hi ? o : hi;
        }
        mn[a] = (double)lo;
        mx[a] = (double)hi;
        // robust box (outliers): the grid covers the bulk only; whatever lies outside is clamped into
        // the edge cells, which the search treats as unbounded outward
        if (box) {
            mn[a] = box[a] > mn[a] ? box[a] : mn[a];
            mx[a] = box[3 + a] < mx[a] ? box[3 + a] : mx[a];
        }
    }
    if (threadIdx.x != 0) return;
    double ext[3], emax = 0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = a < dim ? mx[a] - mn[a] : 0.0;
        if (!(ext[a] >= 0)) ext[a] = 0; // NaN guard
        emax = ext[a] > emax ? ext[a] : emax;
    }
    double c;
    if (emax <= 0) {
        c = 1.0;
    } else {
        double vol = 1.0;
        for (int a = 0; a < dim; ++a) vol *= (ext[a] > emax * 1e-6 ? ext[a] : emax * 1e-6);
        // cell_scale < 1: the caller measured that the occupied cells hold more points than the box
        // average says (graded clouds, surfaces, outliers stretching the box) and shrinks the edge
        c = pow(rho_k * vol / (double)npts, 1.0 / (double)dim) * cell_scale;
        if (radius > 0) {
            double cr = radius * (1.0 + 1.0 / 64.0); // c - margin >= radius
            double cc = pow(2.0 * vol / (double)npts, 1.0 / (double)dim) * cell_scale;
            c = cr > cc ? cr : cc;
        }
        if (radius <= 0 && min_cell > c) c = min_cell; // caller's floor on the cell edge
        if (!(c > 0)) c = emax;
    }
    int nn[3];
    for (int it = 0; it < 400; ++it) {
        double cells = 1;
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            double f = a < dim ? floor(ext[a] / c) + 1.0 : 1.0;
            if (f > (double)kMaxAxisCells) ok = false;
            nn[a] = f > (double)kMaxAxisCells ? kMaxAxisCells : (int)f;
            cells *= (double)nn[a];
        }
        if (ok && cells <= (double)cell_cap) break;
        c *= 1.08;
    }
    // round the cell edge to T once; every kernel uses these exact values
    T cT = (T)c;
    g->c = cT;
    g->inv_c = (T)1 / cT;
    g->margin = cT * (T)(1.0 / 256.0);
    int64_t cells = 1;
    for (int a = 0; a < 3; ++a) {
        g->org[a] = (T)mn[a];
        g->n[a] = nn[a];
        cells *= nn[a];
    }
    g->ncells = (int32_t)cells;
    g->nb[0] = (nn[0] + BX - 1) / BX;
    g->nb[1] = (nn[1] + BY - 1) / BY;
    g->nb[2] = (nn[2] + BZ - 1) / BZ;
    g->nbricks = g->nb[0] * g->nb[1] * g->nb[2];
    g->dim = dim;
    g->npts = (int32_t)npts;
    // RadiusTopology: expected row length at the box-average density.  Rows beyond 32 entries are handed back by the brick
    // kernel one by one after it has staged and scanned them; past ~30 expected neighbours (a graded cloud's average of 40
    // hides a dense part at 65) the wave kernel alone is faster (measured at 1 M points: 20.6 per row 0.64 vs 1.40 ms,
    // 40 graded 2.07 vs 1.82, 68.9 per row 3.07 vs 2.64)
    g->rad_wave_only = 0;
    if (radius > 0 && emax > 0) {
        double vol = 1.0;
        for (int a = 0; a < dim; ++a) vol *= (ext[a] > emax * 1e-6 ? ext[a] : emax * 1e-6);
        const double ball = dim == 3 ? 4.18879 * radius * radius * radius : 3.14159265 * radius * radius;
        g->rad_wave_only = (double)npts / vol * ball > 30.0 ? 1 : 0;
    }
    g->pad_ = 0;
}

template <typename T>
__global__ void cell_rank_kernel(const Pt<T>* __restrict__ pts, int64_t n, const Grid<T>* __restrict__ gp,
                                 int32_t* __restrict__ cell_cnt, int32_t* __restrict__ cell_rank,
                                 uint8_t* __restrict__ dirty, int64_t v_old, int32_t v_fixed_old,
                                 const int32_t* __restrict__ stop) {
    if (stop && *stop) return; // (the counts and the dirty map must stay all-zero for the next real build)
    const Grid<T> g = *gp;
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // On rebuilds the input is the previous sorted order, so consecutive lanes mostly share a
    // cell: one atomic per run of equal cells instead of one per point (~8x fewer atomics).
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const int64_t i = base + threadIdx.x;
        bool valid = i < n;
        int cell = -1;
        if (valid) {
            Pt<T> p = pts[i];
            valid = !(i < v_old && w_to_id(p.w) < v_fixed_old); // stale fixed point of the input view: dropped
            if (valid) {
                int cx = cell_coord(g, p.x, 0), cy = cell_coord(g, p.y, 1), cz = cell_coord(g, p.z, 2);
                cell = (cz * g.n[1] + cy) * g.n[0] + cx;
            }
        }
        const int prev = __shfl_up(cell, 1, 64);
        const bool head = (lane == 0) || (prev != cell);
        const unsigned long long heads = __ballot(head);
        // my run starts at the highest head bit at or below my lane, ends before the next head
        const unsigned long long below = heads & ((2ull << lane) - 1ull);
        const int start = 63 - __builtin_clzll(below);
        const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int len_to_end = above ? __builtin_ctzll(above) + 1 : 64 - lane; // valid for the head lane
        int r0 = 0;
        if (head && valid) {
            r0 = atomicAdd(&cell_cnt[cell], len_to_end);
            // A run keeps the order of the input (ids ascending inside a cell: the original order, or the previous
            // canonical one).  Only a cell that collects SEVERAL runs (a point moved in, or a run split between two
            // waves) can end up out of order: the canonical-order pass visits those cells only.
            if (r0 != 0 && dirty) dirty[cell] = 1; // (topology builds pass no map: their rows are ordered by (d2, id) explicitly)
        }
        r0 = __shfl(r0, start, 64);
        if (valid) cell_rank[i] = r0 + (lane - start); // the cell itself is recomputed by the scatter (4 bytes per point less, twice)
    }
}

// ---- exclusive scan of cell counts (1024 cells per block) ------------------------------------
static constexpr int kScanItems = 16; // 4096 cells per block: the single-block scan of the block sums stays short (~10 M cells at one point per cell)
static constexpr int kScanTile = kThreads * kScanItems;

__device__ inline int wave_incl_scan(int v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// returns exclusive prefix of v within the block and the block total via *total
__device__ inline int block_excl_scan(int v, int* total, int* sm /* [kThreads/64 + 1] */) {
    int incl = wave_incl_scan(v);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            int t = sm[w];
            sm[w] = run;
            run += t;
        }
        sm[kThreads / 64] = run;
    }
    __syncthreads();
    int res = incl - v + sm[wave];
    *total = sm[kThreads / 64];
    __syncthreads();
    return res;
}

template <typename T>
__global__ void scan_reduce_kernel(const int32_t* __restrict__ cnt, const Grid<T>* __restrict__ gp,
                                   int32_t* __restrict__ block_sums) {
    __shared__ int sm[kThreads / 64 + 1];
    int ncells = gp->ncells;
    int base = blockIdx.x * kScanTile;
    if (base >= ncells) return;
    int s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        int idx = base + j * kThreads + threadIdx.x;
        s += idx < ncells ? cnt[idx] : 0;
    }
    int total;
    block_excl_scan(s, &total, sm);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

template <typename T>
__global__ void scan_apply_kernel(int32_t* __restrict__ cnt, const int32_t* __restrict__ block_sums,
                                  const Grid<T>* __restrict__ gp, int32_t* __restrict__ cell_start,
                                  const int32_t* __restrict__ stop) {
    __shared__ int sm[kThreads / 64 + 1];
    if (stop && *stop) return;
    int ncells = gp->ncells;
    int base = blockIdx.x * kScanTile;
    if (base >= ncells) return;
    // thread t owns items base + t*kScanItems .. +kScanItems-1 (contiguous per thread)
    int v[kScanItems];
    int s = 0;
    const bool whole = base + kScanTile <= ncells; // every tile but the last: 16-byte accesses, no bound checks
    if (whole) {
        int4* src = reinterpret_cast<int4*>(cnt + base + threadIdx.x * kScanItems);
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            const int4 w = src[q];
            src[q] = make_int4(0, 0, 0, 0); // the counts are consumed: left all-zero for the next build (no memset of 4 B x cells)
            v[4 * q] = w.x;
            v[4 * q + 1] = w.y;
            v[4 * q + 2] = w.z;
            v[4 * q + 3] = w.w;
            s += (w.x + w.y) + (w.z + w.w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            int idx = base + threadIdx.x * kScanItems + j;
            v[j] = idx < ncells ? cnt[idx] : 0;
            if (idx < ncells) cnt[idx] = 0;
            s += v[j];
        }
    }
    // offset of this tile = sum of the tile sums in front of it (what a separate single-block scan kernel used to
    // produce: one launch less, and the sums are a few KB in L2)
    int front = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kThreads) front += block_sums[b];
    int front_total;
    block_excl_scan(front, &front_total, sm);
    int total;
    int ex = block_excl_scan(s, &total, sm) + front_total;
    if (whole) {
        int4* dst = reinterpret_cast<int4*>(cell_start + base + threadIdx.x * kScanItems);
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            int4 w;
            w.x = ex;
            w.y = w.x + v[4 * q];
            w.z = w.y + v[4 * q + 1];
            w.w = w.z + v[4 * q + 2];
            ex = w.w + v[4 * q + 3];
            dst[q] = w;
        }
        if (base + kScanTile == ncells && threadIdx.x == kThreads - 1) cell_start[ncells] = ex;
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            int idx = base + threadIdx.x * kScanItems + j;
            if (idx < ncells) cell_start[idx] = ex;
            ex += v[j];
            if (idx == ncells - 1) cell_start[ncells] = ex;
        }
    }
}

template <typename T>
__global__ void scatter_kernel(const Pt<T>* __restrict__ pts, int64_t n, const int32_t* __restrict__ cell_rank,
                               const Grid<T>* __restrict__ gp, const int32_t* __restrict__ cell_start,
                               Pt<T>* __restrict__ out, int64_t v_old, int32_t v_fixed_old, int32_t v_id_shift,
                               const int32_t* __restrict__ stop) {
    if (stop && *stop) return;
    const Grid<T> g = *gp;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Pt<T> p = pts[i];
        if (i < v_old) { // input view: stale fixed points are dropped, the others follow the resized head
            const int32_t id = w_to_id(p.w);
            if (id < v_fixed_old) continue;
            p.w = id_to_w((T)0, id + v_id_shift);
        }
        const int cx = cell_coord(g, p.x, 0), cy = cell_coord(g, p.y, 1), cz = cell_coord(g, p.z, 2);
        out[cell_start[(cz * g.n[1] + cy) * g.n[0] + cx] + cell_rank[i]] = p;
    }
}

// Within-cell order by id: the atomic ranks above depend on arrival order; this makes the
// sorted array (and every downstream sum) a pure function of the input.  Cells larger than
// kCanonMax keep arrival order (results stay exact; only fp summation order may vary).
static constexpr int kCanonMax = 96;

// One thread per cell.  On rebuilds the input is the previous canonical order and a run of same-cell
// points keeps its order through the ranking (lanes of one wave are served in lane order), so most cells
// arrive sorted: the pass reads the ids (one 4-byte load per point) and insertion-sorts in place, in
// global memory, the cells that are out of order (those that gained a point from another wave).
// Measured per 10 M-point rebuild: staging every cell through LDS and sorting there 110 us; this 85 us;
// marking candidate cells in cell_rank_kernel (bitmap / byte map) and visiting only those 91-97 us (a third
// of the cells exchange a point per iteration, and the marks cost atomics or a second memset); a register
// rank sort with all loads in flight 125 us.  The pass is bound by touching the lines the scatter just wrote.
// Round 2: the dirty map is read 16 cells per lane (one 16-byte load; a lane that finds a mark clears its 16 bytes),
// the marked cells of a wave are queued in LDS and handed out one per lane — the marks are few (a cell is marked
// only when it collected several runs), so a thread per cell spent its time on byte loads and on lanes waiting
// for the one lane of the wave that had work.  97 -> see DESIGN.md §4 (hash).
template <typename T>
__global__ __launch_bounds__(kThreads) void canon_kernel(Pt<T>* __restrict__ pts,
                                                         const int32_t* __restrict__ cell_start,
                                                         uint8_t* __restrict__ dirty,
                                                         const Grid<T>* __restrict__ gp, const int32_t* __restrict__ stop) {
    __shared__ int32_t queue[kThreads / 64][64 * 16];
    if (stop && *stop) return;
    const int ncells = gp->ncells;
    const int ngroups = (ncells + 15) / 16; // (the map is allocated 64 bytes past the cell capacity and all-zero there)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* q = queue[wave];
    const int nwave_groups = (ngroups + 63) / 64;
    for (int wg = blockIdx.x * (kThreads / 64) + wave; wg < nwave_groups; wg += gridDim.x * (kThreads / 64)) {
        const int grp = wg * 64 + lane;
        uint4 m = make_uint4(0u, 0u, 0u, 0u);
        if (grp < ngroups) m = reinterpret_cast<const uint4*>(dirty)[grp];
        const bool any = (m.x | m.y | m.z | m.w) != 0u;
        if (!__any(any)) continue;
        if (any) reinterpret_cast<uint4*>(dirty)[grp] = make_uint4(0u, 0u, 0u, 0u); // consumed: all-zero again for the next build
        // marks are bytes of value 1: bit 8 j of word w <-> cell 16 grp + 4 w + j
        const uint32_t w[4] = {m.x, m.y, m.z, m.w};
        int mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mine += __popc(w[k] & 0x01010101u);
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        int pos = incl - mine;
        if (any) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((w[k] >> (8 * j)) & 1u) q[pos++] = grp * 16 + k * 4 + j;
        }
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < total; e += 64) {
            const int cell = q[e];
            if (cell >= ncells) continue;
            const int s = cell_start[cell];
            const int n_in = cell_start[cell + 1] - s;
            if (n_in < 2 || n_in > kCanonMax) continue;
            bool sorted = true;
            int prev = w_to_id(pts[s].w);
            for (int i = 1; i < n_in; ++i) {
                const int cur = w_to_id(pts[s + i].w);
                sorted = sorted && prev <= cur;
                prev = cur;
            }
            if (sorted) continue;
            for (int i = 1; i < n_in; ++i) {
                const Pt<T> key = pts[s + i];
                const int kid = w_to_id(key.w);
                int j = i - 1;
                while (j >= 0 && w_to_id(pts[s + j].w) > kid) {
                    pts[s + j + 1] = pts[s + j];
                    --j;
                }
                pts[s + j + 1] = key;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static inline int grid_for(int64_t n, int threads, int cap_blocks) {
    int64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    return (int)(b > cap_blocks ? cap_blocks : b);
}

template <typename T>
int load_points(wtp_ctx* ctx, const T* d_xyz, Pt<T>* out, int64_t n, int dim) {
    hipLaunchKernelGGL(load_points_kernel<T>, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0,
                       ctx->stream, d_xyz, out, n, dim);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

static int cell_capacity(const wtp_ctx* ctx, int64_t n, int k, double cell_scale) {
    double rho = ctx->rho * (k > 0 ? (double)k / 21.0 : 1.0);
    if (rho < 1.0) rho = 1.0;
    double cap = (double)n / rho * 1.6 / (cell_scale * cell_scale * cell_scale) + 4096.0;
    if (cap > 8.0 * (double)n + 4096.0) cap = 8.0 * (double)n + 4096.0; // mostly-empty grids: bounded memory
    if (cap > 1.5e9) cap = 1.5e9;
    return (int)cap;
}

// Occupancy as the points see it: sum cnt^2 / sum cnt = the mean, over points, of the number of
// points sharing their cell (rho + 1 for a Poisson cloud of mean rho).
__global__ void occupancy_kernel(const int32_t* __restrict__ cell_start, const int32_t* __restrict__ ncells_p,
                                 unsigned long long* __restrict__ out /* [sum cnt^2, sum cnt, max] */) {
    const int ncells = *ncells_p;
    unsigned long long s2 = 0, s1 = 0, mx = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncells; i += gridDim.x * blockDim.x) {
        const unsigned long long c = (unsigned long long)(cell_start[i + 1] - cell_start[i]); // (the counts themselves are consumed by the scan)
        s2 += c * c;
        s1 += c;
        mx = c > mx ? c : mx;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        s2 += __shfl_down(s2, d, 64);
        s1 += __shfl_down(s1, d, 64);
        const unsigned long long o = __shfl_down(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0 && s1) {
        atomicAdd(&out[0], s2);
        atomicAdd(&out[1], s1);
        atomicMax(&out[2], mx);
    }
}

template <typename T>
__global__ void sum_kernel(const T* __restrict__ v, int64_t n, double* __restrict__ out) {
    double s = 0, s2 = 0; // out[0] = sum, out[1] = sum of squares
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)v[i];
        s += x;
        s2 += x * x;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_down(s, d, 64);
        s2 += __shfl_down(s2, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, s);
        atomicAdd(out + 1, s2);
    }
}

template <typename T> int launch_sum(wtp_ctx* ctx, const T* d_v, int64_t n, double* d_out) {
    WTP_HIP(ctx, hipMemsetAsync(d_out, 0, 2 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(sum_kernel<T>, dim3(grid_for(n, kThreads, 1024)), dim3(kThreads), 0, ctx->stream, d_v, n, d_out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// ---- fp64 topology through an fp32 candidate search (wtp_api.hip knn_dev_f64) -------------------------
// origin_kernel: bbox partials -> {min x, min y, min z, largest extent};  to_local_f32: double
// points minus that origin, rounded to float, ids kept;  refine: exact fp64 re-ranking of the fp32
// candidate lists with a certificate that no excluded point can belong to the answer.
__global__ void origin_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out4) {
    double mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        double lo = Lim<double>::inf(), hi = -Lim<double>::inf();
        for (int b = threadIdx.x; b < nparts; b += 64) {
            lo = part[b * 6 + a] < lo ? part[b * 6 + a] : lo;
            hi = part[b * 6 + 3 + a] > hi ? part[b * 6 + 3 + a] : hi;
        }
        for (int d = 32; d >= 1; d >>= 1) {
            double o = __shfl_down(lo, d, 64);
            lo = o < lo ? o : lo;
            o = __shfl_down(hi, d, 64);
            hi = o > hi ? o : hi;
        }
        mn[a] = lo;
        mx[a] = hi;
    }
    if (threadIdx.x == 0) {
        double ext = 0;
        for (int a = 0; a < 3; ++a) {
            const double e = mx[a] - mn[a];
            out4[a] = mn[a] == mn[a] && mn[a] > -Lim<double>::inf() && mn[a] < Lim<double>::inf() ? mn[a] : 0.0;
            if (e == e && e > ext && e < Lim<double>::inf()) ext = e;
        }
        out4[3] = ext;
    }
}

__global__ void to_local_f32_kernel(const double4* __restrict__ in, int64_t n, const double* __restrict__ org4,
                                    float4* __restrict__ out) {
    const double ox = org4[0], oy = org4[1], oz = org4[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double4 p = in[i];
        float4 o;
        o.x = (float)(p.x - ox);
        o.y = (float)(p.y - oy);
        o.z = (float)(p.z - oz);
        o.w = id_to_w(0.f, w_to_id(p.w));
        out[i] = o;
    }
}

static constexpr int kRefineMax = 32;

// One thread per query: exact d2 to its kc fp32 candidates (self among them), canonical order, first k.
// Certificate: every point the fp32 search excluded is at least as far, in fp32-local arithmetic, as its
// last candidate (distance dmax32); rounding the coordinates to float and evaluating in float moves a
// distance by less than eps = extent 2^-21 + dmax32 2^-20, so an excluded point's exact distance exceeds
// dmax32 - eps.  If the exact kq-th candidate distance is strictly below that, the first kq candidates
// in exact order are the answer; otherwise the query goes to the exact fp64 path.
__global__ void refine_f64_kernel(const double4* __restrict__ raw, const int32_t* __restrict__ cand,
                                  const float* __restrict__ cdist, int64_t n, int kc, int k, int include_self,
                                  const double* __restrict__ org4, int32_t* __restrict__ idx_out,
                                  double* __restrict__ dist_out, int32_t* __restrict__ fail_list,
                                  int32_t* __restrict__ fail_count) {
    const double extent = org4[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double4 q = raw[i];
        double kd[kRefineMax];
        int32_t ki[kRefineMax];
        int m = 0;
        for (int j = 0; j < kc; ++j) {
            const int32_t c = cand[i * kc + j];
            const double4 p = raw[c];
            const double d = dist2<double>(q.x, q.y, q.z, p.x, p.y, p.z);
            int pos = m++;
            while (pos > 0 && lex_lt(d, c, kd[pos - 1], ki[pos - 1])) {
                kd[pos] = kd[pos - 1];
                ki[pos] = ki[pos - 1];
                --pos;
            }
            kd[pos] = d;
            ki[pos] = c;
        }
        const int kq = include_self ? k : k + 1;
        const double dmax32 = (double)cdist[i * kc + kc - 1];
        const double eps = extent * 0x1p-21 + dmax32 * 0x1p-20;
        const bool certified = (int64_t)kc >= n || wsqrt(kd[kq - 1]) < dmax32 - eps;
        int out = 0;
        for (int j = 0; j < kc && out < k; ++j) {
            if (!include_self && ki[j] == (int32_t)i) continue; // self removed by index (src/topology.jl:82)
            idx_out[i * k + out] = ki[j];
            if (dist_out) dist_out[i * k + out] = wsqrt(kd[j]);
            ++out;
        }
        if (!certified || out < k) {
            const int pos = atomicAdd(fail_count, 1);
            fail_list[pos] = (int32_t)i;
        }
    }
}

// Round 3: the same re-ranking in SLOT order.  The fp32 search ran on points relabelled with their slot in the sorted
// array (relabel_slots_kernel), so its rows are in slot order and name slots: a query's candidates sit next to it in
// `sorted` (the fp64 points in the same order) instead of anywhere in the input — the gathers hit the cache lines the
// neighbouring queries just used — and the candidate list lives in registers (KC at compile time; the rows arrive in fp32
// order, so exchange passes until nothing moves replace the insertion sort whose dynamically indexed arrays lived in scratch).
// Canonical order and the row's place are by ORIGINAL id (kept in sorted[].w); failed queries are listed by original id.
__global__ void relabel_slots_kernel(const double4* __restrict__ raw, float4* __restrict__ sorted32, double4* __restrict__ sorted64,
                                     int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p = sorted32[i];
        sorted64[i] = raw[w_to_id(p.w)];
        p.w = id_to_w(0.f, (int32_t)i);
        sorted32[i] = p;
    }
}

template <int KC>
__global__ __launch_bounds__(128) void refine_f64_slots_kernel(const double4* __restrict__ sorted, const int32_t* __restrict__ cand,
                                                               const float* __restrict__ cdist, int64_t n, int k, int include_self,
                                                               const double* __restrict__ org4, int32_t* __restrict__ idx_out,
                                                               double* __restrict__ dist_out, int32_t* __restrict__ fail_list,
                                                               int32_t* __restrict__ fail_count) {
    const double extent = org4[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double4 q = sorted[i];
        const int32_t qid = w_to_id(q.w);
        double kd[KC];
        int32_t ki[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const double4 p = sorted[cand[i * KC + j]];
            kd[j] = dist2<double>(q.x, q.y, q.z, p.x, p.y, p.z);
            ki[j] = w_to_id(p.w);
        }
        bool again = true;
        while (again) { // (lane-local: the lists arrive almost sorted, one or two passes)
            again = false;
#pragma unroll
            for (int j = 0; j + 1 < KC; ++j) {
                const bool sw = lex_lt(kd[j + 1], ki[j + 1], kd[j], ki[j]);
                const double td = kd[j];
                const int32_t ti = ki[j];
                kd[j] = sw ? kd[j + 1] : td;
                ki[j] = sw ? ki[j + 1] : ti;
                kd[j + 1] = sw ? td : kd[j + 1];
                ki[j + 1] = sw ? ti : ki[j + 1];
                again = again || sw;
            }
        }
        const int kq = include_self ? k : k + 1;
        const double dmax32 = (double)cdist[i * KC + KC - 1];
        const double eps = extent * 0x1p-21 + dmax32 * 0x1p-20;
        double dkq = kd[KC - 1];
#pragma unroll
        for (int j = 0; j < KC; ++j) dkq = (j == kq - 1) ? kd[j] : dkq;
        const bool certified = (int64_t)KC >= n || wsqrt(dkq) < dmax32 - eps;
        int out = 0;
        int32_t* orow = idx_out + (int64_t)qid * k;
        double* drow = dist_out ? dist_out + (int64_t)qid * k : nullptr;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const bool take = out < k && (include_self || ki[j] != qid); // self removed by index (src/topology.jl:82)
            if (take) {
                orow[out] = ki[j];
                if (drow) drow[out] = wsqrt(kd[j]);
                ++out;
            }
        }
        if (!certified || out < k) {
            const int pos = atomicAdd(fail_count, 1);
            fail_list[pos] = qid;
        }
    }
}

int launch_relabel_slots(wtp_ctx* ctx, const double4* raw, float4* sorted32, double4* sorted64, int64_t n) {
    hipLaunchKernelGGL(relabel_slots_kernel, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0, ctx->stream, raw, sorted32,
                       sorted64, n);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// kc must be 24 (k = 21 without self, the reference's default) — the caller checks
int launch_refine_f64_slots(wtp_ctx* ctx, const double4* sorted, const int32_t* cand, const float* cdist, int64_t n, int kc, int k,
                            int include_self, const double* d_org4, int32_t* idx_out, double* dist_out, int32_t* fail_list,
                            int32_t* fail_count) {
    if (kc != 24) return fail(ctx, WTP_ERR_ARG, "launch_refine_f64_slots: kc must be 24");
    WTP_HIP(ctx, hipMemsetAsync(fail_count, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(refine_f64_slots_kernel<24>, dim3(grid_for(n, 128, 16384)), dim3(128), 0, ctx->stream, sorted, cand, cdist, n,
                       k, include_self, d_org4, idx_out, dist_out, fail_list, fail_count);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int launch_origin(wtp_ctx* ctx, const double4* pts, int64_t n, double* d_org4) {
    const int nbb = grid_for(n, kThreads, 1024);
    int rc;
    if ((rc = ensure(ctx, ctx->bbox_part, sizeof(double) * 6 * 1024))) return rc;
    hipLaunchKernelGGL(bbox_kernel<double>, dim3(nbb), dim3(kThreads), 0, ctx->stream, pts, n, (double*)ctx->bbox_part.p);
    hipLaunchKernelGGL(origin_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->bbox_part.p, nbb, d_org4);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int launch_to_local_f32(wtp_ctx* ctx, const double4* in, int64_t n, const double* d_org4, float4* out) {
    hipLaunchKernelGGL(to_local_f32_kernel, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0, ctx->stream, in, n, d_org4, out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int launch_refine_f64(wtp_ctx* ctx, const double4* raw, const int32_t* cand, const float* cdist, int64_t n, int kc, int k,
                      int include_self, const double* d_org4, int32_t* idx_out, double* dist_out, int32_t* fail_list,
                      int32_t* fail_count) {
    WTP_HIP(ctx, hipMemsetAsync(fail_count, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(refine_f64_kernel, dim3(grid_for(n, 128, 16384)), dim3(128), 0, ctx->stream, raw, cand, cdist, n, kc,
                       k, include_self, d_org4, idx_out, dist_out, fail_list, fail_count);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// Per-axis histograms of the coordinates over [lo, hi] (1024 bins each): the host reads them to find
// the quantile box of a cloud whose bounding box is stretched by outliers.
static constexpr int kHistBins = 1024;

template <typename T>
__global__ void axis_hist_kernel(const Pt<T>* __restrict__ pts, int64_t n, int dim, const double* __restrict__ range,
                                 unsigned int* __restrict__ hist) {
    __shared__ unsigned int sh[3 * kHistBins];
    for (int i = threadIdx.x; i < 3 * kHistBins; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    double lo[3], scale[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = range[a];
        const double w = range[3 + a] - range[a];
        scale[a] = w > 0 ? (double)kHistBins / w : 0.0;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const Pt<T> p = pts[i];
        const double v[3] = {(double)p.x, (double)p.y, (double)p.z};
        for (int a = 0; a < dim; ++a) {
            double f = (v[a] - lo[a]) * scale[a];
            int b = f > 0 ? (f < (double)(kHistBins - 1) ? (int)f : kHistBins - 1) : 0; // NaN -> 0
            atomicAdd(&sh[a * kHistBins + b], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * kHistBins; i += blockDim.x)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

template <typename T>
int launch_axis_hist(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int dim, const double* d_range, unsigned int* d_hist) {
    WTP_HIP(ctx, hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * 3 * kHistBins, ctx->stream));
    hipLaunchKernelGGL(axis_hist_kernel<T>, dim3(grid_for(n, kThreads, 512)), dim3(kThreads), 0, ctx->stream, pts, n, dim,
                       d_range, d_hist);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int launch_occupancy(wtp_ctx* ctx, unsigned long long* d_out3) {
    WTP_HIP(ctx, hipMemsetAsync(d_out3, 0, 3 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(occupancy_kernel, dim3(1024), dim3(kThreads), 0, ctx->stream, (const int32_t*)ctx->cell_start.p,
                       (const int32_t*)ctx->ncells_dev, d_out3);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// The per-build scratch (cell counts and starts, one rank per input entry, the dirty map, the scan's block sums), sized for
// a structure of n points read from n_in input entries; *fresh = the counts or the map moved (their contents are gone).
static int hash_scratch(wtp_ctx* ctx, int64_t n, int64_t n_in, int k, double radius, double rho_direct, double cell_scale,
                        int* cap_out, bool* fresh) {
    // k-equivalent of the occupancy the caller fixed (rho = 8 <-> k = 21)
    const int k_cap = rho_direct > 0 ? (int)(rho_direct * 21.0 / ctx->rho) : (radius > 0 ? 6 : k);
    const int cap = cell_capacity(ctx, n, k_cap > 0 ? k_cap : 1, cell_scale);
    int rc;
    if ((rc = ensure(ctx, ctx->grid, sizeof(Grid<double>)))) return rc;
    if ((rc = ensure(ctx, ctx->bbox_part, sizeof(double) * 6 * 1024))) return rc;
    const void* cnt_before = ctx->cell_cnt.p;
    const void* dirty_before = ctx->rank_of.p;
    if ((rc = ensure(ctx, ctx->cell_cnt, sizeof(int32_t) * (size_t)(cap + 1)))) return rc;
    if ((rc = ensure(ctx, ctx->cell_start, sizeof(int32_t) * (size_t)(cap + 2)))) return rc;
    if ((rc = ensure(ctx, ctx->cell_of, sizeof(int32_t) * (size_t)n_in))) return rc;
    if ((rc = ensure(ctx, ctx->rank_of, (size_t)cap + 64))) return rc; // one byte per cell: filled by more than one run of the input
    const int nscan = (cap + kScanTile - 1) / kScanTile;
    if ((rc = ensure(ctx, ctx->scan_tmp, sizeof(int32_t) * (size_t)(nscan + 1)))) return rc;
    *cap_out = cap;
    *fresh = cnt_before != ctx->cell_cnt.p || dirty_before != ctx->rank_of.p;
    return WTP_OK;
}

// First half of the next build_hash, issued early: the entries [0, n_old) of `in` (the snapshot as it stands, its fixed
// head of fixed_old points stale) are ranked into the cells of the grid in place, which the build is going to keep
// (ctx->reuse_grid).  The build that follows appends n_in - n_old new entries and ranks only those (block driver: the
// owned points are ranked while the ghost rows travel).  Nothing depends on the guess being right: build_hash
// checks ctx->prerank against what it is asked to build and otherwise starts over.
template <typename T>
int prerank_old_snapshot(wtp_ctx* ctx, const Pt<T>* in, int64_t n_old, int32_t fixed_old, int64_t n_next, int64_t n_in_next,
                         int k, double rho_direct, double cell_scale) {
    if (!(cell_scale > 0)) cell_scale = 1.0;
    int cap = 0, rc;
    bool fresh = false;
    ctx->prerank.valid = false;
    if ((rc = hash_scratch(ctx, n_next, n_in_next, k, 0.0, rho_direct, cell_scale, &cap, &fresh))) return rc;
    int32_t* cnt = (int32_t*)ctx->cell_cnt.p;
    uint8_t* dirty = (uint8_t*)ctx->rank_of.p;
    if (!ctx->hash_scratch_clean || fresh) {
        WTP_HIP(ctx, hipMemsetAsync(cnt, 0, ctx->cell_cnt.cap, ctx->stream));
        WTP_HIP(ctx, hipMemsetAsync(dirty, 0, ctx->rank_of.cap, ctx->stream));
    }
    ctx->hash_scratch_clean = false; // (the counts hold the first half from here on)
    const int nb = grid_for(n_old, kThreads, 16384);
    hipLaunchKernelGGL(cell_rank_kernel<T>, dim3(nb), dim3(kThreads), 0, ctx->stream, in, n_old, (const Grid<T>*)ctx->grid.p, cnt,
                       (int32_t*)ctx->cell_of.p, ctx->topology_build ? (uint8_t*)nullptr : dirty, n_old, fixed_old, ctx->stop_dev);
    WTP_HIP(ctx, hipGetLastError());
    ctx->prerank.valid = true;
    ctx->prerank.in = in;
    ctx->prerank.n_old = n_old;
    ctx->prerank.fixed_old = fixed_old;
    ctx->prerank.cnt = cnt;
    ctx->prerank.cr = ctx->cell_of.p;
    ctx->prerank.dirty = dirty;
    return WTP_OK;
}
template int prerank_old_snapshot<float>(wtp_ctx*, const Pt<float>*, int64_t, int32_t, int64_t, int64_t, int, double, double);
template int prerank_old_snapshot<double>(wtp_ctx*, const Pt<double>*, int64_t, int32_t, int64_t, int64_t, int, double, double);

template <typename T>
int build_hash(wtp_ctx* ctx, const Pt<T>* in, Pt<T>* out, int64_t n, int dim, int k, double radius, double rho_direct,
               double min_cell, double cell_scale) {
    if (!(cell_scale > 0)) cell_scale = 1.0;
    // n = points of the structure; the input array may be longer (ctx->hash_view: stale fixed points
    // still in place, new ones appended)
    const HashView hv = ctx->hash_view;
    const int64_t n_in = hv.active ? hv.n_in : n;
    const int64_t v_old = hv.active ? hv.n_old : 0;
    const int32_t v_fixed_old = hv.active ? hv.fixed_old : 0, v_shift = hv.active ? hv.id_shift : 0;
    const int nbb = grid_for(n_in, kThreads, 1024);
    int cap = 0, rc;
    bool fresh = false;
    if ((rc = hash_scratch(ctx, n, n_in, k, radius, rho_direct, cell_scale, &cap, &fresh))) return rc;
    const int nscan = (cap + kScanTile - 1) / kScanTile;

    Grid<T>* g = (Grid<T>*)ctx->grid.p;
    T* part = (T*)ctx->bbox_part.p;
    int32_t* cnt = (int32_t*)ctx->cell_cnt.p;
    int32_t* start = (int32_t*)ctx->cell_start.p;
    int32_t* cr = (int32_t*)ctx->cell_of.p;
    uint8_t* dirty = (uint8_t*)ctx->rank_of.p;
    int32_t* bs = (int32_t*)ctx->scan_tmp.p;
    hipStream_t st = ctx->stream;

    // target occupancy: c = 1.17 r_k  (r_k = k-th neighbour distance at uniform density)
    double rho_k = (dim == 3 ? 0.381 : 0.436) * (double)(k > 0 ? k : 21) * (ctx->rho / 8.0);
    if (rho_direct > 0) rho_k = rho_direct; // caller fixes the occupancy (compact-support sweep)
    if (rho_k < 1.0) rho_k = 1.0;

    // Counts and dirty map are consumed (zeroed) by the scan and by the canonical-order pass of every build, so
    // a build normally finds them all-zero: the two fills (4 + 1 bytes per cell) run only after a reallocation or
    // after a build that did not complete.
    // ctx->reuse_grid (one-shot, set by the relax session): keep the Grid of the previous build — origin, cell edge,
    // cell counts — and skip the bounding-box pass.  A point that has left the old box since is clamped into an
    // edge cell, which the kernels treat as unbounded outward, so the search stays exact.
    const bool reuse = ctx->reuse_grid;
    ctx->reuse_grid = false;
    // the old snapshot's entries may have been ranked already (prerank_old_snapshot): same array, same view, same
    // scratch, same grid — then only the appended entries are left
    const Prerank pre = ctx->prerank;
    ctx->prerank.valid = false;
    const bool half_done = pre.valid && reuse && hv.active && !fresh && pre.in == (const void*)in && pre.n_old == v_old &&
                           pre.fixed_old == v_fixed_old && pre.cnt == (const void*)cnt && pre.cr == (const void*)cr &&
                           pre.dirty == (const void*)dirty;
    if (!half_done && (!ctx->hash_scratch_clean || fresh)) {
        WTP_HIP(ctx, hipMemsetAsync(cnt, 0, ctx->cell_cnt.cap, st));
        WTP_HIP(ctx, hipMemsetAsync(dirty, 0, ctx->rank_of.cap, st));
    }
    ctx->hash_scratch_clean = false;
    ctx->preranked_builds += half_done ? 1 : 0;
    if (!reuse) {
        hipLaunchKernelGGL(bbox_kernel<T>, dim3(nbb), dim3(kThreads), 0, st, in, n_in, part, v_old, v_fixed_old);
        hipLaunchKernelGGL(grid_setup_kernel<T>, dim3(1), dim3(64), 0, st, part, nbb, g, n, dim, rho_k, radius, min_cell,
                           cap, cell_scale, ctx->box_active ? (const double*)ctx->box_dev.p : (const double*)nullptr, ctx->stop_dev);
    }
    ctx->ncells_dev = &g->ncells;
    const int nb = grid_for(n_in, kThreads, 16384);
    if (half_done) {
        if (n_in > v_old) // (the appended entries: nothing stale among them)
            hipLaunchKernelGGL(cell_rank_kernel<T>, dim3(grid_for(n_in - v_old, kThreads, 16384)), dim3(kThreads), 0, st, in + v_old,
                               n_in - v_old, g, cnt, cr + v_old, ctx->topology_build ? (uint8_t*)nullptr : dirty, (int64_t)0, 0,
                               ctx->stop_dev);
    } else {
        hipLaunchKernelGGL(cell_rank_kernel<T>, dim3(nb), dim3(kThreads), 0, st, in, n_in, g, cnt, cr,
                           ctx->topology_build ? (uint8_t*)nullptr : dirty, v_old, v_fixed_old, ctx->stop_dev);
    }
    hipLaunchKernelGGL(scan_reduce_kernel<T>, dim3(nscan), dim3(kThreads), 0, st, cnt, g, bs);
    hipLaunchKernelGGL(scan_apply_kernel<T>, dim3(nscan), dim3(kThreads), 0, st, cnt, bs, g, start, ctx->stop_dev);
    hipLaunchKernelGGL(scatter_kernel<T>, dim3(nb), dim3(kThreads), 0, st, in, n_in, cr, g, start, out, v_old, v_fixed_old,
                       v_shift, ctx->stop_dev);
    if (!ctx->topology_build) {
        hipLaunchKernelGGL(canon_kernel<T>, dim3(grid_for((cap + 15) / 16, kThreads, 4096)), dim3(kThreads), 0, st, out, start,
                           dirty, g, ctx->stop_dev);
    }
    WTP_HIP(ctx, hipGetLastError());
    ctx->hash_scratch_clean = true; // (the scan consumed the counts; a topology build wrote no marks, the canonical-order pass cleared the others)
    return WTP_OK;
}

// ---- small utility kernels -----------------------------------------------------------------
template <typename T>
__global__ void unpermute_kernel(const Pt<T>* __restrict__ pts, int64_t n, int64_t n_fixed, int dim,
                                 T* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Pt<T> p = pts[i];
        int64_t id = w_to_id(p.w);
        if (id < n_fixed) continue;
        T* o = out + (id - n_fixed) * dim;
        o[0] = p.x;
        o[1] = p.y;
        if (dim == 3) o[2] = p.z;
    }
}

template <typename T>
__global__ void unpermute_pd_kernel(const Pt<T>* __restrict__ pts, int64_t n, int64_t n_fixed,
                                    const T* __restrict__ forces, const T* __restrict__ nn_dist,
                                    const int32_t* __restrict__ nn_id, T* __restrict__ fo,
                                    T* __restrict__ no, int32_t* __restrict__ io) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int64_t id = w_to_id(pts[i].w);
        if (id < n_fixed) continue;
        int64_t o = id - n_fixed;
        if (fo) fo[o] = forces[i];
        if (no) no[o] = nn_dist[i];
        if (io) io[o] = nn_id[i];
    }
}

template <typename T>
__global__ void set_point_kernel(Pt<T>* __restrict__ pts, int64_t n, int32_t id, int dim, const T* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (w_to_id(pts[i].w) == id) {
            pts[i].x = v[0];
            pts[i].y = v[1];
            pts[i].z = dim == 3 ? v[2] : (T)0;
        }
    }
}

// ---- exclusive scan of row lengths -> CSR offsets (int32 counts, int64 offsets; wtp_radius_offsets) ------
static constexpr int kOffTile = 2048; // elements per block
__global__ __launch_bounds__(kThreads) void offsets_tile_sum_kernel(const int32_t* __restrict__ cnt, int64_t n,
                                                                   int64_t* __restrict__ tile_sum) {
    __shared__ int64_t sm[kThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kOffTile;
    int64_t s = 0;
    for (int i = threadIdx.x; i < kOffTile; i += kThreads)
        if (base + i < n) s += cnt[base + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < kThreads / 64; ++w) t += sm[w];
        tile_sum[blockIdx.x] = t;
    }
}
__global__ void offsets_tile_scan_kernel(int64_t* __restrict__ tile_sum, int64_t ntiles) { // one thread: ntiles <= n / 2048
    int64_t run = 0;
    for (int64_t i = 0; i < ntiles; ++i) {
        const int64_t t = tile_sum[i];
        tile_sum[i] = run;
        run += t;
    }
    tile_sum[ntiles] = run;
}
__global__ __launch_bounds__(kThreads) void offsets_apply_kernel(const int32_t* __restrict__ cnt, int64_t n,
                                                                const int64_t* __restrict__ tile_sum,
                                                                int64_t* __restrict__ off) {
    // a tile = kThreads runs of kOffTile / kThreads consecutive elements, one run per thread
    constexpr int kRun = kOffTile / kThreads;
    __shared__ int64_t sm[kThreads];
    const int64_t base = (int64_t)blockIdx.x * kOffTile + (int64_t)threadIdx.x * kRun;
    int32_t v[kRun];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < kRun; ++i) {
        v[i] = base + i < n ? cnt[base + i] : 0;
        s += v[i];
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = tile_sum[blockIdx.x];
        for (int t = 0; t < kThreads; ++t) {
            const int64_t x = sm[t];
            sm[t] = run;
            run += x;
        }
    }
    __syncthreads();
    int64_t run = sm[threadIdx.x];
#pragma unroll
    for (int i = 0; i < kRun; ++i) {
        if (base + i < n) off[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) off[n] = tile_sum[gridDim.x];
}

// d_off[0..n] = exclusive scan of d_cnt[0..n); d_tmp: (ntiles + 1) int64
int launch_offsets_scan(wtp_ctx* ctx, const int32_t* d_cnt, int64_t n, int64_t* d_tmp, int64_t* d_off) {
    const int64_t ntiles = (n + kOffTile - 1) / kOffTile;
    hipLaunchKernelGGL(offsets_tile_sum_kernel, dim3((unsigned)ntiles), dim3(kThreads), 0, ctx->stream, d_cnt, n, d_tmp);
    hipLaunchKernelGGL(offsets_tile_scan_kernel, dim3(1), dim3(1), 0, ctx->stream, d_tmp, ntiles);
    hipLaunchKernelGGL(offsets_apply_kernel, dim3((unsigned)ntiles), dim3(kThreads), 0, ctx->stream, d_cnt, n,
                       (const int64_t*)d_tmp, d_off);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}
size_t offsets_scan_tmp_bytes(int64_t n) { return sizeof(int64_t) * (size_t)((n + kOffTile - 1) / kOffTile + 2); }

template <typename T>
int launch_unpermute(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t n_fixed, int dim, T* d_out) {
    hipLaunchKernelGGL(unpermute_kernel<T>, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0,
                       ctx->stream, pts, n, n_fixed, dim, d_out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T>
int launch_unpermute_point_data(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t n_fixed,
                                const T* forces, const T* nn_dist, const int32_t* nn_id, T* fo, T* no,
                                int32_t* io) {
    hipLaunchKernelGGL(unpermute_pd_kernel<T>, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0,
                       ctx->stream, pts, n, n_fixed, forces, nn_dist, nn_id, fo, no, io);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// many points at once: ids ascending (snapshot ids), one binary search per snapshot point
template <typename T>
__global__ void set_points_kernel(Pt<T>* __restrict__ pts, int64_t n, const int32_t* __restrict__ ids, int64_t m, int dim,
                                  const T* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int32_t id = w_to_id(pts[i].w);
        int64_t lo = 0, hi = m;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ids[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        if (lo < m && ids[lo] == id) {
            pts[i].x = v[lo * dim];
            pts[i].y = v[lo * dim + 1];
            pts[i].z = dim == 3 ? v[lo * dim + 2] : (T)0;
        }
    }
}

template <typename T>
int launch_set_points(wtp_ctx* ctx, Pt<T>* pts, int64_t n, const int32_t* d_ids, int64_t m, int dim, const T* d_v) {
    hipLaunchKernelGGL(set_points_kernel<T>, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0, ctx->stream, pts, n, d_ids,
                       m, dim, d_v);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T>
int launch_set_point(wtp_ctx* ctx, Pt<T>* pts, int64_t n, int32_t id, int dim, const T* d_v) {
    hipLaunchKernelGGL(set_point_kernel<T>, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0,
                       ctx->stream, pts, n, id, dim, d_v);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// ---- synthetic inputs: splitmix64(seed*2^40 + 3*i + axis) >> 40 * 2^-24 ------
__device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T>
__global__ void gen_uniform_kernel(uint64_t seed, int64_t first, int64_t n, int dim, T* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        for (int a = 0; a < dim; ++a) {
            uint64_t h = splitmix64((seed << 40) + 3ull * (uint64_t)(first + i) + (uint64_t)a);
            out[i * dim + a] = (T)((float)(h >> 40) * (1.0f / 16777216.0f));
        }
}

template <typename T>
int launch_gen_uniform(wtp_ctx* ctx, uint64_t seed, int64_t first, int64_t n, int dim, T* d_out) {
    hipLaunchKernelGGL(gen_uniform_kernel<T>, dim3(grid_for(n, kThreads, 8192)), dim3(kThreads), 0,
                       ctx->stream, seed, first, n, dim, d_out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// ---- final reduction of per-block partials (fixed order => deterministic) --------------------
__global__ void reduce_partials_kernel(const Partial* __restrict__ parts, int n_parts, int used_brick, int wave_base,
                                       int used_wave, int used_generic, const int32_t* __restrict__ fb_count,
                                       const int32_t* __restrict__ uncovered, const int32_t* __restrict__ escaped,
                                       wtp_step_stats* __restrict__ out, int32_t* __restrict__ counters) {
    __shared__ Acc sm[kThreads / 64];
    reduce_partials_block(parts, n_parts, used_brick, wave_base, used_wave, used_generic, fb_count, uncovered, escaped, out,
                          counters, sm);
}

int launch_reduce_partials(wtp_ctx* ctx, const Partial* parts, int n_parts, int used_brick, int used_wave,
                           int used_generic, const int32_t* fb_count, const int32_t* uncovered, const int32_t* escaped,
                           wtp_step_stats* d_slot) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, parts, n_parts, used_brick,
                       brick_partials(), used_wave, used_generic, fb_count, uncovered, escaped, d_slot,
                       (int32_t*)ctx->fb_count.p);
    ctx->counters_clean = true;
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// ---- sharded sessions: boundary layers out, ghost layer in ----------------------
// Ordered compaction in two launches.  A block owns kLayerChunk consecutive slots; the output order
// is (block, wave, pass, lane) — a fixed function of the slot order, so the layer a neighbour
// receives (and with it the ids its ghosts get) does not depend on scheduling.
static constexpr int kLayerPasses = 32;
static constexpr int kLayerChunk = kThreads * kLayerPasses;

int layer_blocks(int64_t n) { return (int)((n + kLayerChunk - 1) / kLayerChunk); }

template <typename T> __device__ inline T axis_of(const Pt<T>& p, int axis) {
    return axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
}

// slot of (wave, pass, lane) inside a block's chunk: a wave reads 64 consecutive points per pass
__device__ inline int64_t layer_slot(int block, int wave, int pass, int lane) {
    return (int64_t)block * kLayerChunk + ((int64_t)wave * kLayerPasses + pass) * 64 + lane;
}

template <typename T>
__global__ void layer_count_kernel(const Pt<T>* __restrict__ pts, int64_t n, int32_t n_fixed, int axis, T lo_in,
                                   T hi_in, T lo_out, T hi_out, int2* __restrict__ blk, int32_t* __restrict__ totals,
                                   const Grid<T>* __restrict__ gp, const int32_t* __restrict__ cell_start, T reach) {
    __shared__ int sm[4][kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (gp) {
        // pts is in the slot order of a grid whose slowest axis is `axis`, and no point has moved more
        // than `reach` since it was binned: only the cell layers within reach of the planes can hold
        // layer points, i.e. a prefix and a suffix of the slot range.  Chunks in between are skipped.
        const Grid<T> g = *gp;
        const int plane_cells = g.ncells / g.n[axis];
        const T lo_plane = lo_in > lo_out ? lo_in : lo_out, hi_plane = hi_in < hi_out ? hi_in : hi_out;
        int64_t lo_end = 0, hi_begin = n; // layer candidates: slots [0, lo_end) and [hi_begin, n)
        if (lo_plane > -Lim<T>::inf()) lo_end = cell_start[(int64_t)(cell_coord(g, lo_plane + reach, axis) + 1) * plane_cells];
        if (hi_plane < Lim<T>::inf()) hi_begin = cell_start[(int64_t)cell_coord(g, hi_plane - reach, axis) * plane_cells];
        const int64_t b0 = (int64_t)blockIdx.x * kLayerChunk, b1 = b0 + kLayerChunk;
        if (b0 >= lo_end && b1 <= hi_begin) {
            if (threadIdx.x == 0) blk[blockIdx.x] = make_int2(0, 0);
            return;
        }
    }
    int c[4] = {0, 0, 0, 0};
    for (int pass = 0; pass < kLayerPasses; ++pass) {
        const int64_t i = layer_slot(blockIdx.x, wave, pass, lane);
        bool in_lo = false, in_hi = false, out_lo = false, out_hi = false;
        if (i < n) {
            const Pt<T> p = pts[i];
            if (w_to_id(p.w) >= n_fixed) {
                const T v = axis_of<T>(p, axis);
                in_lo = v < lo_in;
                in_hi = v >= hi_in;
                out_lo = v < lo_out;
                out_hi = v >= hi_out;
            }
        }
        c[0] += __popcll(__ballot(in_lo));
        c[1] += __popcll(__ballot(in_hi));
        c[2] += __popcll(__ballot(out_lo));
        c[3] += __popcll(__ballot(out_hi));
    }
    if (lane == 0)
        for (int j = 0; j < 4; ++j) sm[j][wave] = c[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        int t[4] = {0, 0, 0, 0};
        for (int j = 0; j < 4; ++j)
            for (int w = 0; w < kThreads / 64; ++w) t[j] += sm[j][w];
        blk[blockIdx.x] = make_int2(t[0], t[1]);
        for (int j = 0; j < 4; ++j)
            if (t[j]) atomicAdd(&totals[j], t[j]);
    }
}

template <typename T>
__global__ void layer_fill_kernel(const Pt<T>* __restrict__ pts, int64_t n, int32_t n_fixed, int axis, T lo_in,
                                  T hi_in, const int2* __restrict__ blk, Pt<T>* __restrict__ out_lo,
                                  Pt<T>* __restrict__ out_hi, int64_t cap) {
    __shared__ int sm_base[2][kThreads / 64];
    __shared__ int sm_wave[2][kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blk[blockIdx.x].x == 0 && blk[blockIdx.x].y == 0) return; // nothing of this chunk is in a layer
    // exclusive prefix of the block counts before this block
    int b0 = 0, b1 = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += kThreads) {
        const int2 v = blk[j];
        b0 += v.x;
        b1 += v.y;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        b0 += __shfl_down(b0, d, 64);
        b1 += __shfl_down(b1, d, 64);
    }
    if (lane == 0) {
        sm_base[0][wave] = b0;
        sm_base[1][wave] = b1;
    }
    // this wave's own counts (same predicate as the count kernel)
    int c0 = 0, c1 = 0;
    for (int pass = 0; pass < kLayerPasses; ++pass) {
        const int64_t i = layer_slot(blockIdx.x, wave, pass, lane);
        bool in_lo = false, in_hi = false;
        if (i < n) {
            const Pt<T> p = pts[i];
            if (w_to_id(p.w) >= n_fixed) {
                const T v = axis_of<T>(p, axis);
                in_lo = v < lo_in;
                in_hi = v >= hi_in;
            }
        }
        c0 += __popcll(__ballot(in_lo));
        c1 += __popcll(__ballot(in_hi));
    }
    if (lane == 0) {
        sm_wave[0][wave] = c0;
        sm_wave[1][wave] = c1;
    }
    __syncthreads();
    int64_t pos0 = 0, pos1 = 0;
    for (int w = 0; w < kThreads / 64; ++w) {
        pos0 += sm_base[0][w];
        pos1 += sm_base[1][w];
        if (w < wave) {
            pos0 += sm_wave[0][w];
            pos1 += sm_wave[1][w];
        }
    }
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < kLayerPasses; ++pass) {
        const int64_t i = layer_slot(blockIdx.x, wave, pass, lane);
        bool in_lo = false, in_hi = false;
        Pt<T> p{};
        if (i < n) {
            p = pts[i];
            const int32_t id = w_to_id(p.w);
            if (id >= n_fixed) {
                const T v = axis_of<T>(p, axis);
                in_lo = v < lo_in;
                in_hi = v >= hi_in;
                p.w = id_to_w((T)0, id - n_fixed);
            }
        }
        const unsigned long long m0 = __ballot(in_lo), m1 = __ballot(in_hi);
        if (in_lo) {
            const int64_t o = pos0 + __popcll(m0 & below);
            if (o < cap) out_lo[o] = p;
        }
        if (in_hi) {
            const int64_t o = pos1 + __popcll(m1 & below);
            if (o < cap) out_hi[o] = p;
        }
        pos0 += __popcll(m0);
        pos1 += __popcll(m1);
    }
}

template <typename T>
int launch_layers(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t n_fixed, int axis, double lo_in, double hi_in,
                  double lo_out, double hi_out, Pt<T>* d_lo, Pt<T>* d_hi, int64_t cap, int2* d_blk, int32_t* d_totals,
                  bool slot_ordered, double reach) {
    const int nblk = layer_blocks(n);
    WTP_HIP(ctx, hipMemsetAsync(d_totals, 0, 4 * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(layer_count_kernel<T>, dim3(nblk), dim3(kThreads), 0, ctx->stream, pts, n, (int32_t)n_fixed, axis,
                       (T)lo_in, (T)hi_in, (T)lo_out, (T)hi_out, d_blk, d_totals,
                       slot_ordered ? (const Grid<T>*)ctx->grid.p : (const Grid<T>*)nullptr,
                       (const int32_t*)ctx->cell_start.p, (T)reach);
    hipLaunchKernelGGL(layer_fill_kernel<T>, dim3(nblk), dim3(kThreads), 0, ctx->stream, pts, n, (int32_t)n_fixed, axis,
                       (T)lo_in, (T)hi_in, (const int2*)d_blk, d_lo, d_hi, cap);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// New snapshot = [new fixed head (ids 0..n_fixed_new) ; movable points of `in` with ids shifted].
// Movable points are appended through a counter, one atomic per block of kRefixChunk slots (a
// per-wave atomic on the single counter serialises: 1.8 ms at 12 M points, measured): their order
// in `out` is arbitrary, the counting sort that follows (and its per-cell canonical order)
// removes it.
static constexpr int kRefixPasses = 8;
static constexpr int kRefixChunk = kThreads * kRefixPasses;

template <typename T>
__global__ void refix_kernel(const Pt<T>* __restrict__ in, int64_t n_old, int32_t n_fixed_old, int32_t n_fixed_new,
                             const Pt<T>* __restrict__ fixed_new, Pt<T>* __restrict__ out,
                             int32_t* __restrict__ counter, int n_chunks) {
    __shared__ int sm_wave[kThreads / 64];
    __shared__ int sm_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if ((int)blockIdx.x < n_chunks) {
        Pt<T> p[kRefixPasses];
        unsigned long long m[kRefixPasses];
        int total = 0;
#pragma unroll
        for (int pass = 0; pass < kRefixPasses; ++pass) {
            const int64_t i = (int64_t)blockIdx.x * kRefixChunk + (int64_t)pass * kThreads + threadIdx.x;
            bool keep = false;
            if (i < n_old) {
                p[pass] = in[i];
                const int32_t id = w_to_id(p[pass].w);
                keep = id >= n_fixed_old;
                p[pass].w = id_to_w((T)0, id - n_fixed_old + n_fixed_new);
            }
            m[pass] = __ballot(keep);
            total += __popcll(m[pass]);
        }
        if (lane == 0) sm_wave[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < kThreads / 64; ++w) t += sm_wave[w];
            sm_base = t ? atomicAdd(counter, t) : 0;
        }
        __syncthreads();
        int64_t pos = (int64_t)n_fixed_new + sm_base;
        for (int w = 0; w < wave; ++w) pos += sm_wave[w];
#pragma unroll
        for (int pass = 0; pass < kRefixPasses; ++pass) {
            if ((m[pass] >> lane) & 1ull) out[pos + __popcll(m[pass] & below)] = p[pass];
            pos += __popcll(m[pass]);
        }
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_fixed_new; j += stride) {
        Pt<T> q = fixed_new[j];
        q.w = id_to_w((T)0, (int32_t)j);
        out[j] = q;
    }
}

template <typename T>
__global__ void append_fixed_kernel(const Pt<T>* __restrict__ src, int64_t n, Pt<T>* __restrict__ dst) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        Pt<T> p = src[j];
        p.w = id_to_w((T)0, (int32_t)j);
        dst[j] = p;
    }
}

template <typename T> int launch_append_fixed(wtp_ctx* ctx, const Pt<T>* d_src, int64_t n, Pt<T>* d_dst) {
    hipLaunchKernelGGL(append_fixed_kernel<T>, dim3(grid_for(n, kThreads, 4096)), dim3(kThreads), 0, ctx->stream, d_src, n,
                       d_dst);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T>
int launch_refix(wtp_ctx* ctx, const Pt<T>* in, int64_t n_old, int64_t n_fixed_old, int64_t n_fixed_new,
                 const Pt<T>* d_fixed_new, Pt<T>* out, int32_t* d_counter) {
    WTP_HIP(ctx, hipMemsetAsync(d_counter, 0, sizeof(int32_t), ctx->stream));
    const int n_chunks = (int)((n_old + kRefixChunk - 1) / kRefixChunk);
    const int blocks = n_chunks > 1024 ? n_chunks : 1024; // >= 1024 so the head copy also fills the chip
    hipLaunchKernelGGL(refix_kernel<T>, dim3(blocks), dim3(kThreads), 0, ctx->stream, in, n_old, (int32_t)n_fixed_old,
                       (int32_t)n_fixed_new, d_fixed_new, out, d_counter, n_chunks);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// explicit instantiations
#define INST(T)                                                                                         \
    template int launch_sum<T>(wtp_ctx*, const T*, int64_t, double*);                                   \
    template int launch_axis_hist<T>(wtp_ctx*, const Pt<T>*, int64_t, int, const double*, unsigned int*); \
    template int launch_layers<T>(wtp_ctx*, const Pt<T>*, int64_t, int64_t, int, double, double, double, double, \
                                  Pt<T>*, Pt<T>*, int64_t, int2*, int32_t*, bool, double);              \
    template int launch_append_fixed<T>(wtp_ctx*, const Pt<T>*, int64_t, Pt<T>*);                       \
    template int launch_refix<T>(wtp_ctx*, const Pt<T>*, int64_t, int64_t, int64_t, const Pt<T>*, Pt<T>*, int32_t*); \
    template int load_points<T>(wtp_ctx*, const T*, Pt<T>*, int64_t, int);                              \
    template int build_hash<T>(wtp_ctx*, const Pt<T>*, Pt<T>*, int64_t, int, int, double, double, double, double); \
    template int launch_unpermute<T>(wtp_ctx*, const Pt<T>*, int64_t, int64_t, int, T*);                \
    template int launch_unpermute_point_data<T>(wtp_ctx*, const Pt<T>*, int64_t, int64_t, const T*,    \
                                                const T*, const int32_t*, T*, T*, int32_t*);            \
    template int launch_set_point<T>(wtp_ctx*, Pt<T>*, int64_t, int32_t, int, const T*);                \
    template int launch_set_points<T>(wtp_ctx*, Pt<T>*, int64_t, const int32_t*, int64_t, int, const T*); \
    template int launch_gen_uniform<T>(wtp_ctx*, uint64_t, int64_t, int64_t, int, T*);
INST(float)
INST(double)
#undef INST

} // namespace wtp
