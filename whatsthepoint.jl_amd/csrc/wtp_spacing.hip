// wtp_spacing.hip — the variable spacing laws evaluated on the device.
//
//   LogLike               h0 x / (a + x),  a = h0 (1 - (g - 1))          src/discretization/spacings.jl:67-72
//   BoundaryLayerSpacing  h_w + (h_b - h_w) / (1 + exp(-(d - δ/2)/(δ/6)))    :121-133
// with x = d = distance to the nearest boundary point (_min_distance, :17-22: a 1-NN query of a
// kd-tree over the boundary points).  The sweep calls the law at every movable point every
// iteration (src/repel.jl:251,260), so the query has to live next to the sweep.
//
// Search structure: the boundary is static for a whole repel call, so it gets a left-balanced
// kd-tree built once on the host (median split on the widest axis, heap order: children of node
// i are 2i+1 and 2i+2, one boundary point per node, no pointers; every node also stores the
// bounding box of its subtree).  A uniform grid would not do: boundary points lie on a surface and
// a query deep inside the domain would have to walk O((d/c)^3) empty cells.  Device traversal:
// depth-first, near child first, far child pushed; a popped subtree is skipped when the distance
// from the query to its box cannot beat the best so far.  (The split-plane distance alone is far
// too weak a bound here: a query 0.3 away from a wall sampled every 0.005 is within 0.3 of
// thousands of split planes — measured 100 ms per million queries, against 0.7 ms with boxes.)
// Exact: the value returned is min over canonical d2 = ((dx*dx + dy*dy) + dz*dz), the same number
// a brute-force scan gives — the box bound is monotone under rounding (each of its terms is <= the
// matching term of any point inside the box, and rounded sums are monotone in their arguments).
// Cell-sorted queries make neighbouring lanes walk nearly the same path.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "wtp_device.hpp"

namespace wtp {

static constexpr int kSpThreads = 256;

#ifndef WTP_DIAG
#define WTP_DIAG 0
#endif
#if WTP_DIAG
__device__ unsigned long long g_kd_steps[2]; // diagnostic builds: node visits of all waves, waves (wtp_debug_kd_steps)
#endif

// nodes in the left subtree of a left-balanced binary tree with n nodes
static int64_t kd_left_size(int64_t n) {
    if (n <= 1) return 0;
    int h = 0;
    while ((int64_t(1) << (h + 1)) <= n) ++h; // h = floor(log2 n): levels 0..h-1 are full
    const int64_t full = (int64_t(1) << h) - 1;
    const int64_t last = n - full;
    const int64_t half = int64_t(1) << (h - 1);
    return (full - 1) / 2 + (last < half ? last : half);
}

template <typename T> struct HostPt { T c[3]; };

template <typename T> struct KdBox { T lo[3], hi[3]; };
// One record per node, fetched with a single (wave-uniform) load per traversal step: the point with
// its split axis in w, and the box of the subtree below it.  48 B (fp32) / 96 B (fp64).
template <typename T> struct KdNode {
    Pt<T> p;
    KdBox<T> box;
    T pad[2];
};

template <typename T>
static void kd_build_rec(std::vector<HostPt<T>>& pts, int64_t lo, int64_t hi, int64_t node, int dim, KdNode<T>* out) {
    while (hi > lo) {
        const int64_t n = hi - lo;
        T mn[3], mx[3];
        for (int a = 0; a < 3; ++a) mn[a] = mx[a] = pts[lo].c[a];
        for (int64_t i = lo + 1; i < hi; ++i)
            for (int a = 0; a < dim; ++a) {
                mn[a] = pts[i].c[a] < mn[a] ? pts[i].c[a] : mn[a];
                mx[a] = pts[i].c[a] > mx[a] ? pts[i].c[a] : mx[a];
            }
        for (int a = 0; a < 3; ++a) {
            out[node].box.lo[a] = mn[a];
            out[node].box.hi[a] = mx[a];
        }
        out[node].pad[0] = out[node].pad[1] = (T)0;
        int sd = 0;
        for (int a = 1; a < dim; ++a)
            if (mx[a] - mn[a] > mx[sd] - mn[sd]) sd = a;
        const int64_t L = kd_left_size(n);
        std::nth_element(pts.begin() + lo, pts.begin() + lo + L, pts.begin() + hi,
                         [sd](const HostPt<T>& a, const HostPt<T>& b) { return a.c[sd] < b.c[sd]; });
        const HostPt<T>& m = pts[lo + L];
        Pt<T> o;
        o.x = m.c[0];
        o.y = m.c[1];
        o.z = m.c[2];
        o.w = id_to_w((T)0, (int32_t)sd);
        out[node].p = o;
        kd_build_rec<T>(pts, lo, lo + L, 2 * node + 1, dim, out); // left: recursion depth = tree height
        lo = lo + L + 1;                                            // right: iterate
        node = 2 * node + 2;
    }
}

// Host build into `out`: m node records in heap order.
// Behind the m node records: the points once more, grouped by bucket.  A bucket is a subtree of at most kKdBucket nodes
// whose parent's subtree is larger; its root's record says where its points start and how many there are (pad[0], pad[1]:
// integers kept as bits).  A walk that wants a bucket evaluates all of its points from that run — one dependent fetch
// instead of a descent of four levels, node by node; next to a finely sampled wall that descent was most of the walk.
constexpr int kKdBucket = 15;
template <typename T> size_t kd_bytes(int64_t m) { return (sizeof(KdNode<T>) + sizeof(Pt<T>)) * (size_t)m; }

template <typename T> static void kd_store_int(T* slot, int64_t v) {
    using I = typename std::conditional<sizeof(T) == 4, int32_t, int64_t>::type;
    const I i = (I)v;
    memcpy(slot, &i, sizeof(T));
}
template <typename T> __host__ __device__ inline int32_t kd_load_int(T v) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(int32_t, v);
    else return (int32_t)__builtin_bit_cast(int64_t, v);
}

template <typename T> void kd_build_host(const T* xyz, int64_t m, int dim, void* out_raw) {
    KdNode<T>* out = (KdNode<T>*)out_raw;
    std::vector<HostPt<T>> pts((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        pts[i].c[0] = xyz[i * dim];
        pts[i].c[1] = xyz[i * dim + 1];
        pts[i].c[2] = dim == 3 ? xyz[i * dim + 2] : (T)0;
    }
    kd_build_rec<T>(pts, 0, m, 0, dim, out);
    // buckets: subtree sizes bottom-up, then every maximal subtree of at most kKdBucket nodes is copied into the run
    Pt<T>* packed = reinterpret_cast<Pt<T>*>(out + m);
    std::vector<int32_t> sz((size_t)m);
    for (int64_t i = m - 1; i >= 0; --i)
        sz[i] = 1 + (2 * i + 1 < m ? sz[2 * i + 1] : 0) + (2 * i + 2 < m ? sz[2 * i + 2] : 0);
    int64_t fill = 0;
    std::vector<int64_t> todo;
    for (int64_t i = 0; i < m; ++i) {
        kd_store_int<T>(&out[i].pad[0], 0);
        kd_store_int<T>(&out[i].pad[1], 0);
        if (sz[i] > kKdBucket || (i > 0 && sz[(i - 1) / 2] <= kKdBucket)) continue;
        kd_store_int<T>(&out[i].pad[0], fill);
        kd_store_int<T>(&out[i].pad[1], sz[i]);
        todo.assign(1, i);
        while (!todo.empty()) {
            const int64_t j = todo.back();
            todo.pop_back();
            Pt<T> q = out[j].p;
            q.w = id_to_w((T)0, (int32_t)j); // (the run keeps the node's index where the node keeps its split axis)
            packed[fill++] = q;
            if (2 * j + 2 < m) todo.push_back(2 * j + 2);
            if (2 * j + 1 < m) todo.push_back(2 * j + 1);
        }
    }
}

template <typename T>
__device__ inline T box_d2(const KdBox<T>& b, T qx, T qy, T qz) {
    const T q[3] = {qx, qy, qz};
    T t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T below = b.lo[a] - q[a], above = q[a] - b.hi[a];
        const T m = below > above ? below : above;
        t[a] = m > (T)0 ? m : (T)0;
    }
    return (t[0] * t[0] + t[1] * t[1]) + t[2] * t[2];
}

// Packet traversal: the 64 queries of a wave walk the tree TOGETHER.  `node` and the stack are
// wave-uniform (the stack is a 64-entry LDS row per wave), so every node and box is fetched once
// per wave through uniform loads
// instead of 64 divergent ones, and there is no lane divergence at all.  A subtree is entered
// when its box can still improve SOME lane's best; each lane keeps its own best over the canonical
// d2, so visiting extra nodes never changes a result.  For cell-sorted queries (session order) a
// wave's queries sit within a few cells of each other and the union of their search paths is
// hardly longer than one path.
// hint: a node believed to be near the query (its nearest node of the previous sweep) or -1; it
// only seeds `best`.  *best_node: the minimiser.  Inactive lanes (tail of the grid) never want a
// subtree.  Measured at 8 M queries x 46 786 boundary points: per-lane traversal (64 divergent
// walks per wave) 3.3 ms, packet traversal 1.0 ms.
template <typename T>
__device__ inline T kd_nearest_d2(const KdNode<T>* __restrict__ nodes, int32_t m, T qx, T qy, T qz, bool active,
                                  int32_t hint, int32_t* best_node, int32_t* __restrict__ stack /* LDS, this wave's */,
                                  T* other_lb = nullptr) {
    T best = active ? Lim<T>::inf() : (T)-1; // box_d2 >= 0 > -1: an inactive lane never asks for anything
    int32_t bn = -1;
    // *other_lb: a lower bound on the canonical d2 of every boundary point OTHER than the winner — the smaller of the
    // second-best point this lane evaluated and the boxes of the subtrees the wave skipped (each lane's own distance to
    // them).  The caller turns it into a certificate that the winner is still the nearest after the query has moved.
    T lb = Lim<T>::inf();
    if (active && hint >= 0 && hint < m) {
        const Pt<T> p = nodes[hint].p;
        best = dist2<T>(qx, qy, qz, p.x, p.y, p.z);
        bn = hint;
    }
    int sp = 0;        // uniform
    int32_t node = 0;  // uniform
#if WTP_DIAG
    int kd_steps_ = 0;
#endif
    for (;;) {
#if WTP_DIAG
        ++kd_steps_;
#endif
        node = __builtin_amdgcn_readfirstlane(node);
        const KdNode<T> nd = nodes[node]; // one uniform fetch per step
        const T bd2 = box_d2<T>(nd.box, qx, qy, qz);
        const bool want = bd2 < best;
        bool descended = false;
        if (!__any(want)) {
            lb = bd2 < lb ? bd2 : lb; // the whole subtree is skipped: nothing in it is nearer than its box
        } else if (kd_load_int<T>(nd.pad[1]) > 0) {
            // a bucket: all points of the subtree from their run, no descent
            const int32_t cnt = __builtin_amdgcn_readfirstlane(kd_load_int<T>(nd.pad[1]));
            const Pt<T>* run = reinterpret_cast<const Pt<T>*>(nodes + m) + __builtin_amdgcn_readfirstlane(kd_load_int<T>(nd.pad[0]));
            for (int e = 0; e < cnt; ++e) {
                const Pt<T> p = run[e];
                const int32_t pn = w_to_id(p.w);
                const T d2 = dist2<T>(qx, qy, qz, p.x, p.y, p.z);
                const bool better = active && d2 < best;
                if (pn != bn) {
                    const T loser = better ? best : d2;
                    lb = loser < lb ? loser : lb;
                }
                bn = better ? pn : bn;
                best = better ? d2 : best;
            }
        } else {
            const Pt<T> p = nd.p;
            const T d2 = dist2<T>(qx, qy, qz, p.x, p.y, p.z);
            const bool better = active && d2 < best;
            if (node != bn) { // (the hint's own node comes up again in the walk)
                const T loser = better ? best : d2;
                lb = loser < lb ? loser : lb;
            }
            bn = better ? node : bn;
            best = better ? d2 : best;
            const int32_t sd = w_to_id(p.w);
            const T diff = sd == 0 ? qx - p.x : (sd == 1 ? qy - p.y : qz - p.z);
            const int32_t left = 2 * node + 1;
            // the side most of the interested lanes lie on goes first
            const int nl = __popcll(__ballot(want && diff < (T)0)), nr = __popcll(__ballot(want && !(diff < (T)0)));
            const int32_t first = left + (nl >= nr ? 0 : 1), second = left + (nl >= nr ? 1 : 0);
            if (second < m && sp < 64) { // one push per level at most: 2^31 points have 31 levels
                if ((threadIdx.x & 63) == 0) stack[sp] = second;
                ++sp;
            }
            if (first < m) {
                node = first;
                descended = true;
            }
        }
        if (!descended) {
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
#if WTP_DIAG
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_kd_steps[0], (unsigned long long)kd_steps_);
        atomicAdd(&g_kd_steps[1], 1ull);
    }
#endif
    *best_node = bn;
    if (other_lb) *other_lb = lb;
    return best;
}

template <typename T> struct SpacingLaw {
    int32_t kind; // WTP_SPACING_LOGLIKE / WTP_SPACING_BOUNDARY_LAYER
    T p0, p1, p2;
};

__device__ inline float wexp(float x) { return expf(x); }
__device__ inline double wexp(double x) { return exp(x); }

template <typename T> __device__ inline T spacing_law(const SpacingLaw<T>& law, T d) {
    if (law.kind == WTP_SPACING_LOGLIKE) { // base_size * x / (a + x), a = base_size * (1 - (growth_rate - 1))
        const T inv_growth = (T)1 - (law.p1 - (T)1);
        const T a = law.p0 * inv_growth;
        return law.p0 * d / (a + d);
    }
    const T center = law.p2 / (T)2, width = law.p2 / (T)6; // at_wall + (bulk - at_wall) * sigma
    const T sig = (T)1 / ((T)1 + wexp(-(d - center) / width));
    return law.p0 + (law.p1 - law.p0) * sig;
}

// Raw AoS points -> spacing values (wtp_spacing_eval).
template <typename T>
__global__ void spacing_eval_kernel(const T* __restrict__ xyz, int64_t n, int dim, const KdNode<T>* __restrict__ nodes,
                                    int32_t m, SpacingLaw<T> law, T* __restrict__ out) {
    __shared__ int32_t kd_stack[kSpThreads / 64][64];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t span = (n + 63) / 64 * 64; // whole waves walk the tree together
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += stride) {
        const bool active = i < n;
        const int64_t ii = active ? i : n - 1;
        const T x = xyz[ii * dim], y = xyz[ii * dim + 1], z = dim == 3 ? xyz[ii * dim + 2] : (T)0;
        int32_t bn;
        const T d2 = kd_nearest_d2<T>(nodes, m, x, y, z, active, -1, &bn, kd_stack[threadIdx.x >> 6]);
        if (active) out[i] = spacing_law<T>(law, wsqrt(d2));
    }
}

// Session points (slot order, id in w) -> spacing_pp[id]; ids below first_id keep their value.
//
// Which 64 points walk the tree together decides how long the walk is (a wave visits the union of what its lanes
// want).  Consecutive slots are one x-row of cells — with the round-2 sweep's one-point cells a wave would be a line
// 64 cells long, and next to a finely sampled wall the union of 64 such searches is several hundred nodes.  With
// the grid at hand the wave takes a compact tile instead: W x H x H cells, as cubic as the mean occupancy allows and
// ~56 points, its row segments (one per lane) enumerated through cell_start.  The grouping only affects speed: every lane still gets the
// minimum over ALL boundary points.
template <typename T>
__global__ void spacing_session_kernel(const Pt<T>* __restrict__ pts, int64_t n, int32_t first_id,
                                       const KdNode<T>* __restrict__ nodes, int32_t m, SpacingLaw<T> law,
                                       T* __restrict__ spacing_pp, int32_t* __restrict__ hint,
                                       const int32_t* __restrict__ stop, const int32_t* __restrict__ cell_start,
                                       const Grid<T>* __restrict__ gp, Pt<T>* __restrict__ cert, int queue_on) {
    __shared__ int32_t kd_stack[kSpThreads / 64][64];
    __shared__ int32_t kd_queue[kSpThreads / 64][128];
    if (stop && *stop) return;
    // Most walks can be skipped.  A walk leaves, per point, where it stood (x_ref) and a lower bound lb on the distance
    // to every boundary point other than the winner b.  After the point has moved to x, any other boundary point q has
    // |x - q| >= |x_ref - q| - |x - x_ref| >= sqrt(lb) - |x - x_ref|; if that still exceeds |x - b|, b is the nearest
    // point of x as well and the minimum is |x - b|^2 — the very expression the walk would return, bit for bit.  Points
    // move a fraction of a spacing per sweep and ever less as the cloud relaxes; only the lanes whose certificate
    // fails walk (a wave of fewer interested lanes visits fewer nodes).
    const T eps = sizeof(T) == 4 ? (T)1e-6 : (T)1e-14;
    int32_t* queue = nullptr; // this wave's queue of slots that must search (128 entries: flushed whenever 64 have gathered)
    int qn = 0;
    // the search proper for up to 64 queued (or, without the queue, current) points: packet walk, results, new certificates
    auto search = [&](int64_t slot, bool need) {
        const Pt<T> p = pts[need ? slot : 0];
        const int32_t id = w_to_id(p.w);
        const int32_t h = need ? hint[id] : -1;
        int32_t bn;
        T lb;
        const T d2 = kd_nearest_d2<T>(nodes, m, p.x, p.y, p.z, need, need ? h : -1, &bn, kd_stack[threadIdx.x >> 6], &lb);
        if (need) {
            spacing_pp[id] = spacing_law<T>(law, wsqrt(d2));
            hint[id] = bn; // points move a fraction of a spacing per sweep: next time this is (nearly) the answer
            if (cert) {
                Pt<T> c;
                c.x = p.x;
                c.y = p.y;
                c.z = p.z;
                c.w = lb;
                cert[id] = c;
            }
        }
    };
    auto one = [&](int64_t slot, bool on) {
        const Pt<T> p = pts[on ? slot : 0];
        const int32_t id = w_to_id(p.w);
        const bool active = on && id >= first_id;
        bool need = active;
        int32_t h = -1;
        if (active) {
            h = hint[id];
            if (cert && h >= 0 && h < m) {
                const Pt<T> c = cert[id];
                if (c.w > (T)0) { // (never walked: the bits of -1)
                    const Pt<T> b = nodes[h].p;
                    const T d2b = dist2<T>(p.x, p.y, p.z, b.x, b.y, b.z);
                    const T moved = wsqrt(dist2<T>(p.x, p.y, p.z, c.x, c.y, c.z));
                    if ((wsqrt(d2b) + moved) * ((T)1 + eps) < wsqrt(c.w) * ((T)1 - eps)) {
                        need = false;
                        spacing_pp[id] = spacing_law<T>(law, wsqrt(d2b));
                    }
                }
            }
        }
        if (!__any(need)) return;
        if (queue) {
            // The points that must search are queued (their slots) and searched 64 at a time: a tile of 56 points walks as a
            // whole if ONE of them must, and in a relaxing cloud one usually must (far from a finely sampled wall the two
            // nearest boundary points are nearly equidistant, so that point's certificate fails whenever it moves).
            const unsigned long long mk = __ballot(need);
            const int lane_ = threadIdx.x & 63;
            if (need) queue[qn + __popcll(mk & ((1ull << lane_) - 1ull))] = (int32_t)slot;
            qn += __popcll(mk);
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64) {
                qn -= 64;
                search(queue[qn + lane_], true);
            }
            return;
        }
        search(slot, need);
    };

    if (!cell_start) { // no grid yet (session setup): slot order
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        const int64_t span = (n + 63) / 64 * 64; // whole waves walk the tree together
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += stride) one(i, i < n);
        return;
    }
    queue = queue_on ? kd_queue[threadIdx.x >> 6] : nullptr;
    const Grid<T> g = *gp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // tile = W x H x Hz cells, as cubic as the occupancy allows and ~56 points: up to 64 row segments, one per lane
    // (points per tile: was one batch of 56; with the queue a tile only orders the certificate checks, and fewer, larger tiles
    // mean fewer chains of cell-table fetches — 112 … 448 measured within 3 % of each other at 10 M points, 160 kept)
    constexpr float kSpTilePts = 160.f;
    const float rho0 = (float)g.npts / (float)(g.ncells > 0 ? g.ncells : 1);
    const float rho = rho0 > 0.125f ? rho0 : 0.125f;
    const bool flat = g.n[2] <= 1;
    int H = (int)(flat ? sqrtf(kSpTilePts / rho) : cbrtf(kSpTilePts / rho) + 0.5f);
    H = H < 1 ? 1 : (H > 8 ? 8 : H);
    const int Hz = flat ? 1 : H;
    int W = (int)(kSpTilePts / (rho * (float)(H * Hz)) + 0.5f);
    W = W < 1 ? 1 : (W > 32 ? 32 : W);
    const int nrow = H * Hz; // <= 64
    const int tx_n = (g.n[0] + W - 1) / W, ty_n = (g.n[1] + H - 1) / H, tz_n = (g.n[2] + Hz - 1) / Hz;
    const int64_t ntiles = (int64_t)tx_n * ty_n * tz_n;
    const int64_t wave_g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    __shared__ int32_t row_end[kSpThreads / 64][64], row_start[kSpThreads / 64][64];
    // a wave takes CONSECUTIVE tiles (neighbours along x): what it queues for one search then lies close together
    const int64_t per_wave = (ntiles + nwaves - 1) / nwaves;
    const int64_t t_end = (wave_g + 1) * per_wave < ntiles ? (wave_g + 1) * per_wave : ntiles;
    for (int64_t t = wave_g * per_wave; t < t_end; ++t) {
        const int tx = (int)(t % tx_n), ty = (int)((t / tx_n) % ty_n), tz = (int)(t / ((int64_t)tx_n * ty_n));
        const int x0 = tx * W, x1 = (x0 + W) < g.n[0] ? (x0 + W) : g.n[0];
        // lane r < nrow: its row segment of the tile
        const int y = ty * H + lane % H, z = tz * Hz + lane / H;
        const bool in = lane < nrow && y < g.n[1] && z < g.n[2];
        const int base = in ? (z * g.n[1] + y) * g.n[0] : 0;
        const int s0 = cell_start[base + (in ? x0 : 0)];
        const int len = in ? cell_start[base + x1] - s0 : 0;
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        __builtin_amdgcn_wave_barrier();
        row_end[wave][lane] = incl;
        row_start[wave][lane] = s0 - (incl - len); // slot = row_start + tile-local index
        __builtin_amdgcn_wave_barrier();
        for (int b = 0; b < total; b += 64) {
            const int i = b + lane;
            const bool on = i < total;
            // first row whose inclusive end exceeds i
            int lo = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) lo += (row_end[wave][lo + step - 1] <= i) ? step : 0;
            lo = lo > 63 ? 63 : lo;
            one((int64_t)row_start[wave][lo] + i, on);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (queue && qn > 0) { // what is left in the queue (fewer than 64)
        const int lane_ = threadIdx.x & 63;
        search(queue[lane_ < qn ? lane_ : 0], lane_ < qn);
    }
}

static int sp_grid(int64_t n) {
    int64_t b = (n + kSpThreads - 1) / kSpThreads;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

template <typename T>
int launch_spacing_eval(wtp_ctx* ctx, const T* d_xyz, int64_t n, int dim, const void* d_nodes, int64_t m, int kind,
                        double p0, double p1, double p2, T* d_out) {
    SpacingLaw<T> law{kind, (T)p0, (T)p1, (T)p2};
    hipLaunchKernelGGL(spacing_eval_kernel<T>, dim3(sp_grid(n)), dim3(kSpThreads), 0, ctx->stream, d_xyz, n, dim,
                       (const KdNode<T>*)d_nodes, (int32_t)m, law, d_out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T>
int launch_spacing_session(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t first_id, const void* d_nodes, int64_t m,
                           int kind, double p0, double p1, double p2, T* d_spacing_pp, int32_t* d_hint,
                           const int32_t* d_cell_start, const void* d_grid, void* d_cert) {
    SpacingLaw<T> law{kind, (T)p0, (T)p1, (T)p2};
    hipLaunchKernelGGL(spacing_session_kernel<T>, dim3(sp_grid(n)), dim3(kSpThreads), 0, ctx->stream, pts, n,
                       (int32_t)first_id, (const KdNode<T>*)d_nodes, (int32_t)m, law, d_spacing_pp, d_hint, ctx->stop_dev,
                       d_cell_start, (const Grid<T>*)d_grid, (Pt<T>*)d_cert, getenv("WTP_SP_QUEUE") ? atoi(getenv("WTP_SP_QUEUE")) : 1);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int debug_kd_steps(unsigned long long out[2]) {
#if WTP_DIAG
    unsigned long long z[2] = {0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kd_steps), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_kd_steps), z, sizeof(z)) != hipSuccess;
#else
    out[0] = out[1] = 0;
    return 0;
#endif
}

#define INST(T)                                                                                              \
    template size_t kd_bytes<T>(int64_t);                                                                    \
    template void kd_build_host<T>(const T*, int64_t, int, void*);                                           \
    template int launch_spacing_eval<T>(wtp_ctx*, const T*, int64_t, int, const void*, int64_t, int, double, \
                                        double, double, T*);                                                 \
    template int launch_spacing_session<T>(wtp_ctx*, const Pt<T>*, int64_t, int64_t, const void*, int64_t, int, \
                                           double, double, double, T*, int32_t*, const int32_t*, const void*, void*);
INST(float)
INST(double)
#undef INST

} // namespace wtp
