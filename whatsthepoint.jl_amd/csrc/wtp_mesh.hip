// wtp_mesh.hip — the triangle-mesh geometry index behind the octree method of repel.
//
//   repel(cloud, spacing, octree)            src/repel.jl:122-181: every point moves; after each sweep
//   _constrain_octree                        :448-469   boundary points are re-projected onto the mesh,
//                                                       volume points that left the domain go back
//   _project_to_boundary                     :522-537   nearest triangle, closest point, nudged inward
//   isinside(p, octree) / classify_point     src/octree/triangle_octree.jl:71-99
//   _compute_signed_distance_octree          :583-607   sign from the angle-weighted pseudonormal of
//                                                       the closest feature (face / edge / vertex)
//   TriangleIndex                            :221-277   face normals + pseudonormals keyed by exact
//                                                       coordinates (triangle soup needs no welding)
//   closest_point_on_triangle_feature        src/octree/geometric_utils.jl:68-136
//
// The reference walks an octree of triangle lists per query.  Here the mesh is static for a whole
// repel call, so it gets a left-balanced bounding-volume tree built once on the host: heap order
// (children of node i are 2i+1, 2i+2), one triangle per node (the median of its subtree along the
// widest axis of the centroids), and the box of everything below it.  One 64-byte record per node
// (fp32; 128 B fp64) = one wave-uniform scalar fetch per traversal step.  The 64 queries of a wave
// walk the tree together (cell-sorted session points sit next to each other, so the union of their
// paths is hardly longer than one path; see wtp_spacing.hip for the measurement that motivated it).
//
// Exactness: every lane keeps the minimum of (d2, triangle index) over the triangles it evaluated,
// d2 computed exactly as the oracle computes it (no contraction).  A subtree is skipped only when
// its box is farther than the lane's best by more than the rounding slack of a computed closest
// point (64 eps x the coordinate scale), so the result is the brute-force minimum.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "wtp_device.hpp"

namespace wtp {

static constexpr int kMeshThreads = 256;

template <typename T> struct MeshNode {
    T v[9];           // the node's triangle
    T lo[3], hi[3];   // box of the whole subtree (own triangle included)
    int32_t tri_axis; // triangle index | split axis << 30
};
static_assert(sizeof(MeshNode<float>) == 64, "one cache line per node");
static_assert(sizeof(MeshNode<double>) == 128, "two cache lines per node");

// ---- host build -----------------------------------------------------------------------------------
template <typename T> struct EpsOf;
template <> struct EpsOf<float> { static constexpr float v = 1.1920928955078125e-07f; };
template <> struct EpsOf<double> { static constexpr double v = 2.220446049250313e-16; };

struct KeyHash {
    size_t operator()(const std::array<uint64_t, 6>& k) const {
        uint64_t h = 1469598103934665603ull;
        for (uint64_t w : k) {
            h ^= w;
            h *= 1099511628211ull;
            h ^= h >> 29;
        }
        return (size_t)h;
    }
};

template <typename T> static uint64_t bits_of(T x) {
    uint64_t b = 0;
    memcpy(&b, &x, sizeof(T));
    return b;
}

template <typename T> static T hdot(const T* a, const T* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

template <typename T> static bool lex_less(const T* a, const T* b) {
    for (int i = 0; i < 3; ++i) {
        if (a[i] < b[i]) return true;
        if (a[i] > b[i]) return false;
    }
    return false;
}

template <typename T> static T corner_angle(const T* vc, const T* va, const T* vb) {
    T u[3], w[3];
    for (int i = 0; i < 3; ++i) {
        u[i] = va[i] - vc[i];
        w[i] = vb[i] - vc[i];
    }
    const T den = std::sqrt(hdot(u, u) * hdot(w, w));
    if (den < EpsOf<T>::v) return (T)0;
    T c = hdot(u, w) / den;
    c = c < (T)-1 ? (T)-1 : (c > (T)1 ? (T)1 : c);
    return std::acos(c);
}

// pn: nt x 7 x 3 = {face, vertex 1..3, edge 12, 13, 23}; sums run in triangle order like the
// reference's Dict updates (src/octree/triangle_octree.jl:255-268).
template <typename T> static void build_pseudonormals(const T* verts, const int32_t* tris, int64_t nt, T* pn) {
    using Key = std::array<uint64_t, 6>;
    using Vec = std::array<T, 3>;
    std::unordered_map<Key, Vec, KeyHash> vmap, emap;
    vmap.reserve((size_t)nt);
    emap.reserve((size_t)(2 * nt));
    auto vkey = [](const T* a) { return Key{bits_of(a[0]), bits_of(a[1]), bits_of(a[2]), 0, 0, 0}; };
    auto ekey = [](const T* a, const T* b) {
        const bool ab = lex_less(a, b);
        const T* p = ab ? a : b;
        const T* q = ab ? b : a;
        return Key{bits_of(p[0]), bits_of(p[1]), bits_of(p[2]), bits_of(q[0]), bits_of(q[1]), bits_of(q[2])};
    };
    static const int ea[3] = {0, 1, 2}, eb[3] = {1, 2, 0}; // (v1,v2), (v2,v3), (v3,v1)
    for (int64_t t = 0; t < nt; ++t) {
        const T* v[3] = {verts + 3 * (int64_t)tris[3 * t], verts + 3 * (int64_t)tris[3 * t + 1],
                         verts + 3 * (int64_t)tris[3 * t + 2]};
        T e1[3], e2[3], nr[3];
        for (int i = 0; i < 3; ++i) {
            e1[i] = v[1][i] - v[0][i];
            e2[i] = v[2][i] - v[0][i];
        }
        nr[0] = e1[1] * e2[2] - e1[2] * e2[1];
        nr[1] = e1[2] * e2[0] - e1[0] * e2[2];
        nr[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const T mag = std::sqrt(hdot(nr, nr));
        T* f = pn + 21 * t;
        for (int i = 0; i < 3; ++i) f[i] = mag < EpsOf<T>::v * 100 ? (T)0 : nr[i] / mag;
        for (int s = 0; s < 3; ++s) {
            Vec& e = emap.try_emplace(ekey(v[ea[s]], v[eb[s]]), Vec{0, 0, 0}).first->second;
            for (int i = 0; i < 3; ++i) e[i] = e[i] + f[i];
        }
        for (int s = 0; s < 3; ++s) {
            const T ang = corner_angle(v[s], v[(s + 1) % 3], v[(s + 2) % 3]);
            Vec& q = vmap.try_emplace(vkey(v[s]), Vec{0, 0, 0}).first->second;
            for (int i = 0; i < 3; ++i) q[i] = q[i] + ang * f[i];
        }
    }
    static const int efeat[3] = {4, 6, 5}; // slots (12), (23), (31) -> features e12, e23, e13
    for (int64_t t = 0; t < nt; ++t) {
        const T* v[3] = {verts + 3 * (int64_t)tris[3 * t], verts + 3 * (int64_t)tris[3 * t + 1],
                         verts + 3 * (int64_t)tris[3 * t + 2]};
        for (int s = 0; s < 3; ++s) {
            const Vec& q = vmap[vkey(v[s])];
            const Vec& e = emap[ekey(v[ea[s]], v[eb[s]])];
            for (int i = 0; i < 3; ++i) {
                pn[21 * t + 3 * (1 + s) + i] = q[i];
                pn[21 * t + 3 * efeat[s] + i] = e[i];
            }
        }
    }
}

static int64_t bvh_left_size(int64_t n) { // nodes in the left subtree of a left-balanced tree of n
    if (n <= 1) return 0;
    int h = 0;
    while ((int64_t(1) << (h + 1)) <= n) ++h;
    const int64_t full = (int64_t(1) << h) - 1, last = n - full, half = int64_t(1) << (h - 1);
    return (full - 1) / 2 + (last < half ? last : half);
}

template <typename T> struct TriItem {
    T c[3];          // centroid (ordering only)
    T lo[3], hi[3];  // extents
    int32_t tri;
};

template <typename T>
static void bvh_build_rec(std::vector<TriItem<T>>& it, int64_t lo, int64_t hi, int64_t node, const T* verts,
                          const int32_t* tris, MeshNode<T>* out) {
    while (hi > lo) {
        T mn[3], mx[3], cmn[3], cmx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = it[lo].lo[a];
            mx[a] = it[lo].hi[a];
            cmn[a] = cmx[a] = it[lo].c[a];
        }
        for (int64_t i = lo + 1; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], it[i].lo[a]);
                mx[a] = std::max(mx[a], it[i].hi[a]);
                cmn[a] = std::min(cmn[a], it[i].c[a]);
                cmx[a] = std::max(cmx[a], it[i].c[a]);
            }
        int sd = 0;
        for (int a = 1; a < 3; ++a)
            if (cmx[a] - cmn[a] > cmx[sd] - cmn[sd]) sd = a;
        const int64_t L = bvh_left_size(hi - lo);
        std::nth_element(it.begin() + lo, it.begin() + lo + L, it.begin() + hi,
                         [sd](const TriItem<T>& a, const TriItem<T>& b) { return a.c[sd] < b.c[sd]; });
        const TriItem<T>& m = it[lo + L];
        MeshNode<T>& nd = out[node];
        for (int c = 0; c < 3; ++c)
            for (int a = 0; a < 3; ++a) nd.v[3 * c + a] = verts[3 * (int64_t)tris[3 * (int64_t)m.tri + c] + a];
        for (int a = 0; a < 3; ++a) {
            nd.lo[a] = mn[a];
            nd.hi[a] = mx[a];
        }
        nd.tri_axis = m.tri | (sd << 30);
        bvh_build_rec<T>(it, lo, lo + L, 2 * node + 1, verts, tris, out);
        lo = lo + L + 1;
        node = 2 * node + 2;
    }
}

template <typename T>
static void mesh_build_host(const T* verts, const int32_t* tris, int64_t nt, MeshNode<T>* nodes, T* pn, double bbox[6],
                            int64_t nv) {
    build_pseudonormals<T>(verts, tris, nt, pn);
    std::vector<TriItem<T>> it((size_t)nt);
    for (int64_t t = 0; t < nt; ++t) {
        TriItem<T>& x = it[t];
        x.tri = (int32_t)t;
        for (int a = 0; a < 3; ++a) {
            const T p0 = verts[3 * (int64_t)tris[3 * t] + a], p1 = verts[3 * (int64_t)tris[3 * t + 1] + a],
                    p2 = verts[3 * (int64_t)tris[3 * t + 2] + a];
            x.lo[a] = std::min(p0, std::min(p1, p2));
            x.hi[a] = std::max(p0, std::max(p1, p2));
            x.c[a] = (p0 + p1 + p2) / (T)3;
        }
    }
    bvh_build_rec<T>(it, 0, nt, 0, verts, tris, nodes);
    // _compute_bbox_raw (src/octree/triangle_octree.jl:279-291): over ALL vertices, flat axes widened
    T mn[3], mx[3];
    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = verts[a];
    for (int64_t i = 1; i < nv; ++i)
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], verts[3 * i + a]);
            mx[a] = std::max(mx[a], verts[3 * i + a]);
        }
    const T e = std::max(EpsOf<T>::v * 100, (T)1.0e-10);
    for (int a = 0; a < 3; ++a) {
        if (mn[a] == mx[a]) {
            mn[a] -= e;
            mx[a] += e;
        }
        bbox[a] = (double)mn[a];
        bbox[3 + a] = (double)mx[a];
    }
}

// ---- device ---------------------------------------------------------------------------------------
template <typename T> __device__ inline T ddot(const T* a, const T* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// closest_point_on_triangle_feature (src/octree/geometric_utils.jl:68-136), expression for
// expression what oracle/wtp_oracle_impl.h tri_closest evaluates.
template <typename T>
__device__ inline int tri_closest(const T* p, const T* a, const T* b, const T* c, T* out) {
    T ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ab[i] = b[i] - a[i];
        ac[i] = c[i] - a[i];
        ap[i] = p[i] - a[i];
    }
    const T d1 = ddot(ab, ap), d2 = ddot(ac, ap);
    if (d1 <= (T)0 && d2 <= (T)0) {
        out[0] = a[0], out[1] = a[1], out[2] = a[2];
        return 1;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) bp[i] = p[i] - b[i];
    const T d3 = ddot(ab, bp), d4 = ddot(ac, bp);
    if (d3 >= (T)0 && d4 <= d3) {
        out[0] = b[0], out[1] = b[1], out[2] = b[2];
        return 2;
    }
    const T vc = d1 * d4 - d3 * d2;
    if (vc <= (T)0 && d1 >= (T)0 && d3 <= (T)0) {
        const T v = d1 / (d1 - d3);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = a[i] + v * ab[i];
        return 4;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) cp[i] = p[i] - c[i];
    const T d5 = ddot(ab, cp), d6 = ddot(ac, cp);
    if (d6 >= (T)0 && d5 <= d6) {
        out[0] = c[0], out[1] = c[1], out[2] = c[2];
        return 3;
    }
    const T vb = d5 * d2 - d1 * d6;
    if (vb <= (T)0 && d2 >= (T)0 && d6 <= (T)0) {
        const T w = d2 / (d2 - d6);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = a[i] + w * ac[i];
        return 5;
    }
    const T va = d3 * d6 - d5 * d4;
    if (va <= (T)0 && (d4 - d3) >= (T)0 && (d5 - d6) >= (T)0) {
        const T w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = b[i] + w * (c[i] - b[i]);
        return 6;
    }
    const T denom = (T)1 / ((va + vb) + vc);
    const T v = vb * denom, w = vc * denom;
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = (a[i] + ab[i] * v) + ac[i] * w;
    return 0;
}

#ifdef WTP_MESH_COUNT
__device__ unsigned long long g_mesh_count[4];
extern "C" int wtp_mesh_count(unsigned long long out[4]) {
    unsigned long long z[4] = {0, 0, 0, 0};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mesh_count), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(g_mesh_count), z, sizeof(z));
    return 0;
}
#endif

template <typename T> struct Nearest {
    T d2;
    T cp[3];
    int32_t tri, feat;
};

template <typename T> __device__ inline T prune_limit(T best, T delta) {
    // (sqrt(best) + 2 delta)^2: a box farther than this cannot hold a triangle whose COMPUTED d2 ties or beats best
    return best + ((T)4 * delta) * (wsqrt(best) + delta);
}

// squared distance from q to the bounding box of triangle v[9] (a lower bound of the distance to it)
template <typename T> __device__ inline T tri_box_d2(const T* v, const T* q) {
    T t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        T lo = v[a] < v[3 + a] ? v[a] : v[3 + a], hi = v[a] < v[3 + a] ? v[3 + a] : v[a];
        lo = lo < v[6 + a] ? lo : v[6 + a];
        hi = hi > v[6 + a] ? hi : v[6 + a];
        const T below = lo - q[a], above = q[a] - hi;
        const T mm = below > above ? below : above;
        t[a] = mm > (T)0 ? mm : (T)0;
    }
    return (t[0] * t[0] + t[1] * t[1]) + t[2] * t[2];
}

// next node in pre-order after skipping the subtree of i (1-based heap index of a left-balanced tree of
// m nodes); 1 = walk finished
__device__ inline uint32_t heap_escape(uint32_t i, uint32_t m) {
    do {
        const uint32_t j = i + 1;
        i = j >> __builtin_ctz(j);
    } while (i > m);
    return i;
}

// A first guess for a query without history: walk down from the root into the child whose box is
// nearer (ties: left), ~log2(m) steps; the node reached holds a triangle close to the query.
template <typename T>
__device__ inline int32_t mesh_greedy_guess(const MeshNode<T>* __restrict__ nodes, int32_t m, const T* q) {
    uint32_t i = 1;
    for (;;) {
        const uint32_t l = 2 * i, rr = 2 * i + 1;
        if (l > (uint32_t)m) break;
        if (rr > (uint32_t)m) {
            i = l;
            break;
        }
        T dl, dr;
        {
            const MeshNode<T>& a = nodes[l - 1];
            const MeshNode<T>& b = nodes[rr - 1];
            T t[3], u[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T b1 = a.lo[c] - q[c], a1 = q[c] - a.hi[c];
                const T m1 = b1 > a1 ? b1 : a1;
                t[c] = m1 > (T)0 ? m1 : (T)0;
                const T b2 = b.lo[c] - q[c], a2 = q[c] - b.hi[c];
                const T m2 = b2 > a2 ? b2 : a2;
                u[c] = m2 > (T)0 ? m2 : (T)0;
            }
            dl = (t[0] * t[0] + t[1] * t[1]) + t[2] * t[2];
            dr = (u[0] * u[0] + u[1] * u[1]) + u[2] * u[2];
        }
        i = dr < dl ? rr : l;
    }
    return (int32_t)(i - 1);
}

// Per-lane walk for a query that brings a good first guess (the tree node of its nearest triangle at the
// last sweep).  The packet walk below visits the UNION of what 64 lanes want — measured 98 node steps
// per wave on the box surface although a lane with a tight bound wants ~30 — so with a guess every lane
// walks alone: stackless (pre-order with arithmetic skips in the heap-ordered tree), own triangle of a
// node evaluated only if its bbox is within the bound.  Same candidates, same (d2, index) minimum.
template <typename T>
__device__ inline Nearest<T> mesh_nearest_guess(const MeshNode<T>* __restrict__ nodes, int32_t m, const T* q, bool active,
                                                T scale, int32_t guess, int32_t* best_node) {
    Nearest<T> r;
    r.d2 = Lim<T>::inf();
    r.cp[0] = q[0], r.cp[1] = q[1], r.cp[2] = q[2];
    r.tri = -1;
    r.feat = 0;
    int32_t bn = -1;
    if (active) {
        T aq = q[0] < 0 ? -q[0] : q[0];
        const T ay = q[1] < 0 ? -q[1] : q[1], az = q[2] < 0 ? -q[2] : q[2];
        aq = aq > ay ? aq : ay;
        aq = aq > az ? aq : az;
        const T delta = (T)64 * EpsOf<T>::v * (aq > scale ? aq : scale);
        T limit;
        int32_t guess2 = -1;
        {
            T v[9], cp[3], dv[3];
#pragma unroll
            for (int a = 0; a < 9; ++a) v[a] = nodes[guess].v[a];
            r.feat = tri_closest<T>(q, v, v + 3, v + 6, cp);
#pragma unroll
            for (int a = 0; a < 3; ++a) dv[a] = q[a] - cp[a];
            r.d2 = ddot(dv, dv);
            r.tri = nodes[guess].tri_axis & 0x3fffffff;
            r.cp[0] = cp[0], r.cp[1] = cp[1], r.cp[2] = cp[2];
            bn = guess;
            if (!(r.d2 == r.d2)) { // a degenerate triangle (0/0 in its edge parameter): no candidate, like `d2 < best`
                r.d2 = Lim<T>::inf();
                r.tri = -1;
                r.feat = 0;
                r.cp[0] = q[0], r.cp[1] = q[1], r.cp[2] = q[2];
                bn = -1;
            }
            // a guess farther away than its own triangle is wide is a poor one (the point moved a long
            // way, e.g. a spacing far coarser than the tessellation): look for a second one from the root
            T e2 = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                T lo = v[a] < v[3 + a] ? v[a] : v[3 + a], hi = v[a] < v[3 + a] ? v[3 + a] : v[a];
                lo = lo < v[6 + a] ? lo : v[6 + a];
                hi = hi > v[6 + a] ? hi : v[6 + a];
                e2 = e2 + (hi - lo) * (hi - lo);
            }
            if (r.d2 > e2 || r.tri < 0) {
                const int32_t g2 = mesh_greedy_guess<T>(nodes, m, q);
#pragma unroll
                for (int a = 0; a < 9; ++a) v[a] = nodes[g2].v[a];
                const int f2 = tri_closest<T>(q, v, v + 3, v + 6, cp);
#pragma unroll
                for (int a = 0; a < 3; ++a) dv[a] = q[a] - cp[a];
                const T d2 = ddot(dv, dv);
                const int32_t t2 = nodes[g2].tri_axis & 0x3fffffff;
                if (d2 < r.d2 || (d2 == r.d2 && t2 < r.tri)) {
                    r.d2 = d2;
                    r.tri = t2;
                    r.feat = f2;
                    r.cp[0] = cp[0], r.cp[1] = cp[1], r.cp[2] = cp[2];
                    bn = g2;
                }
                guess2 = g2;
            }
            limit = prune_limit<T>(r.d2, delta);
        }
        uint32_t i = 1;
        do {
            const MeshNode<T>& nd = nodes[i - 1];
            T t[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const T below = nd.lo[a] - q[a], above = q[a] - nd.hi[a];
                const T mm = below > above ? below : above;
                t[a] = mm > (T)0 ? mm : (T)0;
            }
            if (((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]) <= limit) {
                T v[9];
#pragma unroll
                for (int a = 0; a < 9; ++a) v[a] = nd.v[a];
                if ((int32_t)(i - 1) != guess && (int32_t)(i - 1) != guess2 && tri_box_d2<T>(v, q) <= limit) {
                    T cp[3], dv[3];
                    const int f = tri_closest<T>(q, v, v + 3, v + 6, cp);
#pragma unroll
                    for (int a = 0; a < 3; ++a) dv[a] = q[a] - cp[a];
                    const T d2 = ddot(dv, dv);
                    const int32_t tri = nd.tri_axis & 0x3fffffff;
                    if (d2 < r.d2 || (d2 == r.d2 && tri < r.tri)) {
                        r.d2 = d2;
                        r.tri = tri;
                        r.feat = f;
                        r.cp[0] = cp[0], r.cp[1] = cp[1], r.cp[2] = cp[2];
                        bn = (int32_t)(i - 1);
                        limit = prune_limit<T>(d2, delta);
                    }
                }
                i = 2 * i <= (uint32_t)m ? 2 * i : heap_escape(i, (uint32_t)m);
            } else {
                i = heap_escape(i, (uint32_t)m);
            }
        } while (i != 1);
    }
    *best_node = bn;
    return r;
}

// Packet traversal (all 64 lanes of the wave call this together; inactive lanes never ask for a subtree).
template <typename T>
__device__ inline Nearest<T> mesh_nearest(const MeshNode<T>* __restrict__ nodes, int32_t m, const T* q, bool active,
                                          T scale, int32_t* __restrict__ stack /* LDS row of this wave */,
                                          int32_t hint = -1, int32_t* best_node = nullptr) {
    Nearest<T> r;
    r.d2 = Lim<T>::inf();
    r.cp[0] = q[0], r.cp[1] = q[1], r.cp[2] = q[2];
    r.tri = -1;
    r.feat = 0;
    int32_t bn = -1;
    T aq = q[0] < 0 ? -q[0] : q[0];
    const T ay = q[1] < 0 ? -q[1] : q[1], az = q[2] < 0 ? -q[2] : q[2];
    aq = aq > ay ? aq : ay;
    aq = aq > az ? aq : az;
    const T delta = (T)64 * EpsOf<T>::v * (aq > scale ? aq : scale);
    T limit = active ? Lim<T>::inf() : (T)-1; // box distance >= 0 > -1
    if (active && hint >= 0 && hint < m) { // the nearest triangle of the last sweep: a point moves a fraction of a
        const MeshNode<T>& hn = nodes[hint]; // spacing per sweep, so this already prunes nearly every subtree
        T v[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) v[a] = hn.v[a];
        T cp[3], dv[3];
        r.feat = tri_closest<T>(q, v, v + 3, v + 6, cp);
#pragma unroll
        for (int a = 0; a < 3; ++a) dv[a] = q[a] - cp[a];
        r.d2 = ddot(dv, dv);
        r.tri = hn.tri_axis & 0x3fffffff;
        r.cp[0] = cp[0], r.cp[1] = cp[1], r.cp[2] = cp[2];
        bn = hint;
        if (!(r.d2 == r.d2)) { // degenerate triangle: no candidate
            r.d2 = Lim<T>::inf();
            r.tri = -1;
            bn = -1;
        }
        limit = prune_limit<T>(r.d2, delta);
    }
    int sp = 0;
    int32_t node = 0;
#ifdef WTP_MESH_COUNT
    int steps = 0, evals = 0;
#endif
    for (;;) {
        node = __builtin_amdgcn_readfirstlane(node);
        const MeshNode<T> nd = nodes[node];
#ifdef WTP_MESH_COUNT
        ++steps;
#endif
        T t[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T below = nd.lo[a] - q[a], above = q[a] - nd.hi[a];
            const T mm = below > above ? below : above;
            t[a] = mm > (T)0 ? mm : (T)0;
        }
        const bool want = ((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]) <= limit;
        bool descended = false;
        if (__any(want)) {
#ifdef WTP_MESH_COUNT
            ++evals;
#endif
            const int32_t tri = nd.tri_axis & 0x3fffffff, sd = (nd.tri_axis >> 30) & 3;
            if (want && tri_box_d2<T>(nd.v, q) <= limit) { // the node's own triangle may be far although its subtree is near
                T cp[3], dv[3];
                const int f = tri_closest<T>(q, nd.v, nd.v + 3, nd.v + 6, cp);
#pragma unroll
                for (int a = 0; a < 3; ++a) dv[a] = q[a] - cp[a];
                const T d2 = ddot(dv, dv);
                if (d2 < r.d2 || (d2 == r.d2 && tri < r.tri)) {
                    r.d2 = d2;
                    r.tri = tri;
                    r.feat = f;
                    r.cp[0] = cp[0], r.cp[1] = cp[1], r.cp[2] = cp[2];
                    bn = node;
                    limit = prune_limit<T>(d2, delta);
                }
            }
            const T cen = ((nd.v[sd] + nd.v[3 + sd]) + nd.v[6 + sd]) / (T)3;
            const bool lft = q[sd] < cen;
            const int32_t left = 2 * node + 1;
            const int nl = __popcll(__ballot(want && lft)), nr = __popcll(__ballot(want && !lft));
            const int32_t first = left + (nl >= nr ? 0 : 1), second = left + (nl >= nr ? 1 : 0);
            if (second < m && sp < 64) {
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
#ifdef WTP_MESH_COUNT
    if (best_node && (threadIdx.x & 63) == 0) {
        atomicAdd(g_mesh_count, (unsigned long long)steps);
        atomicAdd(g_mesh_count + 1, (unsigned long long)evals);
        atomicAdd(g_mesh_count + 2, 1ull);
        atomicMax(g_mesh_count + 3, (unsigned long long)steps);
    }
#endif
    if (best_node) *best_node = bn;
    return r;
}

template <typename T> struct MeshView {
    const MeshNode<T>* nodes;
    const T* pn; // nt x 7 x 3
    int32_t m;
    T lo[3], hi[3]; // vertex bbox (classify_point's fast path)
    T scale;        // largest |coordinate| of the mesh
    // inside/outside class per cell of a uniform grid over the bbox (the role of the reference's
    // leaf_classification cache, src/octree/triangle_octree.jl:84-89): 0 = near the surface (exact
    // test), 1 = wholly inside, 2 = wholly outside.  nullptr: every point takes the exact test.
    const uint8_t* cls;
    int32_t cdim[3];
    T cinv; // 1 / cell edge
};

enum : uint8_t { CLS_BOUNDARY = 0, CLS_INTERIOR = 1, CLS_EXTERIOR = 2 };

template <typename T> __device__ inline uint8_t cell_class(const MeshView<T>& mv, const T* q) {
    int64_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int32_t i = (int32_t)((q[a] - mv.lo[a]) * mv.cinv);
        i = i < 0 ? 0 : (i >= mv.cdim[a] ? mv.cdim[a] - 1 : i);
        c[a] = i;
    }
    return mv.cls[(c[2] * mv.cdim[1] + c[1]) * mv.cdim[0] + c[0]];
}

// sign of (p - closest) . pseudonormal(feature): < 0 inside
template <typename T> __device__ inline T side_of(const MeshView<T>& mv, const T* q, const Nearest<T>& r) {
    const T* nrm = mv.pn + 21 * (int64_t)r.tri + 3 * r.feat;
    const T dv[3] = {q[0] - r.cp[0], q[1] - r.cp[1], q[2] - r.cp[2]};
    const T nn[3] = {nrm[0], nrm[1], nrm[2]};
    return ddot(dv, nn);
}

template <typename T> __device__ inline bool in_bbox(const MeshView<T>& mv, const T* q) {
    bool out = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) out = out || (q[a] < mv.lo[a]) || (q[a] > mv.hi[a]);
    return !out;
}

// One exact signed distance per cell centre.  A point of the cell is at most half a diagonal from the
// centre, so if the surface is farther than that (plus rounding slack) no point of the cell can be
// on the other side of it.
template <typename T>
__global__ void __launch_bounds__(kMeshThreads)
mesh_classify_kernel(MeshView<T> mv, T cell, int64_t ncell, uint8_t* __restrict__ cls) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t span = (ncell + 63) / 64 * 64;
    const T half_diag = wsqrt((T)0.75) * cell;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += stride) {
        const bool active = i < ncell;
        const int64_t ii = active ? i : ncell - 1;
        const int64_t cx = ii % mv.cdim[0], cy = (ii / mv.cdim[0]) % mv.cdim[1], cz = ii / ((int64_t)mv.cdim[0] * mv.cdim[1]);
        const T q[3] = {mv.lo[0] + ((T)cx + (T)0.5) * cell, mv.lo[1] + ((T)cy + (T)0.5) * cell,
                        mv.lo[2] + ((T)cz + (T)0.5) * cell};
        int32_t bn;
        const Nearest<T> r =
            mesh_nearest_guess<T>(mv.nodes, mv.m, q, active, mv.scale, mesh_greedy_guess<T>(mv.nodes, mv.m, q), &bn);
        if (!active) continue;
        if (r.tri < 0) {
            cls[i] = CLS_BOUNDARY;
            continue;
        }
        const T s = side_of<T>(mv, q, r);
        const T slack = half_diag * (T)1.001 + (T)256 * EpsOf<T>::v * mv.scale;
        uint8_t c = CLS_BOUNDARY;
        if (wsqrt(r.d2) > slack) c = s < (T)0 ? CLS_INTERIOR : (s > (T)0 ? CLS_EXTERIOR : CLS_BOUNDARY);
        cls[i] = c;
    }
}

// Raw AoS points -> any of {signed distance, triangle, closest point, inside flag, projection}.
template <typename TM, typename TP>
__global__ void __launch_bounds__(kMeshThreads)
mesh_query_kernel(const TP* __restrict__ xyz, int64_t n, MeshView<TM> mv, TM offset, TP* __restrict__ sd_out,
                  int32_t* __restrict__ tri_out, TP* __restrict__ cp_out, uint8_t* __restrict__ inside_out,
                  TP* __restrict__ proj_out, int packet) {
    __shared__ int32_t stk[kMeshThreads / 64][64];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t span = (n + 63) / 64 * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += stride) {
        const bool active = i < n;
        const int64_t ii = active ? i : n - 1;
        const TM q[3] = {(TM)xyz[3 * ii], (TM)xyz[3 * ii + 1], (TM)xyz[3 * ii + 2]}; // seam: convert once at entry
        Nearest<TM> r;
        if (packet) {
            r = mesh_nearest<TM>(mv.nodes, mv.m, q, active, mv.scale, stk[threadIdx.x >> 6]);
        } else {
            int32_t bn;
            r = mesh_nearest_guess<TM>(mv.nodes, mv.m, q, active, mv.scale, mesh_greedy_guess<TM>(mv.nodes, mv.m, q), &bn);
        }
        if (!active) continue;
        if (r.tri < 0) { // every triangle degenerate: no nearest element (closest_idx == 0 in the reference)
            if (sd_out) sd_out[i] = Lim<TP>::inf();
            if (tri_out) tri_out[i] = -1;
            if (inside_out) inside_out[i] = 0;
            for (int a = 0; a < 3; ++a) {
                if (cp_out) cp_out[3 * i + a] = (TP)q[a];
                if (proj_out) proj_out[3 * i + a] = (TP)q[a];
            }
            continue;
        }
        const TM s = side_of<TM>(mv, q, r);
        const TM dist = wsqrt(r.d2);
        if (sd_out) sd_out[i] = (TP)(s < (TM)0 ? -dist : (s > (TM)0 ? dist : (TM)0));
        if (tri_out) tri_out[i] = r.tri;
        if (cp_out)
            for (int a = 0; a < 3; ++a) cp_out[3 * i + a] = (TP)r.cp[a];
        if (inside_out) inside_out[i] = (in_bbox<TM>(mv, q) && s < (TM)0 && dist > (TM)0) ? 1 : 0;
        if (proj_out) {
            const TM* f = mv.pn + 21 * (int64_t)r.tri;
            for (int a = 0; a < 3; ++a) proj_out[3 * i + a] = (TP)(r.cp[a] - offset * f[a]);
        }
    }
}

// The wall rule after a sweep (src/repel.jl:448-469).  old = the sweep's query buffer (x_i), cur = its
// output (x_proposed), same slot order.  Boundary points land on the mesh; volume points that left
// the domain go back to x_i and are flagged.
template <typename TM, typename TP>
__global__ void __launch_bounds__(kMeshThreads)
mesh_constrain_kernel(const Pt<TP>* __restrict__ old, Pt<TP>* __restrict__ cur, int64_t n, int32_t n_fixed,
                      MeshView<TM> mv, TM offset, const uint8_t* __restrict__ is_bnd, uint8_t* __restrict__ escaped,
                      int32_t* __restrict__ tri_idx, int32_t* __restrict__ hint, int32_t* __restrict__ n_escaped) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t span = (n + 63) / 64 * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += stride) {
        const Pt<TP> p = cur[i < n ? i : n - 1];
        const int32_t id = w_to_id(p.w);
        const bool active = i < n && id >= n_fixed;
        if (!__any(active)) continue;
        const TM q[3] = {(TM)p.x, (TM)p.y, (TM)p.z};
        const bool bnd = active && is_bnd[id - n_fixed] != 0;
        // a volume point outside the mesh bbox is outside without a search (classify_point's fast path)
        const bool boxed = in_bbox<TM>(mv, q);
        // ... and one whose grid cell lies wholly on one side of the surface needs none either
        const uint8_t cls = (active && !bnd && boxed && mv.cls) ? cell_class<TM>(mv, q) : (uint8_t)CLS_BOUNDARY;
        const bool search = active && (bnd || (boxed && cls == CLS_BOUNDARY));
        Nearest<TM> r;
        r.tri = -1;
        r.d2 = (TM)1;
        if (__any(search)) {
            const int32_t g = search ? hint[id - n_fixed] : -1;
            const bool guessed = search && g >= 0 && g < mv.m;
            int32_t bn = -1;
            // first sweep, or a point that just came near the surface: a guess found from the root
            const int32_t g1 = search ? (guessed ? g : mesh_greedy_guess<TM>(mv.nodes, mv.m, q)) : -1;
            r = mesh_nearest_guess<TM>(mv.nodes, mv.m, q, search, mv.scale, g1, &bn);
            if (search) hint[id - n_fixed] = bn;
        }
        if (!active) continue;
        if (bnd && r.tri < 0) continue; // no triangle found: the point stays where the sweep put it (src/repel.jl:526)
        if (bnd) {
            const TM* f = mv.pn + 21 * (int64_t)r.tri;
            Pt<TP> o = p;
            o.x = (TP)(r.cp[0] - offset * f[0]);
            o.y = (TP)(r.cp[1] - offset * f[1]);
            o.z = (TP)(r.cp[2] - offset * f[2]);
            cur[i] = o;
            tri_idx[id - n_fixed] = r.tri;
        } else {
            bool inside = false;
            if (boxed)
                inside = cls == CLS_INTERIOR ||
                         (cls == CLS_BOUNDARY && r.tri >= 0 && side_of<TM>(mv, q, r) < (TM)0 && r.d2 > (TM)0);
            if (!inside) {
                const Pt<TP> xo = old[i];
                Pt<TP> o = p;
                o.x = xo.x, o.y = xo.y, o.z = xo.z;
                cur[i] = o;
                escaped[id - n_fixed] = 1;
                atomicAdd(n_escaped, 1);
            }
        }
    }
}

static int mesh_grid(int64_t n) {
    int64_t b = (n + kMeshThreads - 1) / kMeshThreads;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

template <typename TM> static MeshView<TM> make_view(const wtp_ctx* ctx) {
    MeshView<TM> mv;
    mv.nodes = (const MeshNode<TM>*)ctx->mesh_nodes.p;
    mv.pn = (const TM*)ctx->mesh_pn.p;
    mv.m = (int32_t)ctx->mesh_nt;
    for (int a = 0; a < 3; ++a) {
        mv.lo[a] = (TM)ctx->mesh_bbox[a];
        mv.hi[a] = (TM)ctx->mesh_bbox[3 + a];
    }
    mv.scale = (TM)ctx->mesh_scale;
    mv.cls = ctx->mesh_cls_ready ? (const uint8_t*)ctx->mesh_cls.p : nullptr;
    for (int a = 0; a < 3; ++a) mv.cdim[a] = ctx->mesh_cls_dim[a];
    mv.cinv = (TM)(1.0 / (ctx->mesh_cls_cell > 0 ? ctx->mesh_cls_cell : 1.0));
    return mv;
}

template <typename TM, typename TP>
static int launch_mesh_query(wtp_ctx* ctx, const TP* d_xyz, int64_t n, double offset, TP* sd, int32_t* tri, TP* cp,
                             uint8_t* inside, TP* proj) {
    hipLaunchKernelGGL((mesh_query_kernel<TM, TP>), dim3(mesh_grid(n)), dim3(kMeshThreads), 0, ctx->stream, d_xyz, n,
                       make_view<TM>(ctx), (TM)offset, sd, tri, cp, inside, proj, ctx->mesh_packet);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename TP>
int launch_mesh_constrain(wtp_ctx* ctx, const Pt<TP>* old, Pt<TP>* cur, int64_t n, int64_t n_fixed, double offset,
                          const uint8_t* is_bnd, uint8_t* escaped, int32_t* tri_idx, int32_t* hint, int32_t* n_escaped) {
    if (ctx->mesh_dtype == WTP_F32)
        hipLaunchKernelGGL((mesh_constrain_kernel<float, TP>), dim3(mesh_grid(n)), dim3(kMeshThreads), 0, ctx->stream, old,
                           cur, n, (int32_t)n_fixed, make_view<float>(ctx), (float)offset, is_bnd, escaped, tri_idx,
                           hint, n_escaped);
    else
        hipLaunchKernelGGL((mesh_constrain_kernel<double, TP>), dim3(mesh_grid(n)), dim3(kMeshThreads), 0, ctx->stream,
                           old, cur, n, (int32_t)n_fixed, make_view<double>(ctx), (double)offset, is_bnd, escaped,
                           tri_idx, hint, n_escaped);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}
template int launch_mesh_constrain<float>(wtp_ctx*, const Pt<float>*, Pt<float>*, int64_t, int64_t, double,
                                          const uint8_t*, uint8_t*, int32_t*, int32_t*, int32_t*);
template int launch_mesh_constrain<double>(wtp_ctx*, const Pt<double>*, Pt<double>*, int64_t, int64_t, double,
                                           const uint8_t*, uint8_t*, int32_t*, int32_t*, int32_t*);

// Class grid with cells of about `cell` (never more than 2^26 cells); kept when the one already built
// is at least as fine and not more than twice finer.
template <typename TM> static int mesh_classes_t(wtp_ctx* ctx, double cell) {
    double ext[3], vol = 1;
    for (int a = 0; a < 3; ++a) {
        ext[a] = ctx->mesh_bbox[3 + a] - ctx->mesh_bbox[a];
        vol *= ext[a];
    }
    const double floor_cell = std::cbrt(vol / 67108864.0);
    if (!(cell > floor_cell)) cell = floor_cell;
    if (ctx->mesh_cls_ready && ctx->mesh_cls_cell <= cell * 1.0001 && ctx->mesh_cls_cell >= cell * 0.5) return WTP_OK;
    int64_t ncell = 1;
    for (int a = 0; a < 3; ++a) {
        int64_t d = (int64_t)std::ceil(ext[a] / cell);
        d = d < 1 ? 1 : d;
        ctx->mesh_cls_dim[a] = (int)d;
        ncell *= d;
    }
    if (ncell > 200000000LL) return fail(ctx, WTP_ERR_ARG, "mesh class grid too large");
    int rc;
    ctx->mesh_cls_ready = false;
    if ((rc = ensure(ctx, ctx->mesh_cls, (size_t)ncell))) return rc;
    ctx->mesh_cls_cell = cell;
    MeshView<TM> mv = make_view<TM>(ctx);
    hipLaunchKernelGGL((mesh_classify_kernel<TM>), dim3(mesh_grid(ncell)), dim3(kMeshThreads), 0, ctx->stream, mv,
                       (TM)cell, ncell, (uint8_t*)ctx->mesh_cls.p);
    WTP_HIP(ctx, hipGetLastError());
    ctx->mesh_cls_ready = true;
    return WTP_OK;
}

static size_t al256(size_t b) { return (b + 255) / 256 * 256; }

template <typename TM> static int mesh_set_t(wtp_ctx* ctx, const TM* verts, int64_t nv, const int32_t* tris, int64_t nt) {
    std::vector<MeshNode<TM>> nodes((size_t)nt);
    std::vector<TM> pn((size_t)nt * 21);
    mesh_build_host<TM>(verts, tris, nt, nodes.data(), pn.data(), ctx->mesh_bbox, nv);
    double sc = 0;
    for (int a = 0; a < 6; ++a) sc = std::max(sc, std::fabs(ctx->mesh_bbox[a]));
    ctx->mesh_scale = sc;
    int rc;
    if ((rc = ensure(ctx, ctx->mesh_nodes, sizeof(MeshNode<TM>) * (size_t)nt))) return rc;
    if ((rc = ensure(ctx, ctx->mesh_pn, sizeof(TM) * 21 * (size_t)nt))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->mesh_nodes.p, nodes.data(), sizeof(MeshNode<TM>) * (size_t)nt, hipMemcpyHostToDevice,
                                ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(ctx->mesh_pn.p, pn.data(), sizeof(TM) * 21 * (size_t)nt, hipMemcpyHostToDevice,
                                ctx->stream));
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the host vectors go out of scope
    ctx->mesh_face_host.assign((size_t)nt * 3, 0.0);
    for (int64_t t = 0; t < nt; ++t)
        for (int a = 0; a < 3; ++a) ctx->mesh_face_host[3 * t + a] = (double)pn[21 * t + a];
    return WTP_OK;
}

} // namespace wtp

using namespace wtp;
#define WTP_API extern "C"

WTP_API int wtp_mesh_set(wtp_ctx* ctx, const void* vertices, int64_t nv, const int32_t* triangles, int64_t nt,
                         int dtype) {
    if (!ctx) return WTP_ERR_ARG;
    if (dtype != WTP_F32 && dtype != WTP_F64) return fail(ctx, WTP_ERR_ARG, "dtype must be WTP_F32 or WTP_F64");
    if (!vertices || !triangles) return fail(ctx, WTP_ERR_ARG, "NULL array");
    if (nv < 3 || nt < 1) return fail(ctx, WTP_ERR_ARG, "need at least 3 vertices and 1 triangle");
    if (nt >= (int64_t(1) << 30)) return fail(ctx, WTP_ERR_ARG, "more than 2^30 triangles");
    for (int64_t i = 0; i < 3 * nt; ++i)
        if (triangles[i] < 0 || triangles[i] >= nv)
            return fail(ctx, WTP_ERR_ARG, "triangle vertex index out of range (0-based indices expected)");
    if (ctx->relax.active && ctx->relax.wall_active)
        return fail(ctx, WTP_ERR_STATE, "the relax session uses the current mesh: call wtp_relax_end first");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    ctx->mesh_nt = 0;
    ctx->mesh_cls_ready = false;
    const int rc = dtype == WTP_F32 ? mesh_set_t<float>(ctx, (const float*)vertices, nv, triangles, nt)
                                    : mesh_set_t<double>(ctx, (const double*)vertices, nv, triangles, nt);
    if (rc) return rc;
    ctx->mesh_nt = nt;
    ctx->mesh_dtype = dtype;
    return WTP_OK;
}

WTP_API int wtp_mesh_clear(wtp_ctx* ctx) {
    if (!ctx) return WTP_ERR_ARG;
    if (ctx->relax.active && ctx->relax.wall_active)
        return fail(ctx, WTP_ERR_STATE, "the relax session uses the current mesh: call wtp_relax_end first");
    ctx->mesh_nt = 0;
    ctx->mesh_cls_ready = false;
    ctx->mesh_face_host.clear();
    return WTP_OK;
}

WTP_API int wtp_mesh_face_normals(wtp_ctx* ctx, double* normals_out) {
    if (!ctx) return WTP_ERR_ARG;
    if (ctx->mesh_nt < 1) return fail(ctx, WTP_ERR_STATE, "no mesh: call wtp_mesh_set first");
    if (!normals_out) return fail(ctx, WTP_ERR_ARG, "NULL array");
    memcpy(normals_out, ctx->mesh_face_host.data(), sizeof(double) * ctx->mesh_face_host.size());
    return WTP_OK;
}

WTP_API int wtp_mesh_bounds(wtp_ctx* ctx, double bbox_out[6]) {
    if (!ctx) return WTP_ERR_ARG;
    if (ctx->mesh_nt < 1) return fail(ctx, WTP_ERR_STATE, "no mesh: call wtp_mesh_set first");
    if (!bbox_out) return fail(ctx, WTP_ERR_ARG, "NULL array");
    memcpy(bbox_out, ctx->mesh_bbox, sizeof(double) * 6);
    return WTP_OK;
}

WTP_API int wtp_mesh_query(wtp_ctx* ctx, const void* xyz, int64_t n, int dtype, double offset, void* sd_out,
                           int32_t* tri_out, void* closest_out, uint8_t* inside_out, void* projected_out) {
    if (!ctx) return WTP_ERR_ARG;
    if (dtype != WTP_F32 && dtype != WTP_F64) return fail(ctx, WTP_ERR_ARG, "dtype must be WTP_F32 or WTP_F64");
    if (ctx->mesh_nt < 1) return fail(ctx, WTP_ERR_STATE, "no mesh: call wtp_mesh_set first");
    if (n < 0 || n > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "bad n");
    if (n == 0) return WTP_OK;
    if (!xyz) return fail(ctx, WTP_ERR_ARG, "NULL array");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = dtype == WTP_F64 ? 8 : 4;
    // mesh_io: [xyz n*3 | sd n | cp n*3 | proj n*3 | tri n (int32) | inside n (u8)]
    const size_t o_sd = al256(ts * n * 3), o_cp = o_sd + al256(ts * n), o_pr = o_cp + al256(ts * n * 3),
                 o_tr = o_pr + al256(ts * n * 3), o_in = o_tr + al256(4 * (size_t)n);
    int rc;
    if ((rc = ensure(ctx, ctx->mesh_io, o_in + (size_t)n))) return rc;
    char* b = (char*)ctx->mesh_io.p;
    WTP_HIP(ctx, hipMemcpyAsync(b, xyz, ts * n * 3, hipMemcpyHostToDevice, ctx->stream));
    int sp = span_begin(ctx, 2);
    if (dtype == WTP_F32) {
        float *sd = sd_out ? (float*)(b + o_sd) : nullptr, *cp = closest_out ? (float*)(b + o_cp) : nullptr,
              *pr = projected_out ? (float*)(b + o_pr) : nullptr;
        rc = ctx->mesh_dtype == WTP_F32
                 ? launch_mesh_query<float, float>(ctx, (const float*)b, n, offset, sd, tri_out ? (int32_t*)(b + o_tr) : nullptr,
                                                   cp, inside_out ? (uint8_t*)(b + o_in) : nullptr, pr)
                 : launch_mesh_query<double, float>(ctx, (const float*)b, n, offset, sd,
                                                    tri_out ? (int32_t*)(b + o_tr) : nullptr, cp,
                                                    inside_out ? (uint8_t*)(b + o_in) : nullptr, pr);
    } else {
        double *sd = sd_out ? (double*)(b + o_sd) : nullptr, *cp = closest_out ? (double*)(b + o_cp) : nullptr,
               *pr = projected_out ? (double*)(b + o_pr) : nullptr;
        rc = ctx->mesh_dtype == WTP_F32
                 ? launch_mesh_query<float, double>(ctx, (const double*)b, n, offset, sd,
                                                    tri_out ? (int32_t*)(b + o_tr) : nullptr, cp,
                                                    inside_out ? (uint8_t*)(b + o_in) : nullptr, pr)
                 : launch_mesh_query<double, double>(ctx, (const double*)b, n, offset, sd,
                                                     tri_out ? (int32_t*)(b + o_tr) : nullptr, cp,
                                                     inside_out ? (uint8_t*)(b + o_in) : nullptr, pr);
    }
    span_end(ctx, sp);
    if (rc) return rc;
    if (sd_out) WTP_HIP(ctx, hipMemcpyAsync(sd_out, b + o_sd, ts * n, hipMemcpyDeviceToHost, ctx->stream));
    if (closest_out) WTP_HIP(ctx, hipMemcpyAsync(closest_out, b + o_cp, ts * n * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (projected_out)
        WTP_HIP(ctx, hipMemcpyAsync(projected_out, b + o_pr, ts * n * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (tri_out) WTP_HIP(ctx, hipMemcpyAsync(tri_out, b + o_tr, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (inside_out) WTP_HIP(ctx, hipMemcpyAsync(inside_out, b + o_in, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WTP_OK;
}

// Installs the wall rule of the octree method on the current relax session: from now on every sweep
// is followed by _constrain_octree (src/repel.jl:448-469).  Movable points with index below
// n_boundary (counted from the first movable one) start as boundary points.
WTP_API int wtp_relax_set_wall(wtp_ctx* ctx, int64_t n_boundary, double offset_dist) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_wall before wtp_relax_init");
    if (ctx->mesh_nt < 1) return fail(ctx, WTP_ERR_STATE, "no mesh: call wtp_mesh_set first");
    if (r.dim != 3) return fail(ctx, WTP_ERR_ARG, "the wall rule is 3-D only (src/repel.jl:122)");
    const int64_t nm = r.n - r.n_fixed;
    if (n_boundary < 0 || n_boundary > nm) return fail(ctx, WTP_ERR_ARG, "n_boundary out of range");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    // wall_flags: [is_bnd nm | escaped nm | counter (int32, 256-aligned)]; wall_tri: int32 nm, -1 = none
    const size_t o_cnt = al256(2 * (size_t)nm);
    if ((rc = ensure(ctx, ctx->wall_flags, o_cnt + 256))) return rc;
    if ((rc = ensure(ctx, ctx->wall_tri, 4 * (size_t)nm))) return rc;
    if ((rc = ensure(ctx, ctx->wall_hint, 4 * (size_t)nm))) return rc;
    // cells of about one spacing: only the points within a spacing or two of the wall take the exact test
    const double cell = r.spacing_kind == WTP_SPACING_CONSTANT ? r.spacing_const
                        : (r.spacing_kind == WTP_SPACING_BOUNDARY_LAYER ? r.sp_p0 : r.spacing_max / 2);
    if ((rc = ctx->mesh_dtype == WTP_F32 ? mesh_classes_t<float>(ctx, cell) : mesh_classes_t<double>(ctx, cell))) return rc;
    WTP_HIP(ctx, hipMemsetAsync(ctx->wall_hint.p, 0xff, 4 * (size_t)nm, ctx->stream));
    WTP_HIP(ctx, hipMemsetAsync(ctx->wall_flags.p, 0, o_cnt + 256, ctx->stream));
    if (n_boundary > 0) WTP_HIP(ctx, hipMemsetAsync(ctx->wall_flags.p, 1, (size_t)n_boundary, ctx->stream));
    WTP_HIP(ctx, hipMemsetAsync(ctx->wall_tri.p, 0xff, 4 * (size_t)nm, ctx->stream));
    r.wall_active = true;
    r.wall_offset = offset_dist;
    r.wall_nm = nm;
    return WTP_OK;
}

// tri_out: landing triangle of each movable point at its last projection (0-based, -1 = never
// projected); is_bnd_out / escaped_out: the flags of src/repel.jl:146-149.  escaped is cleared by the read
// when clear_escaped != 0 (the deposit pass consumes it, :486-487).
WTP_API int wtp_relax_get_wall(wtp_ctx* ctx, int32_t* tri_out, uint8_t* is_bnd_out, uint8_t* escaped_out,
                               int clear_escaped) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active || !r.wall_active) return fail(ctx, WTP_ERR_STATE, "no wall rule installed");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nm = (size_t)r.wall_nm;
    char* f = (char*)ctx->wall_flags.p;
    if (tri_out) WTP_HIP(ctx, hipMemcpyAsync(tri_out, ctx->wall_tri.p, 4 * nm, hipMemcpyDeviceToHost, ctx->stream));
    if (is_bnd_out) WTP_HIP(ctx, hipMemcpyAsync(is_bnd_out, f, nm, hipMemcpyDeviceToHost, ctx->stream));
    if (escaped_out) WTP_HIP(ctx, hipMemcpyAsync(escaped_out, f + nm, nm, hipMemcpyDeviceToHost, ctx->stream));
    if (clear_escaped) WTP_HIP(ctx, hipMemsetAsync(f + nm, 0, nm, ctx->stream));
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WTP_OK;
}

// Host-side deposition (src/repel.jl:471-520 is serial by design) writes its result back: new
// membership flags and landing triangles for all movable points.
WTP_API int wtp_relax_set_wall_flags(wtp_ctx* ctx, const uint8_t* is_bnd, const int32_t* tri) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active || !r.wall_active) return fail(ctx, WTP_ERR_STATE, "no wall rule installed");
    if (!is_bnd || !tri) return fail(ctx, WTP_ERR_ARG, "NULL array");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nm = (size_t)r.wall_nm;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->wall_flags.p, is_bnd, nm, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(ctx->wall_tri.p, tri, 4 * nm, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WTP_OK;
}
