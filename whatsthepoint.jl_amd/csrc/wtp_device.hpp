// wtp_device.hpp — device functions shared by the generic and the brick kernels.
// Arithmetic restates the reference expressions term by term (file:line cited per function);
// the translation units are compiled with -ffp-contract=off so no FMA is formed.
#pragma once
#include "wtp_internal.hpp"

namespace wtp {

template <typename T> struct Lim;
template <> struct Lim<float> {
    static __device__ __host__ constexpr float inf() { return __builtin_huge_valf(); }
};
template <> struct Lim<double> {
    static __device__ __host__ constexpr double inf() { return __builtin_huge_val(); }
};

__device__ inline float wsqrt(float x) { return __builtin_sqrtf(x); } // correctly rounded (IEEE), like Julia sqrt
__device__ inline double wsqrt(double x) { return __builtin_sqrt(x); }
__device__ inline float wfloor(float x) { return floorf(x); }
__device__ inline double wfloor(double x) { return floor(x); }
__device__ inline float wpow(float a, float b) { return powf(a, b); }
__device__ inline double wpow(double a, double b) { return pow(a, b); }

// canonical squared distance: ((dx*dx + dy*dy) + dz*dz), Distances.jl Euclidean as called
// from src/repel.jl:259 and src/topology.jl:80-81.  For dim == 2 all z are 0 so the third
// term adds an exact zero.
template <typename T>
__device__ inline T dist2(T ax, T ay, T az, T bx, T by, T bz) {
    T dx = ax - bx;
    T dy = ay - by;
    T dz = az - bz;
    T s = dx * dx + dy * dy;
    return s + dz * dz;
}

// (Measured on gfx950: v_pk_add_f32 / v_pk_mul_f32 issue at half the rate of the scalar forms, so
// packing the (x,y) pair buys nothing — profiles/r01_d_ablation.txt.)

template <typename T>
__device__ inline bool lex_lt(T da, int32_t ia, T db, int32_t ib) {
    return (da < db) || (da == db && ia < ib);
}

template <typename T>
__device__ inline int cell_coord(const Grid<T>& g, T v, int axis) {
    T f = wfloor((v - g.org[axis]) * g.inv_c);
    int c = (f > (T)0) ? ((f < (T)(g.n[axis] - 1)) ? (int)f : g.n[axis] - 1) : 0;
    return c;
}

// Squared radius inside which a search over cells [c-r, c+r]^dim around the query's cell is
// provably complete (every point closer than this lies in a searched cell).  Edge cells are
// unbounded outward (clamped map), so a side whose last searched cell is an edge cell
// imposes no bound.  Returns +inf when the searched block covers the grid.
template <typename T>
__device__ inline T safe_radius2(const Grid<T>& g, T qx, T qy, T qz, int cx, int cy, int cz, int r) {
    T best = Lim<T>::inf();
    const T q[3] = {qx, qy, qz};
    const int c[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (a >= g.dim) break;
        T rel = q[a] - g.org[a];
        if (c[a] - r > 0) {
            T lo = rel - (T)(c[a] - r) * g.c;
            best = lo < best ? lo : best;
        }
        if (c[a] + r < g.n[a] - 1) {
            T hi = (T)(c[a] + r + 1) * g.c - rel;
            best = hi < best ? hi : best;
        }
    }
    if (best == Lim<T>::inf()) return best;
    best = best - g.margin;
    if (best < (T)0) best = (T)0;
    return best * best;
}

// force laws, src/repel_forces.jl:37,57-60,96-100,124-127
template <typename T>
__device__ inline T force_law(int kind, T beta, T u0, T gamma, T u) {
    T u2 = u * u;
    T d = u2 + beta;
    if (kind == WTP_FORCE_INVERSE_DISTANCE) return (T)1 / (d * d);
    if (kind == WTP_FORCE_SPACING_EQUILIBRIUM) return ((T)1 - u2) / (d * d);
    if (kind == WTP_FORCE_CLIPPED_SPACING) {
        T f = (u0 * u0 - u2) / (d * d);
        return f > (T)0 ? f : (T)0;
    }
    return ((T)1 - u2) / wpow(d, gamma);
}

// deterministic unit vector for r == 0 (_safe_direction's randn branch, src/repel.jl:358-364,
// is unseeded in the reference; the same counter hash as the oracle keeps GPU == oracle).
template <typename T>
__device__ inline void fallback_dir(int64_t i, int64_t j, int dim, T out[3]) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)j * 0xBF58476D1CE4E5B9ull + 1;
    T v[3] = {0, 0, 0};
    T nn = 0;
    for (int d = 0; d < dim; ++d) {
        z ^= z >> 30;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        z *= 0x94D049BB133111EBull;
        z ^= z >> 31;
        v[d] = (T)((double)(z >> 11) * (1.0 / 9007199254740992.0)) - (T)0.5;
        nn += v[d] * v[d];
    }
    nn = wsqrt(nn);
    if (!(nn > (T)0)) {
        v[0] = 1;
        v[1] = v[2] = 0;
        nn = 1;
    }
    out[0] = v[0] / nn;
    out[1] = v[1] / nn;
    out[2] = v[2] / nn;
}

// One neighbour's contribution, src/repel.jl:276-279:  F += f(r/s) * (xi - xj)/r
template <typename T>
__device__ inline void add_force(const SearchArgs<T>& a, int dim, T s, T xi, T yi, T zi, int32_t self,
                                 T xj, T yj, T zj, int32_t jid, T d2, T& Fx, T& Fy, T& Fz) {
    T r = wsqrt(d2);
    T f = force_law<T>(a.force_kind, a.beta, a.u0, a.gamma, r / s);
    T dir[3];
    if (r > (T)0) {
        dir[0] = (xi - xj) / r;
        dir[1] = (yi - yj) / r;
        dir[2] = (zi - zj) / r;
    } else {
        fallback_dir<T>(self, jid, dim, dir);
    }
    Fx = Fx + f * dir[0];
    Fy = Fy + f * dir[1];
    Fz = Fz + f * dir[2];
}

// Adaptive step and displacement cap, src/repel.jl:282-291.  Returns forces[id] = |F|*s.
template <typename T>
__device__ inline T step_point(const SearchArgs<T>& a, T s, T xi, T yi, T zi, T Fx, T Fy, T Fz,
                               T& xo, T& yo, T& zo) {
    T Fn = wsqrt((Fx * Fx + Fy * Fy) + Fz * Fz);
    T al = (T)1 / (Fn + (T)1.0e-30);
    al = al < a.alpha_lo ? a.alpha_lo : al;
    al = al > a.alpha_max ? a.alpha_max : al;
    T sa = s * al;
    T dx = sa * Fx, dy = sa * Fy, dz = sa * Fz;
    T dn = wsqrt((dx * dx + dy * dy) + dz * dz);
    if (dn > s) {
        T sc = s / dn;
        dx = dx * sc;
        dy = dy * sc;
        dz = dz * sc;
    }
    xo = xi + dx;
    yo = yi + dy;
    zo = zi + dz;
    return Fn * s;
}

// ---- per-thread partial and block reduction -------------------------------------------------
// Sharded sessions: does a neighbourhood of squared radius need2 around the query stay inside the
// coordinate range the snapshot is complete for?  (range ends may be +-inf)
template <typename T>
__device__ inline bool reaches_past_cover(const SearchArgs<T>& a, T qx, T qy, T qz, T need2) {
    if (a.cover_axis < 0) return false;
    if (a.cover_axis == 3) { // a box (block decompositions): the nearest of its six faces decides
        T d = qx - a.cover_lo3[0];
        T e = a.cover_hi3[0] - qx;
        d = e < d ? e : d;
        e = qy - a.cover_lo3[1];
        d = e < d ? e : d;
        e = a.cover_hi3[1] - qy;
        d = e < d ? e : d;
        e = qz - a.cover_lo3[2];
        d = e < d ? e : d;
        e = a.cover_hi3[2] - qz;
        d = e < d ? e : d;
        return !(d > (T)0 && need2 <= d * d);
    }
    const T v = a.cover_axis == 0 ? qx : (a.cover_axis == 1 ? qy : qz);
    const T dl = v - a.cover_lo, dh = a.cover_hi - v;
    const T d = dl < dh ? dl : dh;
    return !(d > (T)0 && need2 <= d * d);
}

struct Acc { // plain aggregate: lives in LDS too
    double max_force;
    double sum_u;
    double sum_u2;
    double argmin_r;
    int64_t argmin_i;
    int64_t argmin_j;
    int64_t n_move;
};

__device__ inline Acc acc_empty() {
    Acc a;
    a.max_force = 0.0;
    a.sum_u = 0.0;
    a.sum_u2 = 0.0;
    a.argmin_r = __builtin_huge_val();
    a.argmin_i = -1;
    a.argmin_j = -1;
    a.n_move = 0;
    return a;
}

// u = nn_dist / spacing is formed in the cloud's float type, like `_dnn_cv` does
// (src/repel.jl:376-381: U = typeof(one(T)/one(eltype(spacings)))), then accumulated in double.
template <typename T>
__device__ inline void acc_point(Acc& acc, T force_t, T nn_dist_t, T s, int64_t id, int64_t nn) {
    const double force = (double)force_t, nn_dist = (double)nn_dist_t;
    acc.max_force = force > acc.max_force ? force : acc.max_force;
    const double u = (double)(nn_dist_t / s);
    acc.sum_u += u;
    acc.sum_u2 += u * u;
    acc.n_move += 1;
    if (nn_dist < acc.argmin_r || (nn_dist == acc.argmin_r && id < acc.argmin_i)) {
        acc.argmin_r = nn_dist;
        acc.argmin_i = id;
        acc.argmin_j = nn;
    }
}

__device__ inline void acc_merge(Acc& a, const Acc& b) {
    a.max_force = b.max_force > a.max_force ? b.max_force : a.max_force;
    a.sum_u += b.sum_u;
    a.sum_u2 += b.sum_u2;
    a.n_move += b.n_move;
    if (b.argmin_i >= 0 &&
        (a.argmin_i < 0 || b.argmin_r < a.argmin_r || (b.argmin_r == a.argmin_r && b.argmin_i < a.argmin_i))) {
        a.argmin_r = b.argmin_r;
        a.argmin_i = b.argmin_i;
        a.argmin_j = b.argmin_j;
    }
}

__device__ inline double shfl_down_d(double v, int d) { return __shfl_down(v, d, 64); }
__device__ inline int64_t shfl_down_i64(int64_t v, int d) {
    return (int64_t)__shfl_down((long long)v, d, 64);
}

// wave64 shuffle tree, then one LDS slot per wave; thread 0 ends up with the block total.
// `smem` must hold (blockDim.x/64) Acc records.
__device__ inline void acc_block_reduce(Acc& acc, Acc* smem) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Acc o;
        o.max_force = shfl_down_d(acc.max_force, d);
        o.sum_u = shfl_down_d(acc.sum_u, d);
        o.sum_u2 = shfl_down_d(acc.sum_u2, d);
        o.argmin_r = shfl_down_d(acc.argmin_r, d);
        o.argmin_i = shfl_down_i64(acc.argmin_i, d);
        o.argmin_j = shfl_down_i64(acc.argmin_j, d);
        o.n_move = shfl_down_i64(acc.n_move, d);
        acc_merge(acc, o);
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) acc_merge(acc, smem[w]);
    }
}

__device__ inline void acc_store(Partial* p, const Acc& acc) {
    p->max_force = acc.max_force;
    p->sum_u = acc.sum_u;
    p->sum_u2 = acc.sum_u2;
    p->argmin_r = acc.argmin_r;
    p->argmin_i = acc.argmin_i;
    p->argmin_j = acc.argmin_j;
    p->n_move = acc.n_move;
}

// The step's final reduction, by one block (any size): only the slots this step's launches wrote, in a fixed
// order => deterministic.  Three ranges: brick [0, used_brick), wave [wave_base, +used_wave), serial
// [n_parts - kGenericPartials, +used_generic).  Ends by zeroing the step's counter block (hand-backs, uncovered,
// escaped, ticket): all-zero for the next step without a memset.
__device__ inline void reduce_partials_block(const Partial* __restrict__ parts, int n_parts, int used_brick, int wave_base,
                                             int used_wave, int used_generic, const int32_t* __restrict__ fb_count,
                                             const int32_t* __restrict__ uncovered, const int32_t* __restrict__ escaped,
                                             wtp_step_stats* __restrict__ out, int32_t* __restrict__ counters, Acc* sm) {
    Acc acc = acc_empty();
    const int total = used_brick + used_wave + used_generic;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
        const int i = t < used_brick ? t
                                     : (t < used_brick + used_wave ? wave_base + (t - used_brick)
                                                                   : n_parts - kGenericPartials + (t - used_brick - used_wave));
        if (parts[i].n_move == 0) continue; // slot of a block that saw no movable point
        Acc o;
        o.max_force = parts[i].max_force;
        o.sum_u = parts[i].sum_u;
        o.sum_u2 = parts[i].sum_u2;
        o.argmin_r = parts[i].argmin_r;
        o.argmin_i = parts[i].argmin_i;
        o.argmin_j = parts[i].argmin_j;
        o.n_move = parts[i].n_move;
        acc_merge(acc, o);
    }
    acc_block_reduce(acc, sm);
    if (threadIdx.x == 0) {
        out->max_force = acc.max_force;
        out->sum_u = acc.sum_u;
        out->sum_u2 = acc.sum_u2;
        out->n_move = acc.n_move;
        out->argmin_i = acc.argmin_i;
        out->argmin_j = acc.argmin_j;
        out->argmin_r = acc.argmin_r;
        out->n_fallback = fb_count ? *fb_count : 0;
        out->n_uncovered = uncovered ? *uncovered : 0;
        out->n_escaped = escaped ? *escaped : 0;
    }
    __syncthreads();
    if (counters && threadIdx.x < 16) counters[threadIdx.x] = 0;
}

} // namespace wtp
