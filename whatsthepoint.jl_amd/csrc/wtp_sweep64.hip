// Float64 sweeps of the laws that need the explicit k nearest neighbours (src/repel.jl:256-292 with InverseDistance, Spacing,
// LennardJones forces: everything but ClippedSpacingForce, which has its compact-support kernels).
//
// Until round 3 these ran on the exact wave-per-query path alone (137 Mpoints/s).  The route that made Float64 KNNTopology
// fast applies: the k + 2 ... 24 nearest CANDIDATES come from the fp32 k-selection kernels (wtp_ksel.hip) on a float copy of
// the snapshot in a local frame, in slot order of that copy's own grid; this file re-ranks them exactly in fp64 — one
// lane per query, the list in registers — certifies the first k (the fp32 search excluded nothing nearer than its last
// candidate minus the rounding bound), and then does what the wave kernel does with its k rows: the forces of the k
// neighbours in ascending (d2, index), added in that order, the step, the statistics.  Same expressions, same order:
// the same bits as the exact path, which still takes every query the certificate turns down.
#include "wtp_device.hpp"
#include "wtp_internal.hpp"

namespace wtp {

// snapshot (session order, w = index) -> float copy in the frame of org4, w = the session slot
__global__ void f64k_local_kernel(const double4* __restrict__ in, int64_t n, const double* __restrict__ org4, float4* __restrict__ out) {
    const double ox = org4[0], oy = org4[1], oz = org4[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double4 p = in[i];
        float4 o;
        o.x = (float)(p.x - ox);
        o.y = (float)(p.y - oy);
        o.z = (float)(p.z - oz);
        o.w = id_to_w(0.f, (int32_t)i);
        out[i] = o;
    }
}

// the float copy sorted by its own grid: entry i came from session slot w.  Afterwards w = i (the search names slots of THIS
// order), sslot[i] = the session slot, s64[i] = the fp64 point (w = the point's index, the tie-break of the canonical order)
__global__ void f64k_relabel_kernel(const double4* __restrict__ snap, float4* __restrict__ sorted32, int32_t* __restrict__ sslot,
                                    double4* __restrict__ s64, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p = sorted32[i];
        const int32_t s = w_to_id(p.w);
        sslot[i] = s;
        s64[i] = snap[s];
        p.w = id_to_w(0.f, (int32_t)i);
        sorted32[i] = p;
    }
}

constexpr int kS64Threads = 128;

template <int KC>
__global__ __launch_bounds__(kS64Threads) void refine_sweep_f64_kernel(SearchArgs<double> a, const double4* __restrict__ s64,
                                                                       const int32_t* __restrict__ sslot,
                                                                       const int32_t* __restrict__ cand,
                                                                       const float* __restrict__ cdist,
                                                                       const double* __restrict__ org4, int part_base) {
    __shared__ Acc sm_acc[kS64Threads / 64];
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    Acc acc = acc_empty();
    const double extent = org4[3];
    const int Kq = a.k; // the k nearest, self among them (src/repel.jl:262: knn(tree, xi, k); :271 skips j == i)
    const int dim = a.grid->dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const double4 q = s64[i];
        const int32_t id = w_to_id(q.w);
        const int32_t slot = sslot[i];
        if (id < a.n_fixed) { // the wall: never moves (src/repel.jl:80,256)
            a.out[slot] = q;
            a.forces[slot] = 0.0;
            a.nn_dist[slot] = Lim<double>::inf();
            a.nn_id[slot] = -1;
            continue;
        }
        double kd[KC];
        int32_t ki[KC], kc[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            kc[j] = cand[i * KC + j];
            const double4 p = s64[kc[j]];
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
                const int32_t ti = ki[j], tc = kc[j];
                kd[j] = sw ? kd[j + 1] : td;
                ki[j] = sw ? ki[j + 1] : ti;
                kc[j] = sw ? kc[j + 1] : tc;
                kd[j + 1] = sw ? td : kd[j + 1];
                ki[j + 1] = sw ? ti : ki[j + 1];
                kc[j + 1] = sw ? tc : kc[j + 1];
                again = again || sw;
            }
        }
        // the certificate of refine_f64_slots_kernel (wtp_hash.hip): rounding the coordinates to float and evaluating in float
        // moves a distance by less than eps, so every point the fp32 search left out is farther than dmax32 - eps
        const double dmax32 = (double)cdist[i * KC + KC - 1];
        const double eps = extent * 0x1p-21 + dmax32 * 0x1p-20;
        double dkq = kd[KC - 1];
#pragma unroll
        for (int j = 0; j < KC; ++j) dkq = (j == Kq - 1) ? kd[j] : dkq;
        if (!(wsqrt(dkq) < dmax32 - eps)) { // not certified: the exact path (its list holds session slots)
            a.fb_list[atomicAdd(a.fb_count, 1)] = slot;
            continue;
        }
        const double s = a.spacing_pp ? a.spacing_pp[id] : a.spacing_const;
        double Fx = 0, Fy = 0, Fz = 0, nd = Lim<double>::inf();
        int32_t nid = -1;
#pragma unroll
        for (int j = 0; j < KC; ++j) { // ascending (d2, id), self skipped by index (:271)
            if (j < Kq && ki[j] != id) {
                const double4 c = s64[kc[j]];
                double fx = 0, fy = 0, fz = 0;
                add_force<double>(a, dim, s, q.x, q.y, q.z, id, c.x, c.y, c.z, ki[j], kd[j], fx, fy, fz);
                if (nid < 0) {
                    nid = ki[j];
                    nd = wsqrt(kd[j]);
                }
                Fx = Fx + fx;
                Fy = Fy + fy;
                Fz = Fz + fz;
            }
        }
        double4 o;
        const double f = step_point<double>(a, s, q.x, q.y, q.z, Fx, Fy, Fz, o.x, o.y, o.z);
        o.w = q.w;
        a.out[slot] = o;
        a.forces[slot] = f;
        a.nn_dist[slot] = nd;
        a.nn_id[slot] = nid;
        acc_point<double>(acc, f, nd, s, id, nid);
        // sharded sessions: what the answer rests on is the k-th neighbour
        if (reaches_past_cover<double>(a, q.x, q.y, q.z, dkq)) atomicAdd(a.uncovered, 1);
    }
    __syncthreads();
    acc_block_reduce(acc, sm_acc);
    if (threadIdx.x == 0) acc_store(&a.partials[part_base + blockIdx.x], acc);
}

static inline int f64k_grid(int64_t n, int threads, int cap) {
    int64_t b = (n + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int launch_f64k_local(wtp_ctx* ctx, const double4* snap, int64_t n, const double* d_org4, float4* out) {
    hipLaunchKernelGGL(f64k_local_kernel, dim3(f64k_grid(n, 256, 8192)), dim3(256), 0, ctx->stream, snap, n, d_org4, out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int launch_f64k_relabel(wtp_ctx* ctx, const double4* snap, float4* sorted32, int32_t* sslot, double4* s64, int64_t n) {
    hipLaunchKernelGGL(f64k_relabel_kernel, dim3(f64k_grid(n, 256, 8192)), dim3(256), 0, ctx->stream, snap, sorted32, sslot, s64, n);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// candidates: 24 per query (the caller searched with k = 24).  Partials go to the brick range [0, used_brick).
int launch_refine_sweep_f64(wtp_ctx* ctx, SearchArgs<double>& a, const double4* s64, const int32_t* sslot, const int32_t* cand,
                            const float* cdist, const double* d_org4) {
    const int blocks = f64k_grid(a.n, kS64Threads, 1024);
    a.used_brick = blocks;
    hipLaunchKernelGGL(refine_sweep_f64_kernel<24>, dim3(blocks), dim3(kS64Threads), 0, ctx->stream, a, s64, sslot, cand, cdist,
                       d_org4, 0);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

} // namespace wtp
