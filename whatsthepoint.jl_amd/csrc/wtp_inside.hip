// wtp_inside.hip — the isinside post-filter of the volume-only repel (src/repel.jl:90).
//
//   3-D  src/isinside.jl:86-106   g(x) = sum_j ((a_j (x - p_j)) . n_j) / |x - p_j|^3 over the boundary
//                                 elements (centroid p, unit normal n, area a); inside iff g < -2 pi.
//   2-D  src/isinside.jl:17-33    winding sum of the signed angles ∠(p_j, x, p_j+1) around the ordered
//                                 polygon; inside iff |sum| >= 1e3 eps(T); coincident points are inside.
//
// Both are dense N x M pair sums: no neighbour structure helps (every element contributes to every
// test point), the work is arithmetic.  One lane owns R test points in registers; the boundary
// elements are wave-uniform, so they stream through the scalar cache (s_load) and every VALU
// instruction works on R x 64 pairs' worth of data per element fetched.  VALU-bound by
// construction: ~16 lane-ops per pair (fp32: 3 sub, 6 mul/fma, rsq at quarter rate, 3 mul/fma).
// Small N with large M: the element range is split over blockIdx.y and the partial sums are added
// in a fixed order by the finishing kernel (deterministic).
#include "wtp_device.hpp"

namespace wtp {

static constexpr int kInsThreads = 256;
static constexpr int kInsR = 4; // test points per lane

template <typename T> struct Elem { // {p, 0, a*n, 0}: two 16-B (fp64: 32-B) records, scalar-loadable
    T px, py, pz, pad0;
    T qx, qy, qz, pad1;
};

template <typename T>
__global__ void pack_elems_kernel(const T* __restrict__ p, const T* __restrict__ nrm, const T* __restrict__ area,
                                  int64_t m, Elem<T>* __restrict__ out) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    Elem<T> e;
    const T a = area[j];
    e.px = p[3 * j];
    e.py = p[3 * j + 1];
    e.pz = p[3 * j + 2];
    e.qx = a * nrm[3 * j];
    e.qy = a * nrm[3 * j + 1];
    e.qz = a * nrm[3 * j + 2];
    e.pad0 = e.pad1 = (T)0;
    out[j] = e;
}

__device__ inline float inv_cube_root2(float r2) { // 1 / r^3 from r^2
    const float i = __builtin_amdgcn_rsqf(r2);
    return (i * i) * i;
}
__device__ inline double inv_cube_root2(double r2) {
    const double i = 1.0 / __builtin_sqrt(r2);
    return (i * i) * i;
}
__device__ inline float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
__global__ void __launch_bounds__(kInsThreads)
greens_kernel(const T* __restrict__ test, int64_t n, const Elem<T>* __restrict__ elems, int64_t m, int64_t chunk,
              T* __restrict__ partial) {
    const int64_t base = (int64_t)blockIdx.x * (kInsThreads * kInsR) + threadIdx.x;
    T x[kInsR], y[kInsR], z[kInsR], g[kInsR];
#pragma unroll
    for (int r = 0; r < kInsR; ++r) {
        const int64_t i = base + (int64_t)r * kInsThreads;
        const int64_t ii = i < n ? i : n - 1; // clamp: tail lanes recompute the last point, never store
        x[r] = test[3 * ii];
        y[r] = test[3 * ii + 1];
        z[r] = test[3 * ii + 2];
        g[r] = (T)0;
    }
    const int64_t j0 = (int64_t)blockIdx.y * chunk;
    const int64_t j1 = j0 + chunk < m ? j0 + chunk : m;
#pragma unroll 4
    for (int64_t j = j0; j < j1; ++j) {
        const Elem<T> e = elems[j]; // uniform address: scalar loads (batched by the unroll)
#pragma unroll
        for (int r = 0; r < kInsR; ++r) {
            const T dx = x[r] - e.px, dy = y[r] - e.py, dz = z[r] - e.pz;
            const T r2 = fma_t(dz, dz, fma_t(dy, dy, dx * dx));
            const T dq = fma_t(dz, e.qz, fma_t(dy, e.qy, dx * e.qx));
            g[r] = fma_t(dq, inv_cube_root2(r2), g[r]); // r2 == 0: 0 * inf = NaN, as the reference's 0/0
        }
    }
#pragma unroll
    for (int r = 0; r < kInsR; ++r) {
        const int64_t i = base + (int64_t)r * kInsThreads;
        if (i < n) partial[(int64_t)blockIdx.y * n + i] = g[r];
    }
}

template <typename T>
__global__ void greens_finish_kernel(const T* __restrict__ partial, int64_t n, int chunks, T* __restrict__ g_out,
                                     uint8_t* __restrict__ inside) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T g = (T)0;
    for (int c = 0; c < chunks; ++c) g = g + partial[(int64_t)c * n + i];
    if (g_out) g_out[i] = g;
    inside[i] = ((double)g < -2.0 * 3.14159265358979323846) ? 1 : 0; // NaN -> outside
}

// ---- 2-D winding ----------------------------------------------------------------------------------
template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920928955078125e-07f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

__device__ inline float atan2_t(float a, float b) { return atan2f(a, b); }
__device__ inline double atan2_t(double a, double b) { return atan2(a, b); }

template <typename T>
__global__ void __launch_bounds__(kInsThreads)
winding_kernel(const T* __restrict__ test, int64_t n, const T* __restrict__ poly, int64_t m, int64_t chunk,
               T* __restrict__ partial, int32_t* __restrict__ coincident) {
    const int64_t i = (int64_t)blockIdx.x * kInsThreads + threadIdx.x;
    const int64_t ii = i < n ? i : n - 1;
    const T x = test[2 * ii], y = test[2 * ii + 1];
    const int64_t j0 = (int64_t)blockIdx.y * chunk;
    const int64_t j1 = j0 + chunk < m ? j0 + chunk : m;
    T sum = (T)0;
    bool hit = false;
    T ux = poly[2 * j0] - x, uy = poly[2 * j0 + 1] - y;
    for (int64_t j = j0; j < j1; ++j) {
        const int64_t jn = j + 1 < m ? j + 1 : 0; // the closing segment (src/isinside.jl:29)
        const T vx = poly[2 * jn] - x, vy = poly[2 * jn + 1] - y;
        hit = hit || (wsqrt(ux * ux + uy * uy) < (T)1.0e2 * Eps<T>::v);
        sum = sum + atan2_t(ux * vy - uy * vx, ux * vx + uy * vy);
        ux = vx;
        uy = vy;
    }
    if (i < n) {
        partial[(int64_t)blockIdx.y * n + i] = sum;
        if (hit) coincident[i] = 1;
    }
}

template <typename T>
__global__ void winding_finish_kernel(const T* __restrict__ partial, int64_t n, int chunks,
                                      const int32_t* __restrict__ coincident, T* __restrict__ sum_out,
                                      uint8_t* __restrict__ inside) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s = (T)0;
    for (int c = 0; c < chunks; ++c) s = s + partial[(int64_t)c * n + i];
    if (sum_out) sum_out[i] = s;
    const T as = s < (T)0 ? -s : s;
    inside[i] = coincident[i] ? 1 : (as < (T)1.0e3 * Eps<T>::v ? 0 : 1);
}

// How many element chunks: enough blocks to fill the chip when N alone does not (<= 64 chunks).
static int pick_chunks(int64_t point_blocks, int64_t m, int sm_count) {
    int64_t want = (4 * (int64_t)sm_count + point_blocks - 1) / point_blocks;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    if (want > m) want = m > 0 ? m : 1;
    return (int)want;
}

template <typename T>
int launch_isinside_greens(wtp_ctx* ctx, const T* d_test, int64_t n, const T* d_p, const T* d_nrm, const T* d_area,
                           int64_t m, void* d_elems, int chunks, T* d_partial, T* d_g, uint8_t* d_inside) {
    Elem<T>* el = (Elem<T>*)d_elems;
    hipLaunchKernelGGL(pack_elems_kernel<T>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_p, d_nrm,
                       d_area, m, el);
    const int64_t pb = (n + kInsThreads * kInsR - 1) / (kInsThreads * kInsR);
    const int64_t chunk = (m + chunks - 1) / chunks;
    hipLaunchKernelGGL(greens_kernel<T>, dim3((unsigned)pb, (unsigned)chunks), dim3(kInsThreads), 0, ctx->stream, d_test,
                       n, (const Elem<T>*)el, m, chunk, d_partial);
    hipLaunchKernelGGL(greens_finish_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const T*)d_partial, n, chunks, d_g, d_inside);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

template <typename T>
int launch_isinside_winding(wtp_ctx* ctx, const T* d_test, int64_t n, const T* d_poly, int64_t m, int chunks,
                            T* d_partial, int32_t* d_coincident, T* d_sum, uint8_t* d_inside) {
    WTP_HIP(ctx, hipMemsetAsync(d_coincident, 0, sizeof(int32_t) * (size_t)n, ctx->stream));
    const int64_t pb = (n + kInsThreads - 1) / kInsThreads;
    const int64_t chunk = (m + chunks - 1) / chunks;
    hipLaunchKernelGGL(winding_kernel<T>, dim3((unsigned)pb, (unsigned)chunks), dim3(kInsThreads), 0, ctx->stream, d_test,
                       n, d_poly, m, chunk, d_partial, d_coincident);
    hipLaunchKernelGGL(winding_finish_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const T*)d_partial, n, chunks, (const int32_t*)d_coincident, d_sum, d_inside);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

int isinside_chunks(wtp_ctx* ctx, int64_t n, int64_t m, int points_per_block) {
    return pick_chunks((n + points_per_block - 1) / points_per_block, m, ctx->sm_count);
}
int isinside_greens_ppb() { return kInsThreads * kInsR; }
int isinside_winding_ppb() { return kInsThreads; }
size_t isinside_elem_bytes(int dtype) { return dtype == WTP_F64 ? sizeof(Elem<double>) : sizeof(Elem<float>); }

#define INST(T)                                                                                              \
    template int launch_isinside_greens<T>(wtp_ctx*, const T*, int64_t, const T*, const T*, const T*, int64_t, void*, \
                                           int, T*, T*, uint8_t*);                                           \
    template int launch_isinside_winding<T>(wtp_ctx*, const T*, int64_t, const T*, int64_t, int, T*, int32_t*, T*, \
                                            uint8_t*);
INST(float)
INST(double)
#undef INST

} // namespace wtp
