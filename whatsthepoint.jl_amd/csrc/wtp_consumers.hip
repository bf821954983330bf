// wtp_consumers.hip — two consumers of the k-NN rows that run where the rows already are.
//
//   compute_normals / update_normals!   src/normals.jl:8-69: per point, the eigenvector of the smallest
//                                       eigenvalue of the covariance of its k nearest points (self
//                                       included) — Hoppe's PCA normals
//   _gradient_limit_field               src/discretization/algorithms/octree.jl:677-717: the g-Lipschitz
//                                       envelope of a per-leaf field by min-plus Jacobi sweeps over the
//                                       k-NN graph of the leaf centres, h[i] <- min(h[i], min_j h[j] + g d_ij),
//                                       until the largest relative change of a sweep drops below tol
//
// Both read the int32 rows (and distances) the topology kernels left in device memory: no host round
// trip between the search and its consumer.
#include "wtp_device.hpp"

namespace wtp {

static constexpr int kConsThreads = 256;

// ---- PCA normals ----------------------------------------------------------------------------------
// Cyclic Jacobi on the symmetric 3x3 (2-D: the z row/column is zero and never rotated).  a = upper
// triangle {xx, xy, xz, yy, yz, zz}; v = eigenvectors as columns.  8 sweeps reach machine precision
// for 3x3 (quadratic convergence); rotations with |apq| == 0 are skipped.
template <typename T> __device__ inline void jacobi_rotate(T& app, T& aqq, T& apq, T& arp, T& arq, T* vp, T* vq) {
    if (apq == (T)0) return;
    const T theta = (aqq - app) / ((T)2 * apq);
    const T at = theta < 0 ? -theta : theta;
    T t = (T)1 / (at + wsqrt(theta * theta + (T)1));
    t = theta < 0 ? -t : t;
    const T c = (T)1 / wsqrt(t * t + (T)1), s = t * c;
    app = app - t * apq;
    aqq = aqq + t * apq;
    apq = (T)0;
    const T rp = arp, rq = arq; // the third index r: rows (r,p) and (r,q)
    arp = c * rp - s * rq;
    arq = s * rp + c * rq;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T a = vp[i], b = vq[i];
        vp[i] = c * a - s * b;
        vq[i] = s * a + c * b;
    }
}

template <typename T>
__global__ void __launch_bounds__(kConsThreads)
pca_normals_kernel(const T* __restrict__ xyz, int64_t n, int dim, const int32_t* __restrict__ rows, int k,
                   T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t* row = rows + i * k;
    T m[3] = {0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const int64_t p = row[j];
        m[0] = m[0] + xyz[p * dim];
        m[1] = m[1] + xyz[p * dim + 1];
        if (dim == 3) m[2] = m[2] + xyz[p * dim + 2];
    }
    const T inv = (T)1 / (T)k;
    m[0] = m[0] * inv, m[1] = m[1] * inv, m[2] = m[2] * inv;
    T xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (int j = 0; j < k; ++j) {
        const int64_t p = row[j];
        const T dx = xyz[p * dim] - m[0], dy = xyz[p * dim + 1] - m[1], dz = dim == 3 ? xyz[p * dim + 2] - m[2] : (T)0;
        xx = xx + dx * dx, xy = xy + dx * dy, xz = xz + dx * dz;
        yy = yy + dy * dy, yz = yz + dy * dz, zz = zz + dz * dz;
    }
    // (the 1/(k-1) of cov() scales the eigenvalues only)
    T vx[3] = {1, 0, 0}, vy[3] = {0, 1, 0}, vz[3] = {0, 0, 1};
    for (int sweep = 0; sweep < 8; ++sweep) {
        jacobi_rotate<T>(xx, yy, xy, xz, yz, vx, vy); // (p,q) = (x,y), r = z
        if (dim == 3) {
            jacobi_rotate<T>(xx, zz, xz, xy, yz, vx, vz); // (x,z), r = y
            jacobi_rotate<T>(yy, zz, yz, xy, xz, vy, vz); // (y,z), r = x
        }
    }
    // eigenvector of the smallest eigenvalue (2-D: among x and y only)
    const T* v = vx;
    T lam = xx;
    if (yy < lam) lam = yy, v = vy;
    if (dim == 3 && zz < lam) lam = zz, v = vz;
    // eigen() fixes no sign; here the component of largest magnitude is made positive
    int big = 0;
    T ab = v[0] < 0 ? -v[0] : v[0];
    for (int c = 1; c < dim; ++c) {
        const T a = v[c] < 0 ? -v[c] : v[c];
        if (a > ab) ab = a, big = c;
    }
    const T sg = v[big] < 0 ? (T)-1 : (T)1;
    for (int c = 0; c < dim; ++c) out[i * dim + c] = sg * v[c];
}

template <typename T>
int launch_pca_normals(wtp_ctx* ctx, const T* d_xyz, int64_t n, int dim, const int32_t* d_rows, int k, T* d_out) {
    hipLaunchKernelGGL(pca_normals_kernel<T>, dim3((unsigned)((n + kConsThreads - 1) / kConsThreads)), dim3(kConsThreads), 0,
                       ctx->stream, d_xyz, n, dim, d_rows, k, d_out);
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

// ---- min-plus gradient limiter -----------------------------------------------------------------------
// state (64-bit words): [0] done flag, [1] sweeps applied, [2] bits of the largest relative change (as a
// double: exact for both types; changes are >= 0, so their bits order like integers) of the sweep in flight
template <typename T>
__global__ void __launch_bounds__(kConsThreads)
minplus_sweep_kernel(const int32_t* __restrict__ rows, const T* __restrict__ dist, int64_t n, int k, T g,
                     const T* __restrict__ h, T* __restrict__ hnew, unsigned long long* __restrict__ state) {
    if (state[0]) return; // converged at an earlier sweep of this batch: the remaining launches are no-ops
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    T rel = 0;
    if (a < n) {
        const T h0 = h[a];
        T hi = h0;
        for (int t = 0; t < k; ++t) {
            const T cand = h[rows[a * k + t]] + g * dist[a * k + t];
            hi = cand < hi ? cand : hi;
        }
        hnew[a] = hi;
        const T d = hi - h0;
        rel = (d < 0 ? -d : d) / h0;
    }
    // block maximum -> one atomic
    __shared__ double sm[kConsThreads / 64];
    double r = (double)rel;
    r = r == r ? r : __builtin_huge_val(); // NaN (a zero field value): never "converged"
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(r, o);
        r = other > r ? other : r;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kConsThreads / 64; ++w) r = sm[w] > r ? sm[w] : r;
        if (r > 0) atomicMax(&state[2], (unsigned long long)__double_as_longlong(r));
    }
}

__global__ void minplus_finish_kernel(double tol, unsigned long long* __restrict__ state) {
    if (state[0]) return;
    state[1] += 1;
    if (__longlong_as_double((long long)state[2]) < tol) state[0] = 1; // maxrel < tol: stop (octree.jl:707)
    state[2] = 0;
}

// One batch of `sweeps` sweeps; buffers alternate, starting with src = bufs[first & 1].
template <typename T>
int launch_minplus_batch(wtp_ctx* ctx, const int32_t* d_rows, const T* d_dist, int64_t n, int k, double g, double tol,
                         T* d_h0, T* d_h1, int first, int sweeps, unsigned long long* d_state) {
    const unsigned nb = (unsigned)((n + kConsThreads - 1) / kConsThreads);
    for (int s = 0; s < sweeps; ++s) {
        T* src = ((first + s) & 1) ? d_h1 : d_h0;
        T* dst = ((first + s) & 1) ? d_h0 : d_h1;
        hipLaunchKernelGGL(minplus_sweep_kernel<T>, dim3(nb), dim3(kConsThreads), 0, ctx->stream, d_rows, d_dist, n, k,
                           (T)g, (const T*)src, dst, d_state);
        hipLaunchKernelGGL(minplus_finish_kernel, dim3(1), dim3(1), 0, ctx->stream, tol, d_state);
    }
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

#define INST(T)                                                                                               \
    template int launch_pca_normals<T>(wtp_ctx*, const T*, int64_t, int, const int32_t*, int, T*);            \
    template int launch_minplus_batch<T>(wtp_ctx*, const int32_t*, const T*, int64_t, int, double, double, T*, T*, \
                                         int, int, unsigned long long*);
INST(float)
INST(double)
#undef INST

} // namespace wtp
