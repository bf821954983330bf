// Micro-benchmark: cost per (query, candidate) of three candidate filters for the sweep's scan loop on gfx950,
// with the LDS traffic each one implies, 4 workgroups of 256 threads per CU like cs2_kernel.
//   F32  ds_read_b128 per candidate; 3 v_sub_f32 + v_mul_f32 + 2 v_fma_f32 + v_cmp + ds_write_b8 + v_addc   (round 2)
//   I16  8 bytes per candidate ([x|y], [z|0] as 16-bit fixed point); 2 v_pk_sub_i16 + 2 v_dot2_i32_i16 + append
//   I8   4 bytes per candidate ([x|y|z|0] bytes); v_sub_u32 + v_xor_b32 + v_dot4_i32_i8 + append
// Build: hipcc --offload-arch=gfx950 -O3 -o scan_filters scan_filters.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

constexpr int kSlots = 1024; // staged candidates per workgroup
constexpr int kRun = 48;     // slots a query scans

template <int KIND> __global__ __launch_bounds__(256, 4) void k(uint32_t* out, int rounds, float thr_f, int thr_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    // fill: pseudo-random coordinates in [0,4) cells
    for (int i = tid; i < kSlots + 16; i += 256) {
        uint32_t h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        const float x = (float)(i / 16) + (float)(h & 1023) / 1024.f, y = (float)((h >> 10) & 4095) / 1024.f,
                    z = (float)((h >> 22) & 1023) / 256.f;
        if (KIND == 0) reinterpret_cast<f4*>(smem)[i] = f4{x, y, z, 0.f};
        if (KIND == 1) {
            const uint32_t xi = (uint32_t)(x * 1024.f) & 0xFFFFu, yi = (uint32_t)(y * 1024.f) & 0xFFFFu, zi = (uint32_t)(z * 1024.f) & 0xFFFFu;
            reinterpret_cast<u2*>(smem)[i] = u2{xi | (yi << 16), zi};
        }
        if (KIND == 2) {
            const uint32_t xi = (uint32_t)(x * 64.f) & 0xFFu, yi = (uint32_t)(y * 64.f) & 0xFFu, zi = (uint32_t)(z * 64.f) & 0xFFu;
            reinterpret_cast<uint32_t*>(smem)[i] = (xi << 24) | (yi << 16) | (zi << 8);
        }
    }
    const uint32_t ring0 = kSlots * 16 + 256 + tid * 36;
    __syncthreads();
    uint32_t total = 0;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
    for (int r = 0; r < rounds; ++r) {
        const int q = (tid * 3 + r * 7) % (kSlots - kRun - 16);
        uint32_t ra = lds_base + ring0;
        const uint32_t ra0 = ra;
        if (KIND == 0) {
            const f4 qp = reinterpret_cast<f4*>(smem)[q + 20];
            uint32_t addr = lds_base + q * 16;
            for (uint32_t i0 = 0; i0 < kRun; i0 += 8, addr += 128) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f4 c[4];
                    asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]) : "v"(addr + h * 64) : "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ex = qp.x - c[u].x, ey = qp.y - c[u].y, ez = qp.z - c[u].z;
                        const float d = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                        const uint32_t val = i0 + h * 4 + u;
                        asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tds_write_b8 %0, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(ra) : "v"(d), "v"(thr_f), "v"(val) : "vcc", "memory");
                    }
                }
            }
        } else if (KIND == 1) {
            const u2 qp = reinterpret_cast<u2*>(smem)[q + 20];
            uint32_t addr = lds_base + q * 8;
            for (uint32_t i0 = 0; i0 < kRun; i0 += 8, addr += 64) {
                u4 c[4];
                asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]) : "v"(addr) : "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t cxy = (u & 1) ? c[u >> 1].z : c[u >> 1].x, cz = (u & 1) ? c[u >> 1].w : c[u >> 1].y;
                    uint32_t dxy, dz, d;
                    asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(dxy) : "v"(qp.x), "v"(cxy));
                    asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(dz) : "v"(qp.y), "v"(cz));
                    asm volatile("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(d) : "v"(dxy));
                    asm volatile("v_dot2_i32_i16 %0, %1, %1, %2" : "=v"(d) : "v"(dz), "v"(d));
                    const uint32_t val = i0 + u;
                    asm volatile("v_cmp_le_i32 vcc, %1, %2\n\tds_write_b8 %0, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(ra) : "v"(d), "v"(thr_i), "v"(val) : "vcc", "memory");
                }
            }
        } else {
            const uint32_t qp = reinterpret_cast<uint32_t*>(smem)[q + 20] | 0x00808080u; // (stand-in for the bias)
            uint32_t addr = lds_base + q * 4;
            for (uint32_t i0 = 0; i0 < kRun; i0 += 16, addr += 64) {
                u4 c[4];
                asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]) : "v"(addr) : "memory");
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t cc = c[u >> 2][u & 3];
                    uint32_t t, d;
                    asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t) : "v"(qp), "v"(cc));
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(0x80808080u));
                    asm volatile("v_dot4_i32_i8 %0, %1, %1, 0" : "=v"(d) : "v"(t));
                    const uint32_t val = i0 + u;
                    asm volatile("v_cmp_le_i32 vcc, %1, %2\n\tds_write_b8 %0, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(ra) : "v"(d), "v"(thr_i), "v"(val) : "vcc", "memory");
                }
            }
        }
        total += ra - ra0;
    }
    out[blockIdx.x * 256 + tid] = total;
}

template <int KIND> static void run(const char* name, uint32_t* d_out, float thr_f, int thr_i) {
    const int rounds = 2000, grid = 256 * 4;
    const size_t smem = kSlots * 16 + 256 + 256 * 36 + 64;
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), smem, 0, d_out, rounds, thr_f, thr_i);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<uint32_t> h(grid * 256);
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    double hits = 0;
    for (auto v : h) hits += v;
    const double pairs = (double)grid * 256 * rounds * (KIND == 2 ? 48 : kRun);
    printf("%-4s %.3f ms  %.1f G (query,candidate)/s  %.4f ns per pair per CU-lane-slot  hits/query %.2f\n", name, ms,
           pairs / ms * 1e-6, ms * 1e6 / ((double)rounds * (KIND == 2 ? 48 : kRun)) / 16.0, hits / ((double)grid * 256 * rounds));
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("F32", d, 1.0f, 0);
    run<1>("I16", d, 0.f, 1024 * 1024);
    run<2>("I8", d, 0.f, 64 * 64);
    run<0>("F32", d, 1.0f, 0);
    run<1>("I16", d, 0.f, 1024 * 1024);
    run<2>("I8", d, 0.f, 64 * 64);
    return 0;
}
