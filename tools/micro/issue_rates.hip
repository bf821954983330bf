// Micro-benchmark: issue cost of wave64 VALU / MFMA instructions on gfx950, in SHADER CYCLES
// (s_memtime stamps around the loop, so no clock assumption enters), for 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip ; run: ./issue_rates
// Output = the table committed under profiles/ (raw), quoted by DESIGN.md §5.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0 v_fma_f32  1 v_pk_fma_f32  2 v_add_u32  3 v_min_u32  4 v_cmp_le_f32+v_cndmask (2 instr)
//      5 v_mfma_f32_16x16x4_f32 alone  6 MFMA + 4 VALU  7 MFMA + 8 VALU  8 MFMA + 12 VALU
//      9 v_mul_f32  10 v_sub_f32  11 v_and_b32  12 v_min_f32  13 v_max_u32  14 v_min_i32  15 v_med3_f32  16 v_min3_u32
constexpr int kUnroll = 16;

template <int KIND>
__global__ __launch_bounds__(256, 8) void k(uint64_t* cyc, float* sink, int iters) {
    float a[kUnroll];
    uint32_t u[kUnroll];
    f32x2 p[kUnroll];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
        a[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
        u[i] = threadIdx.x * 2654435761u + i;
        p[i] = f32x2{a[i], a[i] * 0.5f};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float m = 0.999f, c = 1e-6f;
    const f32x2 pm = {0.999f, 0.998f}, pc = {1e-6f, 2e-6f};
    __syncthreads();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kUnroll]));
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kUnroll]));
        } else if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i)
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
        } else if (KIND == 9) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        } else if (KIND == 10) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 11) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kUnroll]));
        } else if (KIND == 12) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % kUnroll]));
        } else if (KIND == 13) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kUnroll]));
        } else if (KIND == 14) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kUnroll]));
        } else if (KIND == 15) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) % kUnroll]), "v"(m));
        } else if (KIND == 16) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % kUnroll]), "v"(u[(i + 2) % kUnroll]));
        } else {
            constexpr int NV = KIND == 5 ? 0 : (KIND == 6 ? 4 : (KIND == 7 ? 8 : 12));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], a[j + 4], acc[j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[8 + (i & 7)]) : "v"(m), "v"(c));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (KIND == 0 || KIND == 4 || KIND == 9 || KIND == 10 || KIND == 12 || KIND == 15 || (KIND >= 5 && KIND <= 8)) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) s += a[i];
    }
    if (KIND == 2 || KIND == 3 || KIND == 11 || KIND == 13 || KIND == 14 || KIND == 16) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) s += (float)u[i];
    }
    if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) s += p[i].x + p[i].y;
    }
    if (KIND >= 5 && KIND <= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

// MIX: waves 0,1 of a block issue MFMAs only, waves 2,3 VALU only (co-residency across SIMDs is the
// dispatcher's choice; with 4 blocks per CU every SIMD holds both kinds)
__global__ __launch_bounds__(256) void kmix(uint64_t* cyc, float* sink, int iters) {
    float a[kUnroll];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) a[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float m = 0.999f, c = 1e-6f;
    const bool mf = ((threadIdx.x >> 6) & 1) == 0;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], a[j + 4], acc[j], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < kUnroll; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_min_u32", "v_cmp_le_f32+v_cndmask_b32",
                              "v_mfma_f32_16x16x4_f32", "mfma16x16x4 + 4 v_fma", "mfma16x16x4 + 8 v_fma",
                              "mfma16x16x4 + 12 v_fma", "v_mul_f32", "v_sub_f32", "v_and_b32", "v_min_f32", "v_max_u32", "v_min_i32",
                              "v_med3_f32", "v_min3_u32"};

template <int KIND> static void run(uint64_t* dc, float* ds, int wps) {
    const int iters = 100000;
    const int grid = 256 * wps;
    std::vector<uint64_t> h2(grid * 8), h(grid * 4), hr(grid * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, dc, ds, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h2.data(), dc, h2.size() * 8, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) { h[i] = h2[2 * i]; hr[i] = h2[2 * i + 1]; }
    std::sort(h.begin(), h.end());
    std::sort(hr.begin(), hr.end());
    const double med = (double)h[h.size() / 2];
    const double ghz = med / ((double)hr[hr.size() / 2] * 10.0); // s_memrealtime ticks at 100 MHz
    // instructions one wave issued inside the stamps
    double per_wave = 0;
    if (KIND <= 3 || KIND >= 9) per_wave = (double)iters * kUnroll;
    else if (KIND == 4) per_wave = (double)iters * kUnroll * 2;
    else per_wave = (double)iters * 4; // MFMAs
    // a SIMD holds `wps` waves (4 waves of a block go to the 4 SIMDs)
    const double cyc_per_instr_simd = med / (per_wave * wps);
    printf("%-28s waves/SIMD=%d  wall %.3f ms  median wave %.0f cyc  => %.2f cyc per wave-instr per SIMD%s  shader clock %.2f GHz => %.2f ns (wall-based %.2f ns)\n",
           names[KIND], wps, ms, med, cyc_per_instr_simd, (KIND >= 5 && KIND <= 8) ? " [per MFMA]" : "",
           ghz, cyc_per_instr_simd / ghz, ms * 1e6 / (per_wave * wps));
}

int main() {
    uint64_t* dc;
    float* ds;
    hipMalloc(&dc, 256 * 8 * 4 * 8 * 2);
    hipMalloc(&ds, 256 * 8 * 256 * 4);
    for (int wps = 1; wps <= 8; wps *= 2) {
        run<0>(dc, ds, wps);
        run<1>(dc, ds, wps);
        run<2>(dc, ds, wps);
        run<3>(dc, ds, wps);
        run<4>(dc, ds, wps);
        run<9>(dc, ds, wps);
        run<10>(dc, ds, wps);
        run<11>(dc, ds, wps);
        run<12>(dc, ds, wps);
        run<13>(dc, ds, wps);
        run<14>(dc, ds, wps);
        run<15>(dc, ds, wps);
        run<16>(dc, ds, wps);
        run<5>(dc, ds, wps);
        run<6>(dc, ds, wps);
        run<7>(dc, ds, wps);
        run<8>(dc, ds, wps);
    }
    // mixed: MFMA-only and VALU-only waves co-resident
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int iters = 100000, grid = 256 * wps;
        std::vector<uint64_t> h(grid * 4);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(kmix, dim3(grid), dim3(256), 0, 0, dc, ds, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<uint64_t> mf, va;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 4; ++w) ((w & 1) == 0 ? mf : va).push_back(h[b * 4 + w]);
        std::sort(mf.begin(), mf.end());
        std::sort(va.begin(), va.end());
        printf("mixed waves/SIMD=%d: MFMA waves median %.0f cyc for %d MFMAs (%.1f cyc each), VALU waves median %.0f cyc for %d v_fma (%.2f cyc each)\n",
               wps, (double)mf[mf.size() / 2], iters * 4, (double)mf[mf.size() / 2] / (iters * 4.0), (double)va[va.size() / 2],
               iters * kUnroll, (double)va[va.size() / 2] / (iters * (double)kUnroll));
    }
    return 0;
}
