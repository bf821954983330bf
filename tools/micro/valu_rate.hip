// Micro-benchmark: sustained issue rate of plain 32-bit VALU instructions on gfx950 as a function
// of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = seed + i * 7919u + threadIdx.x;
    float f[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) f[i] = (float)a[i] * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (KIND == 0) a[i] = min(a[i] ^ 0x55u, a[(i + 1) & 31] + 3u);      // v_xor + v_add + v_min = 3 VALU
                if (KIND == 1) f[i] = f[i] * 1.0001f + f[(i + 1) & 31];             // v_fma (contracted) = 1 VALU
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(uint32_t) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind)
        for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
            int grid = 256 * blocks_per_cu; // 256 CUs, 4 waves per block -> blocks_per_cu waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1u);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double instr_per_wave = (double)iters * 8 * 32 * (kind == 0 ? 3 : 1);
            double waves = (double)grid * 4;
            double total_wave_instr = instr_per_wave * waves;
            double cycles = ms * 1e-3 * 2.4e9;
            printf("kind=%d waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instr per SIMD (at 2.4 GHz)  %.1f T lane-ops/s\n",
                   kind, blocks_per_cu, ms, cycles * 1024.0 / total_wave_instr, total_wave_instr * 64 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
