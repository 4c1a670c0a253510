// Microbenchmark: do VALU instructions overlap v_mfma_f32_32x32x16_f16 on one SIMD of gfx950?
//   mode 0: every wave issues MFMAs only                    (8 waves / CU = 2 per SIMD)
//   mode 1: every wave issues VALU only
//   mode 2: waves 0-3 MFMA only, waves 4-7 VALU only         (different waves of one SIMD)
//   mode 3: every wave alternates 1 MFMA + K VALU            (same-wave interleave), K = arg
//   mode 4: as 2, the MFMA waves at s_setprio 3;  mode 5: as 2, the VALU waves at s_setprio 3
//   mode 6: waves 4-7 MFMA, waves 0-3 VALU (the YOUNGER wave of a SIMD streams the MFMAs)
//   mode 7: as 2, the VALU waves pause (s_nop 7) after every VALU instruction;  mode 8: as 2, K independent accumulators per MFMA wave
// prints cycles per iteration (s_memtime) of wave 0 and wave 4.     build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int K>
__global__ __launch_bounds__(512, 2) void k(float *out, unsigned long long *cyc, int iters)
{
    const int wid = threadIdx.x >> 6;
    f32x16 acc[4];   // (NACC <= 4 of them in use)
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
    const float c1 = 1.0001f, c2 = 0.5f;
    constexpr bool SPLIT = MODE == 2 || MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8;
    const bool do_mfma = MODE == 0 || MODE == 3 || (SPLIT && wid < 4) || (MODE == 6 && wid >= 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (SPLIT && wid >= 4) || (MODE == 6 && wid < 4);
    if (MODE == 4 && do_mfma) __builtin_amdgcn_s_setprio(3);
    if (MODE == 5 && do_valu) __builtin_amdgcn_s_setprio(3);
    constexpr int NACC = MODE == 8 ? K : 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (do_mfma) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int q = 0; q < (MODE == 3 ? K : 8); ++q) {
                    v[q & 7] = __builtin_fmaf(v[q & 7], c1, c2);
                    if (MODE == 7) { asm volatile("s_nop 7"); }
                }
            }
            if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wid] = t1 - t0;
}

template <int MODE, int K> static void run(const char *what)
{
    float *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-46s wave0 %.1f  wave4 %.1f  memtime ticks per group of 8 (MFMA and/or K VALU each)\n", what, h[0] / (double)iters, h[4] / (double)iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 0>("0: all waves 8 MFMA");
    run<1, 0>("1: all waves 8x8 VALU");
    run<2, 0>("2: waves 0-3 8 MFMA | waves 4-7 8x8 VALU");
    run<3, 1>("3: all waves 8 x (MFMA + 1 VALU)");
    run<3, 2>("3: all waves 8 x (MFMA + 2 VALU)");
    run<3, 4>("3: all waves 8 x (MFMA + 4 VALU)");
    run<3, 8>("3: all waves 8 x (MFMA + 8 VALU)");
    run<4, 0>("4: as 2, MFMA waves at s_setprio 3");
    run<5, 0>("5: as 2, VALU waves at s_setprio 3");
    run<6, 0>("6: waves 4-7 8 MFMA | waves 0-3 8x8 VALU");
    run<7, 0>("7: as 2, s_nop 7 after every VALU");
    run<8, 1>("8: as 2, 1 accumulator (dependent MFMAs)");
    run<8, 2>("8: as 2, 2 accumulators");
    return 0;
}
