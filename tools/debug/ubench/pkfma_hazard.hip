// Root-cause probe for the hazard parked in round 3: a packed-fp32 VALU kernel (v_pk_fma_f32) produced wrong
// single accumulators next to a split-fp16 MFMA kernel.  Minimal form: inside ONE workgroup of 8 waves (two per SIMD, so co-residence on
// a SIMD is guaranteed) waves 0-3 stream v_mfma_f32_32x32x16_f16 while waves 4-7 run a long chain of packed / scalar fp32 FMAs whose
// result is known exactly; also the VALU waves alone, and the two as separate kernels on two streams.  Prints mismatching lanes.
//   build: hipcc --offload-arch=gfx950 -O3 pkfma_hazard.hip -o pkfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the checked computation: 8 independent accumulators per lane (4 packed pairs), ITER rounds of  acc = acc * a + b  with exactly
// representable operands (powers of two and small integers: every intermediate is exact in fp32, so any order / fusion gives the same bits)
template <int PACKED>
__device__ __forceinline__ void chain(float (&acc)[8], int iters, int lane)
{
    const float a = 0.5f, b = (float)(lane & 15);
    for (int it = 0; it < iters; ++it) {
        if (PACKED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x2 v = {acc[2 * q], acc[2 * q + 1]};
                const f32x2 av = {a, a}, bv = {b, b + 1.0f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(av), "v"(bv));
                acc[2 * q] = v[0]; acc[2 * q + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_fmaf(acc[q], a, b + (float)(q & 1));
        }
    }
}
static void host_chain(float *acc, int iters, int lane)
{
    const float a = 0.5f, b = (float)(lane & 15);
    for (int it = 0; it < iters; ++it) for (int q = 0; q < 8; ++q) acc[q] = acc[q] * a + (b + (float)(q & 1));   // exact: no rounding occurs
}

template <int PACKED, int MIXED>      // MIXED: waves 0-3 stream MFMAs in the same workgroup
__global__ __launch_bounds__(512, 2) void probe(float *out, float *sink, int iters)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (MIXED && wid < 4) {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        f16x8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(lane * 0.01f + j); y[j] = (_Float16)(j * 0.25f); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[u & 3], 0, 0, 0);
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    float acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = (float)(lane + q);
    chain<PACKED>(acc, iters * 4, lane);
    for (int q = 0; q < 8; ++q) out[((size_t)blockIdx.x * 512 + threadIdx.x) * 8 + q] = acc[q];
}

template <int PACKED, int MIXED> static long run(const char *what, int reps, int iters, bool side_mfma)
{
    const int blocks = 512;
    float *out, *sink, *sink2;
    hipMalloc(&out, (size_t)blocks * 512 * 8 * 4); hipMalloc(&sink, (size_t)blocks * 512 * 4); hipMalloc(&sink2, (size_t)blocks * 512 * 4);
    float *h = (float *)malloc((size_t)blocks * 512 * 8 * 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    long bad = 0, badlanes = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipMemsetAsync(out, 0xff, (size_t)blocks * 512 * 8 * 4, s1);
        if (side_mfma) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((probe<0, 1>), dim3(blocks), dim3(512), 0, s2, out + 0, sink2, iters);   // MFMA-heavy neighbour (its VALU half writes `out` too -- same values)
        hipLaunchKernelGGL((probe<PACKED, MIXED>), dim3(blocks), dim3(512), 0, s1, out, sink, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, out, (size_t)blocks * 512 * 8 * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < blocks; ++b)
            for (int t = (MIXED ? 256 : 0); t < 512; ++t) {
                float ref[8]; for (int q = 0; q < 8; ++q) ref[q] = (float)((t & 63) + q);
                host_chain(ref, iters * 4, t & 63);
                int lb = 0;
                for (int q = 0; q < 8; ++q) if (memcmp(&ref[q], &h[((size_t)b * 512 + t) * 8 + q], 4)) { ++bad; lb = 1; }
                badlanes += lb;
            }
    }
    printf("%-78s %d launches: %ld wrong accumulators in %ld lanes\n", what, reps, bad, badlanes);
    hipFree(out); hipFree(sink); hipFree(sink2); free(h); hipStreamDestroy(s1); hipStreamDestroy(s2);
    return bad;
}

int main()
{
    const int reps = 40, iters = 400;
    long bad = 0;
    bad += run<1, 0>("v_pk_fma_f32 chains alone", reps, iters, false);
    bad += run<0, 1>("v_fma_f32 chains, waves 4-7, f16 MFMA stream on waves 0-3 of the same workgroup", reps, iters, false);
    bad += run<1, 1>("v_pk_fma_f32 chains, waves 4-7, f16 MFMA stream on waves 0-3 of the same workgroup", reps, iters, false);
    bad += run<1, 0>("v_pk_fma_f32 chains, MFMA-streaming kernel on a second HIP stream", reps, iters, true);
    bad += run<1, 1>("v_pk_fma_f32 + in-workgroup MFMA waves, MFMA kernel on a second stream too", reps, iters, true);
    printf(bad ? "HAZARD REPRODUCED\n" : "no mismatch in any configuration\n");
    return 0;
}
