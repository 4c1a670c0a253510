// debug: does a co-resident kernel corrupt another workgroup's LDS?  Fills `lds_bytes` of dynamic LDS with a pattern, re-checks it `iters`
// times, counts corrupted words per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void canary_kernel(unsigned *bad, unsigned *first_bad, int words, int iters)
{
    extern __shared__ unsigned lds[];
    const unsigned tag = 0xA5000000u | (blockIdx.x << 12);
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = tag ^ i;
    __syncthreads();
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            const unsigned v = ((volatile unsigned *)lds)[i];
            if (v != (tag ^ i)) {
                ++nbad;
                atomicMin(&first_bad[blockIdx.x], (unsigned)i);
                ((volatile unsigned *)lds)[i] = tag ^ i;
            }
        }
        __builtin_amdgcn_s_sleep(20);
    }
    if (nbad) atomicAdd(&bad[blockIdx.x], nbad);
}
extern "C" int canary_launch(void *stream, unsigned *bad, unsigned *first_bad, int blocks, int lds_bytes, int iters)
{
    hipFuncSetAttribute((const void *)canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, bad, first_bad, lds_bytes / 4, iters);
    return (int)hipGetLastError();
}
