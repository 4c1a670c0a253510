// Calibration of rocprofv3's FETCH_SIZE for the access pattern of bsvd::conv3x3_kernel's patch staging (tools/, not product).
//
// The guide's "x2" correction is stated for wide coalesced streaming reads only.  The conv kernel's staging read is different:
// every lane issues one raw_buffer_load_b128 (16 B); 4 adjacent lanes cover the 64 B of one pixel's 16-channel chunk; consecutive
// 4-lane groups walk along a patch row at a pixel stride of Cin*4 bytes; a workgroup reads a PH x PW pixel patch per chunk and
// walks the Cin/16 chunks.  This program replays exactly that (same thread -> (row, column, quad) map, same XCD-aware
// workgroup -> tile order, same descriptor flags) over a tensor of known size, in two forms:
//   halo=0 : PH x PW = TH x TW, tiles do not overlap -> every byte of the tensor is requested exactly once (known byte count)
//   halo=1 : the real (TH+2) x (TW+2) patch -> what share of the 1-pixel halo re-reads reaches the memory-side counter
// and as a plain coalesced stream (mode 2) for reference.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`; the per-kernel
// counter divided into the bytes printed here gives the factor tools/make_traffic.py applies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define OOB 0x7fffffffu
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)

template <int TH, int TW, int HALO>
__global__ __launch_bounds__(256) void cal_patch(const float *x, float *sink, int frames, int H, int W, int Cin, int nty, int ntx)
{
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int ROW_ITEMS = PW * 4, R = 256 / ROW_ITEMS;
    const int tid = threadIdx.x;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tx = lid % ntx; lid /= ntx;
    const int ty = lid % nty;
    const int f = lid / nty;
    const int iy0 = ty * TH - HALO, ix0 = tx * TW - HALO;
    const float *cur = x + (size_t)f * H * W * Cin;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cur), 0, (unsigned)H * W * Cin * 4u, 0x00020000);
    const int r3 = tid / ROW_ITEMS, rem = tid - r3 * ROW_ITEMS, pcol = rem >> 2, pq = rem & 3;
    const int gx = ix0 + pcol;
    const bool x_ok = r3 < R && gx >= 0 && gx < W;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < (Cin >> 4); ++cb)
        for (int row0 = 0; row0 < PH; row0 += R) {
            const int prow = row0 + r3, gy = iy0 + prow;
            const bool ok = x_ok && prow < PH && gy >= 0 && gy < H;
            const unsigned voff = ok ? (unsigned)(gy * W + gx) * (unsigned)Cin * 4u + pq * 16u : OOB;
            acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)cb * 64u, 0));
        }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[bid] = acc[0];       // keeps the loads alive, never true
}

__global__ __launch_bounds__(256) void cal_stream(const f32x4 *x, float *sink, size_t n4)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += x[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x] = acc[0];
}

template <int TH, int TW, int HALO>
static void run(const char *name, const float *x, float *sink, int frames, int H, int W, int Cin, int reps)
{
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW;
    const int nblk = frames * nty * ntx;
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((cal_patch<TH, TW, HALO>), dim3(nblk), dim3(256), 0, 0, x, sink, frames, H, W, Cin, nty, ntx);
    CHECK(hipDeviceSynchronize());
    const double tensor = (double)frames * H * W * Cin * 4.0;
    double req = tensor;
    if (HALO) {          // bytes REQUESTED incl. the in-range part of every tile's 1-pixel halo
        double pix = 0;
        for (int ty = 0; ty < nty; ++ty)
            for (int tx = 0; tx < ntx; ++tx) {
                int y0 = ty * TH - 1, y1 = ty * TH + TH + 1, x0 = tx * TW - 1, x1 = tx * TW + TW + 1;
                y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0; y1 = y1 > H ? H : y1; x1 = x1 > W ? W : x1;
                pix += (double)(y1 - y0) * (x1 - x0);
            }
        req = pix * frames * Cin * 4.0;
    }
    printf("CAL %s tensor_bytes %.0f requested_bytes %.0f launches %d\n", name, tensor, req, reps);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const int mask = argc > 2 ? atoi(argv[2]) : 0xff;
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int F = 10;
    const size_t n = (size_t)F * 540 * 960 * 64;          // 1.327 GB: five times the 256 MiB Infinity Cache
    float *x, *sink;
    CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&sink, 1 << 22));
    CHECK(hipMemset(x, 0x3c, n * 4));
    // 64-channel 540x960 layers: 256-px tiles (16 x 16); 128-channel 270x480 layers: fat tile (16 x 16 px per workgroup pair ...)
    printf("allocated\n");
    if (mask & 1) run<16, 16, 0>("patch16x16_c64_nohalo", x, sink, F, 540, 960, 64, reps);
    if (mask & 2) run<16, 16, 1>("patch16x16_c64_halo", x, sink, F, 540, 960, 64, reps);
    if (mask & 4) run<16, 16, 0>("patch16x16_c128_nohalo", x, sink, 2 * F, 270, 480, 128, reps);      // same bytes, Cin = 128 pixel stride, the 128-channel layers' geometry
    if (mask & 8) run<16, 16, 1>("patch16x16_c128_halo", x, sink, 2 * F, 270, 480, 128, reps);
    if (mask & 16) run<8, 16, 0>("patch8x16_c128_nohalo", x, sink, 2 * F, 270, 480, 128, reps);
    if (mask & 32) run<8, 16, 1>("patch8x16_c128_halo", x, sink, 2 * F, 270, 480, 128, reps);
    if (mask & 64) {
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(cal_stream, dim3(256 * 16), dim3(256), 0, 0, (const f32x4 *)x, sink, n / 4);
        CHECK(hipDeviceSynchronize());
        printf("CAL stream tensor_bytes %.0f requested_bytes %.0f launches %d\n", (double)n * 4, (double)n * 4, reps);
    }
    return 0;
}
