// conv3x3_edge_f32.hip -- the two bandwidth-bound edge layers of the BSVD network (exact fp32, VALU).
//
// layer 1 (Cin = 4 -> 64, K = 36) and layer 32 (64 -> Cout = 3) are < 0.4 % of the FLOPs but
// padding them to MFMA tiles costs ~5 % of the clip time, and they sit next to the NCHW <-> NHWC layout change.
// So they get dedicated kernels that also absorb that change:
//
//   head_kernel<CIN>  : x planar NCHW [f][CIN][H][W] (the caller's tensor, bsvd_arch.py:494-499)
//                       -> y NHWC [f][H][W][Cout_pad], bias + act.              (InputCvBlock conv 0, :208-209)
//   tail_kernel<COUT> : (exact-fp32 mode only; in split mode the exit layer runs on the MFMA kernel with a planar epilogue)
//                       x NHWC [f][H][W][Cin_pad] -> y planar NCHW [f][COUT][H][W], bias (+act),
//                       residual y[c] = base[c] - y[c] (c < resid_ch) and optional clamp.
//                                                   (OutputCvBlock conv 3 :298, none_minus :408-414, clamp of
//                                                    validation_seq_infer.py:24, torch.cat :552)
//
// Both read the SAME packed weight layout as the MFMA kernel ([Cin_pad/16][9][4][Cout_pad][4]); weights are
// wave-uniform and travel through the scalar cache into SGPR operands of v_fma_f32.  Accumulation order is a plain
// fp32 fmaf chain over (chunk, tap, channel) -- exact fp32 like the MFMA path.
#include "bsvd_internal.h"

namespace bsvd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Weights/bias are read-only for the whole launch and indexed wave-uniformly: reading them through the constant
// address space lets the compiler use scalar loads (SGPR operands) even with stores in the same loop.
typedef const __attribute__((address_space(4))) float *cfloat_p;
__device__ __forceinline__ cfloat_p as_const(const float *p) { return (cfloat_p)(uintptr_t)p; }

__device__ __forceinline__ float edge_act(float v, int act)
{
    if (act >= BSVD_ACT_RELU) v = fmaxf(v, 0.f);
    if (act == BSVD_ACT_RELU6) v = fminf(v, 6.f);
    return v;
}

// ------------------------------------------------------------------------------------------------------
// (r03: a two-pixels-per-thread version on v_pk_fma_f32 with the weights broadcast from LDS ran 0.53 ms instead of 0.62 ms per
//  10-frame 540x960 clip and is bit-identical when it runs alone -- but while a split-fp16 MFMA kernel of ANOTHER stream shares the
//  SIMDs (the two-branch graphs of streaming_forward) single accumulators came out wrong in the last 16 lanes of a wave in 30-100 %
//  of the runs (tools/debug/head_stress.py; never with an fp32-MFMA or a copy kernel alongside, never alone).  Ruled out:
//  inline-asm hazards (compiler-generated v_pk_fma_f32 behaves the same), the store-data / LDS-return register reuse of the
//  write-back, the refill distance of the weight registers, LDS corruption by the neighbour kernel (canary workgroups stay
//  intact).  Not understood -> not shipped; this one-pixel kernel is deterministic under the same stress.)
// head: thread = one pixel; workgroup = 64 x 4 pixels (a wave reads 64 consecutive x of one row: coalesced planes)
template <int CIN>
__global__ __launch_bounds__(256) void head_kernel(const ConvParams p)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ntx = (p.W + 63) >> 6, nty = (p.H + 3) >> 2;
    int bid = blockIdx.x;
    const int bx = bid % ntx; bid /= ntx;
    const int by = bid % nty;
    const int f = bid / nty;
    const int ox = bx * 64 + tx, oy = by * 4 + ty;
    const bool live = ox < p.W && oy < p.H;

    const float *xin = p.x + (int64_t)f * p.x_fs;
    const int64_t plane = (int64_t)p.H * p.W;
    float in[9][CIN];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool ok = live && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) in[ky * 3 + kx][c] = ok ? xin[c * plane + (int64_t)iy * p.W + ix] : 0.f;
        }

    // Output staging: a thread owns one pixel, but 64-byte-per-lane stores at a 256-B lane stride reach HBM as partial
    // lines (PMC: 1.6x the algorithmic bytes written).  Each wave therefore parks 32 output channels of its 64 pixels in
    // a private LDS scratch ([px][32 + 4 pad] floats) and writes them back so that 8 consecutive lanes cover one pixel's
    // 128 contiguous bytes (a full L2 line per 8 lanes).
    __shared__ __attribute__((aligned(16))) float stage[4][64 * 36];
    float *sc = stage[ty];
    float *yrow = p.y + (int64_t)f * p.y_fs + ((int64_t)oy * p.W + (ox - tx)) * p.Cout;   // pixel 0 of this wave's row segment
    const bool row_live = oy < p.H;
    const cfloat_p w = as_const(p.w);
    const cfloat_p bias = as_const(p.bias);
    for (int nb = 0; nb < p.Cout; nb += 16) {            // wave-uniform: weights below come through the scalar cache
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[nb + j] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const cfloat_p wt = w + ((int64_t)(tap * 4) * p.Cout + nb) * 4;     // [k4 = 0][n = nb..nb+15][4]
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(in[tap][c], wt[j * 4 + c], acc[j]);
        }
        f32x4 o[4];
        if (p.prec == 1) {                  // split16 output: [hi x16 | lo x16] in the chunk's 64 bytes
            f16x8 hi[2], lo[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = fminf(fmaxf(edge_act(acc[j], p.act), -65504.f), 65504.f);   // saturate, never inf/NaN pairs
                const _Float16 h = (_Float16)v;
                hi[j >> 3][j & 7] = h;
                lo[j >> 3][j & 7] = lo_keep((_Float16)(v - (float)h));
            }
            o[0] = __builtin_bit_cast(f32x4, hi[0]); o[1] = __builtin_bit_cast(f32x4, hi[1]);
            o[2] = __builtin_bit_cast(f32x4, lo[0]); o[3] = __builtin_bit_cast(f32x4, lo[1]);
        } else {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j4][j] = edge_act(acc[j4 * 4 + j], p.act);
        }
        const int half = (nb >> 4) & 1;       // which 16-channel half of the 32-channel staging group
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) *reinterpret_cast<f32x4 *>(sc + tx * 36 + half * 16 + j4 * 4) = o[j4];
        if (half == 1 || nb + 16 >= p.Cout) {              // group complete -> wave-private write-back
            const int g0 = nb - half * 16;                  // first channel of the group
            const int gq = (nb + 16 - g0) >> 2;             // float4 items per pixel in this group (4 or 8)
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = it * 64 + tx;
                const int px = item >> 3, q = item & 7;
                if (row_live && q < gq && bx * 64 + px < p.W)
                    *reinterpret_cast<f32x4 *>(yrow + (int64_t)px * p.Cout + g0 + q * 4) =
                        *reinterpret_cast<const f32x4 *>(sc + px * 36 + q * 4);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// tail: workgroup = 16 x 32 pixels, thread = two pixels 16 rows apart (the weights of a tap are read from LDS once and used
// for both: 5 LDS reads per 24 FMAs instead of 4 per 12 -- the one-pixel version was LDS-issue bound).  Per 16-channel
// chunk the 34 x 18 patch goes through a single LDS buffer: next chunk's loads are in flight during the 9 taps, stored
// between two barriers.
struct TailCfg {
    static constexpr int TW = 16, TH = 32, PWD = TW + 2, PHT = TH + 2, PS = 20, NP = PWD * PHT, NQ = NP * 4;
    static constexpr int PATCH_BYTES = NP * PS * 4;
    static constexpr int MAX_CHUNKS = 16;                       // Cin <= 256
    static int lds_bytes(int ncb, int cout) { return PATCH_BYTES + ncb * 36 * cout * 16; }
};

template <int COUT>
__global__ __launch_bounds__(256, 2) void tail_kernel(const ConvParams p, int y_planar_ch, int do_clamp, float lo, float hi)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using C = TailCfg;
    const int tid = threadIdx.x;
    const int px = tid & 15, py = tid >> 4;
    const int ntx = (p.W + C::TW - 1) / C::TW, nty = (p.H + C::TH - 1) / C::TH;
    int bid = blockIdx.x;
    const int bx = bid % ntx; bid /= ntx;
    const int by = bid % nty;
    const int f = bid / nty;
    const int ox0 = bx * C::TW, oy0 = by * C::TH;
    const float *xin = p.x + (int64_t)f * p.x_fs;
    const int ncb = p.Cin >> 4;

    // staging items of this thread: e = tid + i*256 -> (patch pixel, channel quad); precomputed once
    constexpr int NI = (C::NQ + 255) / 256;
    int g_off[NI], l_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;
        const int pix = e >> 2, q = e & 3;
        const int r = pix / C::PWD, c = pix - r * C::PWD;
        const int iy = oy0 + r - 1, ix = ox0 + c - 1;
        const bool ok = e < C::NQ && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_off[i] = ok ? (iy * p.W + ix) * p.Cin + q * 4 : -1;      // < 2^31 elements per frame (checked by the host)
        l_off[i] = e < C::NQ ? pix * C::PS + q * 4 : -1;
    }
    auto stage_load = [&](int cb, f32x4 (&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g_off[i] >= 0) v[i] = *reinterpret_cast<const f32x4 *>(xin + g_off[i] + cb * 16);
        }
    };
    auto stage_store = [&](float *buf, const f32x4 (&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (l_off[i] >= 0) *reinterpret_cast<f32x4 *>(buf + l_off[i]) = v[i];
    };

    float acc[2][COUT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[u][n] = 0.f;

    // weights of the COUT live output channels -> LDS once per workgroup ([chunk][tap][k4][n][4]); every lane then reads
    // the same 16 bytes (LDS broadcast).  Scalar loads were the bottleneck here: 36 dependent s_load round trips per
    // chunk (1.25 ms per 10-frame clip vs 0.45 ms of HBM time).
    float *wl = smem + C::NP * C::PS;
    for (int e = tid; e < ncb * 36 * COUT; e += 256) {
        const int n = e % COUT, slab = e / COUT;            // slab = (cb*9 + tap)*4 + k4
        *reinterpret_cast<f32x4 *>(wl + e * 4) = *reinterpret_cast<const f32x4 *>(p.w + ((int64_t)slab * p.Cout + n) * 4);
    }

    f32x4 st[NI];
    stage_load(0, st);
    stage_store(smem, st);
    __syncthreads();
    for (int cb = 0; cb < ncb; ++cb) {
        if (cb + 1 < ncb) stage_load(cb + 1, st);          // in flight during this chunk's 9 taps
#pragma unroll 1      // keep the 9 taps rolled: unrolled, hipcc preloads all 432 weight values and spills (512 VGPRs)
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const float *ap = smem + ((py + ky) * C::PWD + (px + kx)) * C::PS;
            f32x4 av[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) av[u][k4] = *reinterpret_cast<const f32x4 *>(ap + u * (16 * C::PWD * C::PS) + k4 * 4);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float *wt = wl + (((cb * 9 + tap) * 4 + k4) * COUT) * 4;               // [n][4], n = 0..COUT-1
#pragma unroll
                for (int n = 0; n < COUT; ++n) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(wt + n * 4);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[u][n] = fmaf(av[u][k4][j], wv[j], acc[u][n]);
                }
            }
        }
        if (cb + 1 < ncb) {
            __syncthreads();                                // everybody is done reading the patch
            stage_store(smem, st);
            __syncthreads();
        }
    }

    const int ox = ox0 + px;
    const int64_t plane = (int64_t)p.H * p.W;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int oy = oy0 + py + 16 * u;
        if (ox >= p.W || oy >= p.H) continue;
        const int64_t opix = (int64_t)oy * p.W + ox;
#pragma unroll
        for (int n = 0; n < COUT; ++n) {
            if (n >= y_planar_ch) break;
            float v = edge_act(acc[u][n] + (p.bias ? as_const(p.bias)[n] : 0.f), p.act);
            if (p.epilogue == BSVD_EPI_RESID && n < p.resid_ch) {
                v = p.extra[(int64_t)f * p.extra_fs + opix * p.extra_ps + (int64_t)n * p.extra_cs] - v;
            }
            if (do_clamp) v = fminf(fmaxf(v, lo), hi);
            p.y[(int64_t)f * p.y_fs + n * plane + opix] = v;
        }
    }
}

int launch_head_f32(const ConvParams &p, int cin_real, hipStream_t stream)
{
    const int64_t nblk = (int64_t)p.frames * ((p.H + 3) / 4) * ((p.W + 63) / 64);
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3(head): grid of %lld workgroups", (long long)nblk); return -1; }
    if (cin_real == 4) hipLaunchKernelGGL(head_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    else if (cin_real == 3) hipLaunchKernelGGL(head_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    else { set_error("bsvd_conv3x3: planar input supports 3 or 4 channels, got %d", cin_real); return -14; }
    return (int)hipGetLastError();
}

int launch_tail_f32(const ConvParams &p, int cout_real, int do_clamp, float lo, float hi, hipStream_t stream)
{
    const int64_t nblk = (int64_t)p.frames * ((p.H + TailCfg::TH - 1) / TailCfg::TH) * ((p.W + TailCfg::TW - 1) / TailCfg::TW);
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3(tail): grid of %lld workgroups", (long long)nblk); return -1; }
    if (cout_real < 1 || cout_real > 4) { set_error("bsvd_conv3x3: planar output supports 1..4 channels, got %d", cout_real); return -15; }
    const int ncb = p.Cin >> 4;
    if (ncb > TailCfg::MAX_CHUNKS) { set_error("bsvd_conv3x3: planar-output layer supports Cin <= %d", TailCfg::MAX_CHUNKS * 16); return -15; }
    const int cc = cout_real == 3 ? 3 : 4;
    const int lds = TailCfg::lds_bytes(ncb, cc);
    static std::atomic<int> granted[2][MAX_DEVICES];
    const void *fn = cc == 3 ? reinterpret_cast<const void *>(&tail_kernel<3>) : reinterpret_cast<const void *>(&tail_kernel<4>);
    if (lds > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds(fn, lds, granted[cc - 3]);
        if (e != hipSuccess) return (int)e;
    }
    if (cc == 3)
        hipLaunchKernelGGL(tail_kernel<3>, dim3((unsigned)nblk), dim3(256), lds, stream, p, cout_real, do_clamp, lo, hi);
    else
        hipLaunchKernelGGL(tail_kernel<4>, dim3((unsigned)nblk), dim3(256), lds, stream, p, cout_real, do_clamp, lo, hi);
    return (int)hipGetLastError();
}

}  // namespace bsvd
