// conv3x3_f32_mfma.hip -- exact-fp32 implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   y[f] = epilogue( act( conv3x3( gather(x[f-1], x[f], x[f+1], fold) ) + bias ) )
//
// GEMM view per frame: M = Ho*Wo output pixels, N = Cout, K = 9*Cin, on v_mfma_f32_32x32x2_f32
// (f32 in / f32 accumulate, bitwise an fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
// Workgroup = 4 waves (256 threads), output tile = (TH x 16) pixels x BN channels.  Each wave owns a
// 64-pixel (4 rows x 16 cols) x 64-channel sub-tile = 2x2 MFMA tiles = 64 accumulator VGPRs.
// K is walked as (channel chunk of 16) x (9 taps):
//   * per channel chunk the (TH*s+2) x (16*s+2) x 16ch input patch incl. the 1-pixel halo is staged
//     ONCE in LDS and reused by all 9 taps and all BN output channels; the temporal-shift gather is
//     folded into this staging load as a per-channel-group source select (no torch.cat copy);
//   * per (chunk, tap) a 16 x BN weight slab is staged in LDS.
// Both are double buffered: the global loads for step s+1 are issued before the 32 MFMAs of step s
// and written to the other LDS buffer after them -> one barrier per step (2048 MFMA cycles).
//
// LDS images
//   patch : [pixel][16 ch + 4 pad] floats (80-B pixel stride keeps ds_read_b128 16-B aligned and spreads
//           consecutive pixels over the 64 banks)
//   weight: [k4 = 4][BN][4] floats, i.e. 4 consecutive input channels per 16-B item
// K-order trick: lane l of a 32x32x2 MFMA supplies k = l>>5.  A ds_read_b128 gives a lane 4 consecutive
// channels; MFMA j (0..3) then uses k = 8g + 4(l>>5) + j on both operands -- any K permutation is legal
// as long as A and B agree, so all LDS reads are 16-byte wide.
//
// Reference ops replaced: see include/bsvd_hip.h (bsvd_conv3x3).
#include "bsvd_internal.h"

namespace bsvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TH_, int WM_, int WN_, int STRIDE_>
struct ConvCfg {
    static constexpr int TH = TH_, TW = 16, WM = WM_, WN = WN_, STRIDE = STRIDE_;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TH == 4 * WM, "a wave covers 4 rows x 16 cols");
    static constexpr int BN = WN * 64;
    static constexpr int PH = (TH - 1) * STRIDE + 3;
    static constexpr int PW = (TW - 1) * STRIDE + 3;
    static constexpr int PS = 20;                      // floats per patch pixel
    static constexpr int NP = PH * PW;                 // patch pixels
    static constexpr int NQ = NP * 4;                  // float4 items per patch chunk
    static constexpr int Q_PER_STEP = (NQ + 8) / 9;    // spread the next patch's loads over 9 taps
    static_assert(Q_PER_STEP <= 256, "one prefetch item per thread per step");
    static constexpr int PATCH_FLOATS = NP * PS;
    static constexpr int W_FLOATS = 16 * BN;
    static constexpr int W_ITEMS = 4 * BN;             // float4 items per weight slab
    static constexpr int W_PER_THREAD = W_ITEMS / 256;
    static_assert(W_ITEMS % 256 == 0, "");
    static constexpr int LDS_BYTES = 2 * PATCH_FLOATS * 4;
};

struct SrcSel {            // per-frame sources of the temporal-shift gather (wave uniform)
    const float *cur, *prev, *next;
    int prev_ps, prev_co, next_ps, next_co;
};

template <class C>
__device__ __forceinline__ f32x4 load_patch_quad(const ConvParams &p, const SrcSel &s, int cb, int e,
                                                 int iy0, int ix0)
{
    const int pix = e >> 2, q = e & 3;
    const int py = pix / C::PW, px = pix - py * C::PW;
    const int iy = iy0 + py, ix = ix0 + px;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
        const int c = cb * 16 + q * 4;
        const int64_t pixi = (int64_t)iy * p.W + ix;
        if (p.vec_ok) {
            const float *src;
            int64_t off;
            if (c < p.fold) {
                src = s.next; off = pixi * s.next_ps + s.next_co + c;
            } else if (c < 2 * p.fold) {
                src = s.prev; off = pixi * s.prev_ps + s.prev_co + (c - p.fold);
            } else {
                src = s.cur; off = pixi * p.Cin + c;
            }
            if (src) v = *reinterpret_cast<const f32x4 *>(src + off);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cj = c + j;
                const float *src;
                int64_t off;
                if (cj < p.fold) {
                    src = s.next; off = pixi * s.next_ps + s.next_co + cj;
                } else if (cj < 2 * p.fold) {
                    src = s.prev; off = pixi * s.prev_ps + s.prev_co + (cj - p.fold);
                } else {
                    src = s.cur; off = pixi * p.Cin + cj;
                }
                v[j] = src ? src[off] : 0.f;
            }
        }
    }
    return v;
}

template <class C>
__device__ __forceinline__ void store_patch_quad(float *patch, int e, f32x4 v)
{
    const int pix = e >> 2, q = e & 3;
    *reinterpret_cast<f32x4 *>(patch + pix * C::PS + q * 4) = v;
}

template <class C>
__device__ __forceinline__ f32x4 load_w_item(const float *slab, int e, int n0, int Cout)
{
    const int k4 = e / C::BN, nn = e % C::BN;
    const int n = n0 + nn;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < Cout) v = *reinterpret_cast<const f32x4 *>(slab + ((int64_t)k4 * Cout + n) * 4);
    return v;
}

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act >= BSVD_ACT_RELU) v = fmaxf(v, 0.f);
    if (act == BSVD_ACT_RELU6) v = fminf(v, 6.f);
    return v;
}

template <class C>
__global__ __launch_bounds__(256, 3) void conv3x3_f32_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const patch_buf = smem;                              // 2 x PATCH_FLOATS

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / C::WN, wn = wid % C::WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- block -> (frame, tile_y, tile_x, cout tile); XCD-aware: block b runs on XCD b%8, give each XCD
    //      a contiguous range of logical tiles so that halo/weight re-reads hit that XCD's L2.
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ct = lid % p.nct; lid /= p.nct;
    const int tx = lid % p.ntx; lid /= p.ntx;
    const int ty = lid % p.nty;
    const int f = lid / p.nty;

    const int oy0 = ty * C::TH, ox0 = tx * C::TW;     // output tile origin
    const int iy0 = oy0 * C::STRIDE - 1, ix0 = ox0 * C::STRIDE - 1;   // patch origin in the input
    const int n0 = ct * C::BN;

    SrcSel s;
    s.cur = p.x + (int64_t)f * p.x_fs;
    if (f > 0) { s.prev = s.cur - p.x_fs; s.prev_ps = p.Cin; s.prev_co = p.fold; }
    else       { s.prev = p.halo_prev; s.prev_ps = p.halo_prev_ps; s.prev_co = p.halo_prev_co; }
    if (f + 1 < p.frames) { s.next = s.cur + p.x_fs; s.next_ps = p.Cin; s.next_co = 0; }
    else                  { s.next = p.halo_next; s.next_ps = p.halo_next_ps; s.next_co = p.halo_next_co; }

    const int ncb = p.Cin >> 4;
    const int64_t slab_stride = (int64_t)16 * p.Cout;           // floats per (chunk, tap) weight slab

    // ---- per-lane offsets.  A fragments come from the LDS patch; B fragments (weights) are read straight
    //      from global/L2 into registers in the packed [k4][Cout][4] layout: lanes = consecutive output
    //      channels -> one coalesced 512-B run per half-wave, no LDS round trip and no per-tap barrier.
    const int a_lane = (((4 * wm + (li >> 4)) * C::STRIDE) * C::PW + (li & 15) * C::STRIDE) * C::PS + lh * 4;
    const int nb0 = n0 + wn * 64 + li;                           // this lane's output channel for nt = 0 (+32 for nt = 1)
    const bool nok0 = nb0 < p.Cout, nok1 = nb0 + 32 < p.Cout;
    const float *wl = p.w + ((int64_t)lh * p.Cout + nb0) * 4;
    const int64_t g_off = (int64_t)8 * p.Cout;                  // floats between k4 = 2g+lh and 2(g+1)+lh

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    auto load_b = [&](int step, f32x4 (&b)[2][2]) {
        const float *sl = wl + (int64_t)step * slab_stride;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            b[0][g] = nok0 ? *reinterpret_cast<const f32x4 *>(sl + g * g_off) : z;
            b[1][g] = nok1 ? *reinterpret_cast<const f32x4 *>(sl + g * g_off + 128) : z;
        }
    };
    auto load_a = [&](const float *pc, int tap, f32x4 (&a)[2][2]) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float *ap = pc + a_lane + (ky * C::PW + kx) * C::PS;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                a[mt][g] = *reinterpret_cast<const f32x4 *>(ap + (2 * mt * C::STRIDE * C::PW) * C::PS + g * 8);
    };

    // ---- prologue: chunk 0 patch -> LDS, weights of step 0 -> registers
    f32x4 bcur[2][2], bnxt[2][2];
    load_b(0, bcur);
    for (int e = tid; e < C::NQ; e += 256) store_patch_quad<C>(patch_buf, e, load_patch_quad<C>(p, s, 0, e, iy0, ix0));
    __syncthreads();

    for (int cb = 0; cb < ncb; ++cb) {
        const float *pcur = patch_buf + (cb & 1) * C::PATCH_FLOATS;
        float *pnext = patch_buf + ((cb + 1) & 1) * C::PATCH_FLOATS;
        const bool more_chunks = cb + 1 < ncb;
        f32x4 acur[2][2], anxt[2][2];
        load_a(pcur, 0, acur);
#ifdef BSVD_ABLATE
        load_a(pcur, 0, anxt);
        load_b(0, bnxt);
#endif
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // (1) global loads for the next step: weights -> registers, 1/9 of the next chunk's patch
#ifdef BSVD_ABLATE   // timing-only ablation build (results invalid): p.ablate bit0 = no weight loads, bit1 = no patch prefetch, bit2 = no A reads
            if ((tap < 8 || more_chunks) && !(p.ablate & 1)) load_b(cb * 9 + tap + 1, bnxt);
            const int ep = tap * C::Q_PER_STEP + tid;
            const bool do_p = more_chunks && tid < C::Q_PER_STEP && ep < C::NQ && !(p.ablate & 2);
#else
            if (tap < 8 || more_chunks) load_b(cb * 9 + tap + 1, bnxt);
            const int ep = tap * C::Q_PER_STEP + tid;
            const bool do_p = more_chunks && tid < C::Q_PER_STEP && ep < C::NQ;
#endif
            f32x4 preg = {0.f, 0.f, 0.f, 0.f};
            if (do_p) preg = load_patch_quad<C>(p, s, cb + 1, ep, iy0, ix0);
            // (2) next tap's A fragments (same patch) while this tap's 32 MFMAs run
#ifdef BSVD_ABLATE
            if (tap < 8 && !(p.ablate & 4)) load_a(pcur, tap + 1, anxt);
#else
            if (tap < 8) load_a(pcur, tap + 1, anxt);
#endif
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[mt][g][j], bcur[nt][g][j],
                                                                               acc[mt][nt], 0, 0, 0);
            // (3) land the prefetched patch slice in the other buffer; rotate registers
            if (do_p) store_patch_quad<C>(pnext, ep, preg);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acur[u][g] = anxt[u][g];
                    bcur[u][g] = bnxt[u][g];
                }
        }
        __syncthreads();   // one barrier per 16-channel chunk (9 taps, 288 MFMAs per wave)
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col (n) = lane&31, row (m) = (r&3) + 8*(r>>2) + 4*(lane>>5);
    //      m -> pixel (row m>>4, col m&15) of the 2x16 pixel block of MFMA tile mt.
    const int Cq = p.Cout >> 2;   // PS_ADD: channels of the shuffled output
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + wn * 64 + nt * 32 + li;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oy = oy0 + 4 * wm + 2 * mt + (m >> 4);
                const int ox = ox0 + (m & 15);
                if (oy >= p.Ho || ox >= p.Wo) continue;
                float v = apply_act(acc[mt][nt][r] + bias, p.act);
                if (p.epilogue == BSVD_EPI_PLAIN) {
                    p.y[(int64_t)f * p.y_fs + ((int64_t)oy * p.Wo + ox) * p.Cout + n] = v;
                } else if (p.epilogue == BSVD_EPI_PS_ADD) {
                    const int sub = n / Cq, ch = n - sub * Cq;
                    const int64_t opix = (int64_t)(2 * oy + (sub >> 1)) * (2 * p.Wo) + (2 * ox + (sub & 1));
                    if (p.extra) v += p.extra[(int64_t)f * p.extra_fs + opix * p.extra_ps + (int64_t)ch * p.extra_cs];
                    p.y[(int64_t)f * p.y_fs + opix * Cq + ch] = v;
                } else {  // BSVD_EPI_RESID
                    const int64_t opix = (int64_t)oy * p.Wo + ox;
                    if (n < p.resid_ch)
                        v = p.extra[(int64_t)f * p.extra_fs + opix * p.extra_ps + (int64_t)n * p.extra_cs] - v;
                    p.y[(int64_t)f * p.y_fs + opix * p.Cout + n] = v;
                }
            }
        }
    }
}

template <class C>
static int launch_cfg(const ConvParams &pin, hipStream_t stream)
{
    ConvParams p = pin;
    p.ntx = (p.Wo + C::TW - 1) / C::TW;
    p.nty = (p.Ho + C::TH - 1) / C::TH;
    p.nct = (p.Cout + C::BN - 1) / C::BN;
    const int64_t nblk = (int64_t)p.frames * p.nty * p.ntx * p.nct;
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3: grid of %lld workgroups", (long long)nblk); return -1; }
    static bool attr_done = false;   // benign race: the call is idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_f32_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3x3_f32_kernel<C>, dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

int launch_conv3x3_f32(const ConvParams &p, int stride, hipStream_t stream)
{
    // Cout <= 64 (the 540x960-level layers of bsvd_c64): 256 px x 64 ch tiles; wider layers: 128 px x 128 ch.
    // Stride 2 always takes the 128 x 128 tile (its 17x33 input patch is what bounds LDS).
    if (stride == 1)
        return p.Cout > 64 ? launch_cfg<ConvCfg<8, 2, 2, 1>>(p, stream) : launch_cfg<ConvCfg<16, 4, 1, 1>>(p, stream);
    return launch_cfg<ConvCfg<8, 2, 2, 2>>(p, stream);
}

}  // namespace bsvd
