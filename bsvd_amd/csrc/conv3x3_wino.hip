// conv3x3_wino.hip -- the wide stride-1 layers of the split-fp16 (BSVD_F16X3) mode as a 1-D Winograd F(M,3) convolution
// along x, M = 2 or 4, for gfx950 (MI355X).  See wino_forms.h for the forms and DESIGN.md §4.1d for the design record.
//
// Why: the direct 3-pass split kernel (conv3x3_mfma.hip) runs at the package power cap with the matrix pipe 0.75 busy; what is
// left is issuing fewer MFMAs.  Along x, F(4,3) turns the 9 tap-GEMMs of a 3x3 conv into 6 positions x 3 rows of GEMMs over a
// quarter of the pixels: 18 / 4 = 4.5 tap-GEMMs per output pixel instead of 9 (F(2,3): 6).
//
//   y[f] = epilogue( act( conv3x3( gather(x[f-1], x[f], x[f+1], fold) ) + bias ) )        (same contract as bsvd_conv3x3)
//
// Work split.  A *sub-tile* is 16 px x SR rows of output (SR = 8 for F(2,3), 16 for F(4,3)) = two 32-"group" MFMA tiles
// (a group = M consecutive output pixels of a row: GW = 16 / M groups per sub-tile row, RM = 32 / GW rows per MFMA tile).
// Workgroup = 4 waves = 2 sub-tiles (waves 0-1 / 2-3) x 2 halves of 128 output channels; a wave owns, for EVERY transformed
// position xi < A = M + 2, 2 x 2 MFMA tiles (64 groups x 64 channels): A * 64 accumulator registers (256 / 384), one wave per
// SIMD.  hipcc keeps at most 256 accumulator registers in AGPRs; for F(4,3) positions 4 and 5 are therefore accumulated by
// inline-asm MFMAs pinned to arch VGPRs (HYB).
//
// K loop: 16-channel chunks; per chunk A x 3 steps (xi, ky) of 12 MFMAs (3 split passes x 2 x 2 tiles).
//   * group operand V[xi]: the transformed activations of the chunk, double-buffered in LDS as planes
//     [xi][quarter: hi c0-7, hi c8-15, lo c0-7, lo c8-15][patch row][group] x 16 B -- the 32 lanes of a fragment read 32
//     consecutive 16-B slots (conflict-free ds_read_b128), and (xi, ky, tile, hi/lo) are immediate offsets from one lane base.
//     Produced during the previous chunk's steps by ALL lanes: an item = (patch row, group, 4 channels) loads A pixels x
//     (8 B hi + 8 B lo) with branch-free raw buffer loads (out of range = 0 = zero padding), decodes to fp32, applies BT,
//     re-splits every transformed value into an fp16 pair and writes 2 A ds_write_b64.  The temporal-shift gather is a
//     per-chunk source select of this load, exactly as in the direct kernel.
//   * weight operand U[xi][ky]: pre-transformed (double precision G) and pre-split at pack time (bsvd_pack_weights_wino),
//     straight from L2 to VGPRs two steps ahead in a 3-deep register ring, layout [chunk][xi][ky][hi|lo][h][Cout][8 fp16].
// Epilogue: AT in fp32 on the accumulators, then the direct kernel's split epilogue per output column j < M (wave-private LDS
// transposition so that 4 adjacent lanes write one pixel's 128 contiguous bytes; bias, activation, PixelShuffle + skip add).
// MEASUREMENT VARIANT: compiled into measurement builds only (-DBSVD_MEASURE, tools/build_measure.sh); the product library
// does not contain this kernel.  Kept as the record of the first design of DESIGN.md 4.1d (wino_m 12 / 14).
#ifdef BSVD_MEASURE
#include <stdio.h>
#include <type_traits>
#include "bsvd_internal.h"
#include "wino_forms.h"

#ifndef BSVD_WINO_OOB
#define BSVD_WINO_OOB 0x7fffffffu
#endif
#ifndef BSVD_WINO_ZSKIP
#define BSVD_WINO_ZSKIP 1       // leave the all-zero temporal-shift chunks of a clip's first / last frame out of K (bit-identical)
#endif

namespace bsvd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int M_>
struct WinoCfg {
    static constexpr int M = M_, A = M + 2;
    static constexpr int GW = 16 / M;            // groups per sub-tile row
    static constexpr int RM = 32 / GW;           // rows of one 32-group MFMA tile
    static constexpr int SR = 2 * RM;            // sub-tile rows
    static constexpr int PR = SR + 2;            // patch rows
    static constexpr int NSLOT = PR * GW;        // (row, group) slots of a plane
    static constexpr int PLANE = NSLOT * 16;     // bytes of one (xi, quarter) plane
    static constexpr int V_SUB = A * 4 * PLANE;  // bytes of one sub-tile's transformed chunk
    static constexpr int V_BUF = 2 * V_SUB;      // a workgroup = two sub-tiles
    static constexpr int LDS_BYTES = 2 * V_BUF;  // double-buffered
    static constexpr int NITEM_SUB = NSLOT * 4;  // transform items (slot, 4 channels) per sub-tile and chunk
    static constexpr int NITEM = 2 * NITEM_SUB;
    static constexpr int NR = 3;                 // rounds per chunk; every wave runs all of them (uniform instruction stream)
    static constexpr int LPR = ((NITEM + NR - 1) / NR + 15) / 16 * 16;   // lanes per round (16-lane ds_write groups stay whole)
    static_assert(LPR <= 256 && NITEM_SUB % 16 == 0 && (NSLOT * 2) % 16 == 0, "");
    static constexpr int NS = 3 * A;             // steps per chunk
    static constexpr int CPS = M == 4 ? 1 : 2;   // channels of an item transformed per step
    static constexpr int CSTEPS = 4 / CPS;       // steps one round's transform is spread over
    static constexpr int WIN = A - CSTEPS;       // steps between a round's loads and its first transform step
    static_assert(WIN >= 0 && LDS_BYTES <= 160 * 1024 && 4 * 32 * 36 * 4 <= LDS_BYTES, "");
    static constexpr bool HYB = A * 64 > 256;    // positions >= 4 accumulate in arch VGPRs through inline-asm MFMAs
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *ptr, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ u32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

struct WChunkSrc {         // wave-uniform source of one 16-channel chunk (temporal-shift gather: next / previous / this frame)
    __amdgpu_buffer_rsrc_t rs;
    unsigned ps4, soff;
};

}  // namespace

template <int M>
__global__ __launch_bounds__(256, 1) void wino_kernel(const ConvParams p)
{
    using C = WinoCfg<M>;
    using F = WinoForm<M>;
    constexpr int A = C::A;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpx = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lh = lane >> 5;

    // ---- block -> (frame, sub-tile pair, channel tile); XCD-aware like the direct kernel (block b runs on XCD b % 8)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    if (p.flip) lid = nblk - 1 - lid;
    const int ct = lid % p.nct; lid /= p.nct;
    const int npair = (p.nty * p.ntx + 1) >> 1;            // pairs never straddle a frame
    const int pair = lid % npair;
    const int f = lid / npair;
    const int nsub = p.nty * p.ntx;
    int sy0[2], sx0[2];
    bool slive[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int si = 2 * pair + s;
        slive[s] = si < nsub;
        const int ty = si / p.ntx, tx = si - ty * p.ntx;
        sy0[s] = ty * C::SR;
        sx0[s] = tx * 16;
    }
    const int n0 = ct * 128;

    // ---- temporal sources of this frame (wave uniform)
    const float *cur = p.x + (int64_t)f * p.x_fs;
    const float *prv, *nxt;
    int prev_ps, prev_co, next_ps, next_co;
    if (f > 0) { prv = cur - p.x_fs; prev_ps = p.Cin; prev_co = p.fold; }
    else       { prv = p.halo_prev; prev_ps = p.halo_prev_ps; prev_co = p.halo_prev_co; }
    if (f + 1 < p.frames) { nxt = cur + p.x_fs; next_ps = p.Cin; next_co = 0; }
    else                  { nxt = p.halo_next; next_ps = p.halo_next_ps; next_co = p.halo_next_co; }

    int ncb = p.Cin >> 4;
    int zs_a = 0, zs_b = 1 << 20, zs_c = 0;
    if (BSVD_WINO_ZSKIP && p.fold >= 16) {
        const int f16 = p.fold >> 4;
        zs_b = f16;
        if (nxt == nullptr) zs_a = f16;
        if (prv == nullptr) zs_c = f16;
        zs_b -= zs_a;                          // in live indices: the prev group starts after the live next-group chunks
        ncb -= zs_a + zs_c;
    }
    const unsigned hw = (unsigned)p.H * (unsigned)p.W;
    const __amdgpu_buffer_rsrc_t rs_cur = make_rsrc(cur, hw * p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prv ? prv : cur, prv ? hw * prev_ps * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_next = make_rsrc(nxt ? nxt : cur, nxt ? hw * next_ps * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_none = make_rsrc(cur, 0u);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.Cin * (unsigned)(3 * A) * (unsigned)p.Cout * 4u);
    auto chunk_src = [&](int cbl) {            // cbl: LIVE chunk index
        WChunkSrc c;
        if (cbl >= ncb) { c.rs = rs_none; c.ps4 = 0; c.soff = 0; return c; }
        const int cb = cbl + zs_a + (cbl >= zs_b ? zs_c : 0);
        const int c0 = cb * 16;
        if (c0 < p.fold)          { c.rs = rs_next; c.ps4 = next_ps * 4u; c.soff = (next_co + c0) * 4u; }
        else if (c0 < 2 * p.fold) { c.rs = rs_prev; c.ps4 = prev_ps * 4u; c.soff = (prev_co + c0 - p.fold) * 4u; }
        else                      { c.rs = rs_cur;  c.ps4 = p.Cin * 4u;   c.soff = c0 * 4u; }
        return c;
    };

    // ---- weights: rows of the MFMA's A operand = output channels (see conv3x3_mfma.hip: `chan`), two 32-channel tiles per wave
    const int rrow = (li & 3) + 4 * (li >> 3);
    const int chan = 8 * (2 * (rrow >> 3) + ((li >> 2) & 1)) + (rrow & 7);
    const int nb0 = n0 + wn * 64 + chan;
    const unsigned slab_bytes = 64u * p.Cout, g_bytes = 32u * p.Cout;
    unsigned vb[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) vb[nt] = nb0 + 32 * nt < p.Cout ? (unsigned)(lh * p.Cout + nb0 + 32 * nt) * 16u : BSVD_WINO_OOB;
    const int nsteps = ncb * C::NS;
    auto load_b = [&](int step, f32x4 (&b)[2][2]) {       // step: LIVE step index (chunk * NS + xi * 3 + ky)
        step = step < nsteps ? step : nsteps - 1;
        const int full = step + C::NS * zs_a + (step >= C::NS * zs_b ? C::NS * zs_c : 0);
        const unsigned so = (unsigned)full * slab_bytes;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            b[nt][0] = buf_load4(rs_w, vb[nt], so);
            b[nt][1] = buf_load4(rs_w, vb[nt], so + g_bytes);
        }
    };

    // ---- transform items.  Round r, lane tid: item E = r * LPR + tid = (sub-tile, quarter bit qb, slot, 8-byte half):
    //      4 channels c4 = 2 qb + half of patch slot (row, group); 16 consecutive lanes write 128 contiguous LDS bytes.
    struct Item { unsigned pix0; int gx0; bool row_ok, active; unsigned c4off; unsigned wr; };
    auto item_of = [&](int r) {
        Item t;
        const int E = r * C::LPR + tid;
        const bool active = tid < C::LPR && E < C::NITEM;
        t.active = active;
        const int sub = E >= C::NITEM_SUB ? 1 : 0;
        const int e = E - sub * C::NITEM_SUB;
        const int qb = e >= C::NSLOT * 2 ? 1 : 0;
        const int r2 = e - qb * C::NSLOT * 2;
        const int slot = r2 >> 1, half = r2 & 1;
        const int row = slot / C::GW, g = slot - row * C::GW;
        const int gy = (sub ? sy0[1] : sy0[0]) - 1 + row;
        t.gx0 = (sub ? sx0[1] : sx0[0]) - 1 + M * g;
        t.row_ok = active && (sub ? slive[1] : slive[0]) && gy >= 0 && gy < p.H;
        t.pix0 = (unsigned)(gy * p.W + t.gx0);
        t.c4off = (unsigned)(qb * 16 + half * 8);
        t.wr = (unsigned)(sub * C::V_SUB + qb * C::PLANE + slot * 16 + half * 8);
        return t;
    };
    auto item_load = [&](const WChunkSrc &c, const Item &t, u32x2 (&rh)[A], u32x2 (&rl)[A]) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const int gx = t.gx0 + i;
            const bool ok = t.row_ok && gx >= 0 && gx < p.W;
            const unsigned voff = ok ? (t.pix0 + (unsigned)i) * c.ps4 + t.c4off : BSVD_WINO_OOB;
            rh[i] = buf_load2(c.rs, voff, c.soff);
            rl[i] = buf_load2(c.rs, voff, c.soff + 32u);
        }
    };
    // transform channels [c_lo, c_hi) of an item (fp32), re-split, keep the fp16 results in vh / vl (A x 4 halves each)
    auto item_transform = [&](const u32x2 (&rh)[A], const u32x2 (&rl)[A], int cl, f16x4 (&vh)[A], f16x4 (&vl)[A]) {
#pragma unroll
        for (int cc = 0; cc < C::CPS; ++cc) {
            const int c = cl + cc;
            float d[A], v[A];
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const f16x4 h = __builtin_bit_cast(f16x4, rh[i]), l = __builtin_bit_cast(f16x4, rl[i]);
                d[i] = (float)h[c] + (float)l[c];
            }
            F::input(d, v);
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const _Float16 hh = (_Float16)v[i];
                vh[i][c] = hh;
                vl[i][c] = lo_keep((_Float16)(v[i] - (float)hh));
            }
        }
    };
    auto item_store = [&](unsigned char *vbuf, const Item &t, const f16x4 (&vh)[A], const f16x4 (&vl)[A]) {
        if (t.active) {
#pragma unroll
            for (int i = 0; i < A; ++i) {
                unsigned char *dst = vbuf + t.wr + i * 4 * C::PLANE;
                *reinterpret_cast<u32x2 *>(dst) = __builtin_bit_cast(u32x2, vh[i]);
                *reinterpret_cast<u32x2 *>(dst + 2 * C::PLANE) = __builtin_bit_cast(u32x2, vl[i]);
            }
        }
    };

    // ---- accumulators
    f32x16 acc[A][2][2];
#pragma unroll
    for (int x = 0; x < A; ++x)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][mt][nt][r] = 0.f;

    // ---- prologue: weights of steps 0, 1 in flight; chunk 0 transformed into buffer 0
    f32x4 bring[3][2][2];
    load_b(0, bring[0]);
    load_b(1, bring[1]);
    {
        const WChunkSrc c0 = chunk_src(0);
#pragma unroll
        for (int r = 0; r < C::NR; ++r) {
            const Item t = item_of(r);
            u32x2 rh[A], rl[A];
            f16x4 vh[A], vl[A];
            item_load(c0, t, rh, rl);
#pragma unroll
            for (int cl = 0; cl < 4; cl += C::CPS) item_transform(rh, rl, cl, vh, vl);
            item_store(wsm, t, vh, vl);
        }
    }
    __syncthreads();

    // group fragments: lane base inside a sub-tile's V image
    const unsigned a_lane = (unsigned)(wpx * C::V_SUB + lh * C::PLANE + li * 16);

    int step = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const unsigned char *pcur = wsm + (cb & 1) * C::V_BUF + a_lane;
        unsigned char *pnext = wsm + ((cb + 1) & 1) * C::V_BUF;
        const WChunkSrc cn = chunk_src(cb + 1);          // beyond the last chunk: zero-size descriptor, results unused
        u32x2 rh[A], rl[A];
        f16x4 vh[A], vl[A];
        Item t;
        static_for<0, C::NS>([&](auto s_) {
            constexpr int S = decltype(s_)::value;
            constexpr int XI = S / 3, KY = S % 3;
            load_b(step + 2, bring[(S + 2) % 3]);
            // transform schedule of this step
            if constexpr (S % A == 0) {                  // a round's loads
                t = item_of(S / A);
                item_load(cn, t, rh, rl);
            }
            f32x4 a[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    a[mt][pt] = *reinterpret_cast<const f32x4 *>(pcur + XI * 4 * C::PLANE + pt * 2 * C::PLANE + (C::RM * mt + KY) * C::GW * 16);
            if constexpr (S % A >= C::WIN) {             // ... its transform, CPS channels per step, and the stores at the end
                item_transform(rh, rl, (S % A - C::WIN) * C::CPS, vh, vl);
                if constexpr (S % A == A - 1) item_store(pnext, t, vh, vl);
            }
            const f32x4 (&b)[2][2] = bring[S % 3];
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        // pass 0: hi(w) x lo(v), 1: lo(w) x hi(v), 2: hi(w) x hi(v)
                        const f16x8 bv = __builtin_bit_cast(f16x8, b[nt][pass == 1 ? 1 : 0]);
                        const f16x8 av = __builtin_bit_cast(f16x8, a[mt][pass == 0 ? 1 : 0]);
                        if constexpr (C::HYB && XI >= 4)
                            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[XI][mt][nt]) : "v"(bv), "v"(av));
                        else
                            acc[XI][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc[XI][mt][nt], 0, 0, 0);
                    }
            ++step;
        });
        __syncthreads();
    }
    if constexpr (C::HYB) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // asm MFMA results -> VALU reads (the compiler does not see the hazard)

    // ---- epilogue
    const bool mylive = wpx ? slive[1] : slive[0];
    if (!mylive) return;
    const int my_sy0 = wpx ? sy0[1] : sy0[0], my_sx0 = wpx ? sx0[1] : sx0[0];
    const int q = lane & 3;
    const int Cq = p.Cout >> 2;
    f32x4 bq[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n8 = n0 + wn * 64 + nt * 32 + 8 * q;
        bq[nt][0] = bq[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n8 < p.Cout) {
            bq[nt][0] = *reinterpret_cast<const f32x4 *>(p.bias + n8);
            bq[nt][1] = *reinterpret_cast<const f32x4 *>(p.bias + n8 + 4);
        }
    }
    float *sc = reinterpret_cast<float *>(wsm) + wid * (32 * 36);
    auto coff16 = [](int c8) { return (c8 >> 4) * 16 + ((c8 >> 3) & 1) * 4; };
    auto finish = [&](auto epi_c, auto act_c) {
        constexpr int EPI = decltype(epi_c)::value, ACT = decltype(act_c)::value;
        const bool has_skip = EPI == BSVD_EPI_PS_ADD && p.extra != nullptr;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 o[M];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float m[A], oo[M];
#pragma unroll
                    for (int x = 0; x < A; ++x) m[x] = acc[x][mt][nt][r];
                    F::output(m, oo);
#pragma unroll
                    for (int j = 0; j < M; ++j) o[j][r] = oo[j];
                }
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    // transposition: lane (li, lh) holds group li's channels 8 (2h + lh) .. + 7 in registers 8h .. 8h + 7
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float *w = sc + li * 36 + 8 * (2 * h + lh);
                        *reinterpret_cast<f32x4 *>(w) = f32x4{o[j][8 * h], o[j][8 * h + 1], o[j][8 * h + 2], o[j][8 * h + 3]};
                        *reinterpret_cast<f32x4 *>(w + 4) = f32x4{o[j][8 * h + 4], o[j][8 * h + 5], o[j][8 * h + 6], o[j][8 * h + 7]};
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
                        const int m = (lane + 64 * sidx) >> 2;                 // group position 0 .. 31 of the MFMA tile
                        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(sc + m * 36 + q * 8);
                        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(sc + m * 36 + q * 8 + 4);
                        const int row = m / C::GW, g = m - row * C::GW;
                        const int oy = my_sy0 + C::RM * mt + row, ox = my_sx0 + M * g + j;
                        const int n8 = n0 + wn * 64 + nt * 32 + 8 * q;
                        const bool live = oy < p.Ho && ox < p.Wo && n8 < p.Cout;
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { v[k] = v0[k] + bq[nt][0][k]; v[4 + k] = v1[k] + bq[nt][1][k]; }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if constexpr (ACT == BSVD_ACT_RELU6) v[k] = __builtin_amdgcn_fmed3f(v[k], 0.f, 6.f);
                            else if constexpr (ACT == BSVD_ACT_RELU) v[k] = fmaxf(v[k], 0.f);
                        }
                        float *dst;
                        if constexpr (EPI == BSVD_EPI_PS_ADD) {
                            const int sub = n8 / Cq, c8 = n8 - sub * Cq;
                            const int64_t upix = (int64_t)(2 * oy + (sub >> 1)) * (2 * p.Wo) + (2 * ox + (sub & 1));
                            dst = p.y + (int64_t)f * p.y_fs + upix * Cq + coff16(c8);
                            if (has_skip && live) {
                                const float *ep = p.extra + (int64_t)f * p.extra_fs + upix * p.extra_ps + coff16(c8);
                                const f16x8 eh = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(ep));
                                const f16x8 el = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(ep + 8));
#pragma unroll
                                for (int k = 0; k < 8; ++k) v[k] += (float)eh[k] + (float)el[k];
                            }
                        } else {
                            dst = p.y + (int64_t)f * p.y_fs + ((int64_t)oy * p.Wo + ox) * p.Cout + coff16(n8);
                        }
                        if (live) {
                            constexpr bool bounded = ACT == BSVD_ACT_RELU6 && EPI == BSVD_EPI_PLAIN;
                            f16x8 hi, lo;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float vs = bounded ? v[k] : __builtin_amdgcn_fmed3f(v[k], -65504.f, 65504.f);
                                hi[k] = (_Float16)vs;
                                lo[k] = lo_keep((_Float16)__builtin_fmaf((float)hi[k], -1.0f, vs));
                            }
                            *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, hi);
                            *reinterpret_cast<f32x4 *>(dst + 8) = __builtin_bit_cast(f32x4, lo);
                        }
                    }
                }
            }
    };
    using std::integral_constant;
    if (p.epilogue == BSVD_EPI_PS_ADD) {
        if (p.act == BSVD_ACT_RELU6) finish(integral_constant<int, BSVD_EPI_PS_ADD>{}, integral_constant<int, BSVD_ACT_RELU6>{});
        else if (p.act == BSVD_ACT_RELU) finish(integral_constant<int, BSVD_EPI_PS_ADD>{}, integral_constant<int, BSVD_ACT_RELU>{});
        else finish(integral_constant<int, BSVD_EPI_PS_ADD>{}, integral_constant<int, BSVD_ACT_NONE>{});
    } else {
        if (p.act == BSVD_ACT_RELU6) finish(integral_constant<int, BSVD_EPI_PLAIN>{}, integral_constant<int, BSVD_ACT_RELU6>{});
        else if (p.act == BSVD_ACT_RELU) finish(integral_constant<int, BSVD_EPI_PLAIN>{}, integral_constant<int, BSVD_ACT_RELU>{});
        else finish(integral_constant<int, BSVD_EPI_PLAIN>{}, integral_constant<int, BSVD_ACT_NONE>{});
    }
}

// Can this layer run on the Winograd kernel?  (nullptr = yes, else the reason)
template <int M>
static int launch_wino_m(const ConvParams &pin, hipStream_t stream, char *name, int name_len)
{
    using C = WinoCfg<M>;
    if (name) {
        snprintf(name, name_len, "wino_kernel<F(%d,3)>[f16x3]", M);
        return 0;
    }
    ConvParams p = pin;
    p.ntx = (p.Wo + 15) / 16;
    p.nty = (p.Ho + C::SR - 1) / C::SR;
    p.nct = (p.Cout + 127) / 128;
    const int64_t npair = ((int64_t)p.nty * p.ntx + 1) / 2;
    const int64_t nblk = (int64_t)p.frames * npair * p.nct;
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3: grid of %lld workgroups", (long long)nblk); return -1; }
    static std::atomic<int> granted[MAX_DEVICES];
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&wino_kernel<M>), C::LDS_BYTES, granted);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((wino_kernel<M>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

int launch_wino(const ConvParams &p, hipStream_t stream, char *name, int name_len)
{
    return launch_wino_m<2>(p, stream, name, name_len);      // (F(4,3) on this kernel -- 264 B of scratch, never parity-green -- is not instantiated)
}

}  // namespace bsvd
#endif  // BSVD_MEASURE
