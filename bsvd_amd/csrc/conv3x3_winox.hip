// conv3x3_winox.hip -- the wide stride-1 layers of the split-fp16 (BSVD_F16X3) mode as a 1-D Winograd F(M,3) convolution along
// x with ONE TRANSFORMED POSITION PER WAVE, for gfx950 (MI355X).  Forms: wino_forms.h (F(2,3), F(4,3), F(6,3)); design record:
// DESIGN.md §4.1d.
//
//   y[f] = epilogue( act( conv3x3( gather(x[f-1], x[f], x[f+1], fold) ) + bias ) )        (same contract as bsvd_conv3x3)
//
// Why this shape.  Along x, F(M,3) turns the 9 tap-GEMMs of a 3x3 conv into A = M + 2 positions x 3 rows of GEMMs over 1/M of
// the pixels: 3 A / M tap-GEMMs per output pixel instead of 9 (6, 4.5, 4 for M = 2, 4, 6) -- but a wave that owns all A
// positions of its output tile needs A / M times the accumulators of the direct kernel, which on a 512-register file means one
// wave per SIMD, half the operand reuse per MFMA and an epilogue nobody overlaps (built and measured first: conv3x3_wino.hip,
// DESIGN.md).  Here a workgroup is A x NH waves; wave (xi, h) accumulates position xi ONLY, for the whole pixel tile of the
// workgroup (4 MFMA tiles = 128 "groups" of M consecutive output pixels) and NTW 32-channel tiles: 4 x NTW accumulator tiles,
// the direct kernel's wave shape, register budget and operand rates (pixel fragments from LDS, weights from L2, 4 x NTW
// reuse), two or three waves per SIMD.  The positions meet in the epilogue, through LDS.
//
// Pixel tile: 8 groups x 16 rows (8 M px wide); MFMA tile mt = rows 4 mt .. 4 mt + 3, lane li <-> (row li >> 3, group li & 7).
// K loop: 16-channel chunks; per chunk and wave 3 steps (ky) of 3 passes x 4 x NTW MFMAs.
//   * group operand V[xi]: the chunk's transformed activations, double-buffered in LDS as planes [xi][quarter: hi c0-7,
//     hi c8-15, lo c0-7, lo c8-15][patch row 0..17][group 0..7] x 16 B.  A fragment's 32 lanes read 32 consecutive slots
//     (conflict-free ds_read_b128); ky / mt / hi-lo are immediate offsets.  Written by all lanes of the workgroup: an item
//     (patch row, group, 4 channels) loads A pixels x (8 B hi + 8 B lo) with branch-free raw buffer loads (out of range = 0 =
//     the conv's zero padding), decodes to fp32, applies BT, re-splits every value into an fp16 pair, writes 2 A ds_write_b64.
//     The temporal-shift gather is a per-chunk source select of this load, as in the direct kernel.
//   * weight operand U[xi][ky]: transformed in double and split at pack time (bsvd_pack_weights_wino), L2 -> VGPRs through a
//     3-slot register ring, two steps ahead.
//   * the waves that share a SIMD (wave w, w + 4, w + 8) run the chunk's two phases -- MFMA steps / transform of the next
//     chunk -- in opposite order, so that one's VALU and memory waits sit under the other's MFMAs.
// Epilogue: two rounds (MFMA tiles 0-1, 2-3).  Every wave publishes its accumulator tiles in LDS ([block][xi][4 regs][lane],
// XOR-swizzled slots), then each wave finishes one block (or half of one: a range of output columns j < M): a lane reads the
// A positions of ONE pixel group x 8 channels -- the exchange doubles as the transposition the direct kernel's split epilogue
// does through a scratch -- applies AT in fp32, bias, activation, PixelShuffle + skip, splits and stores 2 x 16 B per pixel.
#include <stdio.h>
#include <type_traits>
#include "bsvd_internal.h"
#include "wino_forms.h"

#define BSVD_WX_OOB 0x7fffffffu

namespace bsvd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
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

template <int M_, int NH_, int NTW_>
struct XCfg {
    static constexpr int M = M_, A = M + 2, NH = NH_, NTW = NTW_;
    static constexpr int NW = A * NH, NTHREADS = NW * 64;
    static constexpr int MT = 4;                  // MFMA tiles of the pixel tile: 8 groups x 4 rows each
    static constexpr int TR = 16, TWPX = 8 * M;   // pixel tile: 16 rows x 8 M columns
    static constexpr int PR = TR + 2;             // patch rows
    static constexpr int NSLOT = PR * 8;
    static constexpr int PLANE = NSLOT * 16;      // bytes of one (xi, quarter) plane
    static constexpr int V_BUF = A * 4 * PLANE;
    static constexpr int NITEM = NSLOT * 4;       // transform items (row, group, 4 channels) per chunk: 576
    static constexpr int BN = NH * NTW * 32;      // output channels per workgroup
    // epilogue exchange: per round the tiles of 2 MFMA tiles x (NH NTW) channel tiles x A positions, 4 KB each
    static constexpr int NBP = 2 * NH * NTW;      // blocks per round
    static constexpr int XCH_BYTES = NBP * A * 4096;
    static constexpr int LDS_BYTES = 2 * V_BUF > XCH_BYTES ? 2 * V_BUF : XCH_BYTES;
    static constexpr int NPART = NW / NBP >= 2 ? 2 : 1;          // finishers per block (column ranges)
    static_assert(LDS_BYTES <= 160 * 1024 && NW * 64 <= 1024 && NW % 4 == 0 && NW >= NBP, "");
    static constexpr int NPH = NW / 4;            // waves per SIMD = phases of the chunk schedule
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

struct XChunkSrc {         // wave-uniform source of one 16-channel chunk (temporal-shift gather: next / previous / this frame)
    __amdgpu_buffer_rsrc_t rs;
    unsigned ps4, soff;
};

}  // namespace

template <int M, int NH, int NTW>
__global__ __launch_bounds__((XCfg<M, NH, NTW>::NTHREADS), 1) void winox_kernel(const ConvParams p)
{
    using C = XCfg<M, NH, NTW>;
    using F = WinoForm<M>;
    constexpr int A = C::A;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wid % A, hh = wid / A;
    const int li = lane & 31, lh = lane >> 5;

    // ---- block -> (frame, tile y, tile x, channel tile); XCD-aware like the direct kernel (block b runs on XCD b % 8)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    if (p.flip) lid = nblk - 1 - lid;
    const int ct = lid % p.nct; lid /= p.nct;
    const int tx = lid % p.ntx; lid /= p.ntx;
    const int ty = lid % p.nty;
    const int f = lid / p.nty;
    const int oy0 = ty * C::TR, ox0 = tx * C::TWPX;
    const int n0 = ct * C::BN;

    // ---- temporal sources of this frame (wave uniform)
    const float *cur = p.x + (int64_t)f * p.x_fs;
    const float *prv, *nxt;
    int prev_ps, prev_co, next_ps, next_co;
    if (f > 0) { prv = cur - p.x_fs; prev_ps = p.Cin; prev_co = p.fold; }
    else       { prv = p.halo_prev; prev_ps = p.halo_prev_ps; prev_co = p.halo_prev_co; }
    if (f + 1 < p.frames) { nxt = cur + p.x_fs; next_ps = p.Cin; next_co = 0; }
    else                  { nxt = p.halo_next; next_ps = p.halo_next_ps; next_co = p.halo_next_co; }
    // zero-chunk skip (bit-identical): the temporal-shift group of a frame whose neighbour does not exist is all zeros
    int ncb = p.Cin >> 4;
    int zs_a = 0, zs_b = 1 << 20, zs_c = 0;
    if (p.fold >= 16) {
        const int f16 = p.fold >> 4;
        zs_b = f16;
        if (nxt == nullptr) zs_a = f16;
        if (prv == nullptr) zs_c = f16;
        zs_b -= zs_a;
        ncb -= zs_a + zs_c;
    }
    const unsigned hw = (unsigned)p.H * (unsigned)p.W;
    const __amdgpu_buffer_rsrc_t rs_cur = make_rsrc(cur, hw * p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prv ? prv : cur, prv ? hw * prev_ps * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_next = make_rsrc(nxt ? nxt : cur, nxt ? hw * next_ps * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_none = make_rsrc(cur, 0u);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.Cin * (unsigned)(3 * A) * (unsigned)p.Cout * 4u);
    auto full_chunk = [&](int cbl) { return cbl + zs_a + (cbl >= zs_b ? zs_c : 0); };
    auto chunk_src = [&](int cbl) {            // cbl: LIVE chunk index
        XChunkSrc c;
        if (cbl >= ncb) { c.rs = rs_none; c.ps4 = 0; c.soff = 0; return c; }
        const int c0 = full_chunk(cbl) * 16;
        if (c0 < p.fold)          { c.rs = rs_next; c.ps4 = next_ps * 4u; c.soff = (next_co + c0) * 4u; }
        else if (c0 < 2 * p.fold) { c.rs = rs_prev; c.ps4 = prev_ps * 4u; c.soff = (prev_co + c0 - p.fold) * 4u; }
        else                      { c.rs = rs_cur;  c.ps4 = p.Cin * 4u;   c.soff = c0 * 4u; }
        return c;
    };

    // ---- weights of this wave's position: rows of the MFMA's A operand = output channels (conv3x3_mfma.hip: `chan`)
    const int rrow = (li & 3) + 4 * (li >> 3);
    const int chan = 8 * (2 * (rrow >> 3) + ((li >> 2) & 1)) + (rrow & 7);
    const int nb0 = n0 + hh * (NTW * 32) + chan;
    const unsigned slab_bytes = 64u * p.Cout, g_bytes = 32u * p.Cout;
    unsigned vb[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) vb[nt] = nb0 + 32 * nt < p.Cout ? (unsigned)(lh * p.Cout + nb0 + 32 * nt) * 16u : BSVD_WX_OOB;
    const int nsteps = ncb * 3;
    auto load_b = [&](int step, f32x4 (&b)[NTW][2]) {     // step = live chunk * 3 + ky
        step = step < nsteps ? step : nsteps - 1;
        const int cbl = step / 3, ky = step - 3 * cbl;
        const unsigned so = (unsigned)((full_chunk(cbl) * A + xi) * 3 + ky) * slab_bytes;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            b[nt][0] = buf_load4(rs_w, vb[nt], so);
            b[nt][1] = buf_load4(rs_w, vb[nt], so + g_bytes);
        }
    };

    // ---- transform items: E = row * 32 + qb * 16 + group * 2 + half -> 4 channels c4 = 2 qb + half of patch slot (row, group);
    //      16 consecutive lanes write 128 contiguous LDS bytes
    auto transform_item = [&](const XChunkSrc &c, unsigned char *vbuf, int E, bool active) {
        const int row = E >> 5, qb = (E >> 4) & 1, g = (E >> 1) & 7, half = E & 1;
        const int gy = oy0 - 1 + row, gx0 = ox0 - 1 + M * g;
        const bool row_ok = active && gy >= 0 && gy < p.H;
        const unsigned pix0 = (unsigned)(gy * p.W + gx0);
        const unsigned c4off = (unsigned)(qb * 16 + half * 8);
        u32x2 rh[A], rl[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const int gx = gx0 + i;
            const bool ok = row_ok && gx >= 0 && gx < p.W;
            const unsigned voff = ok ? (pix0 + (unsigned)i) * c.ps4 + c4off : BSVD_WX_OOB;
            rh[i] = buf_load2(c.rs, voff, c.soff);
            rl[i] = buf_load2(c.rs, voff, c.soff + 32u);
        }
        // two channels at a time (the fp16 results of a pair are one dword per position and part): fewer live registers than
        // four channels + 8-byte stores, at the price of 4-byte LDS stores (2-way bank aliasing between the qb planes: free)
        unsigned char *dst = vbuf + qb * C::PLANE + (row * 8 + g) * 16 + half * 8;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            f16x2 vh[A], vl[A];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ch = 2 * cp + cc;
                float d[A], v[A];
                __builtin_amdgcn_sched_barrier(0);        // one channel at a time: the scheduler otherwise decodes all four first (64 live floats at M = 6)
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    const f16x4 h = __builtin_bit_cast(f16x4, rh[i]), l = __builtin_bit_cast(f16x4, rl[i]);
                    d[i] = (float)h[ch] + (float)l[ch];
                }
                F::input(d, v);
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    const _Float16 hv = (_Float16)v[i];
                    vh[i][cc] = hv;
                    vl[i][cc] = lo_keep((_Float16)(v[i] - (float)hv));
                }
            }
            if (active) {
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    *reinterpret_cast<unsigned *>(dst + i * 4 * C::PLANE + cp * 4) = __builtin_bit_cast(unsigned, vh[i]);
                    *reinterpret_cast<unsigned *>(dst + i * 4 * C::PLANE + 2 * C::PLANE + cp * 4) = __builtin_bit_cast(unsigned, vl[i]);
                }
            }
        }
    };
    // the workgroup's share of one chunk's 576 items; `rot` rotates the wave that takes the 64 left-over items (rows 16, 17)
    auto transform_chunk = [&](const XChunkSrc &c, unsigned char *vbuf, int rot) {
        // the item geometry is re-derived per chunk from an opaque copy of the lane id: hoisted out of the K loop it would pin
        // ~30 registers (A offsets, masks, store addresses) that the accumulators need
        int tl = lane;
        asm volatile("" : "+v"(tl));
        if constexpr (C::NTHREADS == 768) {
            transform_item(c, vbuf, wid * 48 + tl, tl < 48);
        } else {
            static_assert(C::NTHREADS == 512 || C::NTHREADS == 256, "item map");
#pragma unroll
            for (int r = 0; r < 512 / C::NTHREADS; ++r) transform_item(c, vbuf, r * C::NTHREADS + wid * 64 + tl, true);
            if (wid == rot % C::NW) transform_item(c, vbuf, 512 + tl, true);
        }
    };

    // ---- accumulators
    f32x16 acc[C::MT][NTW];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- prologue
    f32x4 bring[3][NTW][2];
    load_b(0, bring[0]);
    load_b(1, bring[1]);
    transform_chunk(chunk_src(0), xsm, 0);
    __syncthreads();

    const unsigned a_lane = (unsigned)(xi * 4 * C::PLANE + lh * C::PLANE + li * 16);
    const int phase = (wid >> 2) % C::NPH;          // waves w, w + 4, (w + 8) share a SIMD
    for (int cb = 0; cb < ncb; ++cb) {
        const unsigned char *pcur = xsm + (cb & 1) * C::V_BUF + a_lane;
        unsigned char *pnext = xsm + ((cb + 1) & 1) * C::V_BUF;
        auto mfma_phase = [&]() {
            static_for<0, 3>([&](auto k_) {
                constexpr int KY = decltype(k_)::value;
                load_b(cb * 3 + KY + 2, bring[(KY + 2) % 3]);
                const f32x4 (&b)[NTW][2] = bring[KY];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
                    f32x4 a[2];
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
                        a[pt] = *reinterpret_cast<const f32x4 *>(pcur + pt * 2 * C::PLANE + (4 * mt + KY) * 128);
#pragma unroll
                    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) {
                            // pass 0: hi(w) x lo(v), 1: lo(w) x hi(v), 2: hi(w) x hi(v)
                            const f16x8 bv = __builtin_bit_cast(f16x8, b[nt][pass == 1 ? 1 : 0]);
                            const f16x8 av = __builtin_bit_cast(f16x8, a[pass == 0 ? 1 : 0]);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc[mt][nt], 0, 0, 0);
                        }
                }
            });
        };
        const XChunkSrc cn = chunk_src(cb + 1);          // beyond the last chunk: zero-size descriptor, zeros, never read
        // (one copy of the MFMA steps between two conditional transforms: an if / else with the phases in opposite orders made the
        //  register allocator carry the accumulators in two register sets and spill 60-260 registers)
        if (phase != 0) transform_chunk(cn, pnext, cb + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        if (phase == 0) transform_chunk(cn, pnext, cb + 1);
        __syncthreads();
    }

    // ---- epilogue: two rounds of publish -> finish
    const int Cq = p.Cout >> 2;
    auto coff16 = [](int c8) { return (c8 >> 4) * 16 + ((c8 >> 3) & 1) * 4; };
    auto finish = [&](auto epi_c, auto act_c) {
        constexpr int EPI = decltype(epi_c)::value, ACT = decltype(act_c)::value;
        const bool has_skip = EPI == BSVD_EPI_PS_ADD && p.extra != nullptr;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (rnd) __syncthreads();                    // round 0's readers are done
            // publish: block (mtl, channel tile hh * NTW + nt), position xi: [4 register quads][64 lanes] x 16 B, slot XOR-swizzled
#pragma unroll
            for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int blk = mtl * (NH * NTW) + hh * NTW + nt;
                    unsigned char *base = xsm + (blk * A + xi) * 4096;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x16 &t = acc[2 * rnd + mtl][nt];
                        const int slot = (lh * 32 + li) ^ (8 * lh + 4 * (r4 >> 1));
                        *reinterpret_cast<f32x4 *>(base + r4 * 1024 + slot * 16) = f32x4{t[4 * r4], t[4 * r4 + 1], t[4 * r4 + 2], t[4 * r4 + 3]};
                    }
                }
            __syncthreads();
            // finish: wave -> (block, column range)
            const int blk = wid % C::NBP, part = wid / C::NBP;
            if (part >= C::NPART) continue;
            const int mtl = blk / (NH * NTW), ntg = blk % (NH * NTW);
            const int q = lane & 3;
            const int n8 = n0 + ntg * 32 + 8 * q;
            f32x4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = bq0;
            if (p.bias && n8 < p.Cout) {
                bq0 = *reinterpret_cast<const f32x4 *>(p.bias + n8);
                bq1 = *reinterpret_cast<const f32x4 *>(p.bias + n8 + 4);
            }
            constexpr int JN = M / C::NPART;             // columns per finisher
            const int j0 = part * JN;
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const int m = (lane + 64 * sidx) >> 2;   // group position of the MFMA tile: row m >> 3, group m & 7
                // the 8 channels 8 q .. 8 q + 7 of group m: register quads 2 (q >> 1), + 1 of lane (m, lh = q & 1)
                float mv[A][8];
#pragma unroll
                for (int x = 0; x < A; ++x) {
                    const unsigned char *base = xsm + (blk * A + x) * 4096;
                    const int slot = ((q & 1) * 32 + m) ^ (8 * (q & 1) + 4 * (q >> 1));
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(base + (2 * (q >> 1)) * 1024 + slot * 16);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(base + (2 * (q >> 1) + 1) * 1024 + slot * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { mv[x][k] = v0[k]; mv[x][4 + k] = v1[k]; }
                }
                float ov[M][8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float mm[A], oo[M];
#pragma unroll
                    for (int x = 0; x < A; ++x) mm[x] = mv[x][k];
                    F::output(mm, oo);
#pragma unroll
                    for (int j = 0; j < M; ++j) ov[j][k] = oo[j];
                }
                const int oy = oy0 + 4 * (2 * rnd + mtl) + (m >> 3);
#pragma unroll
                for (int jj = 0; jj < JN; ++jj) {
                    // (compile-time column index per part keeps ov[] in registers)
                    float v[8];
                    int jcol = j0 + jj;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float s = 0.f;
#pragma unroll
                        for (int pp = 0; pp < C::NPART; ++pp) s = (part == pp) ? ov[pp * JN + jj][k] : s;
                        v[k] = s + (k < 4 ? bq0[k] : bq1[k - 4]);
                        if constexpr (ACT == BSVD_ACT_RELU6) v[k] = __builtin_amdgcn_fmed3f(v[k], 0.f, 6.f);
                        else if constexpr (ACT == BSVD_ACT_RELU) v[k] = fmaxf(v[k], 0.f);
                    }
                    const int ox = ox0 + M * (m & 7) + jcol;
                    const bool live = oy < p.Ho && ox < p.Wo && n8 < p.Cout;
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

template <int M, int NH, int NTW>
static int launch_winox_cfg(const ConvParams &pin, hipStream_t stream, char *name, int name_len)
{
    using C = XCfg<M, NH, NTW>;
    if (name) {
        snprintf(name, name_len, "winox_kernel<F(%d,3),%dx%d>[f16x3]", M, NH, NTW);
        return 0;
    }
    ConvParams p = pin;
    p.ntx = (p.Wo + C::TWPX - 1) / C::TWPX;
    p.nty = (p.Ho + C::TR - 1) / C::TR;
    p.nct = (p.Cout + C::BN - 1) / C::BN;
    const int64_t nblk = (int64_t)p.frames * p.nty * p.ntx * p.nct;
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3: grid of %lld workgroups", (long long)nblk); return -1; }
    static std::atomic<int> granted[MAX_DEVICES];
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&winox_kernel<M, NH, NTW>), C::LDS_BYTES, granted);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((winox_kernel<M, NH, NTW>), dim3((unsigned)nblk), dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

int launch_winox(const ConvParams &p, hipStream_t stream, char *name, int name_len)
{
    switch (p.wino_m) {
    case 2: return launch_winox_cfg<2, 2, 2>(p, stream, name, name_len);
    case 4: return launch_winox_cfg<4, 2, 1>(p, stream, name, name_len);
    default: return launch_winox_cfg<6, 1, 2>(p, stream, name, name_len);
    }
}

}  // namespace bsvd
