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
// (MT = 2: 8 rows -- the same per-output instruction sequence on half the tile, taken by grids too small to fill the chip.)
// K loop: 16-channel chunks; per chunk and wave 3 steps (ky) of 3 passes x 4 x NTW MFMAs.
//   * group operand V[xi]: the chunk's transformed activations, double-buffered in LDS as planes [xi][quarter: hi c0-7,
//     hi c8-15, lo c0-7, lo c8-15][patch row 0..17][group 0..7] x 16 B.  A fragment's 32 lanes read 32 consecutive slots
//     (conflict-free ds_read_b128); ky / mt / hi-lo are immediate offsets.  Written by all lanes of the workgroup: an item
//     (patch row, group, 4 channels) loads A pixels x (8 B hi + 8 B lo) with branch-free raw buffer loads (out of range = 0 =
//     the conv's zero padding), decodes to fp32, applies BT, re-splits every value into an fp16 pair, writes 2 A ds_write_b64.
//     The temporal-shift gather is a per-chunk source select of this load, as in the direct kernel.
//   * weight operand U[xi][ky]: transformed in double and split at pack time (bsvd_pack_weights_wino), L2 -> VGPRs, the three
//     slabs of a chunk requested at the end of the previous chunk's MFMA steps.
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
#ifndef BSVD_WX_PERSIST_MIN
#define BSVD_WX_PERSIST_MIN (1 << 20)   // tiles per CU from which F(2,3) launches take the persistent form.  Measured SLOWER: never, by default
#endif
#define BSVD_CUS 256       // MI355X: 256 CUs, one 8-wave workgroup of this kernel each (the tile choice of small grids, launch_winox)
#ifndef BSVD_WX_ABL
#define BSVD_WX_ABL 0      // TIMING-ONLY ablations (results wrong): 1 no transform in the K loop, 2 no MFMA steps, 4 no epilogue finish, 8 no chunk barrier, 16 transform without its global loads (constant operands: also removes operand toggling), 32 transform without its LDS stores, 64 the K loop re-transforms the prologue's raw registers (no activation loads in the loop, realistic operand values), 128 the transformed-domain epilogue without its plane stores, 256 without its edge-record stores, 512 no patch pass behind it
#endif

namespace bsvd {

#ifdef BSVD_WX_TL
// Measurement build (tools/debug/wx_timeline.py): per workgroup and wave, shader cycles (s_memtime) summed over the K loop per
// section -- 0 first transform slot, 1 MFMA steps, 2 second transform slot, 3 chunk barrier -- plus 4 prologue, 5 epilogue, 6 total.
#define BSVD_WX_TL_SLOTS 4096
__device__ unsigned long long g_wx_tl[BSVD_WX_TL_SLOTS][12][8];
#define WXT_NOW() ({ unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_; })
#else
#define WXT_NOW() 0ull
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
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

// The tile grid of a launch, shared by the launcher and the kernels.  fold: the grid's last row band (<= 8 live rows of a 16-row tile grid) is walked by
// workgroups that each take TWO horizontally adjacent tiles of it as the two halves of one 16-row tile (XCfg::FOLD): tiles per frame =
// nreg regular ones (nct * ntx * (nty - 1), channel tile fastest) + nct * ceil(ntx / 2) folded ones.
struct WxGrid { int per_frame, nreg; bool fold; };
__host__ __device__ __forceinline__ WxGrid wx_grid(int nty, int ntx, int nct, int Ho, bool may_fold)
{
    WxGrid g;
    g.fold = may_fold && nty >= 2 && Ho - (nty - 1) * 16 <= 8;
    g.nreg = g.fold ? nct * ntx * (nty - 1) : nct * ntx * nty;
    g.per_frame = g.fold ? g.nreg + nct * ((ntx + 1) >> 1) : g.nreg;
    return g;
}

template <int M_, int NH_, int NTW_, int MT_ = 4, bool PERSIST_ = false, bool FOLD_ = false>
struct XCfg {
    static constexpr bool PERSIST = PERSIST_;     // one workgroup per CU walks a list of tiles; the transform pipeline runs across tile boundaries
    static constexpr int M = M_, A = M + 2, NH = NH_, NTW = NTW_;
    static constexpr int NW = A * NH, NTHREADS = NW * 64;
    static constexpr int MT = MT_;                // MFMA tiles of the pixel tile: 8 groups x 4 rows each.  MT = 2: the half-height tile small grids
                                                  // take (launch_winox) -- same arithmetic per output, twice the workgroups
    static constexpr int TR = 4 * MT, TWPX = 8 * M;   // pixel tile: 16 (8) rows x 8 M columns
    // FOLD: the 16 tile rows are two 8-row halves side by side in the image -- rows 0-7 at (oy0, ox0), rows 8-15 at (oy0, ox0 + TWPX): each half
    // has its own 10 patch rows (patch rows 0-9 / 10-19), MFMA tiles 2, 3 read theirs two rows further down, the epilogue's second round
    // stores 8 rows up and TWPX columns to the right.  Everything else -- the item sweeps, the MFMA steps, the exchange -- is the 16-row tile's.
    static constexpr bool FOLD = FOLD_;
    static_assert(!FOLD || (MT_ == 4 && !PERSIST_), "folded tile: the full-height tile, one tile per workgroup");
    static constexpr int PR = TR + 2 + (FOLD ? 2 : 0);     // patch rows
    static constexpr int PRH = TR / 2 + 2;                 // (FOLD) patch rows of a half
    static constexpr int frag_row(int mt, int ky) { return 4 * mt + ky + (FOLD && mt >= 2 ? 2 : 0); }      // patch row of MFMA tile mt's first row, kernel row ky
    static constexpr int NSLOT = PR * 8;
    static constexpr int PLANE = NSLOT * 16;      // bytes of one (xi, quarter) plane.  (A multiple of 256 B, so the two quarters an item wave stores at once meet in
                                                  // the same banks: PMC 0.19 of the LDS-active cycles are conflict cycles; padding the planes apart removes them and
                                                  // changes nothing -- LDS conflicts are not what this kernel waits for.  DESIGN 4.1d, removed knob BSVD_WX_PLANE_PAD)
    static constexpr int V_BUF = A * 4 * PLANE;
    static constexpr int NITEM = NSLOT * 4;       // transform items (row, group, 4 channels) per chunk: 576
    static constexpr int BN = NH * NTW * 32;      // output channels per workgroup
    // epilogue exchange: per round the tiles of MTL MFMA tiles x (NH NTW) channel tiles x A positions, 4 KB each.  It aliases the V
    // buffers -- except in the persistent form, where the next tile's first chunk is already transformed when the epilogue runs:
    // there it sits behind them, and holds one MFMA tile per round so that everything fits 160 KB
    static constexpr int MTL = PERSIST ? 1 : 2;
    static constexpr int NRND = MT / MTL;
    static constexpr int NBP = MTL * NH * NTW;    // blocks per round
    static constexpr int XCH_ROUND = NBP * A * 4096;
    static constexpr int XCH_BYTES = XCH_ROUND;
    static constexpr int XCH_OFF = PERSIST ? 2 * V_BUF : 0;
    static constexpr int LDS_BYTES = PERSIST ? 2 * V_BUF + XCH_BYTES : (2 * V_BUF > XCH_BYTES ? 2 * V_BUF : XCH_BYTES);
    static constexpr int NPART = NW / NBP >= 2 ? 2 : 1;          // finishers per block (column ranges)
    static_assert(LDS_BYTES <= 160 * 1024 && NW * 64 <= 1024 && NW % 4 == 0 && NW >= NBP, "");
    static_assert(MT == 4 || (MT == 2 && NTHREADS != 768), "item map of the half-height tile: 512- and 256-thread workgroups");
    static constexpr int WGS = LDS_BYTES <= 80 * 1024 && NW == 4 && !PERSIST ? 2 : 1;      // workgroups per CU
    static constexpr int NPH = NW / 4 > 1 ? NW / 4 : 2;   // waves per SIMD = phases of the chunk schedule (4-wave workgroups: waves 0-1 / 2-3)
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

// One-instruction forms of the two conversions the transform is made of (v_fma_mix*: fp16 sources read in place, one rounding).
// Written as plain fma()s against OPAQUE +-1.0 scalars so that (a) the compiler cannot fold  x * 1 + y  back into convert + add and
// selects v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16 itself (this file is built with -fno-slp-vectorize: with the SLP
// vectorizer on it prefers v_cvt + v_pk_fma_f32), and (b) the instructions stay visible to the scheduler (sched_group_barrier does
// not see inline asm):
//   decode   d = (float)hi + (float)lo                      -> v_fma_mix_f32   d, hi.h[SEL], 1.0, lo.h[SEL]
//   residual lo' = fp16(v - (float)hi') for a pair (v0, v1) -> v_fma_mixlo_f16 / v_fma_mixhi_f16  (v - hi' is exact in fp32, so this is
//            bit for bit the (_Float16)(v - (float)hi') of the plain form: 3 instructions per value there, 1 here)
#ifndef BSVD_WX_MIXASM
#define BSVD_WX_MIXASM 1
#endif
struct MixConst { float one, mone; };
template <int SEL>
__device__ __forceinline__ float dec_pair(unsigned h, unsigned l, const MixConst &k)
{
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f16x2_t hv = __builtin_bit_cast(f16x2_t, h), lv = __builtin_bit_cast(f16x2_t, l);
    return __builtin_fmaf((float)hv[SEL], k.one, (float)lv[SEL]);
}
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned &hp, unsigned &lp, const MixConst &k)
{
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    static_assert(BSVD_TUNE_LO_BITS >= 10, "the mixed-precision split keeps all bits of the lo halves");
    const f16x2_t hv = {(_Float16)v0, (_Float16)v1};                                  // v_cvt_pk_f16_f32 (round to nearest even)
    hp = __builtin_bit_cast(unsigned, hv);
#if BSVD_WX_MIXASM
    // both residuals straight from the PACKED hi pair (op_sel picks its halves): 3 instructions per channel pair.  Left to itself hipcc converts
    // each channel a second time (v_cvt_f16_f32) to have the fma_mix source in a low half: 5.  Same arithmetic, same bits.
    // ONE asm block: the two partial-register writes (dst_sel) are invisible to the compiler's hazard recognizer, so the block itself ends with
    // the wait state a VALU reader of `lp` would need on gfx950 -- today the only consumer is ds_write (hardware-interlocked), and whatever
    // a later edit or another scheduler puts behind the block is safe too.  BSVD_WX_MIXASM on == off bit for bit:
    // tests/measure_driver.py (`mixasm`).
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#if BSVD_WX_MIXASM != 2      // (2: without the trailing wait state -- A/B of its cost only)
        "\n\ts_nop 0"
#endif
        : "=&v"(lp) : "v"(hp), "s"(k.mone), "v"(v0), "v"(v1));
#else
    const f16x2_t lv = {(_Float16)__builtin_fmaf((float)hv[0], k.mone, v0), (_Float16)__builtin_fmaf((float)hv[1], k.mone, v1)};
    lp = __builtin_bit_cast(unsigned, lv);
#endif
}

struct XChunkSrc {         // wave-uniform source of one 16-channel chunk (temporal-shift gather: next / previous / this frame) + its tile's origin
    __amdgpu_buffer_rsrc_t rs;
    unsigned ps4, soff;
    int oy0, ox0;
    bool x_inside;
};
// The three temporal sources of a frame as plain scalars.  (chunk_src as a lambda over ready-made descriptors made hipcc keep a
// table of them -- and, one nesting level later, the kernel parameters -- in private memory: a free function over a POD passed by
// value, scalar selects, the 128-bit descriptor built at the point of use.)
struct XSources {
    const float *cur, *prv, *nxt;
    unsigned cur_bytes, prev_bytes, next_bytes, cur_ps4, prev_ps4, next_ps4;
    int prev_co, next_co, fold, ncb, zs_a, zs_b, zs_c;      // zs_*: zero-chunk skip (live -> full chunk index)
};
struct XTile {             // one tile of a workgroup's list, wave uniform (a dead tile -- beyond the list -- has zero-size sources and ncb = 0)
    XSources S;
    int oy0, ox0, n0, f;
    bool x_inside;         // every column of the patch is inside the image
};
__device__ __forceinline__ int x_full_chunk(const XSources &s, int cbl) { return cbl + s.zs_a + (cbl >= s.zs_b ? s.zs_c : 0); }
__device__ __forceinline__ XChunkSrc x_chunk_src(const XTile t, int cbl)      // cbl: LIVE chunk index; beyond the last: zero-size descriptor
{
    const XSources &s = t.S;
    XChunkSrc c;
    c.oy0 = t.oy0; c.ox0 = t.ox0; c.x_inside = t.x_inside;
    const int c0 = x_full_chunk(s, cbl) * 16;
    const bool isn = c0 < s.fold, isp = !isn && c0 < 2 * s.fold;
    const float *base = isn ? s.nxt : isp ? s.prv : s.cur;
    unsigned bytes = isn ? s.next_bytes : isp ? s.prev_bytes : s.cur_bytes;
    if (cbl >= s.ncb) bytes = 0u;
    c.rs = make_rsrc(base, bytes);
    c.ps4 = isn ? s.next_ps4 : isp ? s.prev_ps4 : s.cur_ps4;
    c.soff = (unsigned)(isn ? s.next_co + c0 : isp ? s.prev_co + c0 - s.fold : c0) * 4u;
    return c;
}

}  // namespace

// XF (BsvdConvArgs.x_f32): the input tensor (and its halos) holds plain fp32 channels instead of fp16 pairs -- what a producer writes for a
// tensor only Winograd layers read (BsvdConvArgs.y_f32).  The transform then starts from the value itself: no decode (4 of an item's
// 8 v_fma_mix_f32 per channel pair), one 8-byte load per position instead of two 4-byte ones, and BT on channel PAIRS (v_pk_add_f32).
// The whole workgroup body as a device function, so that one launch can run two tile heights: GTR = rows of the tile GRID (p.nty was
// computed for it); the body computes C::TR <= GTR of them from the tile's first row (winox_kernel_tail: the last row band of a
// 135- or 120-row layer has <= 8 live rows and runs the 8-row body on the 16-row grid).
// FG: 0 the plain grid; 1 / 2 the launch's grid may be a folded one (wx_grid): 1 = this workgroup runs a regular tile of it, 2 = a folded tile.
// XF == 2 (BsvdConvArgs.x_v): the input tensor (and its halos) is already in the TRANSFORMED domain -- its producer's epilogue applied BT and split
// the values: per frame [row][16-channel chunk][position xi][quarter: hi c0-7, hi c8-15, lo c0-7, lo c8-15][group of M pixels] x 16 B,
// i.e. exactly the LDS planes of a chunk with the groups of a whole image row side by side.  The K loop's "transform" is then a copy: 16-byte
// units global -> registers -> ds_write_b128, the LDS image linear in the unit index; no VALU work beside the MFMA waves at all.
// YV (BsvdConvArgs.y_v): the epilogue writes the output in the transformed domain of the SAME form (its pixel groups are the reader's groups): the
// finisher of a group holds its M output pixels x 8 channels in fp32, fetches the two neighbouring pixels from the lanes of the neighbouring
// groups (ds_bpermute), applies BT, splits and stores A x (16 B hi + 16 B lo).  The two positions of a tile's first / last group that need a
// pixel of the NEIGHBOURING TILE are written as partial sums and recorded -- with the tile's own edge columns -- in the frame's edge block; the
// patch pass behind the launch (v_patch_kernel) completes them.
template <int M, int NH, int NTW, int MT, bool PERSIST, int XF, int GTR, int FG = 0, bool YV = false>
__device__ __forceinline__ void winox_tile(const ConvParams &p)
{
    constexpr bool XV = XF == 2;
    static_assert(!YV || (FG == 0 && !PERSIST), "transformed-domain output: the plain tile grid");
    static_assert(!XV || (FG == 0 && !PERSIST), "transformed-domain input: the plain tile grid");
    using C = XCfg<M, NH, NTW, MT, PERSIST, FG == 2>;
    using F = WinoForm<M>;
    constexpr int A = C::A;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];

    [[maybe_unused]] unsigned long long tl_acc[7] = {0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] const unsigned long long tl_start = WXT_NOW();
    fp16_saturate_on();        // MODE.FP16_OVFL: see bsvd_internal.h (the transformed values reach 2x .. 4.7x the activation range)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wid % A, hh = wid / A;
    const int li = lane & 31, lh = lane >> 5;

    // ---- workgroup -> tiles (frame, tile y, tile x, channel tile), XCD-aware like the direct kernel: block b runs on XCD b % 8 and
    //      every XCD owns one contiguous range of the tile list (neighbouring tiles share halo rows and weights in that XCD's L2).
    //      One tile per workgroup -- or, PERSIST, the workgroups of an XCD deal its range among themselves round robin.
    const WxGrid G = wx_grid(p.nty, p.ntx, p.nct, p.Ho, FG != 0);
    const int ntiles = p.frames * G.per_frame;
    const int bid = blockIdx.x, xcd = bid & 7;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int xcd_wgs = ((int)gridDim.x - xcd + 7) >> 3;      // workgroups of this launch on this XCD (== xcd_count when not persistent)
    // (XV: a "pixel" of the byte arithmetic below is one (row, group, position) slot: v_wg groups x A positions per row)
    const unsigned hw = XV ? (unsigned)p.H * (unsigned)(p.v_wg >> 3) * ((unsigned)v_block_floats(M) / 16u) : (unsigned)p.H * (unsigned)p.W;
    auto decode_tile = [&](int j) __attribute__((always_inline)) {      // j: index into this XCD's range; beyond it: a dead tile
        XTile t;
        const bool live = j < xcd_count;
        int lid = xcd_first + (live ? j : 0);
        if (p.flip) lid = ntiles - 1 - lid;
        const int f = lid / G.per_frame;
        int r = lid - f * G.per_frame, ct, tx, ty;
        // (channel tile FASTEST: the channel tiles of a pixel tile are concurrent neighbours on one XCD and its input is read once.  Channel tile slowest within a
        //  frame -- an XCD on ONE channel tile's U at a time; the 6.3 MB of U of the 256 -> 512 layer do not fit a 4 MB L2 -- measures 2-3 % slower for F(2,3),
        //  6-8 % for F(6,3): profiles/r06h_tile_order_ct_outer_*.txt)
        if (FG != 0 && G.fold && r >= G.nreg) { r -= G.nreg; ct = r % p.nct; tx = 2 * (r / p.nct); ty = p.nty - 1; }
        else { ct = r % p.nct; r /= p.nct; tx = r % p.ntx; ty = r / p.ntx; }
        t.f = f; t.oy0 = ty * GTR; t.ox0 = tx * C::TWPX; t.n0 = ct * C::BN;
        t.x_inside = t.ox0 >= 1 && t.ox0 + (C::FOLD ? 2 : 1) * C::TWPX + 1 <= p.W;
        // temporal sources of this frame
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
            // the persistent pipeline alternates two V buffers and two raw register sets by chunk parity across tile boundaries:
            // every tile walks an EVEN number of chunks (an odd count keeps its zero chunk; same bits)
            if (PERSIST && (ncb & 1)) { zs_a = 0; zs_c = 0; zs_b = 1 << 20; ncb = p.Cin >> 4; }
        }
        XSources &S = t.S;
        S.cur = cur; S.prv = prv ? prv : cur; S.nxt = nxt ? nxt : cur;
        S.cur_bytes = live ? hw * p.Cin * 4u : 0u; S.prev_bytes = prv && live ? hw * prev_ps * 4u : 0u; S.next_bytes = nxt && live ? hw * next_ps * 4u : 0u;
        S.cur_ps4 = p.Cin * 4u; S.prev_ps4 = prev_ps * 4u; S.next_ps4 = next_ps * 4u;
        S.prev_co = prev_co; S.next_co = next_co; S.fold = p.fold; S.ncb = live ? ncb : 0; S.zs_a = zs_a; S.zs_b = zs_b; S.zs_c = zs_c;
        return t;
    };
    int jt = bid >> 3;
    XTile T = decode_tile(jt);
    // The workgroup's NEXT tile (dead when there is none) is decoded where it is needed -- the requests of the last PP + 1 chunks of a
    // tile and its last weight slabs: ~3 times per tile, a few dozen scalar instructions each.
    // (PERSIST, as built: hipcc hoists the kernel parameters and the epilogues' lane constants out of the tile loop, runs out of scalar
    //  registers -- 140 v_writelane / 2200 v_readlane -- and spills ~20 vector registers; a scratch reload in the epilogue waits, through
    //  the in-order vmcnt, for the previous round's stores.  A tile table in LDS read back with v_readfirstlane removed the scalar spills
    //  and cost more than it saved (the reads' lgkmcnt(0) waits sit inside the MFMA steps).  Measured 6-13 % slower than one tile per
    //  workgroup; kept for the record, never selected: BSVD_WX_PERSIST_MIN.)
    auto next_tile = [&]() __attribute__((always_inline)) {
        // (an L2 prefetch for the CU's next workgroup from here -- the requests behind a tile's last chunk sent to the tile 32 / 64 places
        //  further down the XCD's list -- is 3.7 % slower: DESIGN 4.1d, removed knob BSVD_WX_PFNEXT)
        if constexpr (PERSIST) {
            return decode_tile(jt + xcd_wgs);
        } else {              // no next tile: T with zero-size sources (no decoding -- this sits in the K loop's last iterations)
            XTile t = T;
            t.S.cur_bytes = t.S.prev_bytes = t.S.next_bytes = 0u;
            t.S.ncb = 0;
            return t;
        }
    };
    const unsigned w_bytes = (unsigned)p.Cin * (unsigned)(3 * A) * (unsigned)p.Cout * 4u;

    // ---- weights of this wave's position: rows of the MFMA's A operand = output channels (conv3x3_mfma.hip: `chan`)
    const int rrow = (li & 3) + 4 * (li >> 3);
    const int chan = 8 * (2 * (rrow >> 3) + ((li >> 2) & 1)) + (rrow & 7);
    const unsigned slab_bytes = 64u * p.Cout, g_bytes = 32u * p.Cout;
    unsigned vb[NTW];          // lane part of the address (tile independent); the channel tile's n0 rides in the scalar offset
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) vb[nt] = (unsigned)(lh * p.Cout + hh * (NTW * 32) + chan + 32 * nt) * 16u;
    // step = live chunk * 3 + ky of tile T; beyond T's last step: the first steps of Tn (a dead Tn: any valid slab)
    auto load_b = [&](int step, f32x4 (&b)[NTW][2]) __attribute__((always_inline)) {
        const int ns0 = T.S.ncb * 3;
        int n0, fcb, ky;        // channel tile, full chunk index and kernel row of the slab
        if (step >= ns0) {
            const XTile tn = next_tile();
            int st = step - ns0;
            const int nst = tn.S.ncb * 3;
            st = st < nst ? st : (nst > 0 ? nst - 1 : 0);
            const int cbl = st / 3;
            n0 = tn.n0; fcb = x_full_chunk(tn.S, cbl); ky = st - 3 * cbl;
        } else {
            const int cbl = step / 3;
            n0 = T.n0; fcb = x_full_chunk(T.S, cbl); ky = step - 3 * cbl;
        }
        const unsigned so = (unsigned)((fcb * A + xi) * 3 + ky) * slab_bytes + (unsigned)n0 * 16u;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            // (a channel tile beyond Cout: a zero-size descriptor -- a scalar select, no per-lane address register)
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.w, n0 + hh * (NTW * 32) + 32 * nt < p.Cout ? w_bytes : 0u);
            b[nt][0] = buf_load4(rs, vb[nt], so);
            b[nt][1] = buf_load4(rs, vb[nt], so + g_bytes);
        }
    };

    // ---- transform items.  CH channels of one patch slot (row, group) per item:
    //        CH = 4 (768-thread workgroups): E = row * 32 + group * 4 + qb * 2 + half         -> channels 8 qb + 4 half .. + 3, 8-byte loads
    //        CH = 2 (512-thread workgroups): E = row * 64 + group * 8 + qb * 4 + pr           -> channels 8 qb + 2 pr, + 1,     4-byte loads
    //      (consecutive lanes cover the 32 contiguous bytes of a pixel's hi -- or lo -- halves, then the next group: a load instruction
    //       touches 8 lines for 32 B each.  Quarter-major, 16 B per line and instruction, the texture addresser was as busy as the
    //       matrix pipe: 4608 line accesses per chunk of F(6,3) against 4608 MFMA cycles; BSVD_WX_EMAP)
    //      An item is LOADED one chunk before it is FINISHED (decode, BT, re-split, LDS stores): the raw registers ride through the
    //      MFMA steps in between, so that no wave ever waits for the activation tensor (1-3 us away) with nothing else to issue.
    //      Rows outside the image need no mask -- their byte offsets fall outside the buffer descriptor (negative rows wrap to > 2 GiB),
    //      and an inactive lane starts from the out-of-range sentinel; only columns are compared.
    MixConst mixk = {1.0f, -1.0f};
    asm volatile("" : "+s"(mixk.one), "+s"(mixk.mone));         // opaque: see dec_pair / split_pair
    constexpr int CH = C::NTHREADS == 512 ? 2 : 4;        // channels per transform item (512-thread workgroups: 4-byte loads, two items per lane)
    static_assert(XF != 1 || CH == 2, "fp32 input: the 2-channel items of the 512-thread workgroups");
    constexpr int NDW = CH / 2;                                   // dwords per pixel and part
    struct Raw { unsigned h[A][NDW], l[A][NDW]; };
    // item -> lane order: the two 8-channel quarters of a slot on ADJACENT lanes (8 (4) consecutive lanes read 32 contiguous bytes of one pixel:
    // half the lines per load instruction of the quarter-major order)
    auto item_geom = [](int E, int &row, int &qb, int &g, int &sub) __attribute__((always_inline)) {
        if constexpr (CH == 4) {
            row = E >> 5; sub = (E & 1) * 8;
            g = (E >> 2) & 7; qb = (E >> 1) & 1;
        } else {
            row = E >> 6; sub = (E & 3) * 4;
            g = (E >> 3) & 7; qb = (E >> 2) & 1;
        }
    };
    auto item_load = [&](const XChunkSrc &c, int E, bool active, Raw &r) __attribute__((always_inline)) {
        int row, qb, g, sub;
        item_geom(E, row, qb, g, sub);
        int gx0 = c.ox0 - 1 + M * g;
        if constexpr (C::FOLD) {          // patch rows 10 .. 19: the second half's, TWPX columns to the right
            const bool up = row >= C::PRH;
            row = up ? row - C::PRH : row;
            gx0 = up ? gx0 + C::TWPX : gx0;
        }
        if constexpr (XF == 1) {
            // fp32 channels: the item's two channels 8 qb + sub / 2, + 1 are 8 contiguous bytes (8 adjacent lanes = one pixel's 64-byte chunk)
            const unsigned base = active ? (unsigned)((c.oy0 - 1 + row) * p.W + gx0) * c.ps4 + (unsigned)(qb * 32 + 2 * sub) : BSVD_WX_OOB;
            if (c.x_inside) {            // interior tile: scalar position offsets (see below)
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    const u32x2 v = buf_load2(c.rs, base, c.soff + (unsigned)i * c.ps4);
                    r.h[i][0] = v[0]; r.l[i][0] = v[1];
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const u32x2 v = buf_load2(c.rs, (unsigned)(gx0 + i) < (unsigned)p.W ? base + (unsigned)i * c.ps4 : BSVD_WX_OOB, c.soff);
                r.h[i][0] = v[0]; r.l[i][0] = v[1];
            }
            return;
        }
        const unsigned base = active ? (unsigned)((c.oy0 - 1 + row) * p.W + gx0) * c.ps4 + (unsigned)(qb * 16 + sub) : BSVD_WX_OOB;
        if (c.x_inside) {
            // every column of the patch is inside the image (all tiles but the first and last of a tile row): the A positions differ by a
            // SCALAR offset -- one address register per item, no compare / select per position (3 VALU each; with the compares the address
            // arithmetic was a fifth of the transform's instructions).  Rows still need nothing: the range check is on the vector offset.
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const unsigned so = c.soff + (unsigned)i * c.ps4;
                if constexpr (CH == 4) {
                    const u32x2 hv = buf_load2(c.rs, base, so), lv = buf_load2(c.rs, base, so + 32u);
                    r.h[i][0] = hv[0]; r.h[i][NDW - 1] = hv[1]; r.l[i][0] = lv[0]; r.l[i][NDW - 1] = lv[1];
                } else {
                    r.h[i][0] = __builtin_amdgcn_raw_buffer_load_b32(c.rs, base, so, 0);
                    r.l[i][0] = __builtin_amdgcn_raw_buffer_load_b32(c.rs, base, so + 32u, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const unsigned voff = (unsigned)(gx0 + i) < (unsigned)p.W ? base + (unsigned)i * c.ps4 : BSVD_WX_OOB;
            if constexpr (BSVD_WX_ABL & 16) {
#pragma unroll
                for (int k = 0; k < NDW; ++k) { r.h[i][k] = 0x3c003c00u ^ (voff & 0x00ff00ffu); r.l[i][k] = 0x1c001c00u; }
            } else if constexpr (CH == 4) {
                const u32x2 hv = buf_load2(c.rs, voff, c.soff), lv = buf_load2(c.rs, voff, c.soff + 32u);
                r.h[i][0] = hv[0]; r.h[i][NDW - 1] = hv[1]; r.l[i][0] = lv[0]; r.l[i][NDW - 1] = lv[1];
            } else {
                r.h[i][0] = __builtin_amdgcn_raw_buffer_load_b32(c.rs, voff, c.soff, 0);
                r.l[i][0] = __builtin_amdgcn_raw_buffer_load_b32(c.rs, voff, c.soff + 32u, 0);
            }
        }
    };
    auto item_finish = [&](unsigned char *vbuf, int E, bool active, const Raw &r) __attribute__((always_inline)) {
        int row, qb, g, sub;
        item_geom(E, row, qb, g, sub);
        unsigned char *dst = vbuf + qb * C::PLANE + (row * 8 + g) * 16 + sub;
        if constexpr (XF == 1) {
            // channel by channel (BT on channel PAIRS, v_pk_add_f32: 5-7 % slower per layer -- packed fp32 beside MFMA waves, r05e)
            float d0[A], d1[A], v0[A], v1[A];
            float v[A][2];
#pragma unroll
            for (int i = 0; i < A; ++i) { d0[i] = __builtin_bit_cast(float, r.h[i][0]); d1[i] = __builtin_bit_cast(float, r.l[i][0]); }
            F::input(d0, v0);
            F::input(d1, v1);
#pragma unroll
            for (int i = 0; i < A; ++i) { v[i][0] = v0[i]; v[i][1] = v1[i]; }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                unsigned hp, lp;
                split_pair(v[i][0], v[i][1], hp, lp, mixk);
                if (active) {
                    *reinterpret_cast<unsigned *>(dst + i * 4 * C::PLANE) = hp;
                    *reinterpret_cast<unsigned *>(dst + i * 4 * C::PLANE + 2 * C::PLANE) = lp;
                }
            }
            return;
        }
#pragma unroll
        for (int cp = 0; cp < NDW; ++cp) {            // a channel pair = one dword per position and part
            float v[2][A];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                float d[A];
                __builtin_amdgcn_sched_barrier(0);    // one channel at a time (the scheduler otherwise decodes everything first: 64 live floats at M = 6)
#pragma unroll
                for (int i = 0; i < A; ++i) d[i] = cc ? dec_pair<1>(r.h[i][cp], r.l[i][cp], mixk) : dec_pair<0>(r.h[i][cp], r.l[i][cp], mixk);
                F::input(d, v[cc]);
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                unsigned hp, lp;
                split_pair(v[0][i], v[1][i], hp, lp, mixk);
                if (active && !((BSVD_WX_ABL & 32) && hp == 0x12345678u)) {
                    *reinterpret_cast<unsigned *>(dst + i * 4 * C::PLANE + cp * 4) = hp;
                    *reinterpret_cast<unsigned *>(dst + i * 4 * C::PLANE + 2 * C::PLANE + cp * 4) = lp;
                }
            }
        }
    };
    // A wave's items of chunk `cc` (live index).  768 threads: one 4-channel item on 48 lanes of every wave (576 items).  512 threads:
    // two 2-channel items per lane (1024 of 1152) + 128 left over (rows 16, 17), one more item for the two waves cc % NW, (cc + 1) % NW.
    // 256 threads: two 4-channel items per lane (512 of 576) + 64 left over, one more item for the wave cc % NW.
    // The half-height tile (10 patch rows): one full item per lane instead of two, the same left-over blocks (rows 8, 9).
    // (Measured and removed, DESIGN 4.1d / 11 with the commit that last held them: the transform inside each wave's own MFMA stream in three forms --
    //  scheduler-driven, a stage per MFMA step, instruction by instruction: = / +4 % / +8.5 % --, skipping the dead MFMA tiles of a short last row band
    //  inside the K loop (net zero; the folded / 8-row tail bands below do it per workgroup), uneven dealing of the transform items (slower).)
    constexpr int NITEMS = C::PR * (CH == 4 ? 32 : 64);
    constexpr int NFULL = NITEMS / C::NTHREADS;                                               // whole-workgroup item sweeps per chunk
    constexpr int REM = NITEMS - NFULL * C::NTHREADS;
    constexpr bool PARTIAL = C::NTHREADS != 768 && (REM % 64 != 0 || NFULL == 0);             // the rest as one more sweep with the lanes beyond it idle
    constexpr int NMAIN = C::NTHREADS == 768 ? 1 : PARTIAL ? NFULL + 1 : NFULL;
    constexpr int NROT = (C::NTHREADS == 768 || PARTIAL) ? 0 : REM / 64;               // 64-item blocks left over per chunk
    constexpr int ROT0 = NFULL * C::NTHREADS;                                                 // first left-over item
    static_assert(C::NTHREADS == 768 || C::NTHREADS == 512 || C::NTHREADS == 256, "item map");
    auto lane_id = [&]() __attribute__((always_inline)) {      // opaque per use: the item geometry is re-derived per chunk -- hoisted out of the K loop it pins ~30 registers
        int tl = lane;
        asm volatile("" : "+v"(tl));
        return tl;
    };
    auto main_E = [&](int k, int tl) __attribute__((always_inline)) {
        return C::NTHREADS == 768 ? wid * 48 + tl : (k < NFULL || PARTIAL) ? k * C::NTHREADS + wid * 64 + tl : ROT0 + wid * (REM / C::NW) + tl;
    };
    auto main_active = [&](int k, int tl) __attribute__((always_inline)) {
        return C::NTHREADS == 768 ? tl < 48 : k < NFULL ? true : PARTIAL ? k * C::NTHREADS + wid * 64 + tl < NITEMS : tl < REM / C::NW;
    };
    int rot_base = 0;          // chunks of the workgroup's earlier tiles: the rotation runs on across tiles (a chunk is requested under one tile's
                               // numbering and finished under the next one's)
    // the left-over blocks rotate over all waves (dealt to the MFMA-first waves only, or a whole block more per MFMA-first wave: slower, r05h)
    auto rot_slot = [&](int cc) __attribute__((always_inline)) {                // 0 / 1: this wave takes left-over block 0 / 1 of chunk cc, -1: none
        if constexpr (NROT == 0) return -1;
        const int d = (wid - (cc + rot_base) % C::NW + C::NW) % C::NW;
        return d < NROT ? d : -1;
    };
    // PP raw register sets: chunk cc's items live in set cc % PP.  PP = 2 where the registers are there (F(2,3)): an item is then
    // requested TWO chunks before it is finished
    constexpr int PP = (M == 2 && C::NTHREADS == 512) ? (PERSIST ? 1 : 2) : 1;
    constexpr int NSLOTS = NMAIN;
    Raw raw[PP][NSLOTS], raw_rot[PP];
    // XV: the chunk's LDS image is A * 4 planes of NSLOT 16-byte units, linear in the unit index u = plane * NSLOT + row * 8 + group: a wave's 64 lanes
    // write 1 KB of consecutive LDS and read eight 128-byte lines (one per patch row) of the tensor's [row][tile][chunk] blocks.  (Unit order
    // [row][plane][group] -- 1 KB of consecutive global memory per wave, the planes' segments 2304 B apart in LDS -- measured 3-4 % slower per layer:
    // profiles/r06e_v_layers_blocks.txt.)
    constexpr int VU = A * 4 * C::NSLOT, NVU = (VU + C::NTHREADS - 1) / C::NTHREADS;
    constexpr unsigned BLKB = (unsigned)v_block_floats(M) * 4u;
    [[maybe_unused]] f32x4 rawv[XV ? PP : 1][XV ? NVU : 1];
    [[maybe_unused]] auto v_load = [&](int cc, int SET, int tl) __attribute__((always_inline)) {
        XChunkSrc c;
        if (cc >= T.S.ncb) c = x_chunk_src(next_tile(), cc - T.S.ncb);
        else c = x_chunk_src(T, cc);
        // bytes: the holding tensor's chunks of one tile (ps4 = its channels x 4 -> channels / 16 blocks), one image row of it; this chunk inside it
        const unsigned tilepitch = (c.ps4 >> 6) * BLKB, rowpitch = tilepitch * (unsigned)(p.v_wg >> 3);
        const unsigned soff = (c.soff >> 6) * BLKB;
        const unsigned tbase = (unsigned)(c.ox0 / C::TWPX) * tilepitch;
#pragma unroll
        for (int k = 0; k < NVU; ++k) {
            const int u = k * C::NTHREADS + wid * 64 + tl;
            const int pl = u / C::NSLOT, sl = u - pl * C::NSLOT, g = sl & 7;
            // the two values of a row that need a pixel of the neighbouring tile come from the block's edge line
            const unsigned inblk = ((pl >> 2) == 0 && g == 0) ? (unsigned)(A * 512 + (pl & 3) * 16) : ((pl >> 2) == A - 1 && g == 7) ? (unsigned)(A * 512 + 64 + (pl & 3) * 16)
                                                                                                                                    : (unsigned)(pl * 8 + g) * 16u;
            // rows outside the image fall outside the descriptor (a negative row wraps beyond 2 GiB)
            const unsigned voff = (unsigned)((c.oy0 - 1 + (sl >> 3)) * (int)rowpitch) + tbase + inblk;
            rawv[SET][k] = buf_load4(c.rs, (VU % C::NTHREADS == 0 || u < VU) ? voff : BSVD_WX_OOB, soff);
        }
    };
    [[maybe_unused]] auto v_finish = [&](unsigned char *vbuf, int SET, int tl) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NVU; ++k) {
            const int u = k * C::NTHREADS + wid * 64 + tl;
            if (VU % C::NTHREADS == 0 || u < VU) *reinterpret_cast<f32x4 *>(vbuf + u * 16) = rawv[SET][k];
        }
    };
    using PAll = std::integral_constant<int, 3>;     // part_: 1 the lanes' main items, 2 the rotating left-over block, 3 both
    auto chunk_load = [&](int cc, int SET, auto part_, int tl) __attribute__((always_inline)) {
        if constexpr (XV) { v_load(cc, SET, tl); return; }
        // beyond T's last chunk: the first chunks of the workgroup's next tile (or of a dead tile: zeros)
        XChunkSrc c;
        if (cc >= T.S.ncb) c = x_chunk_src(next_tile(), cc - T.S.ncb);
        else c = x_chunk_src(T, cc);
        if constexpr (decltype(part_)::value & 1) {
#pragma unroll
        for (int k = 0; k < NMAIN; ++k) item_load(c, main_E(k, tl), main_active(k, tl), raw[SET][k]);
        }
        if constexpr (NROT > 0 && (decltype(part_)::value & 2)) {
            const int rs = rot_slot(cc);
            if (rs >= 0) item_load(c, ROT0 + rs * 64 + tl, true, raw_rot[SET]);        // (left-over blocks: patch rows 16, 17)
        }
    };
    auto chunk_finish = [&](int cc, unsigned char *vbuf, int SET, auto part_, int tl) __attribute__((always_inline)) {
        if constexpr (XV) { v_finish(vbuf, SET, tl); return; }
        if constexpr (decltype(part_)::value & 1) {
#pragma unroll
        for (int k = 0; k < NMAIN; ++k) item_finish(vbuf, main_E(k, tl), main_active(k, tl), raw[SET][k]);
        }
        if constexpr (NROT > 0 && (decltype(part_)::value & 2)) {
            const int rs = rot_slot(cc);
            if (rs >= 0) item_finish(vbuf, ROT0 + rs * 64 + tl, true, raw_rot[SET]);
        }
    };

    // ---- prologue (of the workgroup's first tile; a later tile's first chunk is transformed under its predecessor's last MFMA steps)
    f32x16 acc[C::MT][NTW];
    f32x4 bring[3][NTW][2];
    load_b(0, bring[0]);
    load_b(1, bring[1]);
    load_b(2, bring[2]);
    chunk_load(0, 0, PAll{}, lane);
    chunk_finish(0, xsm, 0, PAll{}, lane);
    chunk_load(1, PP - 1, PAll{}, lane);
    if constexpr (PP == 2) chunk_load(2, 0, PAll{}, lane);
    __syncthreads();

    const unsigned a_lane = (unsigned)(xi * 4 * C::PLANE + lh * C::PLANE + li * 16);
    // the waves that share a SIMD (w, w + 4, ..) run the chunk's phases in opposite order.  (Other pairings, both waves in step: +5 .. +7 %;
    //  static wave priorities: 0 .. -6 % -- DESIGN 4.1d, removed knobs BSVD_WX_PHASE / BSVD_WX_SPRIO)
    const int phase = C::NW == 4 ? (wid >> 1) : (wid >> 2) % C::NPH;
    // (the loop body is written out in the loop, not as a lambda: one more level of by-reference closure nesting and hipcc no longer
    //  promotes the captured locals -- kernel parameters, pointers, the raw sets -- out of private memory)
    [[maybe_unused]] const unsigned long long tl_loop = WXT_NOW();
    [[maybe_unused]] unsigned long long tl_epi = 0;
    for (;;) {                 // the workgroup's tiles: one, or (PERSIST) its share of the XCD's range
    const int ncb = T.S.ncb, oy0 = T.oy0, ox0 = T.ox0, n0 = T.n0, f = T.f;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    for (int cb0 = 0; cb0 < ncb; cb0 += PP) {
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        const int cb = cb0 + u;
        if (cb >= ncb) break;
        const int setv = PP == 2 ? ((u + 1) & 1) : 0;          // the raw set of chunk cb + 1 (a constant after unrolling)
        const unsigned char *pcur = xsm + (cb & 1) * C::V_BUF + a_lane;
        unsigned char *pnext = xsm + ((cb + 1) & 1) * C::V_BUF;
        auto mfma_phase = [&](auto nmt_) __attribute__((always_inline)) {      // NMT: the MFMA tiles (4-row bands) of the tile that hold image rows
            constexpr int NMT = decltype(nmt_)::value;
            f32x4 afr[2][2];          // fragments one (ky, mt) step ahead, see the interleaved schedule
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) afr[0][pt] = *reinterpret_cast<const f32x4 *>(pcur + pt * 2 * C::PLANE);
            static_for<0, 3 * NMT>([&](auto k_) __attribute__((always_inline)) {
                constexpr int S_ = decltype(k_)::value, KY = S_ / NMT, mt = S_ % NMT;
                const f32x4 (&b)[NTW][2] = bring[KY];
                {
                    if constexpr (S_ < 3 * NMT - 1) {
                        constexpr int KY1 = (S_ + 1) / NMT, mt1 = (S_ + 1) % NMT;
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt)
                            afr[(S_ + 1) & 1][pt] = *reinterpret_cast<const f32x4 *>(pcur + pt * 2 * C::PLANE + C::frag_row(mt1, KY1) * 128);
                    }
                    const f32x4 (&a)[2] = afr[S_ & 1];
                    __builtin_amdgcn_sched_barrier(0);
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
            // the weights of the NEXT chunk, all three slabs, requested here: a wave's loads return in issue order, so every load a
            // phase consumes must have been issued before the activation loads (1-3 us away) that the following phase waits for --
            // issue order = consumption order, and nobody ever waits for the activation tensor on behalf of a weight slab
            static_for<0, 3>([&](auto k_) __attribute__((always_inline)) { load_b((cb + 1) * 3 + decltype(k_)::value, bring[decltype(k_)::value]); });
        };
        // transform step of this iteration: finish chunk cb + 1 (requested PP iterations ago) into the other buffer, request chunk cb + 1 + PP
        // (hipcc's vmcnt in front of the first MFMA of a phase drains every activation request the wave has in flight -- the branches around the
        //  requests make its counts conservative.  With exact counts, requests behind the MFMA steps in both phase orders, the kernel is 8-10 %
        //  SLOWER: the drain is not what a wave waits for, early requests are what counts.  DESIGN 4.1d, removed knobs BSVD_WX_LOADLAST / _UNCOND)
        auto xform = [&](auto part_) __attribute__((always_inline)) {      // part_: 1 finish, 2 request, 3 both
            constexpr int XP = decltype(part_)::value;
            const int tl = lane_id();
            // (one tile per workgroup: nothing behind the tile's last chunk -- its transform and the requests of the last PP + 1 iterations
            //  would be zeros from zero-size descriptors, a whole transform phase per tile for nothing)
            constexpr bool TS_FIN = !PERSIST, TS_LD = TS_FIN && M == 2;      // (F(6,3) with the request skip too: 25 spills)
            if constexpr (XP & 1) if (!TS_FIN || cb + 1 < ncb) chunk_finish(cb + 1, pnext, setv, PAll{}, tl);
            if constexpr (XP & 2) if (!(BSVD_WX_ABL & 64) && (!TS_LD || cb + 1 + PP < ncb)) chunk_load(cb + 1 + PP, setv, PAll{}, tl);
        };
        // (one copy of the MFMA steps between two conditional transforms: an if / else with the phases in opposite orders made the
        //  register allocator carry the accumulators in two register sets and spill 60-260 registers)
        [[maybe_unused]] const unsigned long long t0 = WXT_NOW();
        using XFin = std::integral_constant<int, 3>;
        if (!(BSVD_WX_ABL & 1) && phase != 0) xform(XFin{});
        __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const unsigned long long t1 = WXT_NOW();
        if (!(BSVD_WX_ABL & 2)) mfma_phase(std::integral_constant<int, C::MT>{});
        __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const unsigned long long t2 = WXT_NOW();
        if (!(BSVD_WX_ABL & 1) && phase == 0) xform(XFin{});
        [[maybe_unused]] const unsigned long long t3 = WXT_NOW();
        if (!(BSVD_WX_ABL & 8)) __syncthreads();
#ifdef BSVD_WX_TL
        { const unsigned long long t4 = WXT_NOW(); tl_acc[0] += t1 - t0; tl_acc[1] += t2 - t1; tl_acc[2] += t3 - t2; tl_acc[3] += t4 - t3; }
#endif
    }
    }

    tl_epi = WXT_NOW();
    // ---- epilogue: NRND rounds of publish -> finish
    const int Cq = p.Cout >> 2;
    auto coff16 = [](int c8) { return (c8 >> 4) * 16 + ((c8 >> 3) & 1) * 4; };
    auto finish = [&](auto epi_c, auto act_c) __attribute__((always_inline)) {
        constexpr int EPI = decltype(epi_c)::value, ACT = decltype(act_c)::value;
        const bool has_skip = EPI == BSVD_EPI_PS_ADD && p.extra != nullptr;
        unsigned char *const xch = xsm + C::XCH_OFF;
        // (lane-derived values of the epilogue from an opaque copy: inside the tile loop they would otherwise be hoisted out of it and
        //  ride through the K loop, where every register counts)
        const int lane = lane_id();
        const int li = lane & 31, lh = lane >> 5;
        // finisher geometry: wave -> (block, column range); the block's channel tile -- and so the bias -- is the same in every round.
        // The bias is requested HERE, once: a load inside a round waits (vmcnt counts loads and stores alike, in order) for the
        // previous round's stores to be acknowledged
        const int blk = wid % C::NBP, part = wid / C::NBP;
        const int mtl = blk / (NH * NTW), ntg = blk % (NH * NTW);
        const int q = lane & 3;
        const int n8 = n0 + ntg * 32 + 8 * q;
        f32x4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = bq0;
        if (p.bias && n8 < p.Cout && part < C::NPART) {
            bq0 = *reinterpret_cast<const f32x4 *>(p.bias + n8);
            bq1 = *reinterpret_cast<const f32x4 *>(p.bias + n8 + 4);
        }
        // publish: block (mtl, channel tile hh * NTW + nt), position xi: [4 register quads][64 lanes] x 16 B, slot XOR-swizzled
        auto publish = [&](auto rnd_) __attribute__((always_inline)) {
            constexpr int rnd = decltype(rnd_)::value;
            unsigned char *const xr = xch;
#pragma unroll
            for (int mtl = 0; mtl < C::MTL; ++mtl)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int blk = mtl * (NH * NTW) + hh * NTW + nt;
                    unsigned char *base = xr + (blk * A + xi) * 4096;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x16 &t = acc[C::MTL * rnd + mtl][nt];
                        const int slot = (lh * 32 + li) ^ (8 * lh + 4 * (r4 >> 1));
                        *reinterpret_cast<f32x4 *>(base + r4 * 1024 + slot * 16) = f32x4{t[4 * r4], t[4 * r4 + 1], t[4 * r4 + 2], t[4 * r4 + 3]};
                    }
                }
        };
        static_for<0, C::NRND>([&](auto rnd_) __attribute__((always_inline)) {
            constexpr int rnd = decltype(rnd_)::value;
            if (rnd) __syncthreads();                // round 0's readers are done
            publish(rnd_);
            // PixelShuffle + skip: the skip values of this round's pixels are requested HERE, in front of the barrier and the exchange reads (asked
            // for at the point of use, each of the 8 requests per wave and tile stood exposed for a DRAM round trip)
            constexpr int JN_ = M / C::NPART;
            // folded tile: this round's MFMA tiles are the second half's (8 rows up, TWPX columns to the right)
            static_assert(!C::FOLD || 2 % C::MTL == 0, "a round holds MFMA tiles of one half");
            constexpr int FDY = C::FOLD && C::MTL * rnd >= 2 ? -8 : 0, FDX = C::FOLD && C::MTL * rnd >= 2 ? C::TWPX : 0;
            [[maybe_unused]] f32x4 skh[2][JN_], skl[2][JN_];
            constexpr bool SKIPPF = M == 2 && !PERSIST;        // (F(4,3) / F(6,3): the 32-48 registers it holds spill)
            if constexpr (EPI == BSVD_EPI_PS_ADD && SKIPPF) {
                if (has_skip && part < C::NPART) {
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
                        const int m = (lane + 64 * sidx) >> 2;
                        const int oy = oy0 + 4 * (C::MTL * rnd + mtl) + (m >> 3) + FDY;
#pragma unroll
                        for (int jj = 0; jj < JN_; ++jj) {
                            const int ox = ox0 + M * (m & 7) + part * JN_ + jj + FDX;
                            const bool live = oy < p.Ho && ox < p.Wo && n8 < p.Cout;
                            const int sub = n8 / Cq, c8 = n8 - sub * Cq;
                            const int64_t upix = (int64_t)(2 * oy + (sub >> 1)) * (2 * p.Wo) + (2 * ox + (sub & 1));
                            const float *ep = p.extra + (int64_t)f * p.extra_fs + upix * p.extra_ps + coff16(c8);
                            skh[sidx][jj] = live ? *reinterpret_cast<const f32x4 *>(ep) : f32x4{0.f, 0.f, 0.f, 0.f};
                            skl[sidx][jj] = live ? *reinterpret_cast<const f32x4 *>(ep + 8) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
            }
            __syncthreads();
            if constexpr (YV) {
            [&]() __attribute__((always_inline)) {
            static_assert(!YV || EPI == BSVD_EPI_PLAIN || true, "");
            if (part >= C::NPART || (BSVD_WX_ABL & 4)) return;
            // one finisher wave per (block, half of the MFMA tile's rows): sidx = part, part + NPART, ..
#pragma unroll
            for (int s0 = 0; s0 < 2; s0 += C::NPART) {
                const int sidx = s0 + (C::NPART == 2 ? part : 0);
                const int m = (lane + 64 * sidx) >> 2, g = m & 7;
                float mv[A][8];
#pragma unroll
                for (int x = 0; x < A; ++x) {
                    const unsigned char *base = xch + (blk * A + x) * 4096;
                    const int slot = ((q & 1) * 32 + m) ^ (8 * (q & 1) + 4 * (q >> 1));
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(base + (2 * (q >> 1)) * 1024 + slot * 16);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(base + (2 * (q >> 1) + 1) * 1024 + slot * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { mv[x][k] = v0[k]; mv[x][4 + k] = v1[k]; }
                }
                const int oy = oy0 + 4 * (C::MTL * rnd + mtl) + (m >> 3);
                const int oxg = ox0 + M * g;
                // d[1 .. M]: the group's pixels (bias, activation; zero beyond the image: the reader's zero padding); d[0], d[M + 1]: the neighbours'
                float d[A][8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float mm[A], oo[M];
#pragma unroll
                    for (int x = 0; x < A; ++x) mm[x] = mv[x][k];
                    F::output(mm, oo);
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        float v = oo[j] + (k < 4 ? bq0[k] : bq1[k - 4]);
                        if constexpr (ACT == BSVD_ACT_RELU6) v = __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
                        else if constexpr (ACT == BSVD_ACT_RELU) v = fmaxf(v, 0.f);
                        d[1 + j][k] = oxg + j < p.Wo ? v : 0.f;
                    }
                }
                // the last pixel of the group to the left, the first of the group to the right: lanes -4 / +4 (same row of the MFMA tile; a tile's
                // first / last group takes 0 here and is completed by the patch pass)
                const int la = ((lane - 4) & 63) * 4, ra = ((lane + 4) & 63) * 4;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float l = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(la, __builtin_bit_cast(int, d[M][k])));
                    const float r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ra, __builtin_bit_cast(int, d[1][k])));
                    d[0][k] = g > 0 ? l : 0.f;
                    d[A - 1][k] = g < 7 ? r : 0.f;
                }
                const bool live = oy < p.Ho && n8 < p.Cout;
                float *const fbase = p.y + (int64_t)f * p.y_fs;
                // edge record of the tile: [row][tile x][0: partial of position 0, 1: partial of position A - 1, 2: first column, 3: last column][Cout]
                constexpr int BLK = v_block_floats(M);
                float *const erow = fbase + (int64_t)p.Ho * p.ntx * (p.Cout >> 4) * BLK + ((int64_t)(oy * p.ntx + ox0 / C::TWPX) * 4) * p.Cout + n8;
                float vv[A][8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float din[A], vout[A];
#pragma unroll
                    for (int x = 0; x < A; ++x) din[x] = d[x][k];
                    F::input(din, vout);
#pragma unroll
                    for (int x = 0; x < A; ++x) vv[x][k] = vout[x];
                }
                if (live && g == 0 && !(BSVD_WX_ABL & 256)) {
                    *reinterpret_cast<f32x4 *>(erow) = f32x4{vv[0][0], vv[0][1], vv[0][2], vv[0][3]};
                    *reinterpret_cast<f32x4 *>(erow + 4) = f32x4{vv[0][4], vv[0][5], vv[0][6], vv[0][7]};
                    *reinterpret_cast<f32x4 *>(erow + 2 * p.Cout) = f32x4{d[1][0], d[1][1], d[1][2], d[1][3]};
                    *reinterpret_cast<f32x4 *>(erow + 2 * p.Cout + 4) = f32x4{d[1][4], d[1][5], d[1][6], d[1][7]};
                }
                if (live && g == 7 && !(BSVD_WX_ABL & 256)) {
                    *reinterpret_cast<f32x4 *>(erow + p.Cout) = f32x4{vv[A - 1][0], vv[A - 1][1], vv[A - 1][2], vv[A - 1][3]};
                    *reinterpret_cast<f32x4 *>(erow + p.Cout + 4) = f32x4{vv[A - 1][4], vv[A - 1][5], vv[A - 1][6], vv[A - 1][7]};
                    *reinterpret_cast<f32x4 *>(erow + 3 * p.Cout) = f32x4{d[M][0], d[M][1], d[M][2], d[M][3]};
                    *reinterpret_cast<f32x4 *>(erow + 3 * p.Cout + 4) = f32x4{d[M][4], d[M][5], d[M][6], d[M][7]};
                }
                // block (row, tile, chunk): [position][quarter][group] x 16 B
                float *const vrow = fbase + ((int64_t)(oy * p.ntx + ox0 / C::TWPX) * (p.Cout >> 4) + (n8 >> 4)) * BLK + (((n8 >> 3) & 1) * 8 + g) * 4;
#pragma unroll
                for (int x = 0; x < A; ++x) {
                    unsigned hp[4], lp[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) split_pair(vv[x][2 * c], vv[x][2 * c + 1], hp[c], lp[c], mixk);
                    if (live && !((BSVD_WX_ABL & 128) && hp[0] != 0x12345678u)) {
                        *reinterpret_cast<f32x4 *>(vrow + x * 128) = __builtin_bit_cast(f32x4, u32x4_t{hp[0], hp[1], hp[2], hp[3]});
                        *reinterpret_cast<f32x4 *>(vrow + x * 128 + 64) = __builtin_bit_cast(f32x4, u32x4_t{lp[0], lp[1], lp[2], lp[3]});
                    }
                }
            }
            }();
            } else
            [&]() __attribute__((always_inline)) {
            // finish
            if (part >= C::NPART || (BSVD_WX_ABL & 4)) return;
            constexpr int JN = M / C::NPART;             // columns per finisher
            const int j0 = part * JN;
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const int m = (lane + 64 * sidx) >> 2;   // group position of the MFMA tile: row m >> 3, group m & 7
                // the 8 channels 8 q .. 8 q + 7 of group m: register quads 2 (q >> 1), + 1 of lane (m, lh = q & 1)
                float mv[A][8];
#pragma unroll
                for (int x = 0; x < A; ++x) {
                    const unsigned char *base = xch + (blk * A + x) * 4096;
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
                const int oy = oy0 + 4 * (C::MTL * rnd + mtl) + (m >> 3) + FDY;
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
                    const int ox = ox0 + M * (m & 7) + jcol + FDX;
                    const bool live = oy < p.Ho && ox < p.Wo && n8 < p.Cout;
                    float *dst;
                    if constexpr (EPI == BSVD_EPI_PS_ADD) {
                        const int sub = n8 / Cq, c8 = n8 - sub * Cq;
                        const int64_t upix = (int64_t)(2 * oy + (sub >> 1)) * (2 * p.Wo) + (2 * ox + (sub & 1));
                        dst = p.y + (int64_t)f * p.y_fs + upix * Cq + (p.y_f32 ? c8 : coff16(c8));
                        if (has_skip && live) {
                            f16x8 eh, el;
                            if constexpr (SKIPPF) {
                                eh = __builtin_bit_cast(f16x8, skh[sidx][jj]);
                                el = __builtin_bit_cast(f16x8, skl[sidx][jj]);
                            } else {
                                const float *ep = p.extra + (int64_t)f * p.extra_fs + upix * p.extra_ps + coff16(c8);
                                eh = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(ep));
                                el = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(ep + 8));
                            }
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += (float)eh[k] + (float)el[k];
                        }
                    } else {
                        dst = p.y + (int64_t)f * p.y_fs + ((int64_t)oy * p.Wo + ox) * p.Cout + (p.y_f32 ? n8 : coff16(n8));
                    }
                    if (live && p.y_f32) {       // BsvdConvArgs.y_f32: plain fp32 channels for a Winograd-only consumer (no split, no range clamp: its transform saturates)
                        *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else
                    if (live) {
                        constexpr bool bounded = (ACT == BSVD_ACT_RELU6 && EPI == BSVD_EPI_PLAIN) || !BSVD_EPI_CLAMP;
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
            }();
        });
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
    if constexpr (!PERSIST) break;
    jt += xcd_wgs;
    if (jt >= xcd_count) break;
    rot_base += ncb;
    T = decode_tile(jt);       // (its first chunk is in V buffer 0, its chunks 1 .. PP in the raw registers, its first weight slabs in `bring`)
    }
#ifdef BSVD_WX_TL
    if (lane == 0 && blockIdx.x < BSVD_WX_TL_SLOTS) {
        const unsigned long long t_end = WXT_NOW();
        unsigned long long *o = g_wx_tl[blockIdx.x][wid];
        o[0] = tl_acc[0]; o[1] = tl_acc[1]; o[2] = tl_acc[2]; o[3] = tl_acc[3];
        o[4] = tl_loop - tl_start; o[5] = t_end - tl_epi; o[6] = t_end - tl_start; o[7] = (unsigned long long)T.S.ncb;
    }
#endif
}

template <int M, int NH, int NTW, int MT, bool PERSIST, int XF = 0, bool YV = false>
__global__ __launch_bounds__((XCfg<M, NH, NTW, MT, PERSIST>::NTHREADS), (XCfg<M, NH, NTW, MT, PERSIST>::NW / 4 * XCfg<M, NH, NTW, MT, PERSIST>::WGS)) void winox_kernel(const ConvParams p)
{
    winox_tile<M, NH, NTW, MT, PERSIST, XF, XCfg<M, NH, NTW, MT, PERSIST>::TR, 0, YV>(p);
}

#ifndef BSVD_WX_TAIL
#define BSVD_WX_TAIL 2     // 16-row tile grids whose last row band has <= 8 live rows: 0 nothing special, 1 that band runs the 8-row body, 2 (F(2,3); F(6,3)'s
                           // transform buffers do not fit two more patch rows: 1) that band is walked by folded tiles, two of its tiles per workgroup
#endif
constexpr bool wx_tail_folds(int m) { return BSVD_WX_TAIL == 2 && m == 2; }

// 16-row tile grid whose LAST row band has <= 8 live rows (Ho mod 16 in 1..8: the 135-row layers of a 540 x 960 frame, the 120-row ones of
// 480 x 856): those workgroups run the 8-row body -- the same instruction sequence per output (bit-identical, like the 8-row tile of
// small grids), 0.57 of a full tile's time for tiles that are <= half alive.  A wave-uniform branch at the very top, two complete
// bodies: nothing inside the K loops knows about it (the in-loop form, BSVD_WX_DEADROWS, cost every tile its MFMA schedule).
template <int M, int NH, int NTW, int XF, bool YV = false>
__global__ __launch_bounds__((XCfg<M, NH, NTW, 4, false>::NTHREADS), (XCfg<M, NH, NTW, 4, false>::NW / 4)) void winox_kernel_tail(const ConvParams p)
{
    // this workgroup's tile, decoded like winox_tile does (XCD-contiguous tile ranges, optional reverse walk)
    constexpr bool FOLDS = wx_tail_folds(M) && XF != 2 && !YV;       // (transformed-domain input / output: the 8-row body for the short band, like F(6,3))
    const WxGrid G = wx_grid(p.nty, p.ntx, p.nct, p.Ho, FOLDS);
    const int ntiles = p.frames * G.per_frame;
    const int bid = blockIdx.x, xcd = bid & 7;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    if (p.flip) lid = ntiles - 1 - lid;
    if constexpr (FOLDS) {
        if (G.fold && lid % G.per_frame >= G.nreg) winox_tile<M, NH, NTW, 4, false, XF, 16, 2>(p);
        else winox_tile<M, NH, NTW, 4, false, XF, 16, 1>(p);
    } else {
        const int ty = (lid / (p.nct * p.ntx)) % p.nty;
        if (p.nty >= 2 && p.Ho - ty * 16 <= 8) winox_tile<M, NH, NTW, 2, false, XF, 16, 0, YV>(p);
        else winox_tile<M, NH, NTW, 4, false, XF, 16, 0, YV>(p);
    }
}

// The patch pass behind a y_v launch: position 0 of a tile's first group and position A - 1 of its last group need one pixel of the
// neighbouring tile -- BT[0][0] * (last column of the tile to the left), BT[A - 1][A - 1] * (first column of the tile to the right).  One
// thread per (frame, row, tile, side, 8 channels): partial sum + coefficient x the neighbour's edge column (0 at the image border) from the frame's
// edge record, split, into the block's EDGE LINE (dense 16-byte writes: a block's line is filled by four threads).  Readers take these two
// values from the edge line only; what the producer left in the planes at those two slots is never read.
template <int M>
__global__ void v_patch_kernel(float *__restrict__ y, int64_t y_fs, int frames, int Ho, int Cout, int ntx)
{
    constexpr int A = M + 2, BLK = v_block_floats(M);
    using F = WinoForm<M>;
    fp16_saturate_on();
    const int c8n = Cout >> 3;
    const int64_t total = (int64_t)frames * Ho * ntx * 2 * c8n;
    const float cl = (float)F::BT[0][0], cr = (float)F::BT[A - 1][A - 1];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        int64_t r = i / c8n;
        const int side = (int)(r & 1); r >>= 1;
        const int t = (int)(r % ntx); r /= ntx;
        const int row = (int)(r % Ho);
        const int f = (int)(r / Ho);
        float *const fbase = y + f * y_fs;
        const float *const e = fbase + (int64_t)Ho * ntx * (Cout >> 4) * BLK + ((int64_t)row * ntx * 4) * Cout + c8 * 8;
        // side 0: partial (record 0 of tile t) + cl * last column of tile t - 1 (record 3); side 1: partial (record 1) + cr * first column of tile t + 1 (record 2)
        const float *const part = e + (int64_t)(t * 4 + side) * Cout;
        const int tn = side ? t + 1 : t - 1;
        const bool has = tn >= 0 && tn < ntx;
        const float *const pix = e + (int64_t)((has ? tn : t) * 4 + (side ? 2 : 3)) * Cout;
        const float c = side ? cr : cl;
        _Float16 hi[8], lo[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = has ? __builtin_fmaf(c, pix[k], part[k]) : part[k];
            hi[k] = (_Float16)v;
            lo[k] = (_Float16)__builtin_fmaf((float)hi[k], -1.0f, v);
        }
        float *const dst = fbase + ((int64_t)(row * ntx + t) * (Cout >> 4) + (c8 >> 1)) * BLK + A * 128 + (side * 4 + (c8 & 1)) * 4;
        *reinterpret_cast<f32x4 *>(dst) = *reinterpret_cast<const f32x4 *>(hi);
        *reinterpret_cast<f32x4 *>(dst + 8) = *reinterpret_cast<const f32x4 *>(lo);
    }
}

template <int M>
static int launch_v_patch(const ConvParams &p, hipStream_t stream)
{
    const int ntx = (p.Wo + 8 * M - 1) / (8 * M);
    const int64_t total = (int64_t)p.frames * p.Ho * ntx * 2 * (p.Cout >> 3);
    int64_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(v_patch_kernel<M>, dim3((unsigned)grid), dim3(256), 0, stream, p.y, p.y_fs, p.frames, p.Ho, p.Cout, ntx);
    return (int)hipGetLastError();
}

template <int M, int NH, int NTW, int MT = 4, bool PERSIST = false, int XF = 0, bool YV = false>
static int launch_winox_cfg(const ConvParams &pin, hipStream_t stream, char *name, int name_len, int max_wgs = BSVD_CUS)
{
    using C = XCfg<M, NH, NTW, MT, PERSIST>;
    if constexpr (XF == 0 && !PERSIST && C::NTHREADS == 512 && (M == 2 || M == 6)) {      // the product configurations exist for every input format
        if (pin.x_f32) return launch_winox_cfg<M, NH, NTW, MT, PERSIST, 1>(pin, stream, name, name_len, max_wgs);
        if (pin.x_v) return launch_winox_cfg<M, NH, NTW, MT, PERSIST, 2>(pin, stream, name, name_len, max_wgs);
    } else if constexpr (XF == 0 && !PERSIST && M == 4) {        // (measurement builds: F(4,3) reads transformed tensors too)
        if (pin.x_v) return launch_winox_cfg<M, NH, NTW, MT, PERSIST, 2>(pin, stream, name, name_len, max_wgs);
        if (pin.x_f32) { set_error("bsvd_conv3x3: x_f32 is not available for this Winograd variant"); return -19; }
    } else if constexpr (XF == 0) {
        if (pin.x_f32 || pin.x_v) { set_error("bsvd_conv3x3: x_f32 / x_v are not available for this Winograd variant"); return -19; }
    }
    // transformed-domain OUTPUT: the product's F(6,3) configuration (its tile grid never folds; the epilogue's pixel groups are the reader's)
    if constexpr (!YV) {
        if (pin.y_v) {
            if constexpr (!PERSIST && C::NTHREADS == 512 && M == 6) return launch_winox_cfg<M, NH, NTW, MT, PERSIST, XF, true>(pin, stream, name, name_len, max_wgs);
            else { set_error("bsvd_conv3x3: y_v is not available for this Winograd variant (F(6,3) only)"); return -19; }
        }
    }
    if (name) {
        snprintf(name, name_len, "winox_kernel<F(%d,3),%dx%d>[f16x3]%s%s%s%s", M, NH, NTW, MT == 2 ? "[8 rows]" : "", PERSIST ? "[persistent]" : "", XF == 2 ? "[V in]" : XF ? "[f32 in]" : "", YV ? "[V out]" : "");
        return 0;
    }
    ConvParams p = pin;
    p.ntx = (p.Wo + C::TWPX - 1) / C::TWPX;
    p.nty = (p.Ho + C::TR - 1) / C::TR;
    p.nct = (p.Cout + C::BN - 1) / C::BN;
    constexpr bool TAILK = BSVD_WX_TAIL && MT == 4 && !PERSIST && C::NTHREADS == 512 && (M == 2 || M == 6);
    const int64_t nblk = (int64_t)p.frames * wx_grid(p.nty, p.ntx, p.nct, p.Ho, TAILK && wx_tail_folds(M) && XF != 2 && !YV).per_frame;
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3: grid of %lld workgroups", (long long)nblk); return -1; }
    // the product's 16-row tile launches are ALL this kernel (one symbol per form in a profile): a grid without such a band never takes
    // the branch
    if constexpr (TAILK) {
        static_assert(XCfg<M, NH, NTW, 4, false, wx_tail_folds(M) && XF != 2 && !YV>::LDS_BYTES == C::LDS_BYTES, "the folded tile's transform buffers fit under the exchange");
        static std::atomic<int> granted_t[MAX_DEVICES];
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&winox_kernel_tail<M, NH, NTW, XF, YV>), C::LDS_BYTES, granted_t);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((winox_kernel_tail<M, NH, NTW, XF, YV>), dim3((unsigned)nblk), dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    } else {
        static std::atomic<int> granted[MAX_DEVICES];
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&winox_kernel<M, NH, NTW, MT, PERSIST, XF, YV>), C::LDS_BYTES, granted);
        if (e != hipSuccess) return (int)e;
        const unsigned grid = PERSIST && nblk > max_wgs ? (unsigned)max_wgs : (unsigned)nblk;
        hipLaunchKernelGGL((winox_kernel<M, NH, NTW, MT, PERSIST, XF, YV>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    }
    const int rc = (int)hipGetLastError();
    if constexpr (YV && !(BSVD_WX_ABL & 512)) { if (rc == 0) return launch_v_patch<M>(p, stream); }
    return rc;
}

// Which wino_m codes this build runs.  Product: 2 / 6 (F(2,3) / F(6,3), each picking the half-height tile for grids that do not fill the
// chip) and 42 / 46 (the same forms, never on the half-height tile: launches that share the chip with another graph branch).  A measurement
// build (-DBSVD_MEASURE, tools/build_measure.sh -> build/measure/libbsvd_hip.so; never the product library) adds F(4,3), the 4-wave
// workgroup, the forced tiles, the persistent form and the all-positions-per-wave kernel of conv3x3_wino.hip -- same arithmetic per form,
// kept for the records in DESIGN.md 4.1d and for tests/measure_driver.py.
static bool wino_code_known(int m)
{
    if (m == 2 || m == 6 || m == 42 || m == 46) return true;
#ifdef BSVD_MEASURE
    if (m == 4 || m == 12 || m == 22 || m == 32 || m == 36 || m == 52 || m == 62) return true;      // (14 = F(4,3) all positions per wave: fails parity, not offered)
#endif
    return false;
}

const char *wino_unsupported(const ConvParams &p, int stride)
{
    if (p.prec != 1) return "dtype must be BSVD_F16X3";
    if (!wino_code_known(p.wino_m))
#ifdef BSVD_MEASURE
        return "wino_m must be 2 or 6 (42 / 46: never the half-height tile; measurement build: 4, 12, 22, 32, 36, 52, 62)";
#else
        return "wino_m must be 2 or 6 (42 / 46: the same forms, never on the half-height tile); the other codes exist in measurement builds (-DBSVD_MEASURE) only";
#endif
    if (stride != 1) return "stride must be 1";
    if (p.epilogue == BSVD_EPI_RESID || p.y_planar_ch > 0 || p.head_w) return "only PLAIN / PS_ADD NHWC layers";
    if ((p.fold & 15) != 0) return "fold must be a multiple of 16";
    if (!p.vec_ok) return "16-byte aligned x / halo pointers and strides";
    if ((p.Cout & 31) != 0) return "Cout must be a multiple of 32";
    if (p.epilogue == BSVD_EPI_PS_ADD && (p.extra != nullptr && p.extra_cs != 1)) return "PS_ADD skip tensor must be split16 NHWC (extra_cstride 1)";
    if ((int64_t)p.H * p.W * p.Cin * 4 >= 0x7fffffffLL) return "frame >= 2 GiB";
    // the halo descriptors are hw * pstride * 4 bytes in 32 bits too: a pstride larger than Cin must not wrap (it would read as zeros)
    if (p.fold > 0 && p.halo_prev && (int64_t)p.H * p.W * p.halo_prev_ps * 4 >= 0x7fffffffLL) return "halo_prev: H * W * pstride * 4 >= 2 GiB";
    if (p.fold > 0 && p.halo_next && (int64_t)p.H * p.W * p.halo_next_ps * 4 >= 0x7fffffffLL) return "halo_next: H * W * pstride * 4 >= 2 GiB";
    if ((int64_t)p.Cin * 3 * (p.wino_m % 10 + 2) * p.Cout * 4 >= 0x7fffffffLL) return "packed weights >= 2 GiB";
    return nullptr;
}

int launch_winox(const ConvParams &p, hipStream_t stream, char *name, int name_len)
{
    // F(2,3) / F(6,3).  Small grids (single-frame launches of the stream schedules): one 8-wave workgroup per CU, so the launch takes
    // ceil(workgroups / 256) rounds -- 270 workgroups (256 -> 256 at 135 x 240) cost two full rounds for 1.05 rounds of work.
    // The half-height tile has twice the workgroups and computes every output with the same instruction sequence (bit-identical:
    // stream == clip stays bitwise), at ~0.87 of the full tile's efficiency (10 patch rows per 8, prologue / epilogue per tile).
    // 42 / 46: never the half-height tile -- launches that share the chip with another graph branch (the lagged two-chain stream
    // step): idle CUs are not idle there.
    auto fill = [](int64_t n) { return (double)n / (double)(((n + BSVD_CUS - 1) / BSVD_CUS) * BSVD_CUS); };
    switch (p.wino_m) {
    case 2:
    case 42: {
        using C4 = XCfg<2, 2, 2, 4>;
        const int ntx4 = (p.Wo + C4::TWPX - 1) / C4::TWPX, nct4 = (p.Cout + C4::BN - 1) / C4::BN;
        const int64_t per_row = (int64_t)p.frames * ntx4 * nct4;
        // (the 16-row grid as it is launched: a short last row band folds -- 256 -> 256 on ONE 135 x 240 frame is 240 + 16 = 256 workgroups, one round)
        const int64_t n4 = (int64_t)p.frames * wx_grid((p.Ho + 15) / 16, ntx4, nct4, p.Ho, wx_tail_folds(2) && !p.x_v).per_frame, n2 = per_row * ((p.Ho + 7) / 8);
        if (p.wino_m == 2 && n4 < 8 * BSVD_CUS && 0.87 * fill(n2) > fill(n4)) return launch_winox_cfg<2, 2, 2, 2>(p, stream, name, name_len);
#ifdef BSVD_MEASURE
        // (large grids as 256 persistent workgroups, the transform pipeline running across tile boundaries: built, bit-identical, 6-13 %
        //  slower, DESIGN 4.1d -- only a measurement build with -DBSVD_WX_PERSIST_MIN=<tiles per CU> ever selects it)
        if (n4 >= (int64_t)BSVD_WX_PERSIST_MIN * BSVD_CUS) return launch_winox_cfg<2, 2, 2, 4, true>(p, stream, name, name_len);
#endif
        return launch_winox_cfg<2, 2, 2>(p, stream, name, name_len);
    }
    case 46: return launch_winox_cfg<6, 1, 2>(p, stream, name, name_len);
    case 6: {
        using C4 = XCfg<6, 1, 2, 4>;
        const int64_t per_row = (int64_t)p.frames * ((p.Wo + C4::TWPX - 1) / C4::TWPX) * ((p.Cout + C4::BN - 1) / C4::BN);
        const int64_t n4 = per_row * ((p.Ho + 15) / 16), n2 = per_row * ((p.Ho + 7) / 8);
        if (n4 < 8 * BSVD_CUS && 0.87 * fill(n2) > fill(n4)) return launch_winox_cfg<6, 1, 2, 2>(p, stream, name, name_len);
        return launch_winox_cfg<6, 1, 2>(p, stream, name, name_len);
    }
#ifdef BSVD_MEASURE
    case 22: return launch_winox_cfg<2, 1, 2>(p, stream, name, name_len);     // 4-wave workgroups, two per CU
    case 32: return launch_winox_cfg<2, 2, 2, 2>(p, stream, name, name_len);  // the half-height tile whatever the grid
    case 52: return launch_winox_cfg<2, 2, 2>(p, stream, name, name_len);     // one tile per workgroup whatever the grid (A/B of the persistent form)
    case 62: {               // the persistent form whatever the grid: small grids on 8 workgroups, so that every one walks several tiles
        using C4 = XCfg<2, 2, 2, 4>;
        const int64_t n4 = (int64_t)p.frames * ((p.Wo + C4::TWPX - 1) / C4::TWPX) * ((p.Cout + C4::BN - 1) / C4::BN) * ((p.Ho + 15) / 16);
        return launch_winox_cfg<2, 2, 2, 4, true>(p, stream, name, name_len, n4 < 2 * BSVD_CUS ? 8 : BSVD_CUS);
    }
    case 4: return launch_winox_cfg<4, 2, 1>(p, stream, name, name_len);
    case 36: return launch_winox_cfg<6, 1, 2, 2>(p, stream, name, name_len);   // F(6,3) on the half-height tile whatever the grid
#endif
    default: set_error("bsvd_conv3x3: wino_m %d is not in this build", p.wino_m); return -19;
    }
}

}  // namespace bsvd

#ifdef BSVD_WX_TL
extern "C" int bsvd_debug_wx_timeline(unsigned long long *dst, int n)     // measurement builds only; not in include/bsvd_hip.h
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bsvd::g_wx_tl), sizeof(unsigned long long) * 12 * 8 * (size_t)n);
}
#endif
