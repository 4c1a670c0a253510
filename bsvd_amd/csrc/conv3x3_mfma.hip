// conv3x3_mfma.hip -- implicit-GEMM 3x3 convolution for gfx950 (MI355X): exact-fp32 and split-fp16 (3-pass) MFMA modes.
//
//   y[f] = epilogue( act( conv3x3( gather(x[f-1], x[f], x[f+1], fold) ) + bias ) )
//
// GEMM view per frame: pixels x Cout x (9*Cin), on v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bitwise an fmaf
// chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).  The WEIGHTS are the MFMA's A operand (rows = output channels) and
// the pixels its B operand, so a lane ends up with 16 output channels of one pixel (see `chan` below and the epilogue).
//
// Workgroup = 4 waves (256 threads), output tile = (TH x 16) pixels x BN channels.  Each wave owns MT x NT 32x32 MFMA
// tiles: 64 px x 64 ch (2x2, 64 accumulators, 3 waves/SIMD) or 128 px x 64 ch (4x2, 128 accumulators, 2 waves/SIMD:
// the wide split-fp16 layers).  K is walked as (channel chunk of 16) x (9 taps):
//   * pixel operand: per channel chunk the input patch incl. the 1-pixel halo is staged ONCE in LDS and reused
//     by all 9 taps and all BN output channels; the temporal-shift gather is folded into this staging
//     load as a source select (no torch.cat copy).  The next chunk's patch is prefetched in row slices
//     during the taps of the current chunk (double-buffered LDS, one barrier per chunk).
//   * weight operand: straight from global/L2 to VGPRs in the pre-packed [k4][Cout][4] layout; no LDS round trip, no
//     per-tap barrier.
//
// FAST path (fold % 16 == 0 -- or fold 8 on the [fold8] instantiation --, 16-B aligned operands): each 16-channel chunk has a
// single temporal source, so all global reads are branch-free raw buffer loads (out-of-range lanes read 0
// = zero padding / masking for free) and the weight fragments run TWO steps ahead in a 3-deep register
// ring, the patch slices two steps ahead in a second ring.  Measured on MI355X the memory latency seen by a
// wave under this load is several thousand cycles (PMC: 23 % of wave time parked in s_waitcnt with a 1-step
// prefetch), which is what the deeper rings hide.  GENERIC path: any fold / alignment, per-element select.
//
// LDS patch image: [pixel][16 ch + 4 pad] floats (80-B pixel stride keeps ds_read_b128 16-B aligned and
// spreads consecutive pixels over the banks).  K-order trick: lane l of a 32x32x2 MFMA supplies k = l>>5;
// a 16-byte read gives a lane 4 consecutive channels and MFMA j (0..3) uses k = 8g + 4(l>>5) + j on both
// operands -- any K permutation is legal as long as A and B agree, so every operand read is 16 bytes wide.
//
// PREC = 1 (BSVD_F16X3, "split16"): the same data movement with every fp32 value carried as an fp16 pair
// v = hi + lo (hi = fp16(v), lo = fp16(v - hi)).  A 16-channel chunk of a pixel is stored as [hi x16 | lo x16] in
// the SAME 64 bytes the fp32 layout uses, weights are pre-split the same way, and each K=16 block issues three
// v_mfma_f32_32x32x16_f16 (hi*hi + lo*hi + hi*lo, fp32 accumulate; the lo*lo term is ~2^-22 relative and dropped).
// 16x the MFMA rate of the fp32 instruction for 3x the instructions; measured max-abs error vs the fp32 reference
// 2-4e-5 on bsvd_c64 (same class as the exact path; plain fp16 gives 1-3e-2).  The split epilogue passes each
// 32x32 accumulator tile through a wave-private LDS scratch so that 4 ADJACENT lanes store the 4 x 8 channels of one
// pixel (128 contiguous bytes) as 16-byte vectors (hi, lo); the exact-fp32 epilogue stores from registers.
//
// Reference ops replaced: see include/bsvd_hip.h (bsvd_conv3x3).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "bsvd_internal.h"

// tuning knobs (compile-time; the defaults are the measured best)
#ifndef BSVD_TUNE_ALIGN
#define BSVD_TUNE_ALIGN 1      // 1: 256-B aligned LDS patch row pitch (conflict-free A reads; +0.5 % in interleaved A/B)
#endif
#ifndef BSVD_TUNE_FAT_OCC
#define BSVD_TUNE_FAT_OCC 2        // waves/SIMD the 128-accumulator tiles are compiled for
#endif
#ifndef BSVD_TUNE_NARROW_OCC
#define BSVD_TUNE_NARROW_OCC 3     // waves/SIMD the 128-px x 32-ch wave tile is compiled for (3: 168 VGPRs + a 12-byte spill; 2: no spill)
#endif
#ifndef BSVD_ABL
#define BSVD_ABL 0             // TIMING-ONLY ablations of the prefetching K loop (results are wrong): 1 no chunk barrier, 2 no patch slices, 4 no weight loads, 8 half the weight loads (lo := hi), 16 all weight loads from two L1-resident slabs, 32 split epilogue without its stores, 64 split epilogue without the conversion, 128 split epilogue storing lane-contiguous 2-KB runs, 256 pixel fragments read from LDS once per chunk instead of once per tap
#endif
#ifndef BSVD_TUNE_ZSKIP
#define BSVD_TUNE_ZSKIP 1      // 128-accumulator split tiles leave the all-zero temporal-shift chunks of a clip's first / last frame out of the K loop
#endif
#ifndef BSVD_TUNE_SKIP_DEAD
#define BSVD_TUNE_SKIP_DEAD 1  // 128-accumulator (LITE) tiles: waves entirely below the image issue no fragment reads / MFMAs
#endif
#ifndef BSVD_TUNE_APFL
#define BSVD_TUNE_APFL 1       // prefetch of the next tap's fragments into the SAME registers (see LITE): bit 0 the 128-accumulator tiles (r03: 20.11 -> 19.56 ms per C1 clip), bit 1 the narrow 64-channel tile (5.05 -> 5.10), bit 2 the exit tile (=), bit 3 the stride-2 tiles (2.38 -> 2.42: third wave per SIMD lost)
#endif
#ifndef BSVD_TUNE_APF
#define BSVD_TUNE_APF 1        // split DBUF tiles that request the NEXT tap's pixel fragments in the middle of the current tap: 0 none, 1 the exit tile (NT == 1), 2 all, 3 fat tiles
#endif

// patch-slice register ring of the K loop: loaded into slot tap % 3, stored one tap later (two taps later: -0.1 %, r02)
#define BSVD_SLICE_D 1
#define S_OLD0 s2
#define S_OLD1 s0
#define S_OLD2 s1

namespace bsvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MT x NT = 32x32 MFMA tiles per wave (a wave covers 2*MT rows x 16 cols of pixels and 32*NT channels),
// WM x WN = waves per workgroup along pixels / channels, RING = depth of the weight register ring (2: one step
// ahead, 3: two steps ahead).
template <int MT_, int NT_, int WM_, int WN_, int STRIDE_, int RING_ = 3, bool DBUF_ = true>
struct ConvCfg {
    static constexpr bool DBUF = DBUF_;   // double-buffered LDS patch (next chunk prefetched during the taps)
    static constexpr int MT = MT_, NT = NT_, WM = WM_, WN = WN_, STRIDE = STRIDE_, RING = RING_;
    static constexpr int TH = 2 * MT * WM, TW = 16;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(RING == 2 || RING == 3, "");
    static constexpr int BN = WN * NT * 32;
    static constexpr int PH = (TH - 1) * STRIDE + 3;
    static constexpr int PW = (TW - 1) * STRIDE + 3;
    // LDS patch layout.  Padded (the original): [row][column][16 ch + 4 pad] floats -- the 80-B pixel pitch spreads the 16
    // columns of a row over all sixteen 16-B slots of a 256-B bank row, but the two rows of a ds_read_b128 lane group only
    // complement each other when the row pitch is a multiple of 256 B, and that pitch did not fit the 256-px tile three
    // times into 160 KiB (r02 PMC: 46-49 % of the LDS cycles of the 64-channel, exit and stride-2 tiles were bank conflicts).
    // Quad-planar (QPL): [row][channel quad q = 0..3][column] x 16 B, row pitch a multiple of 256 B.  A lane group reads ONE
    // quad of 8 pixels of row r (columns 0-3, 12-15 + kx) and 8 of row r+1 (columns 4-11 + kx): sixteen consecutive 16-B slots,
    // conflict-free for every tap; a tap's kx and the hi/lo quad are immediate offsets.  The staging writes are conflict-free
    // too: 8 adjacent lanes = 2 columns x 4 quads, and the 288-B (stride 2: 272-B) plane pitch puts the quads 32 B (16 B)
    // apart modulo the 128-B write bank row.  64 B per pixel instead of 80: 4/5 of the padded layout's LDS.  Stride 2: a tap
    // reads every other column, so even and odd columns get separate planes ([row][quad][parity][column / 2]).
    // quad-planar LDS patch ([row][16-B channel quad][column], row pitch a multiple of 256 B), per tile family (r03 / r05 records):
    //   the 256-px x 64-ch tile <2,2,4,1,1> and the exit tile <2,1,4,1,1>: yes (bank conflicts 0.46 / 0.49 -> 0.06 / 0.00 of the LDS cycles, times =)
    //   the 128-px x 32-ch wave tile <4,1,2,2,1> of the 64-channel layers: yes (8 pixel-fragment reads per 12 MFMAs need conflict-free reads: 6.42 -> 5.94 ms)
    //   the stride-2 tile (even / odd columns in separate planes): no (conflicts 0.47 -> 0.15, and 2.295 -> 2.354 ms: r05a_stride2_qpl_pmc.txt)
    //   the 128-channel stride-1 tiles: no (already conflict-free in the padded layout at a 1536-B pitch; 20.55 -> 20.70 ms)
    static constexpr bool QPL = STRIDE != 2 && ((MT == 2 && WM == 4) || NT == 1);
    static constexpr int PS = QPL ? 16 : 20;           // floats per patch pixel
    static constexpr int PLANE = STRIDE == 2 ? ((PW + 1) / 2) * 4 : PW * 4;      // QPL: floats per (quad[, parity]) plane of a row
    static constexpr int NP = PH * PW;                 // patch pixels
    static constexpr int NQ = NP * 4;                  // float4 items per patch chunk
    // LDS row pitch of the patch.  Stride 1: rounded up to a multiple of 256 B so that the two pixel rows one
    // ds_read_b128 lane group touches land on complementary 16-B slots (conflict-free A reads; PMC showed 48 % of the
    // LDS cycles were bank conflicts with the packed 1440-B pitch).  Not for the 256-px x 64-ch tile at 3 workgroups
    // per CU, whose double-buffered patch would no longer fit three times into 160 KiB.
    static constexpr bool ALIGN_ROWS = QPL || (BSVD_TUNE_ALIGN && STRIDE == 1 && !(MT == 2 && WM == 4));
    static constexpr int ROWP = ALIGN_ROWS ? ((PW * PS * 4 + 255) / 256 * 256) / 4 : PW * PS;   // floats
    // float offset of the 16-B channel quad `quad` of patch pixel (prow, pcol)
    __host__ __device__ static constexpr int lds_off(int prow, int pcol, int quad)
    {
        if (!QPL) return prow * ROWP + pcol * PS + quad * 4;
        if (STRIDE == 2) return prow * ROWP + (quad * 2 + (pcol & 1)) * PLANE + (pcol >> 1) * 4;
        return prow * ROWP + quad * PLANE + pcol * 4;
    }
    // ... and of tap column kx / operand half g (0: quads 0-1 = fp32 channels 0-7 or the hi halves, 1: quads 2-3) relative to it
    __host__ __device__ static constexpr int tap_off(int kx, int g) { return lds_off(0, kx, 2 * g); }
    static constexpr int PATCH_FLOATS = PH * ROWP;
    static constexpr int LDS_BYTES = (DBUF ? 2 : 1) * PATCH_FLOATS * 4 < 4 * 32 * 36 * 4 ? 4 * 32 * 36 * 4
                                                                                          : (DBUF ? 2 : 1) * PATCH_FLOATS * 4;
    // generic path: next patch spread over the 9 taps
    static constexpr int Q_PER_STEP = (NQ + 8) / 9;
    static constexpr int QG = (Q_PER_STEP + 255) / 256;          // items per thread per tap
    // fast path: next patch in row slices; a thread owns one (column, channel-quad) of R rows per pass
    static constexpr int ROW_ITEMS = PW * 4;                     // float4 items per patch row
    static_assert(ROW_ITEMS <= 256, "a patch row must fit one pass of the workgroup");
    static constexpr int R = 256 / ROW_ITEMS;                    // rows covered by one pass
    static constexpr int P = (PH + 8 * R - 1) / (8 * R);         // passes per slice so that <= 8 slices
    static constexpr int ROWS_PER_SLICE = R * P;
    static constexpr int NSLICE = (PH + ROWS_PER_SLICE - 1) / ROWS_PER_SLICE;
    static_assert(NSLICE <= 7, "slices are loaded at taps 0..6 and stored two taps later (2..8)");
    // workgroups per CU the LDS footprint admits (160 KiB) -> register budget for __launch_bounds__
    static constexpr int OCC_LDS = LDS_BYTES > 80 * 1024 ? 1 : (LDS_BYTES > 53 * 1024 ? 2 : 3);
    static constexpr int OCC = (MT * NT >= 8) ? BSVD_TUNE_FAT_OCC : (MT == 4 && NT == 1 && OCC_LDS > BSVD_TUNE_NARROW_OCC) ? BSVD_TUNE_NARROW_OCC : OCC_LDS;   // 128 accumulator registers (AGPRs) + <= 128 VGPRs: two waves per SIMD
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
    const int py = pix / C::PW, px = pix - py * C::PW;
    *reinterpret_cast<f32x4 *>(patch + C::lds_off(py, px, q)) = v;
}

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act >= BSVD_ACT_RELU) v = fmaxf(v, 0.f);
    if (act == BSVD_ACT_RELU6) v = fminf(v, 6.f);
    return v;
}

// ------------------------------------------------------------------------------------------------------
// fast-path helpers: raw buffer loads (lanes with an out-of-range offset read 0)
#define BSVD_OOB 0x7fffffffu

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *ptr, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ptr), 0, bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// (cache policy of the activation loads: nt / sc1 nt are slower -- nt also drops the halo rows from the L2; r03)
__device__ __forceinline__ f32x4 buf_load4_act(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

struct ChunkSrc {          // wave-uniform description of one 16-channel chunk's source
    __amdgpu_buffer_rsrc_t rs;
    unsigned ps4;          // pixel stride in bytes
    unsigned soff;         // byte offset of the chunk's first channel inside a pixel
    // fold == 8 (64-channel temporal-fusion layers of the c32-sized networks): chunk 0 is MIXED -- its channels 0..7 come
    // from the next frame (rs/ps4/soff above), channels 8..15 from the previous frame (the second source below); each
    // staging lane carries one 16-byte piece and picks its source by the 8-channel half the piece belongs to
    bool mixed;
    __amdgpu_buffer_rsrc_t rs2;
    unsigned ps4_2, soff2;
    bool compact1, compact2;   // the source is a compact [H][W][8-channel] slice (split16: [hi x8 | lo x8]), not a full frame
};

// ------------------------------------------------------------------------------------------------------
#ifndef BSVD_TUNE_FAT_MIN_WGS
#define BSVD_TUNE_FAT_MIN_WGS 800   // smallest grid (in 256-px x 128-ch workgroups) that takes the fat split tile (r02: 1020-workgroup launches -- 256-ch layers of a 1080p frame, upc1 of a 540x960 frame -- are 1-3 % faster fat; 510-540 are not)
#endif
#ifndef BSVD_TUNE_S2F32_OCC
#define BSVD_TUNE_S2F32_OCC 2      // waves/SIMD the exact-fp32 stride-2 kernel is compiled for (3 = 168 VGPRs + a 20-B spill: 2.3 % slower)
#endif
// Refill of the split-fp16 stride-2 tile's single LDS patch buffer: a REGISTER DOUBLE BUFFER -- chunk cb + 1 is requested at the boundary before
// chunk cb and waits in 36 VGPRs (flattened items: all 256 lanes carry a piece); a boundary is barrier + ds_write + barrier: 2.29 ms per clip for
// the four stride-2 launches.  Measured and removed (knob BSVD_TUNE_S2_PAIR 0 / 1 / 3, all bit-identical): load -> store at every
// boundary 2.36-2.46; both 16-channel chunks of a 128-byte line fetched at once, the odd one held in registers 2.61 (and requested one chunk period
// ahead: 2.49) -- these halve the memory-side line fetches (2.04x -> 1.1x of the input) and are SLOWER: the re-fetched half lines come from the
// Infinity Cache, the tile lives on occupancy and on not waiting at its boundaries.
constexpr int S2_HOLD_OCC = 2;     // waves/SIMD the register-holding stride-2 tile is compiled for
#ifndef BSVD_TUNE_FOLD8_OCC
#define BSVD_TUNE_FOLD8_OCC 3      // waves/SIMD of the split-fp16 fold-8 instantiation (c32-sized networks): 3 = 168 VGPRs + a 20-byte spill in the
                                   // chunk loop's preheader (outside the taps), 2 = no spill
#endif
template <class C, int PREC, bool MIXF = false>
constexpr int occ_of()
{
    if (MIXF && PREC == 1 && C::OCC > BSVD_TUNE_FOLD8_OCC) return BSVD_TUNE_FOLD8_OCC;
    // exact-fp32 stride 2 (single patch buffer): the refill holds the whole 17x33 patch in registers (72 VGPRs) -> 2 waves/SIMD
    if (C::STRIDE == 2 && PREC == 0 && C::OCC > BSVD_TUNE_S2F32_OCC) return BSVD_TUNE_S2F32_OCC;
    // split-fp16 single-buffer tile with the odd chunk of every 128-byte line held in registers (72 VGPRs): 2 waves/SIMD
    if (!C::DBUF && PREC == 1 && C::OCC > S2_HOLD_OCC) return S2_HOLD_OCC;
    return C::OCC;
}

#ifdef BSVD_TIMELINE
// Measurement build (tools/timeline.py): wave 0 of every workgroup stamps s_memrealtime (100 MHz) at entry, after the
// prologue barrier, after the K loop and at exit, plus its hardware id (XCD / SE / CU / SIMD slot).
#define BSVD_TIMELINE_SLOTS (1 << 16)
__device__ unsigned long long g_timeline[BSVD_TIMELINE_SLOTS][8];
__device__ __forceinline__ void tl_stamp(int slot, int k)
{
    if (threadIdx.x == 0 && slot < BSVD_TIMELINE_SLOTS) {
        g_timeline[slot][k] = __builtin_amdgcn_s_memrealtime();
        if (k == 0) g_timeline[slot][4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20 /* XCC_ID */) << 32) |
                                          (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_ID, 32 bits */);
    }
}
#define TL(k) tl_stamp(blockIdx.x, k)
#else
#define TL(k)
#endif
#if defined(BSVD_TIMELINE) && BSVD_TIMELINE == 2
// epilogue phase split (shader cycles of wave 0, summed over the items): slot 5 staging writes + wait, 6 scratch reads + wait,
// 7 bias/activation/split + store issue.  The stamps serialise the phases: read the proportions, not the total.
#define TLP_BEGIN() unsigned long long tlp_t = __builtin_amdgcn_s_memtime()
#define TLP_MARK(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tlp_acc[k] += n_ - tlp_t; tlp_t = n_; }
#define TLP_WAIT_MARK(k, a, b) { asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(b) : "memory"); TLP_MARK(k) }
#define TLP_FLUSH() if (threadIdx.x == 0 && blockIdx.x < BSVD_TIMELINE_SLOTS) { g_timeline[blockIdx.x][5] = tlp_acc[0]; g_timeline[blockIdx.x][6] = tlp_acc[1]; g_timeline[blockIdx.x][7] = tlp_acc[2]; }
#else
#define TLP_BEGIN()
#define TLP_MARK(k)
#define TLP_WAIT_MARK(k, a, b)
#define TLP_FLUSH()
#endif
// HEADF (fused network entry, BsvdConvArgs.head_w_packed): p.x is the caller's planar fp32 input; the tile computes the
// first conv's output (head_cin -> Cin channels, act, zero outside the image) on its own 18 x 18 patch with MFMAs
// (K = 9 taps x 4 channels, padded to 48) straight into the LDS patch buffers, a pair of 16-channel chunks at a time, and
// never reads an NHWC input tensor.  See head_pair below.
// PREF (fused 64-channel pair, BsvdConvArgs.pre_w_packed): p.x is the input of a FIRST 3 x 3 conv (pre_cin -> Cin channels, act, zero
// outside the image); the tile computes that conv's output on its own 18 x 18 patch -- full K = 9 pre_cin, the input patch staged
// chunk by chunk through a third LDS buffer -- a pair of 16-channel chunks at a time, straight into the two patch buffers of the main
// conv, and the Cin-channel tensor between the two convs never exists in HBM.  See pre_pair below and DESIGN.md 4.1e.
template <class C, bool FAST, int PREC, bool MIXF = false, bool HEADF = false, bool PREF = false>
__global__ __launch_bounds__(256, (PREF ? 2 : occ_of<C, PREC, MIXF>())) void conv3x3_kernel(const ConvParams p)
{
    constexpr bool FRONT = HEADF || PREF;      // the main conv's input chunks are produced inside the tile, not loaded
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const patch_buf = smem;                              // 2 x PATCH_FLOATS

    TL(0);
    if constexpr (PREC == 1) fp16_saturate_on();      // MODE.FP16_OVFL (bsvd_internal.h): every fp16 conversion of the split mode saturates
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
    if (p.flip) lid = nblk - 1 - lid;     // reverse walk: start with the tiles the producer wrote last (still in the Infinity Cache)
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

    // 16-channel chunks of K.  ZSKIP (128-accumulator split tiles): a temporal-shift group whose source frame does not exist -- the
    // `next` group of a clip's / stream's last frame, the `prev` group of its first one (bsvd_arch.py:94,104 feeds zeros there) --
    // is all zeros, so its chunks are left out of the K loop: chunk_src / load_b below take LIVE chunk / step indices and map
    // them (i -> i + zs_a, then + zs_c from zs_b on).  Adding exact zero products leaves an fp32 accumulator bit for bit as it
    // was, so the output is identical; a 10-frame clip saves 2/10 x 1/8 of the MFMAs of its 16 temporal-fusion layers.
    constexpr bool ZSKIP = BSVD_TUNE_ZSKIP && FAST && PREC == 1 && !MIXF && !FRONT && C::DBUF && C::RING == 3 && C::STRIDE == 1 &&
                           (BSVD_TUNE_APFL & 1) && C::MT * C::NT >= 8;
    int ncb = p.Cin >> 4;
    [[maybe_unused]] int zs_a = 0, zs_b = 0, zs_c = 0;
    if constexpr (ZSKIP) {
        const int f16 = p.fold >> 4;
        if (f16 > 0 && (p.fold & 15) == 0) {
            zs_b = f16;
            if (s.next == nullptr) zs_a = f16;
            if (s.prev == nullptr) zs_c = f16;
            zs_b -= zs_a;                       // in live indices: the prev group starts after the live next-group chunks
            ncb -= zs_a + zs_c;
        }
    }

    // A fragments: per-lane offset into the LDS patch (floats)
    const int a_lane = C::lds_off((2 * C::MT * wm + (li >> 4)) * C::STRIDE, (li & 15) * C::STRIDE, lh);
    // The weights are the MFMA's A operand (rows = output channels) and the pixels its B operand (columns), so a lane ends
    // up holding 16 output channels of ONE pixel: D row of register r is (r&3) + 8*(r>>2) + 4*lh.  Row i is fed with
    // channel chan(i) such that registers 0..7 / 8..15 of a lane are 8 consecutive channels each (groups 2h + lh):
    // the exact-fp32 epilogue then stores 16-byte pieces straight from registers, and the split one stages a tile into its
    // transposition scratch with four ds_write_b128 per lane.
    const int rrow = (li & 3) + 4 * (li >> 3);                   // register index that holds row li (in lane half (li>>2)&1)
    const int chan = 8 * (2 * (rrow >> 3) + ((li >> 2) & 1)) + (rrow & 7);
    const int nb0 = n0 + wn * (C::NT * 32) + chan;               // output channel whose weights this lane loads for nt = 0 (+32 per nt)

    f32x16 acc[C::MT][C::NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    if constexpr (!PREF) zero_acc();      // (fused pair: after the first conv -- 64 registers of zeros would ride through it otherwise)

    // pixel fragments of tap (ky, KX): immediate offsets from the lane's base in both layouts
    // (kx is a literal in the fast path's unrolled taps and folds into the instruction's offset field; the generic path passes
    //  it at run time -- a three-way branch on kx there was miscompiled by hipcc 7.2: one arm lost an address register)
    auto a_ptr = [&](const float *pc, int ky, int kx, int mt, int g) {
        return pc + a_lane + ky * C::ROWP + (2 * mt * C::STRIDE) * C::ROWP + C::tap_off(kx, g);
    };
    auto load_a = [&](const float *pc, int ky, int kx, f32x4 (&a)[C::MT][2]) {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                a[mt][g] = *reinterpret_cast<const f32x4 *>(a_ptr(pc, ky, kx, mt, g));
    };
    auto mfma32 = [&](const f32x4 (&a)[C::MT][2], const f32x4 (&b)[C::NT][2]) {
        if constexpr (PREC == 1) {
            // split16: a[mt][0] = 8 hi halves, a[mt][1] = 8 lo halves of this lane's k-slots; same for b[nt][.]
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, a[mt][0]), al = __builtin_bit_cast(f16x8, a[mt][1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, b[nt][0]), bl = __builtin_bit_cast(f16x8, b[nt][1]);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[mt][nt], 0, 0, 0);
                }
            return;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nt][g][j], a[mt][g][j], acc[mt][nt], 0, 0, 0);
    };

    if constexpr (FAST) {
        // ================================================================================ FAST path
        const unsigned hw = (unsigned)p.H * (unsigned)p.W;
        const __amdgpu_buffer_rsrc_t rs_cur = make_rsrc(s.cur, hw * p.Cin * 4u);
        const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(s.prev ? s.prev : s.cur, s.prev ? hw * s.prev_ps * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rs_next = make_rsrc(s.next ? s.next : s.cur, s.next ? hw * s.next_ps * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.Cin * 9u * p.Cout * 4u);
        constexpr bool MIX = MIXF;     // separate instantiation (fold == 8): the mixed-chunk staging costs the plain one 6 %
        auto chunk_src = [&](int cb) {
            ChunkSrc c;
            c.mixed = false;
            if constexpr (ZSKIP) cb = cb + zs_a + (cb >= zs_b ? zs_c : 0);
            const int c0 = cb * 16;
            if constexpr (MIX) {
                if (p.fold == 8 && cb == 0) {
                    c.mixed = true;
                    c.compact1 = s.next_ps == 8; c.compact2 = s.prev_ps == 8;
                    c.rs = rs_next; c.ps4 = s.next_ps * 4u;
                    c.rs2 = rs_prev; c.ps4_2 = s.prev_ps * 4u;
                    // fp32: channel c of a source sits at float offset co + (c - first channel of the slice); split16: the
                    // pieces of a full frame keep their place inside chunk 0 (the channel offset co is not a byte offset)
                    c.soff = PREC == 1 ? 0u : (unsigned)s.next_co * 4u;
                    c.soff2 = PREC == 1 ? 0u : (unsigned)s.prev_co * 4u;
                    return c;
                }
            }
            if (c0 < p.fold)          { c.rs = rs_next; c.ps4 = s.next_ps * 4u; c.soff = (s.next_co + c0) * 4u; }
            else if (c0 < 2 * p.fold) { c.rs = rs_prev; c.ps4 = s.prev_ps * 4u; c.soff = (s.prev_co + c0 - p.fold) * 4u; }
            else                      { c.rs = rs_cur;  c.ps4 = p.Cin * 4u;     c.soff = c0 * 4u; }
            return c;
        };

        // weights: lane's byte offset inside a (chunk, tap) slab; the OOB sentinel masks columns >= Cout
        const unsigned slab_bytes = 64u * p.Cout;                // 4 k4 rows x Cout x 16 B
        const unsigned g_bytes = 32u * p.Cout;                   // k4 -> k4 + 2
        unsigned vb[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
            vb[nt] = nb0 + 32 * nt < p.Cout ? (unsigned)(lh * p.Cout + nb0 + 32 * nt) * 16u : BSVD_OOB;
        auto load_b = [&](int step, f32x4 (&b)[C::NT][2]) {
            if constexpr (ZSKIP) step = step + 9 * zs_a + (step >= 9 * zs_b ? 9 * zs_c : 0);
            const unsigned so = (BSVD_ABL & 16) ? (unsigned)(step & 1) * slab_bytes : (unsigned)step * slab_bytes;   // 16: timing only, L1-resident weights
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) {
                b[nt][0] = buf_load4(rs_w, vb[nt], so);
                if constexpr (BSVD_ABL & 8) b[nt][1] = b[nt][0];      // timing only: half the weight stream, realistic operands
                else b[nt][1] = buf_load4(rs_w, vb[nt], so + g_bytes);
            }
        };

        // patch slices: thread = (row r3 inside a pass, column, channel quad)
        const int r3 = tid / C::ROW_ITEMS, rem = tid - r3 * C::ROW_ITEMS;
        const int pcol = rem >> 2, pq = rem & 3;
        const bool t_ok = r3 < C::R;
        const int gx = ix0 + pcol;
        const bool x_ok = t_ok && gx >= 0 && gx < p.W;
        const int lds_item = C::lds_off(r3, pcol, pq);                        // floats, pass row 0
        auto slice_load = [&](const ChunkSrc &c, int row0, f32x4 (&v)[C::P]) {
#pragma unroll
            for (int i = 0; i < C::P; ++i) {
                const int prow = row0 + i * C::R + r3;
                const int gy = iy0 + prow;
                const bool ok = x_ok && prow < C::PH && gy >= 0 && gy < p.H;
                if constexpr (MIX) {
                    if (c.mixed) {
                        const unsigned pix = (unsigned)(gy * p.W + gx);
                        f32x4 a, b;
                        if constexpr (PREC == 1) {       // pieces: 0 = hi 0..7, 1 = hi 8..15, 2 = lo 0..7, 3 = lo 8..15
                            const bool second = pq & 1;
                            const unsigned o1 = c.compact1 ? (pq >> 1) * 16u : pq * 16u, o2 = c.compact2 ? (pq >> 1) * 16u : pq * 16u;
                            a = buf_load4(c.rs, ok && !second ? pix * c.ps4 + o1 : BSVD_OOB, c.soff);
                            b = buf_load4(c.rs2, ok && second ? pix * c.ps4_2 + o2 : BSVD_OOB, c.soff2);
                            v[i] = second ? b : a;
                        } else {                         // quads: 0,1 = channels 0..7 (next), 2,3 = channels 8..15 (previous)
                            const bool second = pq >> 1;
                            a = buf_load4(c.rs, ok && !second ? pix * c.ps4 + pq * 16u : BSVD_OOB, c.soff);
                            b = buf_load4(c.rs2, ok && second ? pix * c.ps4_2 + (pq - 2) * 16u : BSVD_OOB, c.soff2);
                            v[i] = second ? b : a;
                        }
                        continue;
                    }
                }
                const unsigned voff = ok ? (unsigned)(gy * p.W + gx) * c.ps4 + pq * 16u : BSVD_OOB;
                v[i] = buf_load4_act(c.rs, voff, c.soff);
            }
        };
        auto slice_store = [&](float *pb, int row0, const f32x4 (&v)[C::P]) {
#pragma unroll
            for (int i = 0; i < C::P; ++i) {
                const int prow = row0 + i * C::R + r3;
                if (t_ok && prow < C::PH)
                    *reinterpret_cast<f32x4 *>(pb + lds_item + (row0 + i * C::R) * C::ROWP) = v[i];
            }
        };

        // ---- prologue: weights of steps 0 and 1 in flight, chunk 0 patch -> LDS
        f32x4 b0[C::NT][2], b1[C::NT][2], b2[C::NT][2];
        if constexpr (!PREF) {          // (fused pair: the main conv's first slabs are requested at the end of every pre_pair instead)
            load_b(0, b0);
            if constexpr (C::RING == 3) load_b(1, b1);
        }
        if constexpr ((BSVD_ABL & 4) != 0) load_b(2, b2);     // (timing only: the ring keeps these three slabs for the whole tile)
        // whole patch in flight at once, then published: one HBM latency per tile instead of one per slice
        auto fill_patch = [&](const ChunkSrc &c, float *pb) {
            // G slices in flight at once, then published.  Everything at once where the registers are there (prologue of
            // the double-buffered tiles, exact-fp32 stride 2 at 2 waves/SIMD)
            // (the fold-8 instantiation issues TWO masked loads per item of its mixed chunk: half the slices per round trip, or the
            //  prologue spills 20 bytes of scratch)
            constexpr int G = (C::DBUF || PREC == 0) ? (MIX ? (C::NSLICE + 1) / 2 : C::NSLICE) : 1;
#pragma unroll
            for (int g0 = 0; g0 < C::NSLICE; g0 += G) {
                f32x4 v[G][C::P];
#pragma unroll
                for (int sl = 0; sl < G; ++sl)
                    if (g0 + sl < C::NSLICE) slice_load(c, (g0 + sl) * C::ROWS_PER_SLICE, v[sl]);
#pragma unroll
                for (int sl = 0; sl < G; ++sl)
                    if (g0 + sl < C::NSLICE) slice_store(pb, (g0 + sl) * C::ROWS_PER_SLICE, v[sl]);
            }
        };
        // Single-buffer split tile (stride 2): register double buffer -- chunk cb + 1 in flight during chunk cb (see S2_HOLD_OCC above).
        constexpr bool REGPF = !C::DBUF && PREC == 1;
        // items: the patch flattened to (row, column, 16-byte quad) items, 256 per pass -- the row-slice map above keeps only 132 of 256 lanes busy
        // on the 33-pixel rows of this tile, which would double the registers the held chunk costs
        constexpr int PNITEM = C::PH * C::ROW_ITEMS, PNI = (PNITEM + 255) / 256;
        [[maybe_unused]] f32x4 hold[REGPF ? PNI : 1];
        auto pair_item = [&](int i, unsigned &voff, int &lds_off, bool &in_patch) {
            const int e = tid + 256 * i;
            const int prow = e / C::ROW_ITEMS, rem = e - prow * C::ROW_ITEMS;
            const int pc = rem >> 2, pq4 = rem & 3;
            const int gy = iy0 + prow, gxx = ix0 + pc;
            in_patch = e < PNITEM;
            const bool ok = in_patch && gy >= 0 && gy < p.H && gxx >= 0 && gxx < p.W;
            voff = ok ? (unsigned)(gy * p.W + gxx) * ((unsigned)p.Cin * 4u) + pq4 * 16u : BSVD_OOB;
            lds_off = C::lds_off(prow, pc, pq4);
        };
        auto publish_hold = [&](float *pb) {
#pragma unroll
            for (int i = 0; i < PNI; ++i) {
                unsigned voff; int lo; bool inp;
                pair_item(i, voff, lo, inp);
                if constexpr (REGPF) if (inp) *reinterpret_cast<f32x4 *>(pb + lo) = hold[i];
            }
        };
        // the flattened item map: all 256 lanes carry a 16-byte piece per load instruction (9 items = the whole 17 x 33 patch)
        constexpr bool FLAT = REGPF;
        auto prefetch_hold = [&](int cbn) {            // REGPF: request chunk cbn into registers; it is published at the next boundary
#pragma unroll
            for (int i = 0; i < PNI; ++i) {
                unsigned voff; int lo; bool inp;
                pair_item(i, voff, lo, inp);
                if constexpr (REGPF) hold[i] = buf_load4(cbn < ncb ? rs_cur : make_rsrc(s.cur, 0u), voff, (unsigned)cbn * 64u);
            }
        };
        auto fill_flat = [&](const ChunkSrc &c, float *pb) {
            constexpr int GI = 9;
#pragma unroll
            for (int g0 = 0; g0 < PNI; g0 += GI) {
                f32x4 v[GI];
#pragma unroll
                for (int j = 0; j < GI; ++j)
                    if (g0 + j < PNI) {
                        const int e = tid + 256 * (g0 + j);
                        const int prow = e / C::ROW_ITEMS, rem2 = e - prow * C::ROW_ITEMS;
                        const int gy = iy0 + prow, gxx = ix0 + (rem2 >> 2);
                        const bool ok = e < PNITEM && gy >= 0 && gy < p.H && gxx >= 0 && gxx < p.W;
                        v[j] = buf_load4(c.rs, ok ? (unsigned)(gy * p.W + gxx) * c.ps4 + (rem2 & 3) * 16u : BSVD_OOB, c.soff);
                    }
#pragma unroll
                for (int j = 0; j < GI; ++j)
                    if (g0 + j < PNI) {
                        const int e = tid + 256 * (g0 + j);
                        const int prow = e / C::ROW_ITEMS, rem2 = e - prow * C::ROW_ITEMS;
                        if (e < PNITEM) *reinterpret_cast<f32x4 *>(pb + C::lds_off(prow, rem2 >> 2, rem2 & 3)) = v[j];
                    }
            }
        };
        // ---- fused network entry: the raw input patch (TH+4) x (TW+4) x head_cin, split to fp16 pairs, 16 B per pixel
        //      [hi c0..3 | lo c0..3], behind the two patch buffers; the first conv's output is then computed per chunk pair
        constexpr int RPH = C::TH + 4, RPW = C::TW + 4;
        [[maybe_unused]] unsigned char *const rawp = reinterpret_cast<unsigned char *>(smem) + C::LDS_BYTES;
        [[maybe_unused]] auto head_pair = [&](int pair) {
            static_assert(!HEADF || (C::QPL && C::STRIDE == 1 && C::DBUF && PREC == 1), "fused entry: quad-planar stride-1 split tile");
            constexpr int NPIX = C::PH * C::PW, NRT = (NPIX + 31) / 32;
            const float *hb = p.head_bias + pair * 32 + 8 * lh;      // this lane's two groups of 8 channels: +0 (chunk 2*pair), +16 (chunk 2*pair+1)
            const f32x4 bia[2][2] = {{*reinterpret_cast<const f32x4 *>(hb), *reinterpret_cast<const f32x4 *>(hb + 4)},
                                     {*reinterpret_cast<const f32x4 *>(hb + 16), *reinterpret_cast<const f32x4 *>(hb + 20)}};
            const f32x4 *wbase = reinterpret_cast<const f32x4 *>(p.head_w) + ((int64_t)pair * 3 * 64 + lane) * 2;
            const float vlo = p.act >= BSVD_ACT_RELU ? 0.f : -65504.f, vhi = p.act == BSVD_ACT_RELU6 ? 6.f : 65504.f;
            for (int rt = wid; rt < NRT; rt += 4) {                 // 32-pixel row tiles of the patch, dealt to the 4 waves
                const int m = rt * 32 + li;
                const int mm = m < NPIX ? m : NPIX - 1;
                const int py = mm / C::PW, px = mm - py * C::PW;
                const unsigned char *rp = rawp + (py * RPW + px) * 16;
                f32x16 hacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
                for (int sidx = 0; sidx < 3; ++sidx) {
                    // this lane's 8 k-slots of the step: taps ta, ta + 1 (4 channels each); taps >= 9 carry zero weights
                    int ta = 4 * sidx + 2 * lh, tb = ta + 1;
                    ta = ta < 9 ? ta : 0; tb = tb < 9 ? tb : 0;
                    const int ya = ta / 3, yb = tb / 3;
                    const u32x4 r1 = *reinterpret_cast<const u32x4 *>(rp + (ya * RPW + (ta - 3 * ya)) * 16);
                    const u32x4 r2 = *reinterpret_cast<const u32x4 *>(rp + (yb * RPW + (tb - 3 * yb)) * 16);
                    const f16x8 xh = __builtin_bit_cast(f16x8, u32x4{r1[0], r1[1], r2[0], r2[1]});
                    const f16x8 xl = __builtin_bit_cast(f16x8, u32x4{r1[2], r1[3], r2[2], r2[3]});
                    const f16x8 wh = __builtin_bit_cast(f16x8, wbase[sidx * 128]), wl = __builtin_bit_cast(f16x8, wbase[sidx * 128 + 1]);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hacc, 0, 0, 0);
                }
                const int gy = iy0 + py, gx = ix0 + px;
                const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;      // outside the image: the main conv's zero padding
#pragma unroll
                for (int h = 0; h < 2; ++h) {                      // h: chunk 2*pair + h = patch buffer h
                    f16x8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = hacc[8 * h + j] + bia[h][j >> 2][j & 3];
                        v = inside ? __builtin_amdgcn_fmed3f(v, vlo, vhi) : 0.f;
                        hi[j] = (_Float16)v;
                        lo[j] = lo_keep((_Float16)(v - (float)hi[j]));
                    }
                    float *dst = patch_buf + h * C::PATCH_FLOATS + C::lds_off(py, px, lh);
                    if (m < NPIX) {
                        *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, hi);
                        *reinterpret_cast<f32x4 *>(dst + 2 * C::PLANE) = __builtin_bit_cast(f32x4, lo);
                    }
                }
            }
        };
        // ---- fused first conv of a 64-channel pair (PREF).  Stage buffer behind the two patch buffers: the first conv's input patch
        //      (TH + 4) x (TW + 4) pixels of ONE 16-channel chunk, quad-planar like the patch buffers (row pitch a multiple of 256 B:
        //      conflict-free fragment reads).  Chunk ci + 1 waits in registers (`phold`) while chunk ci is consumed, like the stride-2
        //      tile's register double buffer; every input chunk is fetched once per channel pair (twice per tile for 64 channels: the
        //      second pass comes from L2).
        //      Arithmetic per output of the first conv = the stand-alone kernel's: the same three MFMAs per (chunk, tap) in the same
        //      order on an fp32 accumulator, bias, activation, split -- fused == unfused bit for bit (tests/test_gpu_pair.py).
        constexpr int SPH = C::TH + 4, SPW = C::TW + 4;
        constexpr int S_ROWP = ((SPW * 64 + 255) / 256 * 256) / 4, S_PLANE = SPW * 4;       // floats
        constexpr int SNITEM = SPH * SPW * 4, SNI = (SNITEM + 255) / 256;
        [[maybe_unused]] float *const stg = smem + C::LDS_BYTES / 4;
        [[maybe_unused]] f32x4 phold[PREF ? SNI : 1];
        [[maybe_unused]] const int pre_nci = p.pre_cin >> 4;
        [[maybe_unused]] auto pre_item = [&](int i, unsigned &voff, int &loff, bool &in_patch) {
            const int e = tid + 256 * i;
            const int prow = e / (SPW * 4), rem2 = e - prow * (SPW * 4);
            const int pc = rem2 >> 2, q4 = rem2 & 3;
            const int gy = iy0 - 1 + prow, gxx = ix0 - 1 + pc;
            in_patch = e < SNITEM;
            const bool ok = in_patch && gy >= 0 && gy < p.H && gxx >= 0 && gxx < p.W;
            voff = ok ? (unsigned)(gy * p.W + gxx) * ((unsigned)p.pre_cin * 4u) + q4 * 16u : BSVD_OOB;
            loff = prow * S_ROWP + q4 * S_PLANE + pc * 4;
        };
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_pre = make_rsrc(s.cur, PREF ? hw * (unsigned)p.pre_cin * 4u : 0u);
        [[maybe_unused]] auto pre_request = [&](int ci, bool live) {       // chunk ci of the input patch -> phold (zero-size descriptor: nothing)
            const __amdgpu_buffer_rsrc_t r = live ? rs_pre : make_rsrc(s.cur, 0u);
#pragma unroll
            for (int i = 0; i < SNI; ++i) {
                unsigned voff; int lo; bool inp;
                pre_item(i, voff, lo, inp);
                if constexpr (PREF) phold[i] = buf_load4_act(r, voff, (unsigned)ci * 64u);
            }
        };
        [[maybe_unused]] auto pre_publish = [&]() {
#pragma unroll
            for (int i = 0; i < SNI; ++i) {
                unsigned voff; int lo; bool inp;
                pre_item(i, voff, lo, inp);
                if constexpr (PREF) if (inp) *reinterpret_cast<f32x4 *>(stg + lo) = phold[i];
            }
        };
        // The first conv for ALL Cin (<= 64) channels of the second conv's K in ONE pass over the staged input (every input chunk fetched
        // once per tile): the accumulators of both 32-channel pairs (2 x NJ MFMA tiles per wave) live in registers; pair 0 is written to
        // the patch buffers right away, pair 1 waits in its registers while the main conv consumes pair 0 (its accumulators are the
        // only other large register block then) and is written at the pair boundary.  (First version of the round: one pass per pair --
        // the input staged twice, 3.77 GB fetched per launch for a 1.33 GB input, PMC r05b; this form: r05c.)
        constexpr int PNPIX = C::PH * C::PW, PNRT = (PNPIX + 31) / 32, PNJ = (PNRT + 3) / 4;     // 324 pixels, 11 row tiles, <= 3 per wave
        [[maybe_unused]] f32x16 pacc[PREF ? 2 : 1][PREF ? PNJ : 1];
        // this wave's row tiles wid, wid + 4, ...: lane pixel -> offset in the stage buffer / the patch buffers.  (A wave's row tile
        // beyond the patch -- wave 3's third -- is computed like the others on a clamped pixel and not stored: wave-uniform branches
        // around the MFMAs cost the whole function its register allocation, and the other three waves have a third tile anyway.)
        [[maybe_unused]] auto pre_geom = [&](int j, int &a_s, int &d_p, bool &px_in, bool &px_st) {
            const int rt = wid + 4 * j;
            const int m = rt * 32 + li;
            const int mm = m < PNPIX ? m : PNPIX - 1;
            const int py = mm / C::PW, px = mm - py * C::PW;
            a_s = py * S_ROWP + lh * S_PLANE + px * 4;
            d_p = C::lds_off(py, px, lh);
            px_st = rt < PNRT && m < PNPIX;
            const int gy = iy0 + py, gx = ix0 + px;
            px_in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;      // outside the image: the main conv's zero padding
        };
        [[maybe_unused]] auto pre_all = [&]() {
            static_assert(!PREF || (C::QPL && C::STRIDE == 1 && C::DBUF && PREC == 1 && C::NT == 1), "fused pair: quad-planar stride-1 split tile, 32-channel wave tiles");
            int a_s[PNJ];
#pragma unroll
            for (int j = 0; j < PNJ; ++j) { int d; bool b1_, b2_; pre_geom(j, a_s[j], d, b1_, b2_); }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int j = 0; j < PNJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pacc[n][j][r] = 0.f;
            // weights of the first conv: the standard split pack [chunk][tap][hi, lo][h][Cmid][8]; this lane's channel of each 32-channel
            // pair (a second pair beyond Cin: the out-of-range sentinel = zero weights, its tiles are never stored)
            const __amdgpu_buffer_rsrc_t rs_wp = make_rsrc(p.pre_w, (unsigned)p.pre_cin * 9u * (unsigned)p.Cin * 4u);
            const unsigned pslab = 64u * (unsigned)p.Cin, pg = 32u * (unsigned)p.Cin;
            unsigned vbp[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) vbp[n] = n * 32 < p.Cin ? (unsigned)(lh * p.Cin + n * 32 + chan) * 16u : BSVD_OOB;
            const int pnsteps = pre_nci * 9;
            auto load_wp = [&](int st, f32x4 (&w)[2][2]) {
                const unsigned so = (unsigned)(st < pnsteps ? st : pnsteps - 1) * pslab;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    w[n][0] = buf_load4(rs_wp, vbp[n], so);
                    w[n][1] = buf_load4(rs_wp, vbp[n], so + pg);
                }
            };
            f32x4 w0[2][2], w1[2][2], w2[2][2];
            load_wp(0, w0);
            load_wp(1, w1);
            int pst = 0;
            for (int ci = 0; ci < pre_nci; ++ci) {
                // the stage buffer is free (the barrier behind the previous chunk's taps)
                pre_publish();
                pre_request(ci + 1, ci + 1 < pre_nci);       // the next chunk into the registers (behind the last one: nothing)
                __syncthreads();
#define BSVD_PRE_TAP(T, WCUR, WFILL)                                                                                   \
                {                                                                                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                    load_wp(pst + 2, WFILL);                                                                           \
                    f32x4 xh[PNJ], xl[PNJ];                                                                            \
                    _Pragma("unroll") for (int j = 0; j < PNJ; ++j) {                                                  \
                        const float *ap = stg + a_s[j] + ((T) / 3) * S_ROWP + ((T) % 3) * 4;                           \
                        xh[j] = *reinterpret_cast<const f32x4 *>(ap);                                                  \
                        xl[j] = *reinterpret_cast<const f32x4 *>(ap + 2 * S_PLANE);                                    \
                    }                                                                                                  \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                    \
                        const f16x8 wh = __builtin_bit_cast(f16x8, WCUR[n][0]), wl = __builtin_bit_cast(f16x8, WCUR[n][1]); \
                        _Pragma("unroll") for (int j = 0; j < PNJ; ++j) {                                              \
                            const f16x8 ah = __builtin_bit_cast(f16x8, xh[j]), al = __builtin_bit_cast(f16x8, xl[j]);  \
                            pacc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, pacc[n][j], 0, 0, 0);          \
                            pacc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, pacc[n][j], 0, 0, 0);          \
                            pacc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, pacc[n][j], 0, 0, 0);          \
                        }                                                                                              \
                    }                                                                                                  \
                    ++pst;                                                                                             \
                }
                BSVD_PRE_TAP(0, w0, w2)
                BSVD_PRE_TAP(1, w1, w0)
                BSVD_PRE_TAP(2, w2, w1)
                BSVD_PRE_TAP(3, w0, w2)
                BSVD_PRE_TAP(4, w1, w0)
                BSVD_PRE_TAP(5, w2, w1)
                BSVD_PRE_TAP(6, w0, w2)
                BSVD_PRE_TAP(7, w1, w0)
                BSVD_PRE_TAP(8, w2, w1)
#undef BSVD_PRE_TAP
                __syncthreads();
            }
        };
        // pair PAIR of the first conv's output -> the two patch buffers (chunk 2 PAIR + h = buffer h): bias, activation, zero outside the
        // image, split.  Both buffers are free (the barrier that ended the main conv's previous chunk).
        [[maybe_unused]] auto pre_store = [&](auto pair_c, int bstep) {
            constexpr int PAIR = decltype(pair_c)::value;
            // the main conv's weight ring does not ride through the first conv (24 registers its accumulators need): the slabs of the
            // next two steps are requested here, in front of the conversion that covers their latency
            load_b(bstep, b0);
            load_b(bstep + 1, b1);
            const float *hb = p.pre_bias + PAIR * 32 + 8 * lh;
            const f32x4 bia[2][2] = {{*reinterpret_cast<const f32x4 *>(hb), *reinterpret_cast<const f32x4 *>(hb + 4)},
                                     {*reinterpret_cast<const f32x4 *>(hb + 16), *reinterpret_cast<const f32x4 *>(hb + 20)}};
            const float vlo = p.pre_act >= BSVD_ACT_RELU ? 0.f : -65504.f, vhi = p.pre_act == BSVD_ACT_RELU6 ? 6.f : 65504.f;
#pragma unroll
            for (int j = 0; j < PNJ; ++j) {
                int a_s, d_p; bool px_in, px_st;
                pre_geom(j, a_s, d_p, px_in, px_st);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f16x8 hi, lo;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float v = pacc[PAIR][j][8 * h + k] + bia[h][k >> 2][k & 3];
                        v = px_in ? __builtin_amdgcn_fmed3f(v, vlo, vhi) : 0.f;
                        hi[k] = (_Float16)v;
                        lo[k] = lo_keep((_Float16)__builtin_fmaf((float)hi[k], -1.0f, v));
                    }
                    float *dst = patch_buf + h * C::PATCH_FLOATS + d_p;
                    if (px_st) {
                        *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, hi);
                        *reinterpret_cast<f32x4 *>(dst + 2 * C::PLANE) = __builtin_bit_cast(f32x4, lo);
                    }
                }
            }
        };
        if constexpr (PREF) {
            pre_request(0, true);          // chunk 0 of the first conv's input; everything else happens inside pre_all
        }
        else if constexpr (HEADF) {
            const float *xin = p.x + (int64_t)f * p.x_fs;
            const int64_t plane = (int64_t)p.H * p.W;
            for (int e = tid; e < RPH * RPW; e += 256) {
                const int ry = e / RPW, rx = e - ry * RPW;
                const int gy = iy0 - 1 + ry, gx = ix0 - 1 + rx;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                _Float16 h4[4], l4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = (ok && c < p.head_cin) ? xin[c * plane + (int64_t)gy * p.W + gx] : 0.f;
                    v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);      // fp16 range guard like every split store: saturate, never an inf / NaN pair
                    h4[c] = (_Float16)v;
                    l4[c] = lo_keep((_Float16)(v - (float)h4[c]));
                }
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                u32x4 o;
                o[0] = __builtin_bit_cast(unsigned, f16x2{h4[0], h4[1]}); o[1] = __builtin_bit_cast(unsigned, f16x2{h4[2], h4[3]});
                o[2] = __builtin_bit_cast(unsigned, f16x2{l4[0], l4[1]}); o[3] = __builtin_bit_cast(unsigned, f16x2{l4[2], l4[3]});
                *reinterpret_cast<u32x4 *>(rawp + e * 16) = o;
            }
        }
        else if constexpr (FLAT) fill_flat(chunk_src(0), patch_buf);
        else fill_patch(chunk_src(0), patch_buf);
        if constexpr (REGPF) prefetch_hold(1);
        __syncthreads();
        TL(1);

        [[maybe_unused]] auto refill_single = [&](int cb, const ChunkSrc &cn) {      // single LDS buffer: after the chunk's barrier
            if (cb + 1 < ncb) {        // single buffer: everybody is done reading it -> refill, publish
                if constexpr (REGPF) {
                    publish_hold(patch_buf);
                    prefetch_hold(cb + 2);
                } else if constexpr (FLAT) {
                    fill_flat(cn, patch_buf);
                } else {
                    fill_patch(cn, patch_buf);
                }
            }
            __syncthreads();
        };
        const int nsteps = ncb * 9;
        int step = 0;
        // APF (A-operand prefetch): the plain loop below gives every tap one load phase (8 ds_read_b128 of the pixel fragments +
        // the weight / slice requests) and then its 24 MFMAs; a wave's LDS latency is covered only while the OTHER wave of the
        // SIMD happens to be in its MFMA phase (timeline: the K loop runs at 63 % of the two-wave MFMA issue rate).  Here the
        // pixel fragments of tap k+1 are requested in the middle of tap k: the `lo` halves are dead after the first of the three
        // passes (their registers take the next tap's `lo`), the `hi` halves are double-buffered (+16 VGPRs), so a wave enters a
        // tap with its operands already in registers.  Same MFMAs in the same order on the same accumulators: bit-identical.
        // Measured r02 (interleaved A/B, ms per clip): fat 128-accumulator tile 20.05 -> 20.31 (SLOWER: its two waves per SIMD already
        // alternate load and MFMA phases, and the scheduling fences cost more than the exposed latency), 64-channel tile 6.33 =,
        // the 32-channel exit tile 0.63 -> 0.58.  Default: the exit tile only.
        constexpr bool APF = PREC == 1 && (C::DBUF || ((BSVD_TUNE_APFL & 8) && C::STRIDE == 2)) && C::RING == 3 &&
                             (((BSVD_TUNE_APFL & 8) && C::STRIDE == 2) || BSVD_TUNE_APF == 2 || (BSVD_TUNE_APF == 1 && C::NT == 1) || (BSVD_TUNE_APF == 3 && C::MT * C::NT >= 8) ||
                              ((BSVD_TUNE_APFL & 1) && C::MT * C::NT >= 8));
        // LITE: no second register set at all -- `lo` of tap k+1 is requested after pass 1 of tap k (as above), `hi` of tap k+1
        // after pass 3 of tap k into the registers its MFMAs have just read; pass 1 of tap k+1 (8 MFMAs) covers that latency.
        constexpr bool LITE = ((BSVD_TUNE_APFL & 1) && C::MT * C::NT >= 8) || ((BSVD_TUNE_APFL & 2) && C::MT == 4 && C::NT == 1) || ((BSVD_TUNE_APFL & 8) && C::STRIDE == 2) ||
                              ((BSVD_TUNE_APFL & 4) && C::MT == 2 && C::NT == 1);
        if constexpr (APF) {
            auto load_hi = [&](const float *pc, int ky, int kx, f32x4 (&h)[C::MT]) {
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) h[mt] = *reinterpret_cast<const f32x4 *>(a_ptr(pc, ky, kx, mt, 0));
            };
            auto load_lo = [&](const float *pc, int ky, int kx, f32x4 (&l)[C::MT]) {
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) l[mt] = *reinterpret_cast<const f32x4 *>(a_ptr(pc, ky, kx, mt, 1));
            };
            auto pass = [&](const f32x4 (&av)[C::MT], const f32x4 (&bv)[C::NT][2], int bpart) {
                // pixel-tile major (channel-tile major -- the weight operand held across consecutive MFMAs: 19.52 -> 19.62 ms, r03)
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[nt][bpart]),
                                                                             __builtin_bit_cast(f16x8, av[mt]), acc[mt][nt], 0, 0, 0);
            };
            // A wave whose 2*MT output rows all lie below the image (135-row layers: the lower half of the last 16-row tile) runs
            // its own loop: it stages its share of every chunk and meets the chunk barriers, but reads no fragments and issues
            // no MFMAs.  (Guarding the reads / MFMAs of the common loop with a wave-uniform branch instead cost every wave its
            // cross-tap schedule: 19.49 -> 19.83 ms per C1 clip.)
            // (never with the fused entry: its live loop has the head_pair barriers the staging-only loop lacks)
            const bool wlive = (BSVD_TUNE_SKIP_DEAD && LITE && !FRONT) ? oy0 + 2 * C::MT * wm < p.Ho : true;
            if (!wlive) {
                for (int cb = 0; cb < ncb; ++cb) {
                    ChunkSrc cn = chunk_src(cb + 1 < ncb ? cb + 1 : cb);
                    if (cb + 1 >= ncb) cn.rs = make_rsrc(s.cur, 0u);
                    fill_patch(cn, patch_buf + ((cb + 1) & 1) * C::PATCH_FLOATS);
                    __syncthreads();
                }
            } else {
            if constexpr (PREF) { pre_all(); zero_acc(); }       // the whole first conv, once per tile, in front of the main conv's K loop
            for (int cb = 0; cb < ncb; ++cb) {
                if constexpr (FRONT) {
                    // both patch buffers are free here (the barrier that ended chunk cb - 1): fill them with chunks cb, cb + 1
                    if ((cb & 1) == 0) {
                        if constexpr (HEADF) head_pair(cb >> 1);
                        else if (cb == 0) pre_store(std::integral_constant<int, 0>{}, step);
                        else pre_store(std::integral_constant<int, 1>{}, step);
                        __syncthreads();
                    }
                }
                const float *pcur = patch_buf + (C::DBUF ? (cb & 1) * C::PATCH_FLOATS : 0);
                [[maybe_unused]] float *pnext = patch_buf + (C::DBUF ? ((cb + 1) & 1) * C::PATCH_FLOATS : 0);
                ChunkSrc cn = chunk_src(cb + 1 < ncb ? cb + 1 : cb);
                if (cb + 1 >= ncb) cn.rs = make_rsrc(s.cur, 0u);
                [[maybe_unused]] f32x4 s0[C::P], s1[C::P], s2[C::P];
#pragma unroll
                for (int i = 0; i < C::P; ++i) s1[i] = s2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 hiA[C::MT], hiB[C::MT], loA[C::MT], loB[C::MT];
                load_hi(pcur, 0, 0, hiA);
                load_lo(pcur, 0, 0, loA);
#define BSVD_APF_TAP(T, HC, LC, HN, LN, BCUR, BFILL, SNEW, SOLD)                                                           \
                {                                                                                                        \
                    if constexpr (!(BSVD_ABL & 4)) load_b(step + 2 < nsteps ? step + 2 : nsteps - 1, BFILL);             \
                    if constexpr (C::DBUF && !FRONT && !(BSVD_ABL & 2)) slice_load(cn, (T) * C::ROWS_PER_SLICE, SNEW);    /* rows >= PH: zeros */ \
                    __builtin_amdgcn_sched_barrier(0);                                                                   \
                    pass(LC, BCUR, 0);                                                      /* hi(w) x lo(x) */           \
                    __builtin_amdgcn_sched_barrier(0);                                                                   \
                    if constexpr ((T) < 8 && !(BSVD_ABL & 256)) {                                                        \
                        load_lo(pcur, ((T) + 1) / 3, ((T) + 1) % 3, LN);                                                 \
                        if constexpr (!LITE) load_hi(pcur, ((T) + 1) / 3, ((T) + 1) % 3, HN);                            \
                    }                                                                                                    \
                    __builtin_amdgcn_sched_barrier(0);                                      /* reads stay HERE: 16 MFMAs of cover */ \
                    pass(HC, BCUR, 1);                                                      /* lo(w) x hi(x) */           \
                    pass(HC, BCUR, 0);                                                      /* hi(w) x hi(x) */           \
                    if constexpr (LITE && (T) < 8 && !(BSVD_ABL & 256)) {                                                \
                        __builtin_amdgcn_sched_barrier(0);                                                               \
                        load_hi(pcur, ((T) + 1) / 3, ((T) + 1) % 3, HN);                                                 \
                        __builtin_amdgcn_sched_barrier(0);                                                               \
                    }                                                                                                    \
                    if constexpr (C::DBUF && !FRONT && !(BSVD_ABL & 2))                                                  \
                        if ((T) >= BSVD_SLICE_D && (T) <= C::NSLICE - 1 + BSVD_SLICE_D) slice_store(pnext, ((T) - BSVD_SLICE_D) * C::ROWS_PER_SLICE, SOLD); \
                    ++step;                                                                                              \
                }
                if constexpr (LITE) {
                    BSVD_APF_TAP(0, hiA, loA, hiA, loA, b0, b2, s0, S_OLD0)
                    BSVD_APF_TAP(1, hiA, loA, hiA, loA, b1, b0, s1, S_OLD1)
                    BSVD_APF_TAP(2, hiA, loA, hiA, loA, b2, b1, s2, S_OLD2)
                    BSVD_APF_TAP(3, hiA, loA, hiA, loA, b0, b2, s0, S_OLD0)
                    BSVD_APF_TAP(4, hiA, loA, hiA, loA, b1, b0, s1, S_OLD1)
                    BSVD_APF_TAP(5, hiA, loA, hiA, loA, b2, b1, s2, S_OLD2)
                    BSVD_APF_TAP(6, hiA, loA, hiA, loA, b0, b2, s0, S_OLD0)
                    BSVD_APF_TAP(7, hiA, loA, hiA, loA, b1, b0, s1, S_OLD1)
                    BSVD_APF_TAP(8, hiA, loA, hiA, loA, b2, b1, s2, S_OLD2)
                } else {
                BSVD_APF_TAP(0, hiA, loA, hiB, loB, b0, b2, s0, S_OLD0)
                BSVD_APF_TAP(1, hiB, loB, hiA, loA, b1, b0, s1, S_OLD1)
                BSVD_APF_TAP(2, hiA, loA, hiB, loB, b2, b1, s2, S_OLD2)
                BSVD_APF_TAP(3, hiB, loB, hiA, loA, b0, b2, s0, S_OLD0)
                BSVD_APF_TAP(4, hiA, loA, hiB, loB, b1, b0, s1, S_OLD1)
                BSVD_APF_TAP(5, hiB, loB, hiA, loA, b2, b1, s2, S_OLD2)
                BSVD_APF_TAP(6, hiA, loA, hiB, loB, b0, b2, s0, S_OLD0)
                BSVD_APF_TAP(7, hiB, loB, hiA, loA, b1, b0, s1, S_OLD1)
                BSVD_APF_TAP(8, hiA, loA, hiB, loB, b2, b1, s2, S_OLD2)
                }
#undef BSVD_APF_TAP
                if constexpr (!(BSVD_ABL & 1)) __syncthreads();
                if constexpr (!C::DBUF) refill_single(cb, cn);
            }
            }
        } else {
        if constexpr (PREF) { pre_all(); zero_acc(); }
        for (int cb = 0; cb < ncb; ++cb) {
            if constexpr (FRONT) {
                // both patch buffers are free here (the barrier that ended chunk cb - 1): fill them with chunks cb, cb + 1
                if ((cb & 1) == 0) {
                    if constexpr (HEADF) head_pair(cb >> 1);
                    else if (cb == 0) pre_store(std::integral_constant<int, 0>{}, step);
                    else pre_store(std::integral_constant<int, 1>{}, step);
                    __syncthreads();
                }
            }
            const float *pcur = patch_buf + (C::DBUF ? (cb & 1) * C::PATCH_FLOATS : 0);
            float *pnext = patch_buf + (C::DBUF ? ((cb + 1) & 1) * C::PATCH_FLOATS : 0);
            // next chunk's source; after the last chunk a zero-size descriptor turns the slice loads into no-ops
            ChunkSrc cn = chunk_src(cb + 1 < ncb ? cb + 1 : cb);
            if (cb + 1 >= ncb) cn.rs = make_rsrc(s.cur, 0u);
            f32x4 s0[C::P], s1[C::P], s2[C::P];
#pragma unroll
            for (int i = 0; i < C::P; ++i) s1[i] = s2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                // three taps per trip so that both register rings rotate statically:
                //   weights  b0 -> b1 -> b2 (filled RING-1 steps ahead),  slices  s0 -> s1 -> s2 (stored two steps later)
#define BSVD_TAP(KX, BCUR, BFILL, SNEW, SOLD)                                                                  \
                {                                                                                              \
                    const int tap = ky * 3 + (KX);                                                             \
                    f32x4 a[C::MT][2];                                                                         \
                    load_a(pcur, ky, (KX), a);                                                                 \
                    load_b(step + C::RING - 1 < nsteps ? step + C::RING - 1 : nsteps - 1, BFILL);              \
                    if constexpr (C::DBUF && !FRONT) slice_load(cn, tap * C::ROWS_PER_SLICE, SNEW);   /* rows >= PH: zeros */ \
                    mfma32(a, BCUR);                                                                           \
                    if constexpr (C::DBUF && !FRONT)                                                           \
                        if (tap >= BSVD_SLICE_D && tap <= C::NSLICE - 1 + BSVD_SLICE_D) slice_store(pnext, (tap - BSVD_SLICE_D) * C::ROWS_PER_SLICE, SOLD); \
                    ++step;                                                                                    \
                }
                if constexpr (C::RING == 3) {
                    BSVD_TAP(0, b0, b2, s0, S_OLD0)
                    BSVD_TAP(1, b1, b0, s1, S_OLD1)
                    BSVD_TAP(2, b2, b1, s2, S_OLD2)
                } else {      // 2-deep ring: period 2 does not divide 3 taps -> alternate the roles by trip parity
                    if (((cb + ky) & 1) == 0) {      // step parity: step = 9 cb + 3 ky + kx
                        BSVD_TAP(0, b0, b1, s0, S_OLD0)
                        BSVD_TAP(1, b1, b0, s1, S_OLD1)
                        BSVD_TAP(2, b0, b1, s2, S_OLD2)
                    } else {
                        BSVD_TAP(0, b1, b0, s0, S_OLD0)
                        BSVD_TAP(1, b0, b1, s1, S_OLD1)
                        BSVD_TAP(2, b1, b0, s2, S_OLD2)
                    }
                }
#undef BSVD_TAP
            }
            __syncthreads();   // one barrier per 16-channel chunk (9 taps, 288 MFMAs per wave)
            if constexpr (!C::DBUF) refill_single(cb, cn);
        }
        }
    } else {
        // ============================================================================= GENERIC path
        const int64_t slab_stride = (int64_t)16 * p.Cout;           // floats per (chunk, tap) weight slab
        const float *wl = p.w + ((int64_t)lh * p.Cout + nb0) * 4;
        const int64_t g_off = (int64_t)8 * p.Cout;
        auto load_b = [&](int step, f32x4 (&b)[C::NT][2]) {
            const float *sl = wl + (int64_t)step * slab_stride;
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    b[nt][g] = nb0 + 32 * nt < p.Cout ? *reinterpret_cast<const f32x4 *>(sl + g * g_off + 128 * nt) : z;
                }
        };
        f32x4 bcur[C::NT][2], bnxt[C::NT][2];
        load_b(0, bcur);
        for (int e = tid; e < C::NQ; e += 256) store_patch_quad<C>(patch_buf, e, load_patch_quad<C>(p, s, 0, e, iy0, ix0));
        __syncthreads();

        for (int cb = 0; cb < ncb; ++cb) {
            const float *pcur = patch_buf + (cb & 1) * C::PATCH_FLOATS;
            float *pnext = patch_buf + ((cb + 1) & 1) * C::PATCH_FLOATS;
            const bool more_chunks = cb + 1 < ncb;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (tap < 8 || more_chunks) load_b(cb * 9 + tap + 1, bnxt);
                f32x4 preg[C::QG];
#pragma unroll
                for (int i = 0; i < C::QG; ++i) {
                    const int ep = tap * C::Q_PER_STEP + tid + i * 256;
                    const bool do_p = more_chunks && tid + i * 256 < C::Q_PER_STEP && ep < C::NQ;
                    preg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (do_p) preg[i] = load_patch_quad<C>(p, s, cb + 1, ep, iy0, ix0);
                }
                const int ky = tap / 3, kx = tap - ky * 3;
                f32x4 a[C::MT][2];
                load_a(pcur, ky, kx, a);
                mfma32(a, bcur);
#pragma unroll
                for (int i = 0; i < C::QG; ++i) {
                    const int ep = tap * C::Q_PER_STEP + tid + i * 256;
                    const bool do_p = more_chunks && tid + i * 256 < C::Q_PER_STEP && ep < C::NQ;
                    if (do_p) store_patch_quad<C>(pnext, ep, preg[i]);
                }
#pragma unroll
                for (int u = 0; u < C::NT; ++u)
#pragma unroll
                    for (int g = 0; g < 2; ++g) bcur[u][g] = bnxt[u][g];
            }
            __syncthreads();
        }
    }

    TL(2);
    if constexpr (BSVD_TUNE_SKIP_DEAD && FAST && PREC == 1 && (BSVD_TUNE_APFL & 1) && C::MT * C::NT >= 8 && C::DBUF && C::RING == 3) {
        // a wave entirely below the image (see wlive in the K loop) has nothing to store; no workgroup barrier follows
#ifndef BSVD_TIMELINE
        if (oy0 + 2 * C::MT * wm >= p.Ho) return;
#endif
    }
    // The epilogue's lane-derived values come from a FRESH lane id (v_mbcnt, opaque to the optimiser) instead of the kernel entry's
    // threadIdx: carried through the K loop they cost the 168-register tiles a 12-16 byte scratch spill (r04 kernel_resources).
    int elane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(elane));
    const int eli = elane & 31, elh = elane >> 5;
    // ---- epilogue.  Lane (eli, elh) holds pixel eli of the 2 x 16 pixel block of MFMA tile mt (row eli>>4, column eli&15)
    //      and, per (mt, nt), two groups of 8 consecutive output channels: registers 8h..8h+7 = channels 8*(2h + elh)..+7
    //      of the 32-channel tile.
    //      exact fp32 : stored straight from registers, two 16-byte pieces per group (was 16 scalar stores per tile).
    //      split16    : one pixel's 32 channels are 128 contiguous bytes and only ADJACENT lanes coalesce, so the tile is
    //                   transposed through a wave-private LDS scratch ([32 px][32 ch + 4 pad] floats, four ds_write_b128
    //                   per elane; the patch buffers are free after the last chunk's barrier) and 4 adjacent lanes finish
    //                   the 4 x 8 channels of one pixel.  (Register-direct stores were tried here too: 4x the write
    //                   transactions, 3-11 % slower on the 64-channel and stride-2 layers.)
    //      Everything read from global memory is requested ahead of its use (bias once per tile, the PixelShuffle skip
    //      operand one item ahead), and the item loop is specialised at compile time on (epilogue, activation): at 2-3
    //      waves/SIMD its VALU work is not hidden behind other waves' MFMAs.
    if constexpr (PREC == 1 && C::NT == 1) {
        // network exit in split mode: channels 0..7 of the single 32-channel tile sit in registers 0..7 of elane half 0;
        // the y_planar_ch live ones go out as planar fp32 [frames][ch][H][W] with the residual (DenBlock.none_minus,
        // bsvd_arch.py:408-414) and the callers' clamp (validation_seq_infer.py:24) fused.  32 lanes = 2 rows x 16
        // consecutive pixels: 64-byte runs per channel.
        if (p.y_planar_ch > 0) {
            if (elh == 0 && n0 == 0) {
                const int64_t plane = (int64_t)p.Ho * p.Wo;
                // the residual base of every pixel of this elane FIRST (split16: channels 0..3 are one 8-byte hi and one 8-byte lo piece): read at
                // the point of use, between stores the compiler must assume alias them, the tile's bases were C::MT x channels dependent round trips
                typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
                [[maybe_unused]] f16x4_t bh[C::MT], bl[C::MT];
                const bool split_base = p.epilogue == BSVD_EPI_RESID && p.extra_split;
                if (split_base) {
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt) {
                        const int oy = oy0 + 2 * C::MT * wm + 2 * mt + (eli >> 4), ox = ox0 + (eli & 15);
                        bh[mt] = bl[mt] = f16x4_t{0, 0, 0, 0};
                        if (oy >= p.Ho || ox >= p.Wo) continue;
                        const _Float16 *e = reinterpret_cast<const _Float16 *>(p.extra + (int64_t)f * p.extra_fs + ((int64_t)oy * p.Wo + ox) * p.extra_ps);
                        bh[mt] = *reinterpret_cast<const f16x4_t *>(e);
                        bl[mt] = *reinterpret_cast<const f16x4_t *>(e + 16);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
                    const int oy = oy0 + 2 * C::MT * wm + 2 * mt + (eli >> 4), ox = ox0 + (eli & 15);
                    if (oy >= p.Ho || ox >= p.Wo) continue;
                    const int64_t opix = (int64_t)oy * p.Wo + ox;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (n >= p.y_planar_ch) break;
                        float v = apply_act(acc[mt][0][n] + (p.bias ? p.bias[n] : 0.f), p.act);
                        if (p.epilogue == BSVD_EPI_RESID && n < p.resid_ch) {
                            float base;
                            if (split_base) {          // split16 NHWC base: channel n < 16 lives in chunk 0
                                base = (float)bh[mt][n] + (float)bl[mt][n];
                            } else {
                                base = p.extra[(int64_t)f * p.extra_fs + opix * p.extra_ps + (int64_t)n * p.extra_cs];
                            }
                            v = base - v;
                        }
                        if (p.y_clamp) v = fminf(fmaxf(v, p.y_lo), p.y_hi);
                        p.y[(int64_t)f * p.y_fs + n * plane + opix] = v;
                    }
                }
            }
            return;
        }
    }
    const int Cq = p.Cout >> 2;   // PS_ADD: channels of the shuffled output
    constexpr int NB = PREC == 1 ? 1 : 2;                    // 8-channel groups per (elane, nt) whose bias is kept
    const int q = elane & 3;
    f32x4 bq[C::NT][NB][2];
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int n8 = n0 + wn * (C::NT * 32) + nt * 32 + (PREC == 1 ? 8 * q : 8 * (2 * h + elh));
            bq[nt][h][0] = bq[nt][h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n8 < p.Cout) {
                bq[nt][h][0] = *reinterpret_cast<const f32x4 *>(p.bias + n8);
                bq[nt][h][1] = *reinterpret_cast<const f32x4 *>(p.bias + n8 + 4);
            }
        }
    float *sc = smem + wid * (32 * 36);
    auto finish = [&](auto epi_c, auto act_c, auto psf_c) {
        constexpr int EPI = decltype(epi_c)::value, ACT = decltype(act_c)::value;
        constexpr bool PSF = decltype(psf_c)::value;      // PixelShuffle items with wave-uniform sub-pixel / channel base
        struct Item { bool live; int64_t opix; int n8, c8; float *dst; const float *esrc; };
        // PLAIN / RESID: everything that depends on the elane is computed once; an item only adds compile-time multiples of
        // the (wave-uniform) row stride -- every instruction of a finishing wave waits for a gap between the co-resident
        // wave's MFMAs, so per-item 64-bit address arithmetic was a measurable part of the epilogue
        const int l_ox = PREC == 1 ? ox0 + (elane >> 2) : ox0 + (eli & 15);
        const int l_oy = oy0 + 2 * C::MT * wm + (PREC == 1 ? 0 : (eli >> 4));
        const int l_ch = n0 + wn * (C::NT * 32) + (PREC == 1 ? 8 * q : 8 * elh);
        const int64_t l_pix = (int64_t)l_oy * p.Wo + l_ox;
        const int64_t rowstride = (int64_t)p.Wo * p.Cout;
        // (BsvdConvArgs.y_f32, PLAIN split layers: plain fp32 channels for a consumer that runs the Winograd form -- no split, no clamp)
        const bool y_f32 = PREC == 1 && EPI == BSVD_EPI_PLAIN && p.y_f32 != 0;
        float *const l_base = p.y + (int64_t)f * p.y_fs + l_pix * p.Cout +
                              ((PREC == 1 && !y_f32) ? (l_ch >> 4) * 16 + ((l_ch >> 3) & 1) * 4 : l_ch);
        // PixelShuffle items of the split mode, Cq % 32 == 0 (every c64 / c32-sized network): a 32-channel MFMA tile lies inside
        // ONE sub-pixel plane, so the sub-pixel, the channel base and the row are wave-uniform and an item's address is
        // (elane part, computed once) + (scalar part).  The generic form below divides by Cq and does 64-bit multiplies per
        // elane and item: 150-230 instructions per item against 60 here -- each of them waits for a gap between the
        // co-resident wave's MFMAs (the 128->256 layer ran 12 % below the temporal-fusion layers of the same shape).
        [[maybe_unused]] int ps_sub0 = 0, ps_rem0 = 0, ps_r[C::NT] = {}, ps_sub[C::NT] = {};
        [[maybe_unused]] float *ps_ybase = nullptr;
        [[maybe_unused]] const float *ps_ebase = nullptr;
        if constexpr (PSF) {
            ps_sub0 = n0 / Cq;
            ps_rem0 = n0 - ps_sub0 * Cq;
            const int lq = (q >> 1) * 16 + (q & 1) * 4;              // this elane's 8-channel piece inside a 32-channel tile
            const int64_t lpix = (int64_t)(2 * l_oy) * (2 * p.Wo) + 2 * l_ox;
            ps_ybase = p.y + (int64_t)f * p.y_fs + lpix * Cq + lq;
            ps_ebase = p.extra ? p.extra + (int64_t)f * p.extra_fs + lpix * p.extra_ps + lq : nullptr;
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) {                     // sub-pixel plane and channel base of each 32-channel tile
                int r = ps_rem0 + wn * (C::NT * 32) + nt * 32, sub = ps_sub0;
                while (r >= Cq) { r -= Cq; ++sub; }
                ps_r[nt] = r;
                ps_sub[nt] = sub;
            }
        }
        auto item_of = [&](int i) {          // i = ((mt * NT + nt) * 2 + s), compile-time after unrolling
            const int sidx = i & 1, nt = (i >> 1) % C::NT, mt = (i >> 1) / C::NT;
            int oy, ox;
            Item t;
            t.esrc = nullptr;
            if constexpr (PSF) {
                {
                    const int off = wn * (C::NT * 32) + nt * 32;                     // wave-uniform
                    const int r = ps_r[nt], sub = ps_sub[nt];
                    const int row = 2 * mt + sidx;                                   // conv-output rows below l_oy
                    const int64_t upix = (int64_t)(2 * row + (sub >> 1)) * (2 * p.Wo) + (sub & 1);
                    t.n8 = n0 + off + 8 * q;
                    t.c8 = r + 8 * q;
                    t.live = l_oy + row < p.Ho && l_ox < p.Wo && n0 + off < p.Cout;
                    t.opix = 0;
                    t.dst = ps_ybase + upix * Cq + r;
                    t.esrc = ps_ebase ? ps_ebase + upix * p.extra_ps + r : nullptr;
                    return t;
                }
            }
            if constexpr (EPI != BSVD_EPI_PS_ADD) {
                const int row = 2 * mt + (PREC == 1 ? sidx : 0);             // rows below the elane's base row
                const int chadd = nt * 32 + (PREC == 1 ? 0 : 16 * sidx);     // channels (= floats in both layouts) above l_ch
                t.n8 = t.c8 = l_ch + chadd;
                t.live = l_oy + row < p.Ho && l_ox < p.Wo && t.n8 < p.Cout;
                t.opix = l_pix + (int64_t)row * p.Wo;
                t.dst = l_base + row * rowstride + chadd;
                return t;
            }
            if constexpr (PREC == 1) {       // s = which 16 of the tile's 32 pixels; 4 adjacent lanes share a pixel
                const int m = (elane + 64 * sidx) >> 2;
                oy = oy0 + 2 * C::MT * wm + 2 * mt + (m >> 4);
                ox = ox0 + (m & 15);
                t.n8 = n0 + wn * (C::NT * 32) + nt * 32 + 8 * q;
            } else {                         // s = h: which of the elane's two channel groups
                oy = oy0 + 2 * C::MT * wm + 2 * mt + (eli >> 4);
                ox = ox0 + (eli & 15);
                t.n8 = n0 + wn * (C::NT * 32) + nt * 32 + 8 * (2 * sidx + elh);
            }
            t.live = oy < p.Ho && ox < p.Wo && t.n8 < p.Cout;
            if constexpr (EPI == BSVD_EPI_PS_ADD) {
                const int sub = t.n8 / Cq;
                t.c8 = t.n8 - sub * Cq;                                      // first of 8 channels in the shuffled tensor
                t.opix = (int64_t)(2 * oy + (sub >> 1)) * (2 * p.Wo) + (2 * ox + (sub & 1));
                t.dst = p.y + (int64_t)f * p.y_fs + t.opix * Cq + (PREC == 1 ? (t.c8 >> 4) * 16 + ((t.c8 >> 3) & 1) * 4 : t.c8);
            } else {
                t.c8 = t.n8;
                t.opix = (int64_t)oy * p.Wo + ox;
                t.dst = nullptr;     // (not reached: the fast path above returns first)
            }
            return t;
        };
        // split16: 8 channels = half a 16-channel chunk: hi at chunk*16 + half*4 floats, lo 8 floats further
        auto coff16 = [](int c8) { return (c8 >> 4) * 16 + ((c8 >> 3) & 1) * 4; };
        const bool has_skip = EPI == BSVD_EPI_PS_ADD && p.extra != nullptr;
        auto skip_load = [&](const Item &t, f32x4 (&e)[2]) {
            e[0] = e[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(has_skip && t.live)) return;
            if constexpr (PREC == 1) {                                       // split16 skip tensor, same layout as y
                const float *ep = t.esrc ? t.esrc : p.extra + (int64_t)f * p.extra_fs + t.opix * p.extra_ps + coff16(t.c8);
                e[0] = *reinterpret_cast<const f32x4 *>(ep);
                e[1] = *reinterpret_cast<const f32x4 *>(ep + 8);
            } else {                                                         // fp32 skip tensor with generic strides
                const float *ep = p.extra + (int64_t)f * p.extra_fs + t.opix * p.extra_ps + (int64_t)t.c8 * p.extra_cs;
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j >> 2][j & 3] = ep[(int64_t)j * p.extra_cs];
            }
        };
        // RESID (DenBlock 1's last layer, bsvd_arch.py:408-414): the residual base of an item, requested one item ahead like the skip operand
        // (read at its point of use it sat between two items' stores, which the compiler must assume alias it: a dependent round trip per item)
        auto resid_load = [&](const Item &t, f32x4 (&e)[2]) {
            e[0] = e[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(t.live && t.n8 == 0)) return;
            const float *ep = p.extra + (int64_t)f * p.extra_fs + t.opix * p.extra_ps;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < p.resid_ch) e[j >> 2][j & 3] = ep[(int64_t)j * p.extra_cs];
        };
        constexpr int NITEM = C::MT * C::NT * 2;
        f32x4 ecur[2], enxt[2];
        [[maybe_unused]] unsigned long long tlp_acc[3] = {0, 0, 0};
        if constexpr (EPI == BSVD_EPI_PS_ADD) skip_load(item_of(0), ecur);
        if constexpr (EPI == BSVD_EPI_RESID) resid_load(item_of(0), ecur);
#pragma unroll
        for (int i = 0; i < NITEM; ++i) {
            const int sidx = i & 1, nt = (i >> 1) % C::NT, mt = (i >> 1) / C::NT;
            if constexpr (EPI == BSVD_EPI_PS_ADD)
                if (i + 1 < NITEM) skip_load(item_of(i + 1), enxt);
            if constexpr (EPI == BSVD_EPI_RESID)
                if (i + 1 < NITEM) resid_load(item_of(i + 1), enxt);
            float v[8];
            TLP_BEGIN();
            if constexpr (PREC == 1) {
                // (r03: requesting both items of tile k, staging tile k+1 behind those reads with no wait in between -- a wave's LDS
                //  instructions execute in order -- and only then converting moved nothing, 19.65 vs 19.67 ms: the timeline build
                //  (tools/timeline.py, -DBSVD_TIMELINE=2) puts 80 % of the epilogue into the convert + store part, 20 % into LDS.)
                auto stage = [&](int smt, int snt) {      // row = pixel eli, 8 consecutive channels per write
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float *w = sc + eli * 36 + 8 * (2 * h + elh);
                        const f32x16 &a = acc[smt][snt];
                        *reinterpret_cast<f32x4 *>(w) = f32x4{a[8 * h], a[8 * h + 1], a[8 * h + 2], a[8 * h + 3]};
                        *reinterpret_cast<f32x4 *>(w + 4) = f32x4{a[8 * h + 4], a[8 * h + 5], a[8 * h + 6], a[8 * h + 7]};
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                };
                if (sidx == 0) {
                    stage(mt, nt);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                TLP_MARK(0);
                const int m = (elane + 64 * sidx) >> 2;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(sc + m * 36 + q * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(sc + m * 36 + q * 8 + 4);
                TLP_WAIT_MARK(1, v0, v1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = v0[j] + bq[nt][0][0][j]; v[4 + j] = v1[j] + bq[nt][0][1][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc[mt][nt][8 * sidx + j] + bq[nt][sidx][j >> 2][j & 3];
            }
            const Item t = item_of(i);
            if (t.live) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (ACT == BSVD_ACT_RELU6) v[j] = __builtin_amdgcn_fmed3f(v[j], 0.f, 6.f);
                    else if constexpr (ACT == BSVD_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
                }
                if constexpr (EPI == BSVD_EPI_PS_ADD) {
                    if (has_skip) {
                        if constexpr (PREC == 1) {
                            const f16x8 eh = __builtin_bit_cast(f16x8, ecur[0]), el = __builtin_bit_cast(f16x8, ecur[1]);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)eh[j] + (float)el[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += ecur[j >> 2][j & 3];
                        }
                    }
                }
                if constexpr (EPI == BSVD_EPI_RESID) {
                    if (t.n8 == 0) {                                         // base: fp32 with generic strides
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < p.resid_ch) v[j] = ecur[j >> 2][j & 3] - v[j];
                    }
                }
                if (y_f32) {
                    float *dst = t.dst;
                    *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else if constexpr (PREC == 1) {
                    float *dst = t.dst;
                    constexpr bool bounded = (ACT == BSVD_ACT_RELU6 && EPI == BSVD_EPI_PLAIN) || !BSVD_EPI_CLAMP;
                    f16x8 hi, lo;
                    // (lo = fp16(v - hi) as inline-asm v_fma_mix{lo,hi}_f16 was tried in r03: 12 instead of 20 conversion instructions
                    //  per 8 channels, bit-identical, 64-channel tile 6.22 -> 6.18 ms per clip but the fat tile 19.74 -> 19.87 -- and an
                    //  inline-asm partial-register write is invisible to the compiler's hazard recognizer, so it was dropped.)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {        // fp16 range guard: saturate instead of inf/NaN pairs
                        const float vs = bounded ? v[j] : __builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
                        hi[j] = (_Float16)vs;
                        lo[j] = lo_keep((_Float16)__builtin_fmaf((float)hi[j], -1.0f, vs));
                    }
                    if constexpr ((BSVD_ABL & 32) != 0) {      // timing only: conversion kept alive, stores never executed
                        if (p.Cout < 0) { *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, hi); *reinterpret_cast<f32x4 *>(dst + 8) = __builtin_bit_cast(f32x4, lo); }
                    } else if constexpr ((BSVD_ABL & 128) != 0) {  // timing only: the same bytes as fully coalesced 2-KB runs per wave and item
                        float *d2 = p.y + (int64_t)f * 0 + ((int64_t)((blockIdx.x % (gridDim.x - gridDim.x / 16)) * 4 + wid) * NITEM + i) * 512 + elane * 8;   // (stays inside the tensor: edge tiles fold back)
                        *reinterpret_cast<f32x4 *>(d2) = __builtin_bit_cast(f32x4, hi); *reinterpret_cast<f32x4 *>(d2 + 4) = __builtin_bit_cast(f32x4, lo);
                    } else if constexpr ((BSVD_ABL & 64) != 0) {   // timing only: stores kept, no split conversion
                        *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4 *>(dst + 8) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                    *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, hi);
                    *reinterpret_cast<f32x4 *>(dst + 8) = __builtin_bit_cast(f32x4, lo);
                    }
                } else {
                    float *dst = t.dst;
                    *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
            if constexpr (EPI == BSVD_EPI_PS_ADD || EPI == BSVD_EPI_RESID) { ecur[0] = enxt[0]; ecur[1] = enxt[1]; }
            TLP_MARK(2);
#if !defined(BSVD_TIMELINE) || BSVD_TIMELINE != 2
            if (i == 0) TL(5);
            if (i == NITEM / 2 - 1) TL(6);
#endif
        }
        TLP_FLUSH();
    };
    using std::integral_constant;
    auto with_act = [&](auto epi_c, auto psf_c) {
        if (p.act == BSVD_ACT_RELU6) finish(epi_c, integral_constant<int, BSVD_ACT_RELU6>{}, psf_c);
        else if (p.act == BSVD_ACT_RELU) finish(epi_c, integral_constant<int, BSVD_ACT_RELU>{}, psf_c);
        else finish(epi_c, integral_constant<int, BSVD_ACT_NONE>{}, psf_c);
    };
    using std::false_type;
    using std::true_type;
    if (p.epilogue == BSVD_EPI_PLAIN) with_act(integral_constant<int, BSVD_EPI_PLAIN>{}, false_type{});
    else if (p.epilogue == BSVD_EPI_PS_ADD) {
        if constexpr (PREC == 1) {
            // UpBlock has no activation (bsvd_arch.py:263-267): the uniform-address form is instantiated for act 'none' only
            if ((Cq & 31) == 0 && (p.extra == nullptr || p.extra_cs == 1) && p.act == BSVD_ACT_NONE)
                finish(integral_constant<int, BSVD_EPI_PS_ADD>{}, integral_constant<int, BSVD_ACT_NONE>{}, true_type{});
            else with_act(integral_constant<int, BSVD_EPI_PS_ADD>{}, false_type{});
        } else {
            with_act(integral_constant<int, BSVD_EPI_PS_ADD>{}, false_type{});
        }
    } else with_act(integral_constant<int, BSVD_EPI_RESID>{}, false_type{});
    TL(3);
}

template <class C, bool FAST, int PREC, bool MIXF = false, bool HEADF = false, bool PREF = false>
static int launch_cfg(const ConvParams &pin, hipStream_t stream, char *name = nullptr, int name_len = 0)
{
    if (name) {      // dry run: report the instantiation bsvd_conv3x3 would launch (used by bench.py's per-kernel timing)
        snprintf(name, name_len, "conv3x3_kernel<%d,%d,%d,%d,%d>[%s]%s%s%s%s", C::MT, C::NT, C::WM, C::WN, C::STRIDE,
                 PREC == 1 ? "f16x3" : "f32", FAST ? (MIXF ? "[fold8]" : "") : "[generic]", pin.y_planar_ch > 0 ? "[planar out]" : "",
                 HEADF ? "[fused entry]" : "", PREF ? "[fused pair]" : "");
        return 0;
    }
    ConvParams p = pin;
    p.ntx = (p.Wo + C::TW - 1) / C::TW;
    p.nty = (p.Ho + C::TH - 1) / C::TH;
    p.nct = (p.Cout + C::BN - 1) / C::BN;
    // (A persistent variant -- one round of resident workgroups walking all tiles -- was measured 4 % slower: it loses
    //  the dispatcher's dynamic balancing and exposes every tile's prologue.)
    const int64_t nblk = (int64_t)p.frames * p.nty * p.ntx * p.nct;
    if (nblk <= 0 || nblk > 0x7fffffff) { set_error("bsvd_conv3x3: grid of %lld workgroups", (long long)nblk); return -1; }
    static std::atomic<int> granted[MAX_DEVICES];
    // + the raw input patch of the fused entry / the first conv's staged input chunk of a fused pair (20 rows x 1280 B)
    constexpr int LDS = C::LDS_BYTES + (HEADF ? (C::TH + 4) * (C::TW + 4) * 16 : 0) +
                        (PREF ? (C::TH + 4) * (((C::TW + 4) * 64 + 255) / 256 * 256) : 0);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv3x3_kernel<C, FAST, PREC, MIXF, HEADF, PREF>), LDS, granted);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_kernel<C, FAST, PREC, MIXF, HEADF, PREF>), dim3((unsigned)nblk), dim3(256), LDS, stream, p);
    return (int)hipGetLastError();
}

// the fused network entry exists for the quad-planar 64-channel tile only (static_assert in head_pair)
template <class C>
static int launch_headf(const ConvParams &p, hipStream_t stream, char *name, int name_len)
{
    if constexpr (C::QPL) return launch_cfg<C, true, 1, false, true>(p, stream, name, name_len);
    else { set_error("bsvd_conv3x3: no fused-entry kernel for this tile"); return -18; }
}

// the fused pair exists for the quad-planar 32-channel-wave tiles (the 64-channel layers' tile and the exit tile)
template <class C>
static int launch_pref(const ConvParams &p, hipStream_t stream, char *name, int name_len)
{
    if constexpr (C::QPL) return launch_cfg<C, true, 1, false, false, true>(p, stream, name, name_len);
    else { set_error("bsvd_conv3x3: no fused-pair kernel for this tile"); return -20; }
}

static bool fast_ok(const ConvParams &p, bool honour_force_generic = true)
{
    // FAST needs: 16-B aligned vector gather (vec_ok), single-source 16-channel chunks (fold % 16 == 0) and
    // 32-bit byte offsets inside one frame / the packed weights.  (ablate == 8: timing builds force GENERIC.)
    const bool fold_ok = (p.fold & 15) == 0 || (p.fold == 8 && p.Cout <= 64);     // fold 8: mixed chunk 0 in the 64-channel tile
    return p.vec_ok && fold_ok && (int64_t)p.H * p.W * p.Cin * 4 < 0x7fffffffLL &&
           (int64_t)p.Cin * 9 * p.Cout * 4 < 0x7fffffffLL && !(honour_force_generic && p.ablate == 8);
}

template <class C>
static int launch_f32(const ConvParams &p, hipStream_t stream, char *name, int name_len)
{
    if (!fast_ok(p)) return launch_cfg<C, false, 0>(p, stream, name, name_len);
    if constexpr (C::STRIDE == 1 && C::BN == 64)
        if (p.fold == 8) return launch_cfg<C, true, 0, true>(p, stream, name, name_len);
    return launch_cfg<C, true, 0>(p, stream, name, name_len);
}

int launch_conv3x3(const ConvParams &p, int stride, hipStream_t stream, char *name, int name_len)
{
    if (p.prec == 1) {
        // split16.  Wide layers: 128-px x 64-ch wave tiles (half the weight bytes per MFMA, twice the step length) at ONE
        // wave per SIMD with the full 512-register file: 430 vs 414 TFLOP/s for the 64x64 tile at 3 waves/SIMD.  (At 2
        // waves/SIMD the same tile needs > 256 registers and spills in the main loop: 10x slower.  <4,1,2,2,1,3> for
        // the 64-channel layers: no gain over <2,2,4,1,1,3>.)  Stride 2: single patch buffer -> 3 workgroups/CU
        // instead of 1 (175 -> 287 TFLOP/s).
        if (!fast_ok(p, false)) {
            // say WHICH requirement failed: the split kernel has no generic (per-element gather / 64-bit offset) variant
            if ((int64_t)p.H * p.W * p.Cin * 4 >= 0x7fffffffLL)
                set_error("bsvd_conv3x3: BSVD_F16X3 addresses one frame with 32-bit byte offsets: H*W*Cin*4 = %lld bytes >= 2 GiB "
                          "(%d x %d x %d channels); tile the frame spatially or use BSVD_F32, whose generic path has 64-bit offsets",
                          (long long)((int64_t)p.H * p.W * p.Cin * 4), p.H, p.W, p.Cin);
            else if ((int64_t)p.Cin * 9 * p.Cout * 4 >= 0x7fffffffLL)
                set_error("bsvd_conv3x3: BSVD_F16X3 packed weights of %d x %d channels exceed 2 GiB", p.Cin, p.Cout);
            else if (!p.vec_ok)
                set_error("bsvd_conv3x3: BSVD_F16X3 needs 16-byte aligned x / halo pointers and strides that are multiples of 4 elements");
            else
                set_error("bsvd_conv3x3: BSVD_F16X3 needs fold %% 16 == 0 (or fold 8 with Cout <= 64), got fold %d with Cout %d", p.fold, p.Cout);
            return -17;
        }
        if (p.head_w) {                  // fused network entry (validated by the ABI layer): the 64-channel tile with the first conv inside
            if (stride != 1 || p.fold != 0 || p.Cout > 64 || (p.Cin & 31)) { set_error("bsvd_conv3x3: fused entry needs stride 1, fold 0, Cin %% 32 == 0, Cout <= 64"); return -18; }
            return launch_headf<ConvCfg<4, 1, 2, 2, 1, 3>>(p, stream, name, name_len);
        }
        if (p.y_planar_ch > 0) {         // network exit: 256 px x 32 ch tiles, planar fp32 epilogue
            if (stride != 1 || p.fold != 0 || p.Cout > 32) { set_error("bsvd_conv3x3: planar split output needs stride 1, fold 0, Cout <= 32"); return -16; }
            if (p.pre_w && p.Cin > 64) { set_error("bsvd_conv3x3: fused pair needs Cin = 32 or 64 (two 32-channel pairs of the first conv)"); return -20; }
            if (p.pre_w) return launch_pref<ConvCfg<2, 1, 4, 1, 1, 3>>(p, stream, name, name_len);       // fused pair: out0 -> exit
            return launch_cfg<ConvCfg<2, 1, 4, 1, 1, 3>, true, 1>(p, stream, name, name_len);
        }
        if (p.pre_w) {                   // fused 64-channel pair (validated by the ABI layer): the narrow tile with the first conv inside
            if (stride != 1 || p.fold != 0 || p.Cout > 64 || (p.Cin & 31) || p.Cin > 64) { set_error("bsvd_conv3x3: fused pair needs stride 1, fold 0, Cin = 32 or 64, Cout <= 64"); return -20; }
            return launch_pref<ConvCfg<4, 1, 2, 2, 1, 3>>(p, stream, name, name_len);
        }
        if (stride == 2) {
            // (the tile's register double buffer reads every chunk from the frame itself: a stride-2 layer with a temporal shift -- none exists in the
            //  reference, DownBlock is a plain conv, bsvd_arch.py:229-255 -- is refused, never computed from the wrong frames)
            if (p.fold != 0) { set_error("bsvd_conv3x3: BSVD_F16X3 stride-2 layers are plain convs (fold must be 0, got %d)", p.fold); return -17; }
            // 8 x 16-px x 128-ch workgroup, ONE patch buffer (a double-buffered patch leaves 1-2 workgroups per CU: slower, r01 / r02), every wave
            // 128 px x 32 ch (<4,1,1,4,2>) since round 6: what this tile waits for is its weight stream -- TCP_PENDING_STALL_CYCLES = 41-51 % of
            // the kernel's time (profiles/r06f_stride2_pmc.txt), the K loop at 55 % of the MFMA pipe with three waves per SIMD
            // (r06f_stride2_timeline.txt) -- and the 32-channel wave tile pulls half the weight bytes per MFMA through the L1 (twice the pixel-fragment
            // reads from LDS): 2.32 -> 2.21 ms per C1 clip for the four launches, same bits (r06g_stride2_variants*.txt).  Refuted before: memory-side
            // re-fetches (r04), LDS bank conflicts (r05), and in round 6 the item decode at the chunk boundaries (a map with one v_add per item: +1 %).
            // (Small grids: the 64 -> 128 layer on ONE 540 x 960 frame is 1020 workgroups on 768 slots -- two rounds for 1.33 rounds of work.  The same tile with 64
            //  channels per workgroup (<2,1,2,2,2>: twice the workgroups at half the size, same bits) is SLOWER, 0.0640 -> 0.0670 ms, per-frame API 359.7 -> 357.7
            //  frames/s: a workgroup with half the MFMAs pays the same prologue and epilogue.  profiles/r06k_stride2_small_grid_*.txt)
            // (tools/kernel_resources.sh shows a 68-byte private segment for <4,1,1,4,2>: a reservation only -- the kernel's ISA contains no scratch, flat-scratch or
            //  private buffer instruction; 154 VGPRs, three workgroups per CU)
            return launch_cfg<ConvCfg<4, 1, 1, 4, 2, 3, false>, true, 1>(p, stream, name, name_len);
        }
        // The fat tiles run one workgroup per CU, so they need a grid of several rounds of 256; small launches
        // (streaming mode: one frame per launch) keep the 64x64 tiles at 3 workgroups per CU.
        const int64_t fat_wide = (int64_t)p.frames * ((p.Ho + 15) / 16) * ((p.Wo + 15) / 16) * ((p.Cout + 127) / 128);
        const int fat_min = p.fat_min_wgs > 0 ? p.fat_min_wgs : BSVD_TUNE_FAT_MIN_WGS;      // BsvdConvArgs.fat_min_wgs (0 = the measured default); no hidden state
        if (p.Cout > 64) {
            // wave tile of the 256-px x 128-ch workgroup: 128 px x 64 ch (256 px x 32 ch -- half the weight bytes per MFMA, twice the pixel-fragment
            // reads -- 19.21 -> 19.60 ms, r03); small grids (single-frame launches): 128 px x 128 ch workgroups (256 px x 64 ch: 291 -> 281 frames/s, r03)
            if (fat_wide >= fat_min) return launch_cfg<ConvCfg<4, 2, 2, 2, 1, 3>, true, 1>(p, stream, name, name_len);
            return launch_cfg<ConvCfg<2, 2, 2, 2, 1, 3>, true, 1>(p, stream, name, name_len);
        }
        if (p.fold == 8) return launch_cfg<ConvCfg<2, 2, 4, 1, 1, 3>, true, 1, true>(p, stream, name, name_len);   // c32-sized nets
        // the 64-channel layers: 128-px x 32-ch wave tiles (<4,1,2,2,1>, 2 waves per SIMD) instead of 64 px x 64 ch at 3 waves per SIMD: half the weight
        // bytes per MFMA (the 64 x 64 tile pulled 4 KB of weights per wave and tap through the L1: ~42 B/clk/CU of its 64), twice the pixel-fragment
        // reads -- a loss with the padded LDS layout (r01), a 7 % gain with the conflict-free quad-planar one (r03: 6.42 -> 5.94 ms per clip).
        // (512-px fat tiles were tried: 9.6 vs 6.8 ms)
        return launch_cfg<ConvCfg<4, 1, 2, 2, 1, 3>, true, 1>(p, stream, name, name_len);
    }
    // exact fp32.  Cout <= 64 (the 540x960-level layers of bsvd_c64): 256 px x 64 ch tiles; wider layers: 128 px x 128 ch.
    // Stride 2 always takes the 128 x 128 tile with a single patch buffer (its 17x33 input patch is what bounds LDS).
    if (stride == 1)
        return p.Cout > 64 ? launch_f32<ConvCfg<2, 2, 2, 2, 1>>(p, stream, name, name_len) : launch_f32<ConvCfg<2, 2, 4, 1, 1>>(p, stream, name, name_len);
    return launch_f32<ConvCfg<2, 2, 2, 2, 2, 3, false>>(p, stream, name, name_len);
}

}  // namespace bsvd

#ifdef BSVD_TIMELINE
extern "C" int bsvd_debug_timeline(unsigned long long *dst, int n)     // measurement builds only; not in include/bsvd_hip.h
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bsvd::g_timeline), sizeof(unsigned long long) * 8 * (size_t)n);
}
#endif
