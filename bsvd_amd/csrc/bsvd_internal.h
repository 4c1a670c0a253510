// Internal declarations shared by the translation units of libbsvd_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "bsvd_hip.h"

namespace bsvd {

// Kernel-side view of BsvdConvArgs (validated, typed, with derived tiling numbers).
struct ConvParams {
    const float *x;
    const float *halo_prev;
    const float *halo_next;
    const float *w;
    const float *bias;
    const float *extra;
    float *y;
    int64_t x_fs, extra_fs, y_fs;
    int32_t halo_prev_ps, halo_prev_co, halo_next_ps, halo_next_co;
    int32_t extra_ps, extra_cs, resid_ch;
    int32_t fold, frames, H, W, Ho, Wo, Cin, Cout;
    int32_t act, epilogue;
    int32_t ntx, nty, nct;   // tiles in x, y and output-channel tiles
    int32_t vec_ok;          // 1: every 4-channel group of the gather comes from one source, 16-B aligned
    int32_t ablate;          // only read by -DBSVD_ABLATE timing builds (tools/), always 0 in the product
    int32_t flip;            // 1: walk the tiles in reverse order (BsvdConvArgs.tile_order; see launch_cfg)
    int32_t prec;            // 0 = exact fp32, 1 = split16 (BSVD_F16X3)
    int32_t extra_split;     // planar-output layer in split mode: the residual base is a split16 NHWC tensor
    int32_t y_planar_ch;     // > 0: y is planar [frames][y_planar_ch][H][W] fp32 (split mode: written by the MFMA kernel)
    int32_t y_clamp;         // clamp the planar output to [y_lo, y_hi]
    float y_lo, y_hi;
    // fused network entry (BsvdConvArgs.head_w_packed): x is the planar fp32 input with head_cin channels
    const void *head_w;
    const float *head_bias;
    int32_t head_cin;
    // fused 64-channel pair (BsvdConvArgs.pre_w_packed): x is the FIRST conv's NHWC input with pre_cin channels, pre_w its split pack
    // (pre_cin -> Cin channels), pre_bias [Cin] fp32, pre_act its activation; w / bias / act / epilogue describe the second conv
    const void *pre_w;
    const float *pre_bias;
    int32_t pre_cin, pre_act;
    // BsvdConvArgs.x_f32 / y_f32 (split mode): the input (Winograd form only) / the NHWC output holds plain fp32 channels instead of fp16 pairs
    int32_t x_f32, y_f32;
    // BsvdConvArgs.x_v / y_v (Winograd form, split mode): the input / output tensor lives in the TRANSFORMED domain of F(wino_m % 10, 3) -- per frame
    // [row][16-channel chunk][position][quarter][group] x 16 B (conv3x3_winox.hip); v_wg = groups per image row (ceil(W / M) rounded up to 8)
    int32_t x_v, y_v, v_wg;
    // Winograd form of the wide split-fp16 layers (BsvdConvArgs.w_wino_packed): w then points at the transformed pack
    int32_t fat_min_wgs;     // BsvdConvArgs.fat_min_wgs (0 = default): smallest grid that takes the 128-accumulator split tile
    int32_t wino_m;          // 0 = direct convolution; 2 | 4 | 6 = F(wino_m, 3) along x (conv3x3_winox.hip); 12 | 14 = the all-positions-per-wave kernel (conv3x3_wino.hip)
};

// Transformed-domain layout (include/bsvd_hip.h, BsvdConvArgs.x_v / y_v): a frame is [row][tile of 8 groups][16-channel chunk] BLOCKS of
// v_block_floats(m): (m + 2) positions x 4 quarters x 8 groups x 16 B (the LDS planes of one patch row of one chunk, 4 KB for F(6,3)) + one
// 128-byte EDGE LINE [side 0 | 1][4 quarters] x 16 B: position 0 of the tile's first group resp. position m + 1 of its last group, which need a pixel of
// the neighbouring tile -- readers take these two values from the edge line (written by the patch pass behind the producer), never from the planes.
__host__ __device__ constexpr int v_block_floats(int m) { return ((m + 2) * 4 * 8 + 8) * 4; }

void set_error(const char *fmt, ...);

// Experiment knob: keep only the top BSVD_TUNE_LO_BITS mantissa bits of the `lo` half of a split16 pair (10 = all;
// -1 = lo := 0).  Probes how much of the power-limited split kernel's energy is operand toggling.
#ifndef BSVD_TUNE_LO_BITS
#define BSVD_TUNE_LO_BITS 10
#endif
__host__ __device__ inline _Float16 lo_keep(_Float16 lo)
{
#if BSVD_TUNE_LO_BITS >= 10
    return lo;
#elif BSVD_TUNE_LO_BITS < 0
    return (_Float16)0.f;
#else
    unsigned short u = __builtin_bit_cast(unsigned short, lo);
    u &= (unsigned short)~((1u << (10 - BSVD_TUNE_LO_BITS)) - 1u);
    return __builtin_bit_cast(_Float16, u);
#endif
}

// fp16 range of the split mode.  Every value is carried as hi = fp16(v), lo = fp16(v - hi); a conversion that overflows to +-inf turns
// the pair into (inf, NaN) and every product it meets into NaN.  The stores of the direct kernel clamp to +-65504 first (v_med3_f32),
// but the Winograd kernel converts TRANSFORMED values -- sums of up to 2x (F(2,3)), 3x (F(4,3)), 4.7x (F(6,3)) the activation range --
// 12 to 24 times per transform item, where one more VALU instruction per value costs 3-4 % of the layer.  gfx9's MODE.FP16_OVFL
// (bit 23: "an overflowed fp16 result is clamped to +-MAX_FP16, true infinities preserved") does it in the conversion itself:
// hi saturates at +-65504 and lo = fp16(v - hi) carries the rest, up to another 65504: below 65504 nothing changes (same bits), in
// (65504, 131008] a pair still represents v, at lo's own 11-bit precision (relative error <= 2^-13 instead of 2^-22: graceful), and
// beyond it saturates -- no instruction, no inf, no NaN from finite inputs.  Set once at kernel entry by every kernel that produces
// fp16 pairs; MFMAs (fp32 results) are not affected.  Verified on MI355X: tests/test_gpu_range.py.
#ifndef BSVD_FP16_OVFL
#define BSVD_FP16_OVFL 1
#endif
// The split STORES keep their explicit clamp to +-65504 (one v_med3_f32 per value of a layer without a bounded activation): a stored pair
// then always has |lo| <= ulp(hi) / 2, i.e. full pair precision, whatever produced it.  0 = rely on the saturating conversions alone
// (A/B knob: the epilogues do not wait for these instructions, r03 / r05 records).
#ifndef BSVD_EPI_CLAMP
#define BSVD_EPI_CLAMP 1
#endif
__device__ __forceinline__ void fp16_saturate_on()
{
#if BSVD_FP16_OVFL
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1 /* hwreg(HW_REG_MODE, 23, 1) */, 1);
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember, per device, the largest
// size already granted for one kernel (`granted` = a zero-initialised static array of MAX_DEVICES atomics owned by
// the launcher).  Thread-safe; a lost race only repeats an idempotent call.
constexpr int MAX_DEVICES = 64;
inline hipError_t ensure_dynamic_lds(const void *fn, int bytes, std::atomic<int> *granted)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) dev = 0;
    if (bytes <= granted[dev].load(std::memory_order_relaxed)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) granted[dev].store(bytes, std::memory_order_relaxed);
    return e;
}

// conv3x3_mfma.hip
// name != nullptr: dry run, only writes the kernel instantiation that would be launched
int launch_conv3x3(const ConvParams &p, int stride, hipStream_t stream, char *name = nullptr, int name_len = 0);

// conv3x3_winox.hip (one transformed position per wave; wino_m 2 | 6, 42 | 46; measurement builds: more codes)
const char *wino_unsupported(const ConvParams &p, int stride);     // nullptr = the Winograd kernel can run this layer
int launch_winox(const ConvParams &p, hipStream_t stream, char *name = nullptr, int name_len = 0);
#ifdef BSVD_MEASURE
// conv3x3_wino.hip (all positions per wave: the rejected first design, measurement builds only)
int launch_wino(const ConvParams &p, hipStream_t stream, char *name = nullptr, int name_len = 0);
#endif

// conv3x3_edge_f32.hip
int launch_head_f32(const ConvParams &p, int cin_real, hipStream_t stream);
int launch_tail_f32(const ConvParams &p, int cout_real, int do_clamp, float lo, float hi, hipStream_t stream);

}  // namespace bsvd
