/*
 * bsvd_hip.h -- C ABI of libbsvd_hip.so: the MI355X (gfx950) kernels behind the BSVD hot path.
 *
 * The reference has no FFI on this path: everything below BSVD.forward is torch ops
 * (/root/reference/Experimental_root/archs/bsvd_arch.py).  This header is the boundary the
 * MI355X engine puts in their place; each entry point names the reference op(s) it replaces.
 * The Python host (bsvd_amd/) binds it with ctypes; any other host binds it the same way
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types; no exceptions cross the ABI.
 *   - The library allocates nothing and keeps no global state besides a thread-local error string.
 *     Every buffer (activations, packed weights, halos) is owned and sized by the caller.
 *   - All work is enqueued asynchronously on the caller's HIP stream (pass it as void*, i.e. a
 *     hipStream_t; NULL = the default stream).  Calls on distinct streams may run concurrently.
 *   - Return value: 0 ok; <0 invalid argument (text via bsvd_last_error()); >0 a hipError_t.
 *   - Activations are NHWC ("channels last"): element (f, y, x, c) of a clip lives at
 *         base + f*frame_stride + (y*W + x)*C + c        (strides in ELEMENTS)
 *     with C already padded by the caller to a multiple of 16 (padded channels hold zeros).
 *   - dtype: BSVD_F32 is the exact-fp32 mode (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain); BSVD_F16X3 the
 *     split-fp16 3-pass mode (see the enum).  In BSVD_F16X3 the edge layers convert: planar fp32 in -> split16,
 *     split16 -> planar fp32 out; the packed weights of MFMA layers are split16 too (bsvd_pack_weights dtype).
 */
#ifndef BSVD_HIP_H
#define BSVD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSVD_ABI_VERSION 11

/* dtype.  BSVD_F32: exact fp32 (v_mfma_f32_32x32x2_f32).  BSVD_F16X3 ("split16"): every fp32 value v is carried as
 * an fp16 pair hi = fp16(v), lo = fp16(v - hi); a 16-channel chunk of a pixel is stored as [hi x16 | lo x16] in the
 * same 64 bytes the fp32 layout uses (so shapes, strides and halo offsets are identical), and each K = 16 block is
 * three fp16 MFMAs (hi*hi + lo*hi + hi*lo, fp32 accumulate).  fp32-class accuracy (2-4e-5 max-abs on bsvd_c64) at
 * several times the fp32-MFMA rate.  BSVD_F16 (plain fp16) is reserved and not implemented: it misses the 1e-3
 * parity bar (1-3e-2 measured). */
enum { BSVD_F32 = 0, BSVD_F16 = 1, BSVD_F16X3 = 2 };
enum { BSVD_ACT_NONE = 0, BSVD_ACT_RELU = 1, BSVD_ACT_RELU6 = 2 }; /* get_act_function, bsvd_arch.py:185-192 */
enum {
    BSVD_EPI_PLAIN = 0,   /* y = act(conv + bias)                                                     */
    BSVD_EPI_PS_ADD = 1,  /* nn.PixelShuffle(2) (+ skip add): UpBlock :265-266 + DenBlock.none_add :402 */
    BSVD_EPI_RESID = 2    /* y[:resid_ch] = extra[:resid_ch] - y[:resid_ch]: DenBlock.none_minus :408-414 */
};

/*
 * One fused layer over a whole clip (or a single frame in streaming mode):
 *
 *     y[f] = epilogue( act( conv3x3( gather(x[f-1], x[f], x[f+1], fold), w, stride, pad 1 ) + bias ) )
 *
 * Replaces, per call: nn.Conv2d 3x3 (bsvd_arch.py:31-38, 208-213, 238, 265, 295-298), the
 * torch.cat temporal-shift gather of ShiftConv.forward (:42-50), the zero tensors BiBufferConv makes
 * at stream start/end (:94, :104), nn.ReLU6/ReLU (:185-192), nn.PixelShuffle(2) (:266), the skip add
 * (:402-406) and the residual (:408-414).
 *
 * Temporal-shift gather (fold > 0): input channel c of frame f is read from
 *     frame f+1  if c <  fold            (for f == frames-1: from halo_next, zeros if NULL)
 *     frame f-1  if fold <= c < 2*fold   (for f == 0:        from halo_prev, zeros if NULL)
 *     frame f    otherwise.
 * Halo element (pixel p, channel c) is read at
 *     halo_prev[p*halo_prev_pstride + halo_prev_coff + (c - fold)],
 *     halo_next[p*halo_next_pstride + halo_next_coff + c].
 * (compact [H][W][fold] slice: pstride = fold, coff = 0; a full NHWC frame: pstride = Cin,
 *  coff = fold resp. 0.)  This is what a neighbouring frame-window shard sends over RCCL, and in
 *  streaming mode it is BiBufferConv's left_fold_2fold / input_right.
 */
typedef struct BsvdConvArgs {
    const void *x;              /* [frames][H][W][Cin]                                              */
    int64_t x_frame_stride;     /* elements between consecutive frames of x                          */
    const void *halo_prev;      /* nullable */
    const void *halo_next;      /* nullable */
    int32_t halo_prev_pstride, halo_prev_coff;
    int32_t halo_next_pstride, halo_next_coff;
    int32_t fold;               /* 0 = plain conv (no temporal shift)                                */
    const void *w_packed;       /* from bsvd_pack_weights()                                          */
    const void *bias_packed;    /* [Cout] from bsvd_pack_weights(); nullable                         */
    const void *extra;          /* PS_ADD: skip tensor laid out like y (nullable); RESID: base tensor */
    int64_t extra_frame_stride;
    int32_t extra_pstride;      /* elements between pixels of extra                                  */
    int32_t extra_cstride;      /* elements between channels of extra                                */
    int32_t resid_ch;           /* RESID: number of leading channels replaced (3 in the reference)   */
    void *y;                    /* PLAIN/RESID: [frames][Ho][Wo][Cout]; PS_ADD: [frames][2Ho][2Wo][Cout/4] */
    int64_t y_frame_stride;
    int32_t frames, H, W;       /* input spatial size; Ho = (H-1)/stride + 1                          */
    int32_t Cin, Cout;          /* padded channel counts: Cin % 16 == 0, Cout % 16 == 0 (PS_ADD: Cout % 64 == 0) */
    int32_t stride;             /* 1 or 2                                                            */
    int32_t act, epilogue, dtype;
    /* Clip entry/exit fused into the edge layers (optional; 0 = plain NHWC as described above):
     *   x_planar_ch > 0: x is the caller's planar tensor [frames][x_planar_ch][H][W] fp32 (3 or 4 channels,
     *                    BSVD.forward's reshaped input, bsvd_arch.py:494-499); needs Cin == 16, stride 1, fold 0, PLAIN.
     *   y_planar_ch > 0: y is planar [frames][y_planar_ch][H][W] fp32 (1..4 channels, the tensor torch.cat(out_seq)
     *                    returns, :552); needs Cout == 16, stride 1, fold 0, PLAIN or RESID; y_clamp != 0 additionally
     *                    clamps to [y_lo, y_hi] (the callers' torch.clamp, validation_seq_infer.py:24).
     * Weights of these two layers: the planar-INPUT layer always takes an fp32 pack (bsvd_pack_weights dtype BSVD_F32; it
     * runs on a VALU kernel and, in BSVD_F16X3, writes split16).  The planar-OUTPUT layer takes the pack of its dtype: in
     * BSVD_F16X3 it is an MFMA layer like any other (split-packed weights, split16 input). */
    int32_t x_planar_ch, y_planar_ch, y_clamp;
    float y_lo, y_hi;
    int32_t extra_split;        /* BSVD_F16X3 + y_planar_ch: the RESID base `extra` is a split16 NHWC tensor (extra_pstride
                                 * floats per pixel) instead of fp32 with generic strides                              */
    int32_t tile_order;         /* 0: workgroups walk the output tiles first frame / first row first; 1: in reverse.  Results
                                 * are identical; a layer-major host alternates it layer by layer so that every layer starts with
                                 * the part of its input the previous layer wrote LAST -- still in the 256 MiB Infinity Cache when
                                 * the clip's tensors (0.3-1.3 GB) are not (+0.3 % on the 10-frame 540x960 clip)          */
    /* Fused network entry (ABI v8, BSVD_F16X3 only): InputCvBlock's two convs (bsvd_arch.py:207-216) in ONE launch.
     * With x_planar_ch > 0 AND head_w_packed != NULL, x is the caller's planar tensor [frames][x_planar_ch][H][W] fp32 (3 or 4
     * channels) and the kernel computes  t = act(conv3x3(x, head_w) + head_bias)  (x_planar_ch -> Cin channels, zero outside
     * the image) on every tile's 18 x 18 patch itself -- on the matrix cores, K = 36 padded to 48, straight into the LDS
     * patch of the main conv -- and then  y = epilogue(act(conv3x3(t, w_packed) + bias)).  The Cin-channel tensor t never
     * exists in HBM (1.33 GB written + read per 10-frame 540x960 clip).  Needs Cin % 32 == 0, Cout <= 64, stride 1, fold 0,
     * BSVD_EPI_PLAIN; both convs share `act`.  head_w_packed: bsvd_pack_head_weights(); head_bias: [Cin] fp32. */
    const void *head_w_packed;
    const void *head_bias;
    /* Winograd form of a wide layer (ABI v9, BSVD_F16X3 only).  w_wino_packed != NULL selects the 1-D Winograd F(wino_m, 3)
     * kernel (conv3x3_winox.hip, one transformed position per wave) with the TRANSFORMED weights of bsvd_pack_weights_wino();
     * w_packed is then ignored (may be NULL).  wino_m:
     *    2 | 6    F(2,3) | F(6,3).  The kernel picks its half-height pixel tile for grids that do not fill the chip (single-frame
     *             launches); both tiles compute every output with the same instruction sequence (bit-identical).  On the 16-row
     *             tile grid a last row band with <= 8 live rows (Ho mod 16 in 1 .. 8) is walked two tiles per workgroup (F(2,3):
     *             folded tiles) or on the 8-row body (F(6,3)) -- again the same instruction sequence per output.
     *   42 | 46   the same two forms, never on the half-height tile: for launches that run beside another stream's or graph
     *             branch's kernels (idle CUs are not idle there).  Same bits as 2 | 6.
     *   Other codes (F(4,3), forced tiles, 4-wave workgroups, the persistent form, the all-positions-per-wave kernel) exist in
     *   MEASUREMENT builds of the library only (-DBSVD_MEASURE: bsvd_build_info() & BSVD_BUILD_MEASURE; tools/build_measure.sh);
     *   the product library answers them with -19.
     * Same contract and tensors as the direct form -- gather, halos, bias, activation, PLAIN / PS_ADD epilogues -- but 6 (F(2,3)) or
     * 4 (F(6,3)) instead of 9 tap-GEMMs per output pixel; results differ from the direct form in the last bits (fp32 transforms,
     * both inside the same error class: 5e-5 max-abs on bsvd_c64), so a host must use ONE form per layer in every schedule it wants
     * bit-identical (clip / stream / sharded).  Needs stride 1, fold % 16 == 0, Cout % 32 == 0, 16-byte aligned x / halo pointers
     * and strides, H*W*Cin*4 and H*W*halo_pstride*4 < 2 GiB, no planar / fused-entry / RESID options; anything else returns -19
     * with the reason -- never a silent direct launch.
     * fp16 range: the kernel converts TRANSFORMED activations (sums of up to 2x / 4.7x the inputs for F(2,3) / F(6,3)) with
     * saturating conversions (MODE.FP16_OVFL): unchanged below 65504, a pair still carries a transformed value up to 131008 (at
     * reduced precision, relative 2^-13) and saturates beyond; finite inputs never produce inf / NaN.  Weights: bsvd_pack_weights_wino saturates U = G g at +-65504; a host should keep a
     * layer whose max |w| x (largest |G| row sum: 1.5 / 15.04) leaves fp16's range on the direct form (bsvd_amd.engine does). */
    const void *w_wino_packed;
    int32_t wino_m;
    /* Tuning (ABI v9; 0 = the library's measured default, 800): the smallest grid, in 256-px x 128-channel workgroups, for which a
     * wide direct-form BSVD_F16X3 layer takes the 128-accumulator tile instead of the 64-accumulator one.  Both tiles compute the
     * same bits; the field replaces an environment variable the library used to read once per process. */
    int32_t fat_min_wgs;
    /* Fused pair of plain stride-1 convs (ABI v10, BSVD_F16X3 only): with pre_w_packed != NULL the launch computes
     *     t = pre_act(conv3x3(x, pre_w) + pre_bias)            (pre_cin -> Cin channels, zero outside the image)
     *     y = epilogue(act(conv3x3(t, w_packed) + bias))       (Cin -> Cout channels; PLAIN, RESID or the planar exit)
     * in ONE kernel: every tile computes the first conv on its own 18 x 18 patch (full K = 9 pre_cin on the matrix cores, the input
     * patch staged chunk by chunk through LDS) a pair of 16-channel chunks at a time, straight into the LDS patch of the second
     * conv; the Cin-channel tensor t never exists in HBM (1.33 GB written + read per 10-frame 540x960 clip at 64 channels).  The
     * price is the halo: 324 patch pixels on 12 MFMA row-tile slots per 256 output pixels = 1.5x the first conv's MFMAs -- on the
     * MI355X the pair is 11-13 % SLOWER than the two launches while moving half their HBM bytes: a choice for
     * hosts short of bandwidth, not of matrix time.  x is the first
     * conv's NHWC input [frames][H][W][pre_cin] (x_frame_stride its frame stride); pre_w_packed = bsvd_pack_weights(dtype
     * BSVD_F16X3) of the first conv (pre_cin -> Cin), pre_bias its bias_packed [Cin].  Both convs: OutputCvBlock / InputCvBlock,
     * bsvd_arch.py:194-226, 287-306.  Needs pre_cin % 16 == 0, Cin = 32 or 64 (the kernel carries two 32-channel pairs of t), Cout <= 64, stride 1, fold 0, no planar INPUT, no
     * head_w_packed / w_wino_packed; anything else returns -20 with the reason.  Per output the arithmetic of both convs is the
     * stand-alone kernels': fused == unfused bit for bit. */
    const void *pre_w_packed;
    const void *pre_bias;
    int32_t pre_cin, pre_act;
    /* Plain-fp32 hand-over between split-mode layers (ABI v10, BSVD_F16X3 only).  A tensor that only Winograd-form layers read need
     * not be carried as fp16 pairs: its producer stores the fp32 value itself (y_f32 != 0: PLAIN layers of the direct kernel, PLAIN and
     * PS_ADD layers of the Winograd kernel; same [frames][H][W][C] shape, strides and bytes: 4 per value either way) and its consumer
     * reads it as such (x_f32 != 0, with w_wino_packed only: x AND both halos hold fp32 channels).  The consumer's input transform then
     * starts from the value -- no decode, one 8-byte load per position, the transform on channel pairs -- which takes about a fifth off
     * the transform's instructions; values are exact fp32 instead of 22-bit pairs.  Anything else returns -21. */
    int32_t x_f32, y_f32;
    /* Transformed-domain hand-over between Winograd-form layers (ABI v11, BSVD_F16X3 + w_wino_packed only).  The input
     * transform of F(m,3) -- V = BT d over the m + 2 pixels of a group, re-split into fp16 pairs -- is a property of the TENSOR, not of its
     * reader: done in the reader's K loop it is repeated per output-channel tile and its VALU time adds to the MFMA time of the SIMD it
     * shares.  With y_v != 0 the producer's epilogue does it ONCE, on the fp32 values it holds anyway, and stores the
     * tensor in the transformed domain; with x_v != 0 the reader's K loop is a plain 16-byte copy into LDS.  Layout of such a tensor, per
     * frame (bsvd_v_frame_elems(H, W, C, m) floats):
     *     B[row][T tiles of 8 groups][C / 16 chunks] blocks of (m + 2) x 512 + 128 bytes (4224 for F(6,3)):
     *         [m + 2 positions][4 quarters: hi c0-7, hi c8-15, lo c0-7, lo c8-15][8 groups] x 16 bytes -- the reader's LDS planes of one patch row of
     *         one chunk, contiguous; a group = m consecutive pixels of a row, T = ceil(W / 8m); position xi of group g =
     *         sum_i BT[xi][i] * d[m g - 1 + i], pixels outside the image = 0 -- followed by the block's EDGE LINE [side 0 | 1][4 quarters] x 16
     *         bytes: position 0 of the tile's first group (side 0) and position m + 1 of its last group (side 1), the two values of a tile row
     *         that need a pixel of the NEIGHBOURING tile.  Readers take these two values from the edge line, never from the planes;
     *     E[row][T][4][C] fp32 behind the blocks -- the producer's edge record (partial sums of those two positions + the tile's own first / last
     *         pixel column), consumed by the patch pass that bsvd_conv3x3 issues right behind a y_v launch on the same stream (one small
     *         second kernel that fills the edge lines: the only cross-tile dependency of the transform).
     * x_v / y_v carry the form's m (2, 4 or 6) and must equal wino_m % 10 of the layer (x_v) resp. of the tensor's reader (y_v).
     * x_v: x AND both halos are such tensors -- halo_*_pstride = channels of the holding tensor, halo_*_coff = first channel, both
     * multiples of 16, exactly as for NHWC halos; x_frame_stride = elements between frames (>= bsvd_v_frame_elems).
     * y_v: PLAIN layers of the Winograd kernel whose own form has the same m (the epilogue's pixel groups are the reader's groups);
     * y_frame_stride likewise.  Not with x_f32 / y_f32 on the same tensor.  Anything else returns -22. */
    int32_t x_v, y_v;
} BsvdConvArgs;

int bsvd_abi_version(void);
int bsvd_conv_args_size(void);   /* sizeof(BsvdConvArgs) as compiled into the library (binding sanity check) */
/* ABI v10: bit set of how this library was built.  BSVD_BUILD_MEASURE: a measurement build (-DBSVD_MEASURE) that also contains
 * the kernel variants DESIGN.md 4.1d records as slower (more wino_m codes); the product library returns 0. */
#define BSVD_BUILD_MEASURE 1
int bsvd_build_info(void);
const char *bsvd_last_error(void);

/* Transformed-domain tensors (BsvdConvArgs.x_v / y_v): floats per frame (V planes + edge record), groups per row, and a stand-alone
 * transform of an NHWC tensor (x_f32 != 0: plain fp32 channels, else fp16 pairs) into that layout -- what a y_v producer writes, computed
 * directly (no edge record needed; the E block is zero-filled).  Test / measurement aid and the fallback for a producer that cannot
 * write the layout itself.  m = 2, 4 or 6; C % 16 == 0. */
int64_t bsvd_v_frame_elems(int32_t H, int32_t W, int32_t C, int32_t m);
int32_t bsvd_v_groups(int32_t W, int32_t m);
int bsvd_to_v(const void *x, int64_t x_frame_stride, int32_t x_f32, void *v, int64_t v_frame_stride, int32_t frames, int32_t H, int32_t W,
              int32_t C, int32_t m, void *stream);

/* The fused layer above (BSVD_F32: exact fp32 MFMA; BSVD_F16X3: split-fp16 3-pass MFMA). */
int bsvd_conv3x3(const BsvdConvArgs *args, void *stream);

/* Dry run: validates args exactly like bsvd_conv3x3 and writes the name of the kernel instantiation it would launch
 * (e.g. "conv3x3_kernel<4,2,2,2,1>[f16x3]" = <MT,NT,WM,WN,STRIDE>) -- profiling aid, launches nothing. */
int bsvd_conv3x3_variant(const BsvdConvArgs *args, char *name, int32_t name_len);

/*
 * Re-orders one nn.Conv2d weight [Cout][Cin][3][3] (+ bias [Cout]) into the layout bsvd_conv3x3
 * streams through LDS: w_packed[Cin_pad/16][9][4][Cout_pad][4] (the last index runs over 4
 * consecutive input channels), zero-filled padding.  pixel_shuffle != 0 additionally permutes the
 * output channels so that the four PixelShuffle sub-pixels become four contiguous channel groups:
 *   packed column sub*(Cout_pad/4) + c   <-   original output channel 4*c + sub.
 * Sizes: w_packed needs bsvd_packed_weight_elems(Cin_pad, Cout_pad) elements, bias_packed Cout_pad.
 * (One-time transform of the tensors BSVD.load() produces, bsvd_arch.py:462-474.)
 */
int64_t bsvd_packed_weight_elems(int32_t Cin_pad, int32_t Cout_pad);
int bsvd_pack_weights(const float *w_oihw, const float *bias, int32_t Cin, int32_t Cout,
                      int32_t Cin_pad, int32_t Cout_pad, int32_t pixel_shuffle, int32_t dtype,
                      void *w_packed, void *bias_packed, void *stream);

/*
 * Weights of the Winograd form (BsvdConvArgs.w_wino_packed): U[xi][ky] = sum_kx G[xi][kx] * w[.][.][ky][kx] for the
 * m + 2 transformed positions xi of F(m, 3), m = 2 | 4 | 6 (bsvd_amd/csrc/wino_forms.h), computed in double precision, split into fp16
 * pairs and laid out [Cin_pad/16][m + 2][3][hi, lo][2][Cout_pad][8 fp16]; pixel_shuffle permutes the output channels like
 * bsvd_pack_weights.  w_packed needs bsvd_packed_wino_weight_elems(Cin_pad, Cout_pad, m) 4-byte elements; bias_packed
 * (nullable) [Cout_pad] fp32 as in bsvd_pack_weights.
 */
int64_t bsvd_packed_wino_weight_elems(int32_t Cin_pad, int32_t Cout_pad, int32_t m);
int bsvd_pack_weights_wino(const float *w_oihw, const float *bias, int32_t Cin, int32_t Cout, int32_t Cin_pad,
                           int32_t Cout_pad, int32_t pixel_shuffle, int32_t m, void *w_packed, void *bias_packed,
                           void *stream);

/*
 * Weights of the fused network entry (BsvdConvArgs.head_w_packed): the first conv's [Cmid][Cin][3][3] (Cin = 3 or 4) as the
 * A operand of v_mfma_f32_32x32x16_f16 -- [Cmid_pad/32 channel pairs][3 k-steps][64 lanes][hi x8 | lo x8] fp16 with
 * k = tap*4 + channel (36 real of 48) -- plus head_bias[Cmid_pad] fp32 (zero padded).  w_packed needs
 * bsvd_packed_head_weight_bytes(Cmid_pad) bytes.
 */
int64_t bsvd_packed_head_weight_bytes(int32_t Cmid_pad);
int bsvd_pack_head_weights(const float *w_oihw, const float *bias, int32_t Cin, int32_t Cmid, int32_t Cmid_pad,
                           void *w_packed, float *bias_packed, void *stream);

/*
 * Clip entry/exit: the reference feeds NCHW tensors (BSVD.forward reshape, bsvd_arch.py:494-499,
 * and torch.cat(out_seq_clip) :552); callers then clamp to [0,1] (validation_seq_infer.py:24).
 *   nchw_to_nhwc: src [frames][C][H][W] fp32  ->  dst [frames][H][W][C_pad] (zero padded channels)
 *   nhwc_to_nchw: src [frames][H][W][C_pad]   ->  dst [frames][C][H][W] fp32, optional clamp to [lo,hi]
 */
int bsvd_nchw_to_nhwc(const float *src, void *dst, int32_t frames, int32_t C, int32_t H, int32_t W,
                      int32_t C_pad, int32_t dtype, void *stream);
int bsvd_nhwc_to_nchw(const void *src, float *dst, int32_t frames, int32_t C, int32_t H, int32_t W,
                      int32_t C_pad, int32_t dtype, int32_t do_clamp, float lo, float hi, void *stream);

/*
 * uint8 frame I/O on device (the reference normalises on the host, utils_common.py:184, and rounds
 * with tensor2img, img_util.py:66,87-90).  u8_to_planar: src uint8 [frames][H][W][C] (src_hwc != 0) or
 * [frames][C][H][W] -> dst planar fp32 [frames][C + const_channels][H][W] = src/255 with `const_channels` trailing
 * channels filled with const_val (the constant sigma map of temp_denoise, validation_seq_infer.py:19-21).
 * planar_to_u8: src planar fp32 [frames][C][H][W] -> clamp to [0,1], x255, round half to even -> uint8, HWC or planar,
 * optionally with the channel order reversed (RGB -> BGR like tensor2img).
 */
int bsvd_u8_to_planar(const uint8_t *src, float *dst, int32_t frames, int32_t C, int32_t H, int32_t W, int32_t src_hwc,
                      int32_t const_channels, float const_val, void *stream);
int bsvd_planar_to_u8(const float *src, uint8_t *dst, int32_t frames, int32_t C, int32_t H, int32_t W, int32_t dst_hwc,
                      int32_t reverse_channels, void *stream);

/*
 * Frame-window sharding: gathers the channel slice [c0, c0+n) of one NHWC frame into
 * a compact [H*W][n] buffer -- the message a rank sends to its temporal neighbour
 * (first frame, c0 = 0 -> the left neighbour's halo_next; last frame, c0 = fold -> the right
 * neighbour's halo_prev).  dtype BSVD_F32: a plain float range (also right for WHOLE 16-channel chunks of a split16 frame);
 * dtype BSVD_F16X3: one 8-channel half chunk of a split16 frame (n == 8, c0 % 8 == 0; the fold-8 layers of the c32-sized
 * networks) -> dst [H*W][hi x8 | lo x8], which bsvd_conv3x3 reads as a compact halo (pstride = 8, coff = 0).
 */
int bsvd_halo_pack(const void *frame, void *dst, int32_t HW, int32_t C, int32_t c0, int32_t n,
                   int32_t dtype, void *stream);

/*
 * Inverse of bsvd_halo_pack: scatters a compact [H*W][n] slice into channels [c0, c0+n) of an NHWC frame, for hosts that
 * keep a materialised neighbour frame (the reference's own `left_fold_2fold` / `center` buffers, bsvd_arch.py:112-113)
 * instead of handing the slice to bsvd_conv3x3 as halo_prev / halo_next (which reads compact slices in place:
 * pstride = n, coff = 0).  The other channels of `frame` are left untouched.
 */
int bsvd_halo_unpack(const void *src, void *frame, int32_t HW, int32_t C, int32_t c0, int32_t n,
                     int32_t dtype, void *stream);

/*
 * Scratch bytes bsvd_conv3x3(args) needs beyond the caller's tensors.  0 for every configuration of this ABI version
 * (all staging is LDS / registers; the library never allocates) -- exported so a host that sizes its arena from the
 * library keeps working if a later version wants a workspace.  Validates `args` like bsvd_conv3x3: < 0 = bad argument.
 */
int64_t bsvd_workspace_bytes(const BsvdConvArgs *args);

/*
 * Steady-state streaming step as ONE submission (SURVEY.md §7.1 "hipGraph-capture one steady-state step"; replaces the
 * per-layer Python/torch dispatch under BSVD.feedin_one_element, bsvd_arch.py:485-488).
 *
 * bsvd_conv3x3_batch: validates and enqueues args[0..n-1] in order on `stream` (same semantics as n calls of
 * bsvd_conv3x3); stops at the first failing layer and returns its code (bsvd_last_error() names the index).
 *
 * Graph capture: the host brackets a batch with bsvd_graph_begin / bsvd_graph_end on a NON-default capture stream (nothing
 * executes while capturing), gets an executable graph handle and replays it with bsvd_graph_launch on any stream, once
 * per frame.  All device pointers are baked into the graph, so the host keeps every buffer of a step in fixed rings
 * (bsvd_amd/stream_plan.py).  bsvd_graph_fork / bsvd_graph_join make a second stream part of the capture so that two
 * independent layer chains (the two DenBlocks of consecutive pipeline steps) become parallel branches of one graph.
 * A graph handle owns no device memory; bsvd_graph_destroy releases it.  Capture mode is "relaxed": other threads of
 * the process are not restricted while a capture is open.
 */
int bsvd_conv3x3_batch(const BsvdConvArgs *args, int32_t n, void *stream);
int bsvd_graph_begin(void *capture_stream);
int bsvd_graph_fork(void *capture_stream, void *side_stream);
int bsvd_graph_join(void *capture_stream, void *side_stream);
int bsvd_graph_end(void *capture_stream, void **graph_exec, int32_t *num_nodes);
int bsvd_graph_abort(void *capture_stream);   /* ends a capture after a failed launch and discards it */
int bsvd_graph_launch(void *graph_exec, void *stream);
int bsvd_graph_destroy(void *graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* BSVD_HIP_H */
