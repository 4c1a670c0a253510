/* CPU ORACLE (plain C) -- TEST INFRASTRUCTURE ONLY, never linked into the product library.
 *
 * Independent restatement of the per-layer arithmetic on the BSVD hot path, accumulating in
 * double so that it can arbitrate between the fp32 HIP kernels and the torch/oneDNN oracle:
 *
 *   y = epilogue( act( conv3x3( gather(prev, cur, next, fold) ) + bias ) )
 *
 *   gather   : ShiftConv's torch.cat  (/root/reference/Experimental_root/archs/bsvd_arch.py:42-50)
 *              channel c comes from the NEXT frame if c < fold, from the PREVIOUS frame if
 *              fold <= c < 2*fold, else from the current frame; absent neighbours are zeros
 *              (BiBufferConv stream start/end, bsvd_arch.py:94,104).
 *   conv3x3  : nn.Conv2d(k=3, pad=1, stride 1|2), weight [Cout][Cin][3][3]  (bsvd_arch.py:31-38,238)
 *   act      : 0 none, 1 ReLU, 2 ReLU6                                       (bsvd_arch.py:185-192)
 *   epilogue : 0 plain
 *              1 PixelShuffle(2) then + skip      out[c][2h+i][2w+j] = in[4c+2i+j][h][w]  (:266, :402-406)
 *              2 residual on the first 3 channels out[c] = base[c] - out[c], c < 3        (:408-414)
 *
 * Layout here is the reference's NCHW, one frame per call (pointers may be NULL as documented).
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/conv_ref.c -o oracle/libconv_ref.so
 */
#include <stddef.h>
#include <stdint.h>

static inline float act_f(double v, int act)
{
    if (act >= 1 && v < 0.0) v = 0.0;
    if (act == 2 && v > 6.0) v = 6.0;
    return (float)v;
}

/* returns 0 on success, -1 on bad arguments */
int oracle_conv3x3(const float *cur,      /* [Cin][H][W]                         */
                   const float *prev_sl,  /* [fold][H][W] = prev frame ch fold..2fold-1, or NULL */
                   const float *next_sl,  /* [fold][H][W] = next frame ch 0..fold-1,     or NULL */
                   int fold,              /* 0 -> plain conv                      */
                   const float *w,        /* [Cout][Cin][3][3]                    */
                   const float *bias,     /* [Cout] or NULL                       */
                   int Cin, int Cout, int H, int W, int stride, int act, int epilogue,
                   const float *extra,    /* epilogue 1: skip [Cout/4][2Ho][2Wo] or NULL; 2: base [>=3][Ho][Wo] */
                   float *out)            /* epi 0/2: [Cout][Ho][Wo]; epi 1: [Cout/4][2Ho][2Wo] */
{
    if (!cur || !w || !out || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return -1;
    if (fold < 0 || 2 * fold > Cin) return -1;
    if (epilogue == 1 && (Cout % 4)) return -1;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t plane = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        for (int oy = 0; oy < Ho; ++oy) {
            for (int ox = 0; ox < Wo; ++ox) {
                double acc = bias ? (double)bias[co] : 0.0;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *src;
                    if (ci < fold)            src = next_sl ? next_sl + (size_t)ci * plane : NULL;
                    else if (ci < 2 * fold)   src = prev_sl ? prev_sl + (size_t)(ci - fold) * plane : NULL;
                    else                      src = cur + (size_t)ci * plane;
                    if (!src) continue;
                    const float *wk = w + ((size_t)co * Cin + ci) * 9;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int iy = oy * stride + ky - 1;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int ix = ox * stride + kx - 1;
                            if (ix < 0 || ix >= W) continue;
                            acc += (double)src[(size_t)iy * W + ix] * (double)wk[ky * 3 + kx];
                        }
                    }
                }
                float v = act_f(acc, act);
                if (epilogue == 0) {
                    out[((size_t)co * Ho + oy) * Wo + ox] = v;
                } else if (epilogue == 1) {
                    const int c = co >> 2, i = (co >> 1) & 1, j = co & 1;
                    const size_t o = ((size_t)c * (2 * Ho) + (2 * oy + i)) * (2 * Wo) + (2 * ox + j);
                    out[o] = v + (extra ? extra[o] : 0.0f);
                } else {
                    const size_t o = ((size_t)co * Ho + oy) * Wo + ox;
                    out[o] = (co < 3 && extra) ? (float)((double)extra[o] - (double)v) : v;
                }
            }
        }
    }
    return 0;
}

/* layout helpers used by the tests to talk to the NHWC device buffers */
void oracle_nchw_to_nhwc(const float *src, float *dst, int C, int H, int W, int Cpad)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < Cpad; ++c)
                dst[((size_t)y * W + x) * Cpad + c] = c < C ? src[((size_t)c * H + y) * W + x] : 0.0f;
}

void oracle_nhwc_to_nchw(const float *src, float *dst, int C, int H, int W, int Cpad)
{
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                dst[((size_t)c * H + y) * W + x] = src[((size_t)y * W + x) * Cpad + c];
}
