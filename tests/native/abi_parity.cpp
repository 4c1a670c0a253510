// Torch-free consumer of the C ABI (include/bsvd_hip.h): what a non-Python host would write.
// TEST INFRASTRUCTURE: links libbsvd_hip.so (the product) and oracle/conv_ref.c (the plain-C double-accumulating
// checker) and compares them layer by layer -- device memory comes straight from the HIP runtime, no PyTorch anywhere.
//
//   case 1  temporal-fusion conv 128->128 (ShiftConv + ReLU6, bsvd_arch.py:42-50,133-141) over a 3-frame clip with both
//           neighbour-shard halos, exact-fp32 mode
//   case 2  stride-2 conv 64->128 + ReLU6 (DownBlock, :238)
//   case 3  conv 128->256 + PixelShuffle(2) + skip add (UpBlock :265-266, none_add :402)
//   case 4  3-layer chain in split-fp16 mode: planar 4-channel input -> head 4->64 -> conv 64->64 (BSVD_F16X3) -> tail
//           64->3 with residual + clamp, planar output  (InputCvBlock / OutputCvBlock + none_minus :408-414)
//
//   case 8  the same temporal-fusion conv in split-fp16 mode in its Winograd F(2,3) form (ABI v9: bsvd_pack_weights_wino +
//           BsvdConvArgs.w_wino_packed), the host doing the split16 encode / decode itself
//
// Built by __graft_entry__.build() (tests/native/Makefile); run by tests/test_gpu_native.py on the MI355X.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "bsvd_hip.h"

extern "C" int oracle_conv3x3(const float *cur, const float *prev_sl, const float *next_sl, int fold, const float *w,
                              const float *bias, int Cin, int Cout, int H, int W, int stride, int act, int epilogue,
                              const float *extra, float *out);

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define ABI_OK(x) do { int r_ = (x); if (r_ != 0) { printf("ABI error %d (%s) at %s:%d\n", r_, bsvd_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

static uint32_t rng_state = 12345u;
static float frand() {                       // uniform in [-1, 1)
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}
static std::vector<float> randv(size_t n, float scale) { std::vector<float> v(n); for (auto &x : v) x = frand() * scale; return v; }

template <class T> static T *dev(const std::vector<T> &h) {
    T *d = nullptr; HIP_OK(hipMalloc((void **)&d, h.size() * sizeof(T) + 16));
    HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}
static float *dev_zeros(size_t n) { float *d = nullptr; HIP_OK(hipMalloc((void **)&d, n * 4 + 16)); HIP_OK(hipMemset(d, 0, n * 4)); return d; }
static std::vector<float> host(const float *d, size_t n) { std::vector<float> h(n); HIP_OK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; }

// NCHW frame [C][H][W] -> NHWC [H][W][Cp] (zero-padded channels) and back
static std::vector<float> to_nhwc(const std::vector<float> &x, int T, int C, int H, int W, int Cp) {
    std::vector<float> y((size_t)T * H * W * Cp, 0.f);
    for (int t = 0; t < T; ++t) for (int c = 0; c < C; ++c) for (int p = 0; p < H * W; ++p)
        y[((size_t)t * H * W + p) * Cp + c] = x[((size_t)t * C + c) * H * W + p];
    return y;
}
static std::vector<float> to_nchw(const std::vector<float> &y, int T, int C, int H, int W, int Cp) {
    std::vector<float> x((size_t)T * C * H * W);
    for (int t = 0; t < T; ++t) for (int c = 0; c < C; ++c) for (int p = 0; p < H * W; ++p)
        x[((size_t)t * C + c) * H * W + p] = y[((size_t)t * H * W + p) * Cp + c];
    return x;
}
static double maxabs(const std::vector<float> &a, const std::vector<float> &b) {
    double m = 0; if (a.size() != b.size()) return 1e30;
    for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - (double)b[i]); if (!(d <= m)) m = d; }
    return m;
}

struct Packed { float *w, *b; };
static Packed pack(const std::vector<float> &w, const std::vector<float> &b, int Cin, int Cout, int Cinp, int Coutp, int ps, int dtype) {
    float *dw = dev(w), *db = dev(b);
    Packed p; p.w = dev_zeros((size_t)bsvd_packed_weight_elems(Cinp, Coutp)); p.b = dev_zeros(Coutp);
    ABI_OK(bsvd_pack_weights(dw, db, Cin, Cout, Cinp, Coutp, ps, dtype, p.w, p.b, nullptr));
    HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(dw)); HIP_OK(hipFree(db));
    return p;
}

static int failures = 0;
static void report(const char *name, double err, double tol) {
    printf("%-72s max-abs %.3e (tol %.0e) %s\n", name, err, tol, err < tol ? "ok" : "FAIL");
    if (!(err < tol)) ++failures;
}

static void case_tsm() {
    const int T = 3, C = 128, H = 10, W = 19, fold = C / 8;
    auto x = randv((size_t)T * C * H * W, 1.f), w = randv((size_t)C * C * 9, 0.04f), b = randv(C, 0.1f);
    auto prevf = randv((size_t)C * H * W, 1.f), nextf = randv((size_t)C * H * W, 1.f);      // neighbour shards' boundary frames
    float *dx = dev(to_nhwc(x, T, C, H, W, C)), *dp = dev(to_nhwc(prevf, 1, C, H, W, C)), *dn = dev(to_nhwc(nextf, 1, C, H, W, C));
    float *dy = dev_zeros((size_t)T * H * W * C);
    Packed pk = pack(w, b, C, C, C, C, 0, BSVD_F32);
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)H * W * C; a.fold = fold;
    a.halo_prev = dp; a.halo_prev_pstride = C; a.halo_prev_coff = fold;      // full NHWC frame: channels [fold, 2fold)
    a.halo_next = dn; a.halo_next_pstride = C; a.halo_next_coff = 0;         //                 channels [0, fold)
    a.w_packed = pk.w; a.bias_packed = pk.b; a.y = dy; a.y_frame_stride = (int64_t)H * W * C;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.epilogue = BSVD_EPI_PLAIN; a.dtype = BSVD_F32;
    if (bsvd_workspace_bytes(&a) != 0) { printf("workspace bytes != 0\n"); ++failures; }
    ABI_OK(bsvd_conv3x3(&a, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = to_nchw(host(dy, (size_t)T * H * W * C), T, C, H, W, C);
    std::vector<float> want((size_t)T * C * H * W);
    const size_t fr = (size_t)C * H * W, pl = (size_t)H * W;
    for (int t = 0; t < T; ++t) {
        const float *pv = (t == 0 ? prevf.data() : x.data() + (t - 1) * fr) + fold * pl;     // channels fold..2fold-1
        const float *nx = (t == T - 1 ? nextf.data() : x.data() + (t + 1) * fr);              // channels 0..fold-1
        if (oracle_conv3x3(x.data() + t * fr, pv, nx, fold, w.data(), b.data(), C, C, H, W, 1, 2, 0, nullptr, want.data() + t * fr)) exit(4);
    }
    report("temporal-fusion conv 128->128, 3 frames + both halos, fp32", maxabs(got, want), 1e-4);
}

static void case_stride2() {
    const int T = 2, Ci = 64, Co = 128, H = 21, W = 36, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    auto x = randv((size_t)T * Ci * H * W, 1.f), w = randv((size_t)Co * Ci * 9, 0.05f), b = randv(Co, 0.1f);
    float *dx = dev(to_nhwc(x, T, Ci, H, W, Ci)), *dy = dev_zeros((size_t)T * Ho * Wo * Co);
    Packed pk = pack(w, b, Ci, Co, Ci, Co, 0, BSVD_F32);
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)H * W * Ci; a.w_packed = pk.w; a.bias_packed = pk.b; a.y = dy; a.y_frame_stride = (int64_t)Ho * Wo * Co;
    a.frames = T; a.H = H; a.W = W; a.Cin = Ci; a.Cout = Co; a.stride = 2; a.act = BSVD_ACT_RELU6; a.dtype = BSVD_F32;
    ABI_OK(bsvd_conv3x3(&a, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = to_nchw(host(dy, (size_t)T * Ho * Wo * Co), T, Co, Ho, Wo, Co);
    std::vector<float> want((size_t)T * Co * Ho * Wo);
    for (int t = 0; t < T; ++t)
        if (oracle_conv3x3(x.data() + (size_t)t * Ci * H * W, nullptr, nullptr, 0, w.data(), b.data(), Ci, Co, H, W, 2, 2, 0, nullptr, want.data() + (size_t)t * Co * Ho * Wo)) exit(4);
    report("stride-2 conv 64->128 on 21x36 (odd height), fp32", maxabs(got, want), 1e-4);
}

static void case_pixel_shuffle() {
    const int T = 2, Ci = 128, Co = 256, Cq = Co / 4, H = 9, W = 13;
    auto x = randv((size_t)T * Ci * H * W, 1.f), w = randv((size_t)Co * Ci * 9, 0.04f), b = randv(Co, 0.1f);
    auto skip = randv((size_t)T * Cq * 4 * H * W, 1.f);                                       // [T][Cq][2H][2W]
    float *dx = dev(to_nhwc(x, T, Ci, H, W, Ci)), *ds = dev(to_nhwc(skip, T, Cq, 2 * H, 2 * W, Cq)), *dy = dev_zeros((size_t)T * 4 * H * W * Cq);
    Packed pk = pack(w, b, Ci, Co, Ci, Co, 1, BSVD_F32);
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)H * W * Ci; a.w_packed = pk.w; a.bias_packed = pk.b;
    a.extra = ds; a.extra_frame_stride = (int64_t)4 * H * W * Cq; a.extra_pstride = Cq; a.extra_cstride = 1;
    a.y = dy; a.y_frame_stride = (int64_t)4 * H * W * Cq;
    a.frames = T; a.H = H; a.W = W; a.Cin = Ci; a.Cout = Co; a.stride = 1; a.act = BSVD_ACT_NONE; a.epilogue = BSVD_EPI_PS_ADD; a.dtype = BSVD_F32;
    ABI_OK(bsvd_conv3x3(&a, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = to_nchw(host(dy, (size_t)T * 4 * H * W * Cq), T, Cq, 2 * H, 2 * W, Cq);
    std::vector<float> want((size_t)T * Cq * 4 * H * W);
    for (int t = 0; t < T; ++t)
        if (oracle_conv3x3(x.data() + (size_t)t * Ci * H * W, nullptr, nullptr, 0, w.data(), b.data(), Ci, Co, H, W, 1, 0, 1,
                           skip.data() + (size_t)t * Cq * 4 * H * W, want.data() + (size_t)t * Cq * 4 * H * W)) exit(4);
    report("conv 128->256 + PixelShuffle(2) + skip add, fp32", maxabs(got, want), 1e-4);
}

static void case_split_chain() {
    const int T = 2, H = 24, W = 40, C = 64;
    auto x = randv((size_t)T * 4 * H * W, 1.f);                                               // planar [T][4][H][W]
    auto w1 = randv((size_t)C * 4 * 9, 0.3f), b1 = randv(C, 0.1f);
    auto w2 = randv((size_t)C * C * 9, 0.06f), b2 = randv(C, 0.1f);
    auto w3 = randv((size_t)3 * C * 9, 0.06f), b3 = randv(3, 0.1f);
    float *dx = dev(x), *d1 = dev_zeros((size_t)T * H * W * C), *d2 = dev_zeros((size_t)T * H * W * C), *dy = dev_zeros((size_t)T * 3 * H * W);
    Packed p1 = pack(w1, b1, 4, C, 16, C, 0, BSVD_F32);          // the entry layer keeps an fp32 pack (VALU kernel)
    Packed p2 = pack(w2, b2, C, C, C, C, 0, BSVD_F16X3);
    Packed p3 = pack(w3, b3, C, 3, C, 16, 0, BSVD_F16X3);        // the exit layer is an MFMA layer too
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)4 * H * W; a.x_planar_ch = 4; a.w_packed = p1.w; a.bias_packed = p1.b;
    a.y = d1; a.y_frame_stride = (int64_t)H * W * C; a.frames = T; a.H = H; a.W = W; a.Cin = 16; a.Cout = C; a.stride = 1;
    a.act = BSVD_ACT_RELU6; a.dtype = BSVD_F16X3;
    ABI_OK(bsvd_conv3x3(&a, nullptr));
    memset(&a, 0, sizeof(a));
    a.x = d1; a.x_frame_stride = (int64_t)H * W * C; a.w_packed = p2.w; a.bias_packed = p2.b; a.y = d2; a.y_frame_stride = (int64_t)H * W * C;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.dtype = BSVD_F16X3;
    ABI_OK(bsvd_conv3x3(&a, nullptr));
    memset(&a, 0, sizeof(a));
    a.x = d2; a.x_frame_stride = (int64_t)H * W * C; a.w_packed = p3.w; a.bias_packed = p3.b; a.y = dy; a.y_frame_stride = (int64_t)3 * H * W;
    a.extra = dx; a.extra_frame_stride = (int64_t)4 * H * W; a.extra_pstride = 1; a.extra_cstride = H * W; a.resid_ch = 3;   // planar base
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = 16; a.stride = 1; a.act = BSVD_ACT_NONE; a.epilogue = BSVD_EPI_RESID;
    a.dtype = BSVD_F16X3; a.y_planar_ch = 3; a.y_clamp = 1; a.y_lo = -0.5f; a.y_hi = 0.75f;
    ABI_OK(bsvd_conv3x3(&a, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = host(dy, (size_t)T * 3 * H * W);
    std::vector<float> want((size_t)T * 3 * H * W), t1((size_t)C * H * W), t2((size_t)C * H * W);
    for (int t = 0; t < T; ++t) {
        const float *xf = x.data() + (size_t)t * 4 * H * W;
        if (oracle_conv3x3(xf, nullptr, nullptr, 0, w1.data(), b1.data(), 4, C, H, W, 1, 2, 0, nullptr, t1.data())) exit(4);
        if (oracle_conv3x3(t1.data(), nullptr, nullptr, 0, w2.data(), b2.data(), C, C, H, W, 1, 2, 0, nullptr, t2.data())) exit(4);
        if (oracle_conv3x3(t2.data(), nullptr, nullptr, 0, w3.data(), b3.data(), C, 3, H, W, 1, 0, 2, xf, want.data() + (size_t)t * 3 * H * W)) exit(4);
    }
    for (auto &v : want) v = fminf(fmaxf(v, -0.5f), 0.75f);
    report("split-fp16 chain: planar in -> 4->64 -> 64->64 (f16x3) -> 64->3 resid+clamp", maxabs(got, want), 1e-3);
}

// The same chain with InputCvBlock's two convs as ONE launch (ABI v8, BsvdConvArgs.head_w_packed): planar in -> [4->64 -> 64->64]
// fused -> 64->3.  Against the plain-C oracle, and against the three-launch chain above (same arithmetic class, not bitwise:
// the fused entry computes the 4->64 conv with split-fp16 MFMAs, the stand-alone entry kernel with exact fp32 FMAs).
static void case_fused_entry() {
    const int T = 2, H = 24, W = 40, C = 64;
    auto x = randv((size_t)T * 4 * H * W, 1.f);
    auto w1 = randv((size_t)C * 4 * 9, 0.3f), b1 = randv(C, 0.1f);
    auto w2 = randv((size_t)C * C * 9, 0.06f), b2 = randv(C, 0.1f);
    auto w3 = randv((size_t)3 * C * 9, 0.06f), b3 = randv(3, 0.1f);
    float *dx = dev(x), *d2 = dev_zeros((size_t)T * H * W * C), *dy = dev_zeros((size_t)T * 3 * H * W);
    float *dw1 = dev(w1), *db1 = dev(b1);
    void *hw = nullptr; float *hb = nullptr;
    HIP_OK(hipMalloc(&hw, (size_t)bsvd_packed_head_weight_bytes(C))); HIP_OK(hipMalloc((void **)&hb, sizeof(float) * C));
    ABI_OK(bsvd_pack_head_weights(dw1, db1, 4, C, C, hw, hb, nullptr));
    if (bsvd_pack_head_weights(dw1, db1, 5, C, C, hw, hb, nullptr) != -5) { printf("pack_head_weights must refuse 5 input channels\n"); ++failures; }
    Packed p2 = pack(w2, b2, C, C, C, C, 0, BSVD_F16X3);
    Packed p3 = pack(w3, b3, C, 3, C, 16, 0, BSVD_F16X3);
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)4 * H * W; a.x_planar_ch = 4; a.head_w_packed = hw; a.head_bias = hb;
    a.w_packed = p2.w; a.bias_packed = p2.b; a.y = d2; a.y_frame_stride = (int64_t)H * W * C;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.dtype = BSVD_F16X3;
    char name[96];
    ABI_OK(bsvd_conv3x3_variant(&a, name, sizeof(name)));
    if (!strstr(name, "[fused entry]")) { printf("variant of the fused entry: %s\n", name); ++failures; }
    ABI_OK(bsvd_conv3x3(&a, nullptr));
    BsvdConvArgs bad = a; bad.dtype = BSVD_F32;
    if (bsvd_conv3x3(&bad, nullptr) != -18) { printf("the fused entry must refuse BSVD_F32: %s\n", bsvd_last_error()); ++failures; }
    memset(&a, 0, sizeof(a));
    a.x = d2; a.x_frame_stride = (int64_t)H * W * C; a.w_packed = p3.w; a.bias_packed = p3.b; a.y = dy; a.y_frame_stride = (int64_t)3 * H * W;
    a.extra = dx; a.extra_frame_stride = (int64_t)4 * H * W; a.extra_pstride = 1; a.extra_cstride = H * W; a.resid_ch = 3;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = 16; a.stride = 1; a.act = BSVD_ACT_NONE; a.epilogue = BSVD_EPI_RESID;
    a.dtype = BSVD_F16X3; a.y_planar_ch = 3;
    ABI_OK(bsvd_conv3x3(&a, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = host(dy, (size_t)T * 3 * H * W);
    std::vector<float> want((size_t)T * 3 * H * W), t1((size_t)C * H * W), t2((size_t)C * H * W);
    for (int t = 0; t < T; ++t) {
        const float *xf = x.data() + (size_t)t * 4 * H * W;
        if (oracle_conv3x3(xf, nullptr, nullptr, 0, w1.data(), b1.data(), 4, C, H, W, 1, 2, 0, nullptr, t1.data())) exit(4);
        if (oracle_conv3x3(t1.data(), nullptr, nullptr, 0, w2.data(), b2.data(), C, C, H, W, 1, 2, 0, nullptr, t2.data())) exit(4);
        if (oracle_conv3x3(t2.data(), nullptr, nullptr, 0, w3.data(), b3.data(), C, 3, H, W, 1, 0, 2, xf, want.data() + (size_t)t * 3 * H * W)) exit(4);
    }
    report("fused entry: planar in -> [4->64 -> 64->64] in one launch (f16x3) -> 64->3 resid", maxabs(got, want), 1e-3);
}

// One streaming step as one submission (ABI v6): the three-layer split chain issued with bsvd_conv3x3_batch, then captured
// into a HIP graph on a created stream and replayed on another; both must reproduce the layer-by-layer result bit for bit.
static void case_batch_and_graph() {
    const int T = 1, H = 24, W = 40, C = 64;
    auto x = randv((size_t)T * 4 * H * W, 1.f);
    auto w1 = randv((size_t)C * 4 * 9, 0.3f), b1 = randv(C, 0.1f);
    auto w2 = randv((size_t)C * C * 9, 0.06f), b2 = randv(C, 0.1f);
    auto w3 = randv((size_t)3 * C * 9, 0.06f), b3 = randv(3, 0.1f);
    float *dx = dev(x), *d1 = dev_zeros((size_t)T * H * W * C), *d2 = dev_zeros((size_t)T * H * W * C), *dy = dev_zeros((size_t)T * 3 * H * W);
    Packed p1 = pack(w1, b1, 4, C, 16, C, 0, BSVD_F32), p2 = pack(w2, b2, C, C, C, C, 0, BSVD_F16X3), p3 = pack(w3, b3, C, 3, C, 16, 0, BSVD_F16X3);
    BsvdConvArgs a[3]; memset(a, 0, sizeof(a));
    a[0].x = dx; a[0].x_frame_stride = (int64_t)4 * H * W; a[0].x_planar_ch = 4; a[0].w_packed = p1.w; a[0].bias_packed = p1.b;
    a[0].y = d1; a[0].y_frame_stride = (int64_t)H * W * C; a[0].Cin = 16; a[0].Cout = C; a[0].act = BSVD_ACT_RELU6;
    a[1].x = d1; a[1].x_frame_stride = (int64_t)H * W * C; a[1].w_packed = p2.w; a[1].bias_packed = p2.b; a[1].y = d2;
    a[1].y_frame_stride = (int64_t)H * W * C; a[1].Cin = C; a[1].Cout = C; a[1].act = BSVD_ACT_RELU6;
    a[2].x = d2; a[2].x_frame_stride = (int64_t)H * W * C; a[2].w_packed = p3.w; a[2].bias_packed = p3.b; a[2].y = dy;
    a[2].y_frame_stride = (int64_t)3 * H * W; a[2].extra = dx; a[2].extra_frame_stride = (int64_t)4 * H * W; a[2].extra_pstride = 1;
    a[2].extra_cstride = H * W; a[2].resid_ch = 3; a[2].Cin = C; a[2].Cout = 16; a[2].act = BSVD_ACT_NONE; a[2].epilogue = BSVD_EPI_RESID; a[2].y_planar_ch = 3;
    for (auto &l : a) { l.frames = T; l.H = H; l.W = W; l.stride = 1; l.dtype = BSVD_F16X3; }
    for (auto &l : a) ABI_OK(bsvd_conv3x3(&l, nullptr));
    HIP_OK(hipDeviceSynchronize());
    auto ref = host(dy, (size_t)T * 3 * H * W);
    HIP_OK(hipMemset(dy, 0, sizeof(float) * T * 3 * H * W));
    ABI_OK(bsvd_conv3x3_batch(a, 3, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = host(dy, (size_t)T * 3 * H * W);
    report("bsvd_conv3x3_batch == three bsvd_conv3x3 calls (bitwise)", maxabs(got, ref), 1e-30);
    BsvdConvArgs bad = a[1]; bad.stride = 3;
    BsvdConvArgs seq[2] = {a[0], bad};
    if (bsvd_conv3x3_batch(seq, 2, nullptr) != -6 || !strstr(bsvd_last_error(), "layer 1 of 2")) { printf("batch error reporting: %s\n", bsvd_last_error()); ++failures; }
    hipStream_t cap, run; HIP_OK(hipStreamCreate(&cap)); HIP_OK(hipStreamCreate(&run));
    if (bsvd_graph_begin(nullptr) != -1) { printf("capture on the default stream must be refused\n"); ++failures; }
    void *g = nullptr; int32_t nodes = 0;
    ABI_OK(bsvd_graph_begin(cap)); ABI_OK(bsvd_conv3x3_batch(a, 3, cap)); ABI_OK(bsvd_graph_end(cap, &g, &nodes));
    if (nodes != 3) { printf("graph has %d nodes, expected 3\n", nodes); ++failures; }
    for (int rep = 0; rep < 2; ++rep) {
        HIP_OK(hipMemset(dy, 0, sizeof(float) * T * 3 * H * W));
        ABI_OK(bsvd_graph_launch(g, run)); HIP_OK(hipStreamSynchronize(run));
        got = host(dy, (size_t)T * 3 * H * W);
        report(rep ? "HIP graph replay #2 == direct launches (bitwise)" : "HIP graph replay #1 == direct launches (bitwise)", maxabs(got, ref), 1e-30);
    }
    ABI_OK(bsvd_graph_destroy(g)); HIP_OK(hipStreamDestroy(cap)); HIP_OK(hipStreamDestroy(run));
}

// BiBufferConv (bsvd_arch.py:53-114) as a torch-free host would stream it: the two frame buffers of the reference become a ring
// of three device frames (next / pending / past), every step is one bsvd_conv3x3 on fixed addresses, and the three ring phases
// of the steady state are three HIP graphs captured once and replayed -- compared bit for bit with ONE clip launch over the
// same frames (the kernel reads t-1 / t+1 inside the clip).  Stream start / end (no past / no next frame: the zeros of :94,104)
// are NULL halos, launched directly.
static void case_stream_ring_graphs() {
    const int T = 11, C = 128, H = 12, W = 20, fold = C / 8;
    const size_t fr = (size_t)H * W * C;
    auto x = randv((size_t)T * C * H * W, 1.f), w = randv((size_t)C * C * 9, 0.04f), b = randv(C, 0.1f);
    float *dx = dev(to_nhwc(x, T, C, H, W, C)), *dclip = dev_zeros(T * fr), *dstream = dev_zeros(T * fr);
    Packed pk = pack(w, b, C, C, C, C, 0, BSVD_F32);
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x_frame_stride = (int64_t)fr; a.fold = fold; a.w_packed = pk.w; a.bias_packed = pk.b; a.y_frame_stride = (int64_t)fr;
    a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.epilogue = BSVD_EPI_PLAIN; a.dtype = BSVD_F32;
    BsvdConvArgs clip = a; clip.x = dx; clip.y = dclip; clip.frames = T;
    ABI_OK(bsvd_conv3x3(&clip, nullptr)); HIP_OK(hipDeviceSynchronize());
    float *ring[3], *oring[3];
    for (int i = 0; i < 3; ++i) { ring[i] = dev_zeros(fr); oring[i] = dev_zeros(fr); }
    hipStream_t cap, run; HIP_OK(hipStreamCreate(&cap)); HIP_OK(hipStreamCreate(&run));
    void *graph[3] = {nullptr, nullptr, nullptr};
    int replays = 0, direct = 0;
    for (int s = 0; s <= T; ++s) {                       // step s: frame s arrives (s < T), frame s-1 leaves
        if (s < T) HIP_OK(hipMemcpyAsync(ring[s % 3], dx + s * fr, fr * 4, hipMemcpyDeviceToDevice, run));
        if (s == 0) continue;                            // first call of BiBufferConv returns None (:86-100)
        const int t = s - 1;
        BsvdConvArgs st = a; st.frames = 1; st.x = ring[t % 3]; st.y = oring[t % 3];
        if (t > 0) { st.halo_prev = ring[(t - 1) % 3]; st.halo_prev_pstride = C; st.halo_prev_coff = fold; }
        if (s < T) { st.halo_next = ring[s % 3]; st.halo_next_pstride = C; st.halo_next_coff = 0; }
        if (t > 0 && s < T) {                            // steady state: all three ring slots in play, addresses repeat with period 3
            const int ph = s % 3;
            if (!graph[ph]) {
                int32_t nodes = 0;
                ABI_OK(bsvd_graph_begin(cap)); ABI_OK(bsvd_conv3x3_batch(&st, 1, cap)); ABI_OK(bsvd_graph_end(cap, &graph[ph], &nodes));
            } else ++replays;
            ABI_OK(bsvd_graph_launch(graph[ph], run));
        } else { ABI_OK(bsvd_conv3x3(&st, run)); ++direct; }
        HIP_OK(hipMemcpyAsync(dstream + t * fr, oring[t % 3], fr * 4, hipMemcpyDeviceToDevice, run));
    }
    HIP_OK(hipStreamSynchronize(run));
    report("BiBufferConv streamed on a 3-frame ring with per-phase HIP graphs == one clip launch (bitwise)",
           maxabs(host(dstream, T * fr), host(dclip, T * fr)), 1e-30);
    if (replays != T - 2 - 3 || direct != 2) { printf("stream bookkeeping: %d replays, %d direct launches\n", replays, direct); ++failures; }
    for (auto g : graph) ABI_OK(bsvd_graph_destroy(g));
    HIP_OK(hipStreamDestroy(cap)); HIP_OK(hipStreamDestroy(run));
}

// ---- split16 on the host (what a non-Python consumer of BSVD_F16X3 tensors writes): fp32 <-> IEEE half, round to nearest even
static uint16_t f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));      // overflow / inf / nan
    if (x < 0x38800000u) {                                                                     // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23), sh = 126 - e;                                            // 14 .. 24
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t r = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        return (uint16_t)(sign | (r + ((rem > half || (rem == half && (r & 1))) ? 1 : 0)));
    }
    const uint32_t r = x + 0xc8000fffu + ((x >> 13) & 1);                                      // rebias + round to nearest even
    return (uint16_t)(sign | (r >> 13));
}
static float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 1023u;
    uint32_t x;
    if (e == 0) { if (!m) x = sign; else { float v = (float)m * 5.9604644775390625e-8f; memcpy(&x, &v, 4); x |= sign; } }
    else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}
// NHWC fp32 [..][C] (C % 16 == 0) -> split16: per 16-channel chunk [hi x16 | lo x16] halves in the same 64 bytes
static std::vector<float> to_split16(const std::vector<float> &v) {
    std::vector<float> o(v.size());
    uint16_t *q = reinterpret_cast<uint16_t *>(o.data());
    for (size_t c = 0; c < v.size(); c += 16)
        for (int j = 0; j < 16; ++j) { const uint16_t hi = f2h(v[c + j]); q[2 * c + j] = hi; q[2 * c + 16 + j] = f2h(v[c + j] - h2f(hi)); }
    return o;
}
static std::vector<float> from_split16(const std::vector<float> &s) {
    std::vector<float> o(s.size());
    const uint16_t *q = reinterpret_cast<const uint16_t *>(s.data());
    for (size_t c = 0; c < s.size(); c += 16)
        for (int j = 0; j < 16; ++j) o[c + j] = h2f(q[2 * c + j]) + h2f(q[2 * c + 16 + j]);
    return o;
}

static void case_winograd() {
    const int T = 3, C = 128, H = 21, W = 38, fold = C / 8;
    auto x = randv((size_t)T * C * H * W, 1.f), w = randv((size_t)C * C * 9, 0.04f), b = randv(C, 0.1f);
    auto prevf = randv((size_t)C * H * W, 1.f);                                              // left neighbour's frame; no right neighbour (zeros)
    // the oracle sees the values the split16 tensors encode
    auto xs = to_split16(to_nhwc(x, T, C, H, W, C)), ps = to_split16(to_nhwc(prevf, 1, C, H, W, C));
    x = to_nchw(from_split16(xs), T, C, H, W, C); prevf = to_nchw(from_split16(ps), 1, C, H, W, C);
    float *dx = dev(xs), *dp = dev(ps), *dy = dev_zeros((size_t)T * H * W * C);
    float *dw = dev(w), *db = dev(b);
    float *wq = dev_zeros((size_t)bsvd_packed_wino_weight_elems(C, C, 2)), *bq = dev_zeros(C);
    ABI_OK(bsvd_pack_weights_wino(dw, db, C, C, C, C, 0, 2, wq, bq, nullptr));
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)H * W * C; a.fold = fold;
    a.halo_prev = dp; a.halo_prev_pstride = C; a.halo_prev_coff = fold;
    a.w_wino_packed = wq; a.wino_m = 2; a.bias_packed = bq; a.y = dy; a.y_frame_stride = (int64_t)H * W * C;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.epilogue = BSVD_EPI_PLAIN; a.dtype = BSVD_F16X3;
    char name[96]; ABI_OK(bsvd_conv3x3_variant(&a, name, sizeof(name)));
    if (!strstr(name, "F(2,3)")) { printf("variant %s\n", name); ++failures; }
    ABI_OK(bsvd_conv3x3(&a, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto got = to_nchw(from_split16(host(dy, (size_t)T * H * W * C)), T, C, H, W, C);
    std::vector<float> want((size_t)T * C * H * W);
    const size_t fr = (size_t)C * H * W, pl = (size_t)H * W;
    std::vector<float> zeros(fr, 0.f);
    for (int t = 0; t < T; ++t) {
        const float *pv = (t == 0 ? prevf.data() : x.data() + (t - 1) * fr) + fold * pl;
        const float *nx = (t == T - 1 ? zeros.data() : x.data() + (t + 1) * fr);
        if (oracle_conv3x3(x.data() + t * fr, pv, nx, fold, w.data(), b.data(), C, C, H, W, 1, 2, 0, nullptr, want.data() + t * fr)) exit(4);
    }
    report("temporal-fusion conv 128->128 as Winograd F(2,3), split16 tensors, left halo only", maxabs(got, want), 2e-4);
    a.stride = 2;
    if (bsvd_conv3x3(&a, nullptr) != -19 || !strstr(bsvd_last_error(), "stride")) { printf("w_wino_packed + stride 2 not refused\n"); ++failures; }
}

// case 9  OutputCvBlock's two convs (bsvd_arch.py:287-306) 64 -> 64 -> 64, ReLU6 / none, as ONE launch (ABI v10: BsvdConvArgs.pre_w_packed):
//         against the two launches (bit for bit) and against the plain-C oracle's two convs (the tensor between them re-encoded as fp16 pairs)
static void case_fused_pair() {
    const int T = 2, C = 64, H = 21, W = 38;
    auto x = randv((size_t)T * C * H * W, 1.5f), wa = randv((size_t)C * C * 9, 0.05f), ba = randv(C, 0.1f), wb = randv((size_t)C * C * 9, 0.05f), bb = randv(C, 0.1f);
    auto xs = to_split16(to_nhwc(x, T, C, H, W, C));
    x = to_nchw(from_split16(xs), T, C, H, W, C);
    Packed pa = pack(wa, ba, C, C, C, C, 0, BSVD_F16X3), pb = pack(wb, bb, C, C, C, C, 0, BSVD_F16X3);
    float *dx = dev(xs), *dmid = dev_zeros((size_t)T * H * W * C), *dy2 = dev_zeros((size_t)T * H * W * C), *dy1 = dev_zeros((size_t)T * H * W * C);
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = (int64_t)H * W * C; a.y_frame_stride = (int64_t)H * W * C;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.epilogue = BSVD_EPI_PLAIN; a.dtype = BSVD_F16X3;
    BsvdConvArgs a1 = a, a2 = a, af = a;
    a1.w_packed = pa.w; a1.bias_packed = pa.b; a1.act = BSVD_ACT_RELU6; a1.y = dmid;
    a2.x = dmid; a2.w_packed = pb.w; a2.bias_packed = pb.b; a2.act = BSVD_ACT_NONE; a2.y = dy2;
    af.w_packed = pb.w; af.bias_packed = pb.b; af.act = BSVD_ACT_NONE; af.y = dy1;
    af.pre_w_packed = pa.w; af.pre_bias = pa.b; af.pre_cin = C; af.pre_act = BSVD_ACT_RELU6;
    char name[96]; ABI_OK(bsvd_conv3x3_variant(&af, name, sizeof(name)));
    if (!strstr(name, "[fused pair]")) { printf("variant %s\n", name); ++failures; }
    ABI_OK(bsvd_conv3x3(&a1, nullptr)); ABI_OK(bsvd_conv3x3(&a2, nullptr)); ABI_OK(bsvd_conv3x3(&af, nullptr)); HIP_OK(hipDeviceSynchronize());
    auto y1 = host(dy1, (size_t)T * H * W * C), y2 = host(dy2, (size_t)T * H * W * C);
    if (memcmp(y1.data(), y2.data(), y1.size() * 4) != 0) { printf("FAIL fused pair != the two launches bit for bit\n"); ++failures; }
    auto got = to_nchw(from_split16(y1), T, C, H, W, C);
    std::vector<float> mid((size_t)T * C * H * W), want((size_t)T * C * H * W);
    const size_t fr = (size_t)C * H * W;
    for (int t = 0; t < T; ++t)
        if (oracle_conv3x3(x.data() + t * fr, nullptr, nullptr, 0, wa.data(), ba.data(), C, C, H, W, 1, 2, 0, nullptr, mid.data() + t * fr)) exit(4);
    mid = to_nchw(from_split16(to_split16(to_nhwc(mid, T, C, H, W, C))), T, C, H, W, C);
    for (int t = 0; t < T; ++t)
        if (oracle_conv3x3(mid.data() + t * fr, nullptr, nullptr, 0, wb.data(), bb.data(), C, C, H, W, 1, 0, 0, nullptr, want.data() + t * fr)) exit(4);
    report("OutputCvBlock pair 64->64->64 in one launch (pre_w_packed), split16 tensors", maxabs(got, want), 2e-4);
    af.fold = 16;
    if (bsvd_conv3x3(&af, nullptr) != -20 || !strstr(bsvd_last_error(), "fold")) { printf("pre_w_packed + fold not refused\n"); ++failures; }
}

// case 10  two temporal-fusion convs 128 -> 128 -> 128 (MemCvBlock, bsvd_arch.py:116-149) as F(6,3) layers with the tensor BETWEEN them in the transformed
// domain (ABI v11: the first launch writes y_v -- its epilogue applies the second layer's input transform, the patch pass completes the tile edges --, the
// second reads x_v): against the plain-C oracle's two convs, and bit for bit against the same chain with a plain-fp32 tensor between the layers.
static void case_transformed_handover() {
    const int T = 2, C = 128, H = 19, W = 100, fold = C / 8;        // three 48-pixel tiles per row: two interior tile boundaries to patch
    auto x = randv((size_t)T * C * H * W, 1.f), w1 = randv((size_t)C * C * 9, 0.04f), b1 = randv(C, 0.1f), w2 = randv((size_t)C * C * 9, 0.04f), b2 = randv(C, 0.1f);
    auto xn = to_nhwc(x, T, C, H, W, C);                              // plain fp32 input of the first layer (x_f32)
    float *dx = dev(xn), *dw1 = dev(w1), *db1 = dev(b1), *dw2 = dev(w2), *db2 = dev(b2);
    float *wq1 = dev_zeros((size_t)bsvd_packed_wino_weight_elems(C, C, 6)), *bq1 = dev_zeros(C), *wq2 = dev_zeros((size_t)bsvd_packed_wino_weight_elems(C, C, 6)), *bq2 = dev_zeros(C);
    ABI_OK(bsvd_pack_weights_wino(dw1, db1, C, C, C, C, 0, 6, wq1, bq1, nullptr));
    ABI_OK(bsvd_pack_weights_wino(dw2, db2, C, C, C, C, 0, 6, wq2, bq2, nullptr));
    const int64_t vfe = bsvd_v_frame_elems(H, W, C, 6);
    if (vfe <= 0 || bsvd_v_groups(W, 6) != 24) { printf("bsvd_v_frame_elems / bsvd_v_groups\n"); ++failures; return; }
    float *dv = dev_zeros((size_t)T * vfe), *dm = dev_zeros((size_t)T * H * W * C), *dy_v = dev_zeros((size_t)T * H * W * C), *dy_f = dev_zeros((size_t)T * H * W * C);
    auto layer = [&](const float *in, int in_v, int in_f32, int64_t in_fs, float *wq, float *bq, float *out, int out_v, int out_f32, int64_t out_fs, const char *tag) {
        BsvdConvArgs a; memset(&a, 0, sizeof(a));
        a.x = in; a.x_frame_stride = in_fs; a.x_v = in_v; a.x_f32 = in_f32; a.fold = fold;       // no halos: zeros past both ends of the clip
        a.w_wino_packed = wq; a.wino_m = 6; a.bias_packed = bq; a.y = out; a.y_frame_stride = out_fs; a.y_v = out_v; a.y_f32 = out_f32;
        a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.epilogue = BSVD_EPI_PLAIN; a.dtype = BSVD_F16X3;
        char name[96]; ABI_OK(bsvd_conv3x3_variant(&a, name, sizeof(name)));
        if (!strstr(name, tag)) { printf("variant %s (expected %s)\n", name, tag); ++failures; }
        ABI_OK(bsvd_conv3x3(&a, nullptr));
    };
    const int64_t fs = (int64_t)H * W * C;
    layer(dx, 0, 1, fs, wq1, bq1, dv, 6, 0, vfe, "[V out]");            // transformed tensor between the layers
    layer(dv, 6, 0, vfe, wq2, bq2, dy_v, 0, 0, fs, "[V in]");
    layer(dx, 0, 1, fs, wq1, bq1, dm, 0, 1, fs, "[f32 in]");             // plain-fp32 tensor between the layers
    layer(dm, 0, 1, fs, wq2, bq2, dy_f, 0, 0, fs, "[f32 in]");
    HIP_OK(hipDeviceSynchronize());
    auto gv = host(dy_v, (size_t)T * H * W * C), gf = host(dy_f, (size_t)T * H * W * C);
    auto got = to_nchw(from_split16(gv), T, C, H, W, C);
    std::vector<float> mid((size_t)T * C * H * W), want((size_t)T * C * H * W);
    const size_t fr = (size_t)C * H * W, pl = (size_t)H * W;
    std::vector<float> zeros(fr, 0.f);
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<float> &in = pass ? mid : x; std::vector<float> &out = pass ? want : mid;
        for (int t = 0; t < T; ++t) {
            const float *pv = (t == 0 ? zeros.data() : in.data() + (t - 1) * fr) + (t == 0 ? 0 : fold * pl);
            const float *nx = (t == T - 1 ? zeros.data() : in.data() + (t + 1) * fr);
            if (oracle_conv3x3(in.data() + t * fr, pv, nx, fold, pass ? w2.data() : w1.data(), pass ? b2.data() : b1.data(), C, C, H, W, 1, 2, 0, nullptr, out.data() + t * fr)) exit(4);
        }
    }
    report("MemCvBlock 128->128->128 as F(6,3) layers, transformed-domain tensor between them (y_v -> x_v)", maxabs(got, want), 3e-4);
    // the two chains differ only at the patched positions (one fp32 rounding later): their outputs agree to the last bits
    report("... against the same chain with a plain-fp32 tensor between the layers", maxabs(got, to_nchw(from_split16(gf), T, C, H, W, C)), 2e-5);
    // refusals: y_v needs the frame stride of a transformed frame; x_v of another form
    BsvdConvArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.x_frame_stride = fs; a.x_f32 = 1; a.w_wino_packed = wq1; a.wino_m = 6; a.bias_packed = bq1; a.y = dv; a.y_frame_stride = fs; a.y_v = 6;
    a.frames = T; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.stride = 1; a.act = BSVD_ACT_RELU6; a.dtype = BSVD_F16X3;
    if (bsvd_conv3x3(&a, nullptr) != -22 || !strstr(bsvd_last_error(), "y_frame_stride")) { printf("short y_frame_stride not refused\n"); ++failures; }
    a.y_frame_stride = vfe; a.x_f32 = 0; a.x_v = 2;
    if (bsvd_conv3x3(&a, nullptr) != -22) { printf("x_v of another form not refused\n"); ++failures; }
}

int main() {
    if (bsvd_abi_version() != BSVD_ABI_VERSION || bsvd_conv_args_size() != (int)sizeof(BsvdConvArgs)) { printf("ABI mismatch\n"); return 1; }
    int n = 0; HIP_OK(hipGetDeviceCount(&n)); if (n < 1) { printf("no HIP device\n"); return 1; }
    HIP_OK(hipSetDevice(0));
    case_tsm(); case_stride2(); case_pixel_shuffle(); case_split_chain(); case_fused_entry(); case_batch_and_graph(); case_stream_ring_graphs(); case_winograd(); case_fused_pair(); case_transformed_handover();
    printf(failures ? "abi_parity: %d FAILED\n" : "abi_parity: all cases ok\n", failures);
    return failures ? 1 : 0;
}
