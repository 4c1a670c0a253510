// wino_forms.h -- the 1-D Winograd forms F(M,3) the split-fp16 wide-layer kernel (conv3x3_wino.hip) runs along x.
//
//   out[y][M q + j] = sum_xi AT[j][xi] * ( sum_{ky, c} U[xi][ky][c] * V[xi][y + ky][q][c] ),
//   V[xi] = sum_i BT[xi][i] * d[M q - 1 + i]   (input transform, A = M + 2 consecutive pixels of a row),
//   U[xi][ky] = sum_kx G[xi][kx] * g[ky][kx]   (weight transform, done once at pack time in double).
//
// F(2,3): the classic points {0, 1, -1, inf}.  F(6,3): see below.  F(4,3): points {0, 1, -1, 1/2, -1/2, inf} -- NOT the usual {.., 2, -2, ..}:
// with half-integer points every entry of BT and AT is a small dyadic number (1/8 .. 5/4) and, carried in fp16 pairs with
// fp32 transforms, the whole bsvd_c64 network comes out in the same error class as the direct 3-pass split kernel
// (4.6e-5 vs 5.0e-5 max-abs against float64; the {+-2} form: 2.0e-4) -- tools/debug/winograd_emul.py.
//
// Plain C++ (no HIP types): included by the kernel, by the pack kernel and by the CPU unit test tests/native/wino_forms_test.cpp,
// which checks the matrices against the defining identity and the hand-factored transforms below against the matrices.
#pragma once

#if defined(__HIPCC__)
#define BSVD_HD __host__ __device__ __forceinline__
#else
#define BSVD_HD inline
#endif

namespace bsvd {

template <int M> struct WinoForm;

template <> struct WinoForm<2> {
    static constexpr int M = 2, A = 4;
    static constexpr double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    static constexpr double BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    static constexpr double AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    template <class T> static BSVD_HD void input(const T (&d)[4], T (&v)[4])
    {
        v[0] = d[0] - d[2];
        v[1] = d[1] + d[2];
        v[2] = d[2] - d[1];
        v[3] = d[1] - d[3];
    }
    template <class T> static BSVD_HD void output(const T (&m)[4], T (&o)[2])
    {
        o[0] = m[0] + m[1] + m[2];
        o[1] = m[1] - m[2] - m[3];
    }
};

template <> struct WinoForm<4> {
    static constexpr int M = 4, A = 6;
    static constexpr double G[6][3] = {{4, 0, 0},
                                       {2.0 / 3, 2.0 / 3, 2.0 / 3},
                                       {2.0 / 3, -2.0 / 3, 2.0 / 3},
                                       {-8.0 / 3, -4.0 / 3, -2.0 / 3},
                                       {-8.0 / 3, 4.0 / 3, -2.0 / 3},
                                       {0, 0, 1}};
    static constexpr double BT[6][6] = {{0.25, 0, -1.25, 0, 1, 0},  {0, -0.25, -0.25, 1, 1, 0}, {0, 0.25, -0.25, -1, 1, 0},
                                        {0, -0.5, -1, 0.5, 1, 0},   {0, 0.5, -1, -0.5, 1, 0},   {0, 0.25, 0, -1.25, 0, 1}};
    static constexpr double AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5, -0.5, 0}, {0, 1, 1, 0.25, 0.25, 0}, {0, 1, -1, 0.125, -0.125, 1}};
    // 12 fused multiply-adds / adds per channel (the matrix form has 20 non-trivial terms)
    template <class T> static BSVD_HD void input(const T (&d)[6], T (&v)[6])
    {
        const T a = d[4] - T(0.25) * d[2];
        const T b = d[3] - T(0.25) * d[1];
        v[1] = a + b;
        v[2] = a - b;
        const T c = d[4] - d[2];
        const T e = d[3] - d[1];
        v[3] = c + T(0.5) * e;
        v[4] = c - T(0.5) * e;
        v[0] = T(0.25) * d[0] + (d[4] - T(1.25) * d[2]);
        v[5] = T(0.25) * d[1] + (d[5] - T(1.25) * d[3]);
    }
    template <class T> static BSVD_HD void output(const T (&m)[6], T (&o)[4])
    {
        const T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        o[0] = m[0] + s12 + s34;
        o[1] = d12 + T(0.5) * d34;
        o[2] = s12 + T(0.25) * s34;
        o[3] = d12 + T(0.125) * d34 + m[5];
    }
};

// F(6,3): points {0, 1, -1, 1/2, -1/2, 3/4, -3/4, inf}.  All entries of BT and AT are dyadic (exact in fp32); the usual {.., 2, -2, ..}
// set gives 5.3e-4 max-abs on the whole network, {.., 1/4, -1/4} 2.7e-4, this one 6.9e-5 (direct: 4.8e-5) -- winograd_emul.py.
template <> struct WinoForm<6> {
    static constexpr int M = 6, A = 8;
    static constexpr double G[8][3] = {{-64.0 / 9, 0, 0},
                                       {32.0 / 21, 32.0 / 21, 32.0 / 21},
                                       {32.0 / 21, -32.0 / 21, 32.0 / 21},
                                       {128.0 / 15, 64.0 / 15, 32.0 / 15},
                                       {128.0 / 15, -64.0 / 15, 32.0 / 15},
                                       {-2048.0 / 315, -512.0 / 105, -128.0 / 35},
                                       {-2048.0 / 315, 512.0 / 105, -128.0 / 35},
                                       {0, 0, 1}};
    static constexpr double BT[8][8] = {{-9.0 / 64, 0, 61.0 / 64, 0, -29.0 / 16, 0, 1, 0},
                                        {0, 9.0 / 64, 9.0 / 64, -13.0 / 16, -13.0 / 16, 1, 1, 0},
                                        {0, -9.0 / 64, 9.0 / 64, 13.0 / 16, -13.0 / 16, -1, 1, 0},
                                        {0, 9.0 / 32, 9.0 / 16, -25.0 / 32, -25.0 / 16, 0.5, 1, 0},
                                        {0, -9.0 / 32, 9.0 / 16, 25.0 / 32, -25.0 / 16, -0.5, 1, 0},
                                        {0, 3.0 / 16, 0.25, -15.0 / 16, -1.25, 0.75, 1, 0},
                                        {0, -3.0 / 16, 0.25, 15.0 / 16, -1.25, -0.75, 1, 0},
                                        {0, -9.0 / 64, 0, 61.0 / 64, 0, -29.0 / 16, 0, 1}};
    static constexpr double AT[6][8] = {{1, 1, 1, 1, 1, 1, 1, 0},
                                        {0, 1, -1, 0.5, -0.5, 0.75, -0.75, 0},
                                        {0, 1, 1, 0.25, 0.25, 9.0 / 16, 9.0 / 16, 0},
                                        {0, 1, -1, 0.125, -0.125, 27.0 / 64, -27.0 / 64, 0},
                                        {0, 1, 1, 1.0 / 16, 1.0 / 16, 81.0 / 256, 81.0 / 256, 0},
                                        {0, 1, -1, 1.0 / 32, -1.0 / 32, 243.0 / 1024, -243.0 / 1024, 1}};
    // even / odd parts per +-point pair: 27 fused multiply-adds / adds per channel
    template <class T> static BSVD_HD void input(const T (&d)[8], T (&v)[8])
    {
        T e = d[6] + (T(9.0 / 64) * d[2] - T(13.0 / 16) * d[4]);
        T o = d[5] + (T(9.0 / 64) * d[1] - T(13.0 / 16) * d[3]);
        v[1] = e + o;
        v[2] = e - o;
        e = d[6] + (T(9.0 / 16) * d[2] - T(25.0 / 16) * d[4]);
        o = T(0.5) * d[5] + (T(9.0 / 32) * d[1] - T(25.0 / 32) * d[3]);
        v[3] = e + o;
        v[4] = e - o;
        e = d[6] + (T(0.25) * d[2] - T(1.25) * d[4]);
        o = T(0.75) * d[5] + (T(3.0 / 16) * d[1] - T(15.0 / 16) * d[3]);
        v[5] = e + o;
        v[6] = e - o;
        v[0] = d[6] + (T(61.0 / 64) * d[2] - T(9.0 / 64) * d[0] - T(29.0 / 16) * d[4]);
        v[7] = d[7] + (T(61.0 / 64) * d[3] - T(9.0 / 64) * d[1] - T(29.0 / 16) * d[5]);
    }
    template <class T> static BSVD_HD void output(const T (&m)[8], T (&o)[6])
    {
        const T s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4], s3 = m[5] + m[6], d3 = m[5] - m[6];
        o[0] = m[0] + s1 + s2 + s3;
        o[1] = d1 + T(0.5) * d2 + T(0.75) * d3;
        o[2] = s1 + T(0.25) * s2 + T(9.0 / 16) * s3;
        o[3] = d1 + T(0.125) * d2 + T(27.0 / 64) * d3;
        o[4] = s1 + T(1.0 / 16) * s2 + T(81.0 / 256) * s3;
        o[5] = d1 + T(1.0 / 32) * d2 + T(243.0 / 1024) * d3 + m[7];
    }
};

}  // namespace bsvd
