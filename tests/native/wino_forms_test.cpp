// CPU check of bsvd_amd/csrc/wino_forms.h (g++, no HIP): (1) AT [(G g) . (BT d)] equals the 3-tap correlation for random d, g in
// double; (2) the hand-factored input / output transforms equal the matrix forms.  Prints "ok" lines; exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "wino_forms.h"

template <int M> static int check()
{
    using F = bsvd::WinoForm<M>;
    constexpr int A = F::A;
    double worst_id = 0, worst_in = 0, worst_out = 0;
    srand(7 + M);
    for (int it = 0; it < 2000; ++it) {
        double d[A], g[3], u[A], v[A], vf[A], m[A], o[M], of[M];
        for (int i = 0; i < A; ++i) d[i] = rand() / (double)RAND_MAX * 12.0 - 6.0;
        for (int i = 0; i < 3; ++i) g[i] = rand() / (double)RAND_MAX - 0.5;
        for (int x = 0; x < A; ++x) {
            u[x] = 0; v[x] = 0;
            for (int k = 0; k < 3; ++k) u[x] += F::G[x][k] * g[k];
            for (int i = 0; i < A; ++i) v[x] += F::BT[x][i] * d[i];
            m[x] = u[x] * v[x];
        }
        F::input(d, vf);
        for (int x = 0; x < A; ++x) worst_in = fmax(worst_in, fabs(vf[x] - v[x]));
        F::output(m, of);
        for (int j = 0; j < M; ++j) {
            o[j] = 0;
            for (int x = 0; x < A; ++x) o[j] += F::AT[j][x] * m[x];
            worst_out = fmax(worst_out, fabs(of[j] - o[j]));
            const double ref = d[j] * g[0] + d[j + 1] * g[1] + d[j + 2] * g[2];
            worst_id = fmax(worst_id, fabs(o[j] - ref));
        }
    }
    printf("F(%d,3): identity %.2e  input-factoring %.2e  output-factoring %.2e\n", M, worst_id, worst_in, worst_out);
    return (worst_id < 1e-12 && worst_in < 1e-12 && worst_out < 1e-12) ? 0 : 1;
}

int main()
{
    const int bad = check<2>() + check<4>() + check<6>();
    puts(bad ? "FAIL" : "ok");
    return bad;
}
