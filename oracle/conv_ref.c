/* TEST / BASELINE INFRASTRUCTURE — not product, never linked into libsup3r_hip.so.
 *
 * Plain C + OpenMP restatement of the one operation that carries 99.9 % of the
 * generator's arithmetic as TensorFlow executes it: keras Conv2D / Conv3D with
 * padding "valid" on a channels-last fp32 tensor (cross-correlation, kernel
 * layout (k0, k1, k2, C_in, C_out), bias add) — the op the reference reaches
 * through `self.generator.layers[i](x)` in sup3r/models/abstract.py:1157-1165.
 * The layer loop around it (REFLECT pad 3, crop 2, LeakyReLU, skip adds,
 * expansions) stays in numpy (oracle/c_ref.py); nothing is fused, the padded
 * tensor is materialised, the cropped positions are computed — exactly the
 * work TF does (SURVEY.md §8a "as-TF-executes", 474 GMAC per C2 sample).
 *
 * Two uses: (1) a second, independently written CPU implementation the numpy
 * oracle is checked against (tests/test_c_ref.py); (2) the "port" CPU
 * baseline of bench.py, timed on all host cores.
 *
 * Build: oracle/build_ref.py (gcc -O3 -fopenmp).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TB 4 /* output positions along the innermost spatial axis per block */

int s3ref_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* y[n, o0, o1, o2, co] = b[co] + sum_{a,b,c,ci} x[n, o0 s0 + a, o1 s1 + b, o2 s2 + c, ci]
 *                                              * w[a, b, c, ci, co]
 * x: (N, D0, D1, D2, Cin); y: (N, O0, O1, O2, Cout), O = (D - k) / s + 1.
 * A 2-D conv is the case D2 = k2 = 1. */
int s3ref_conv_valid(const float* x, int64_t N, int64_t D0, int64_t D1, int64_t D2, int64_t Cin,
                     const float* w, const float* bias, int k0, int k1, int k2, int s0, int s1,
                     int s2, int64_t Cout, float* y) {
  if (Cout > 1024 || Cout < 1 || Cin < 1) return -1;
  const int64_t O0 = (D0 - k0) / s0 + 1, O1 = (D1 - k1) / s1 + 1, O2 = (D2 - k2) / s2 + 1;
  if (O0 < 1 || O1 < 1 || O2 < 1) return -1;
  const int64_t rows = N * O0 * O1;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t o1 = r % O1, o0 = (r / O1) % O0, n = r / (O1 * O0);
    float acc[TB][1024];
    for (int64_t t0 = 0; t0 < O2; t0 += TB) {
      const int nb = (int)(O2 - t0 < TB ? O2 - t0 : TB);
      for (int p = 0; p < nb; ++p)
        for (int64_t co = 0; co < Cout; ++co) acc[p][co] = bias ? bias[co] : 0.f;
      for (int a = 0; a < k0; ++a)
        for (int b = 0; b < k1; ++b)
          for (int c = 0; c < k2; ++c) {
            const float* wt = w + (((size_t)a * k1 + b) * k2 + c) * Cin * Cout;
            const float* xr = x + ((((size_t)n * D0 + (o0 * s0 + a)) * D1 + (o1 * s1 + b)) * D2 + c) * Cin;
            for (int64_t ci = 0; ci < Cin; ++ci) {
              const float* wr = wt + (size_t)ci * Cout;
              for (int p = 0; p < nb; ++p) {
                const float xv = xr[(size_t)(t0 + p) * s2 * Cin + ci];
                float* ap = acc[p];
#pragma omp simd
                for (int64_t co = 0; co < Cout; ++co) ap[co] += xv * wr[co];
              }
            }
          }
      for (int p = 0; p < nb; ++p)
        memcpy(y + ((size_t)r * O2 + t0 + p) * Cout, acc[p], (size_t)Cout * sizeof(float));
    }
  }
  return 0;
}
