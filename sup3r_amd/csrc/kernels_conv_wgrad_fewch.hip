// Weight gradient of the 2-channel hi-res conv that opens the discriminator
// (C_in = 2 -> C_out = 32 over N x 78 x 78 x 286 positions at C2) on bf16 MFMA,
// without LDS staging — S3_PREC_BF16 training plans.
//
//   dW[m = tap * 2 + ci][co] = sum_p X[p * s + tap][ci] * dPre[p][co]
//
// M = 54 (4 blocks of 16), N = C_out, contraction over positions.  The layer is
// bound by reading dPre once (128 B per position) — the kernel's job is to do
// exactly that: every WAVE owns the whole M x N gradient (4 x NB accumulators)
// and walks k-steps of 32 consecutive t of one (n, s1, s2) row, so each operand
// element is loaded once: lane (i, kg) of the A fragment loads the float2 cells
// of its own tap at t = 8 kg .. 8 kg + 7 (lanes 2 tap, 2 tap + 1 share them),
// lane (j, kg) of the B fragment 8 dPre values of its channel; immediates carry
// the t offsets.  Rows m >= 54 and t >= O2 are masked through the B operand.
// Waves of a workgroup are summed in LDS; one partial per workgroup, reduced in
// fixed order.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

constexpr int C2_WAVES = 4;

template <int CIN, int NB>
__global__ __launch_bounds__(C2_WAVES * 64) void conv_wgrad_c2_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int chunks, int64_t n_steps) {
  constexpr int MT = 27 * CIN;                  // rows of the gradient
  constexpr int MB = (MT + 15) / 16;
  __shared__ float red[C2_WAVES][MB * NB][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int S1 = g.D[1], S2 = g.D[2];
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2], Cout = g.Cout;
  int tinfo[MB];                                // (ta, tb, tc, ci) of row m, -1 = padding row
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + i;
    const int tap = m / CIN;
    const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
    tinfo[mb] = m < MT ? (ta | (tb << 4) | (tc << 8) | ((m % CIN) << 12)) : -1;
  }
  const int D0 = g.D[0];
  const bool reflect = g.pad_mode == S3_PAD_REFLECT;
  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t n_waves = (int64_t)gridDim.x * C2_WAVES;
  for (int64_t step = (int64_t)blockIdx.x * C2_WAVES + wave; step < n_steps; step += n_waves) {
    int64_t row = step / chunks;
    const int t0 = (int)(step % chunks) * 32 + kg * 8;
    const int o1 = (int)(row % O1); row /= O1;
    const int o0 = (int)(row % O0); row /= O0;
    const int n = (int)row;
    // clamp the lane's first t so that the 8 loads stay inside the row; the
    // shifted-out positions are masked through dPre
    const int tl = t0 + 8 <= O2 ? t0 : (O2 - 8 > 0 ? O2 - 8 : 0);
    const int shift = t0 - tl;                 // 0, or how many of the 8 slots repeat earlier t
    const float* dr = dy + ((((int64_t)n * O0 + o0) * O1 + o1) * O2 + tl) * Cout;
    const int xstep = CIN * g.s[2];
    bf16x8 bfr[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int co = nb * 16 + i;
        const float t = co < Cout ? dr[(int64_t)e * Cout + co] : 0.f;
        // slot e holds t = tl + e; it belongs to this k-step iff e >= shift
        v[e] = (e >= shift && tl + e < O2) ? t : 0.f;
      }
      const uint4 u = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
      bfr[nb] = __builtin_bit_cast(bf16x8, u);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float v[8];
      {
        // this lane's tap: source row (i0, i1) under the (virtual) padding,
        // then 8 cells along t — linear when they are all inside the row
        const int ti = tinfo[mb];
        const int ta = ti & 15, tb = (ti >> 4) & 15, tc = (ti >> 8) & 15, ci = (ti >> 12) & 15;
        int i0 = o0 * g.s[0] + ta - g.lo[0], i1 = o1 * g.s[1] + tb - g.lo[1];
        if (reflect) { i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, S1); }
        const bool rok = ti >= 0 && i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < S1;
        i0 = i0 < 0 ? 0 : (i0 > D0 - 1 ? D0 - 1 : i0);
        i1 = i1 < 0 ? 0 : (i1 > S1 - 1 ? S1 - 1 : i1);
        const float* xrow = x + (((int64_t)n * D0 + i0) * S1 + i1) * S2 * CIN + ci;
        const int t_first = tl * g.s[2] + tc - g.lo[2];
        if (t_first >= 0 && t_first + 7 * g.s[2] < S2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rok ? xrow[(int64_t)t_first * CIN + e * xstep] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            int i2 = t_first + e * g.s[2];
            if (reflect) i2 = s3_reflect(i2, S2);
            const bool ok = rok && i2 >= 0 && i2 < S2;
            i2 = i2 < 0 ? 0 : (i2 > S2 - 1 ? S2 - 1 : i2);
            v[e] = ok ? xrow[(int64_t)i2 * CIN] : 0.f;
          }
        }
      }
      const uint4 u = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
      const bf16x8 afr = __builtin_bit_cast(bf16x8, u);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nb], acc[mb][nb], 0, 0, 0);
    }
  }
  // ---- sum the workgroup's waves, write partial[bid][m][co]
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][mb * NB + nb][lane][r] = acc[mb][nb][r];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * MT * Cout;
  for (int item = threadIdx.x; item < MB * NB * 64 * 4; item += C2_WAVES * 64) {
    const int r = item & 3, ln = (item >> 2) & 63, f = item >> 8;
    const int mb = f / NB, nb = f % NB;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < C2_WAVES; ++w) t += red[w][f][ln][r];
    const int m = mb * 16 + (ln >> 4) * 4 + r, co = nb * 16 + (ln & 15);
    if (m < MT && co < Cout) out[(size_t)m * Cout + co] = t;
  }
}

__global__ void wgrad_c2_partial_reduce(const float* __restrict__ partial, int n_part,
                                        int wsize, float* __restrict__ dw, int accumulate) {
  // one workgroup per 64 elements, 4 lanes-groups split the partials
  __shared__ float s[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float t = 0.f;
  if (e < wsize)
    for (int p = q; p < n_part; p += 4) t += partial[(size_t)p * wsize + e];
  s[q][threadIdx.x & 63] = t;
  __syncthreads();
  if (q == 0 && e < wsize) {
    const float v = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
    dw[e] = accumulate ? dw[e] + v : v;
  }
}

int c2_grid(const s3_ctx* ctx, int64_t n_steps) {
  int64_t grid = 4 * (int64_t)ctx->num_cu;
  const int64_t need = (n_steps + C2_WAVES - 1) / C2_WAVES;
  if (grid > need) grid = need;
  return (int)(grid < 1 ? 1 : grid);
}

int64_t c2_steps(const ConvGeom& g, int* chunks) {
  *chunks = (g.O[2] + 31) / 32;
  return (int64_t)g.N * g.O[0] * g.O[1] * *chunks;
}

}  // namespace

bool conv_wgrad_c2_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || getenv("SUP3R_AMD_NO_WGRAD_C2")) return false;
  // 2 -> 16 / 32 / 64 (discriminator input layer) or 8 -> C_out <= 16 (hi-res tail)
  const bool a = g.Cin == 2 && (g.Cout == 16 || g.Cout == 32 || g.Cout == 64);
  const bool b = g.Cin == 8 && g.Cout <= 16;
  if (!a && !b) return false;
  if (g.d2s != 1) return false;
  for (int d = 0; d < 3; ++d) if (g.k[d] != 3) return false;
  for (int d = 0; d < 3; ++d)
    if (g.lo[d] < 0 || g.lo[d] > 2) return false;
  return g.O[2] >= 8 && (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] >= 4096;
}

size_t conv_wgrad_c2_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  int chunks;
  const int64_t n_steps = c2_steps(g, &chunks);
  return (size_t)c2_grid(ctx, n_steps) * 27 * g.Cin * g.Cout * sizeof(float);
}

int launch_conv_wgrad_c2(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                         float* dw, float* partial, size_t partial_bytes, int accumulate) {
  int chunks;
  const int64_t n_steps = c2_steps(g, &chunks);
  const int grid = c2_grid(ctx, n_steps);
  if (partial_bytes < conv_wgrad_c2_partial_bytes(ctx, g))
    S3_FAIL(ctx, S3_EINVAL, "wgrad_c2: partial buffer too small");
  const int nb = (g.Cout + 15) / 16;
#define S3_C2(C, B)                                                                          \
  hipLaunchKernelGGL((conv_wgrad_c2_kernel<C, B>), dim3(grid), dim3(C2_WAVES * 64), 0, ctx->stream, \
                     x, dy, partial, g, chunks, n_steps)
  if (g.Cin == 8) S3_C2(8, 1);
  else if (nb == 1) S3_C2(2, 1);
  else if (nb == 2) S3_C2(2, 2);
  else S3_C2(2, 4);
#undef S3_C2
  S3_HIP(ctx, hipGetLastError());
  const int wsize = 27 * g.Cin * g.Cout;
  hipLaunchKernelGGL(wgrad_c2_partial_reduce, dim3((wsize + 63) / 64), dim3(256), 0, ctx->stream,
                     partial, grid, wsize, dw, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
