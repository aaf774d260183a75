// Generic direct convolution (fp32 VALU) for the HBM-/latency-bound layers of
// the Sup3rGan path: low-channel convs (C_in <= 8 or C_out <= 2: generator head
// / tail, discriminator entry), strided discriminator convs, SAME/VALID
// padding, 2-D nets (t = 1).  NDHWC, reflect / zero padding evaluated as index
// math at load time (never materialised), bias + activation + residual +
// depth-to-space fused into the store.
//
// Mapping (wave64): one wave = 64 consecutive output positions x CO_T output
// channels.  The filter block w[tap][ci][co0 .. co0+CO_T) is identical for
// every lane of the wave -> address is wave-uniform (readfirstlane) so the
// compiler emits scalar (SMEM) loads; x is read per lane, vectorised over
// C_in when C_in % 4 == 0.  No MFMA here on purpose: these layers have
// arithmetic intensity <= ~20 FLOP/B and are bound by HBM / L2, not math.
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace {

__device__ inline float act_f(float v, int act, float alpha) {
  // one select for every kind (slope 1 = identity, 0 = ReLU, alpha = Leaky):
  // testing the kind per element compiles to two scalar branches per value
  const float s = act == S3_ACT_LEAKY ? alpha : (act == S3_ACT_RELU ? 0.f : 1.f);
  return v > 0.f ? v : s * v;
}

__device__ inline int src_index(int o, int t, int stride, int lo, int n,
                                int pad_mode, bool& valid) {
  int i = o * stride + t - lo;
  if (pad_mode == S3_PAD_REFLECT) {
    i = s3_reflect(i, n);
    // ragged tiles never index outside: clamp is a no-op for legal plans
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  } else if (i < 0 || i >= n) {
    valid = false;
    i = 0;
  }
  return i;
}

template <int CO_T, int CI_V>
__global__ __launch_bounds__(256) void conv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ res,
    void* __restrict__ yv, ConvGeom g, int n_cgroups, int out_bf16) {
  float* __restrict__ y = reinterpret_cast<float*>(yv);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t wave_task = (int64_t)blockIdx.x * 4 + wave;
  const int64_t chunk = wave_task / n_cgroups;
  const int cg = (int)(wave_task % n_cgroups);
  const int co0 = cg * CO_T;
  const int64_t pos = chunk * 64 + lane;
  if (chunk * 64 >= P) return;
  const bool live = pos < P;
  int64_t r = live ? pos : P - 1;
  const int o2 = (int)(r % g.O[2]); r /= g.O[2];
  const int o1 = (int)(r % g.O[1]); r /= g.O[1];
  const int o0 = (int)(r % g.O[0]); r /= g.O[0];
  const int n = (int)r;

  float acc[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) acc[j] = 0.f;

  const int Cin = g.Cin, Cout = g.Cout;
  for (int a = 0; a < g.k[0]; ++a) {
    bool v0 = true;
    const int i0 = src_index(o0, a, g.s[0], g.lo[0], g.D[0], g.pad_mode, v0);
    for (int b = 0; b < g.k[1]; ++b) {
      bool v1 = v0;
      const int i1 = src_index(o1, b, g.s[1], g.lo[1], g.D[1], g.pad_mode, v1);
      for (int c = 0; c < g.k[2]; ++c) {
        bool v2 = v1;
        const int i2 = src_index(o2, c, g.s[2], g.lo[2], g.D[2], g.pad_mode, v2);
        const float* xp = x + ((((int64_t)n * g.D[0] + i0) * g.D[1] + i1) *
                                   g.D[2] + i2) * Cin;
        const float* wp = w + (int64_t)((a * g.k[1] + b) * g.k[2] + c) * Cin * Cout + co0;
        const float m = v2 ? 1.f : 0.f;
        for (int ci = 0; ci < Cin; ci += CI_V) {
          float xv[CI_V];
          if (CI_V == 4) {
            float4 t = *reinterpret_cast<const float4*>(xp + ci);
            xv[0] = t.x * m; xv[1] = t.y * m; xv[2] = t.z * m; xv[3] = t.w * m;
          } else if (CI_V == 2) {
            float2 t = *reinterpret_cast<const float2*>(xp + ci);
            xv[0] = t.x * m; xv[1] = t.y * m;
          } else {
            xv[0] = xp[ci] * m;
          }
#pragma unroll
          for (int q = 0; q < CI_V; ++q) {
            const float* wr = wp + (int64_t)(ci + q) * Cout;
#pragma unroll
            for (int j = 0; j < CO_T; ++j) {
              float wv = (co0 + j < Cout) ? wr[j] : 0.f;
              acc[j] = fmaf(xv[q], wv, acc[j]);
            }
          }
        }
      }
    }
  }
  if (!live) return;
  // epilogue: bias, [d2s permutation], act, residual
  const int b = g.d2s;
  const int cpo = Cout / (b * b);  // channels after depth-to-space
#pragma unroll
  for (int j = 0; j < CO_T; ++j) {
    const int co = co0 + j;
    if (co >= Cout) break;
    float v = acc[j] + (bias ? bias[co] : 0.f);
    int64_t dst;
    if (b == 1) {
      dst = pos * Cout + co;
    } else {
      const int blk = co / cpo, cc = co % cpo;
      dst = ((((int64_t)n * g.O[0] * b + o0 * b + blk / b) * (g.O[1] * b) +
              o1 * b + blk % b) * g.O[2] + o2) * cpo + cc;
    }
    v = act_f(v, g.act, g.alpha);
    if (res) v += res[dst];
    if (out_bf16) {
      unsigned u = __float_as_uint(v);
      u += 0x7FFFu + ((u >> 16) & 1u);
      reinterpret_cast<unsigned short*>(yv)[dst] = (unsigned short)(u >> 16);
    } else {
      y[dst] = v;
    }
  }
}

// ---- small-channel 3x3x3 stride-1 conv (generator tail 8->2 on the hi-res
// grid: AI ~ 21 FLOP/B, HBM-bound).  One thread owns TT consecutive outputs
// along t at one (s1, s2): for each of the 9 (a, b) neighbour columns it loads
// the TT+2 input cells once (32-B cells, contiguous along t -> coalesced 16-B
// loads, adjacent lanes overlap in L1) and slides the 3 t-taps over them in
// registers; the 27*CIN*COUT filter lives in LDS and is read as wave-uniform
// broadcasts.  Index / boundary math is paid once per column, not per tap.
template <int CIN, int COUT, int TT, bool IN16>
__global__ __launch_bounds__(256) void conv_small_kernel(
    const void* __restrict__ xv, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, ConvGeom g) {
  static_assert(!IN16 || CIN == 8, "bf16 cells are one 16-B load of 8 channels");
  __shared__ __attribute__((aligned(16))) float ws[27 * CIN * COUT];
  for (int i = threadIdx.x; i < 27 * CIN * COUT; i += 256) ws[i] = w[i];
  __syncthreads();
  const int chunks2 = (g.O[2] + TT - 1) / TT;
  const int64_t total = (int64_t)g.N * g.O[0] * g.O[1] * chunks2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int64_t r = idx;
  const int tc = (int)(r % chunks2); r /= chunks2;
  const int o1 = (int)(r % g.O[1]); r /= g.O[1];
  const int o0 = (int)(r % g.O[0]); r /= g.O[0];
  const int n = (int)r;
  const int t0 = tc * TT;
  float acc[TT][COUT];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[t][co] = bias ? bias[co] : 0.f;
  // source t indices of the TT+2 cells of a column (shared by all 9 columns)
  int i2s[TT + 2];
  bool v2s[TT + 2];
#pragma unroll
  for (int q = 0; q < TT + 2; ++q) {
    bool v = true;
    i2s[q] = src_index(t0 + q, 0, 1, g.lo[2], g.D[2], g.pad_mode, v);
    v2s[q] = v;
  }
  for (int a = 0; a < 3; ++a) {
    bool v0 = true;
    const int i0 = src_index(o0, a, 1, g.lo[0], g.D[0], g.pad_mode, v0);
    for (int b = 0; b < 3; ++b) {
      bool v1 = v0;
      const int i1 = src_index(o1, b, 1, g.lo[1], g.D[1], g.pad_mode, v1);
      const int64_t col = (((int64_t)n * g.D[0] + i0) * g.D[1] + i1) * (int64_t)g.D[2] * CIN;
      float xc[TT + 2][CIN];
#pragma unroll
      for (int q = 0; q < TT + 2; ++q) {
        const float m = (v1 && v2s[q]) ? 1.f : 0.f;
        if (IN16) {
          const uint4 v = *reinterpret_cast<const uint4*>(
              reinterpret_cast<const unsigned short*>(xv) + col + (int64_t)i2s[q] * CIN);
          const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xc[q][(2 * e) % CIN] = __uint_as_float(u[e] << 16) * m;
            xc[q][(2 * e + 1) % CIN] = __uint_as_float(u[e] & 0xFFFF0000u) * m;
          }
        } else {
          const float* p = reinterpret_cast<const float*>(xv) + col + (int64_t)i2s[q] * CIN;
#pragma unroll
          for (int ci = 0; ci < CIN; ci += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + ci);
            xc[q][ci] = v.x * m; xc[q][ci + 1] = v.y * m;
            xc[q][ci + 2] = v.z * m; xc[q][ci + 3] = v.w * m;
          }
        }
      }
      const float* wt = ws + (a * 3 + b) * 3 * CIN * COUT;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float wv = wt[(c * CIN + ci) * COUT + co];
#pragma unroll
            for (int t = 0; t < TT; ++t)
              acc[t][co] = fmaf(xc[t + c][ci], wv, acc[t][co]);
          }
        }
      }
    }
  }
  const int64_t base = ((((int64_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + t0) * COUT;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    if (t0 + t >= g.O[2]) break;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      y[base + t * COUT + co] = act_f(acc[t][co], g.act, g.alpha);
  }
}

// ---- dgrad: dX[n,i,ci] = sum over pre-images q of i under the (virtual)
// padding, taps k, co:  dY[n,(q+lo-k)/s,co] * W[k][ci][co]
template <int CI_T>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(
    const float* __restrict__ dy, const float* __restrict__ w,
    float* __restrict__ dx, ConvGeom g, int n_cgroups) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t P = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  const int64_t wave_task = (int64_t)blockIdx.x * 4 + wave;
  const int64_t chunk = wave_task / n_cgroups;
  const int cg = (int)(wave_task % n_cgroups);
  const int ci0 = cg * CI_T;
  const int64_t pos = chunk * 64 + lane;
  if (chunk * 64 >= P) return;
  const bool live = pos < P;
  int64_t r = live ? pos : P - 1;
  int ii[3];
  ii[2] = (int)(r % g.D[2]); r /= g.D[2];
  ii[1] = (int)(r % g.D[1]); r /= g.D[1];
  ii[0] = (int)(r % g.D[0]); r /= g.D[0];
  const int n = (int)r;
  const int Cin = g.Cin, Cout = g.Cout;

  // candidate virtual (padded-frame) coordinates q with source(q) == i
  int cand[3][3], cnt[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    cnt[d] = 0;
    cand[d][cnt[d]++] = ii[d];
    if (g.pad_mode == S3_PAD_REFLECT) {
      const int nI = g.D[d];
      const int vmax = (g.O[d] - 1) * g.s[d] + g.k[d] - 1 - g.lo[d];
      if (ii[d] >= 1 && -ii[d] >= -g.lo[d]) cand[d][cnt[d]++] = -ii[d];
      const int mq = 2 * (nI - 1) - ii[d];
      if (ii[d] <= nI - 2 && mq <= vmax && mq >= nI) cand[d][cnt[d]++] = mq;
    }
  }
  float acc[CI_T];
#pragma unroll
  for (int j = 0; j < CI_T; ++j) acc[j] = 0.f;

  for (int e0 = 0; e0 < cnt[0]; ++e0)
    for (int a = 0; a < g.k[0]; ++a) {
      int t0 = cand[0][e0] + g.lo[0] - a;
      if (t0 < 0 || t0 % g.s[0] != 0) continue;
      t0 /= g.s[0];
      if (t0 >= g.O[0]) continue;
      for (int e1 = 0; e1 < cnt[1]; ++e1)
        for (int b = 0; b < g.k[1]; ++b) {
          int t1 = cand[1][e1] + g.lo[1] - b;
          if (t1 < 0 || t1 % g.s[1] != 0) continue;
          t1 /= g.s[1];
          if (t1 >= g.O[1]) continue;
          for (int e2 = 0; e2 < cnt[2]; ++e2)
            for (int c = 0; c < g.k[2]; ++c) {
              int t2 = cand[2][e2] + g.lo[2] - c;
              if (t2 < 0 || t2 % g.s[2] != 0) continue;
              t2 /= g.s[2];
              if (t2 >= g.O[2]) continue;
              const float* dyp = dy + ((((int64_t)n * g.O[0] + t0) * g.O[1] + t1) *
                                           g.O[2] + t2) * Cout;
              const float* wp = w + ((int64_t)((a * g.k[1] + b) * g.k[2] + c) * Cin + ci0) * Cout;
              for (int co = 0; co < Cout; ++co) {
                const float d = dyp[co];
#pragma unroll
                for (int j = 0; j < CI_T; ++j) {
                  float wv = (ci0 + j < Cin) ? wp[(int64_t)j * Cout + co] : 0.f;
                  acc[j] = fmaf(d, wv, acc[j]);
                }
              }
            }
        }
    }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < CI_T; ++j)
    if (ci0 + j < Cin) dx[pos * Cin + ci0 + j] = acc[j];
}

// ---- wgrad: dW[tap][ci][co] = sum_{n,o} Xp[n, o*s + k - lo][ci] * dY[n,o][co]
// block = (position slab, tap, 64x64 (ci,co) tile); x / dy slabs of 32
// positions staged through LDS; each thread owns a 4x4 (ci,co) register block.
// Partials per slab go to scratch and are summed in fixed order (deterministic,
// identical on every rank).
constexpr int WG_TILE = 64;
constexpr int WG_POS = 32;

__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int n_slabs, int tiles_ci,
    int tiles_co) {
  __shared__ float xs[WG_POS][WG_TILE + 4];
  __shared__ float ds[WG_POS][WG_TILE + 4];
  const int slab = blockIdx.x;
  const int tap = blockIdx.y;
  const int tile = blockIdx.z;
  const int tci = (tile / tiles_co) * WG_TILE, tco = (tile % tiles_co) * WG_TILE;
  const int a = tap / (g.k[1] * g.k[2]), b = (tap / g.k[2]) % g.k[1], c = tap % g.k[2];
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t per = (P + n_slabs - 1) / n_slabs;
  const int64_t p_begin = (int64_t)slab * per;
  const int64_t p_end = p_begin + per < P ? p_begin + per : P;
  const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;  // 16x16 threads
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  // the source cell of each of the step's 32 positions, decoded ONCE per
  // position (round 5: every one of its 64 channel lanes used to redo the three
  // divisions and the boundary rule — the kernel was bound by that index math
  // on the few-channel head / output convs that still come here)
  __shared__ int64_t xcell[WG_POS];       // input cell index, -1: outside / past the slab
  for (int64_t p0 = p_begin; p0 < p_end; p0 += WG_POS) {
    if (threadIdx.x < WG_POS) {
      const int64_t pos = p0 + threadIdx.x;
      int64_t cell = -1;
      if (pos < p_end) {
        int64_t r = pos;
        const int o2 = (int)(r % g.O[2]); r /= g.O[2];
        const int o1 = (int)(r % g.O[1]); r /= g.O[1];
        const int o0 = (int)(r % g.O[0]); r /= g.O[0];
        const int n = (int)r;
        bool v = true;
        const int i0 = src_index(o0, a, g.s[0], g.lo[0], g.D[0], g.pad_mode, v);
        const int i1 = src_index(o1, b, g.s[1], g.lo[1], g.D[1], g.pad_mode, v);
        const int i2 = src_index(o2, c, g.s[2], g.lo[2], g.D[2], g.pad_mode, v);
        if (v) cell = (((int64_t)n * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + i2;
      }
      xcell[threadIdx.x] = cell;
    }
    __syncthreads();
    // stage: 32 positions x 64 channels each for x (tap-shifted) and dy
    for (int e = threadIdx.x; e < WG_POS * WG_TILE; e += 256) {
      const int pr = e / WG_TILE, ch = e % WG_TILE;
      const int64_t pos = p0 + pr;
      float xv = 0.f, dv = 0.f;
      if (pos < p_end) {
        const int64_t cell = xcell[pr];
        if (cell >= 0 && tci + ch < g.Cin) xv = x[cell * g.Cin + tci + ch];
        if (tco + ch < g.Cout) dv = dy[pos * g.Cout + tco + ch];
      }
      xs[pr][ch] = xv;
      ds[pr][ch] = dv;
    }
    __syncthreads();
#pragma unroll 4
    for (int pr = 0; pr < WG_POS; ++pr) {
      float4 xv = *reinterpret_cast<const float4*>(&xs[pr][ty * 4]);
      float4 dv = *reinterpret_cast<const float4*>(&ds[pr][tx * 4]);
      const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
      const float da[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xa[i], da[j], acc[i][j]);
    }
    __syncthreads();
  }
  // partial[slab][tap][ci][co]
  float* out = partial + ((int64_t)slab * (g.k[0] * g.k[1] * g.k[2]) + tap) * g.Cin * g.Cout;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = tci + ty * 4 + i;
    if (ci >= g.Cin) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = tco + tx * 4 + j;
      if (co < g.Cout) out[(int64_t)ci * g.Cout + co] = acc[i][j];
    }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial,
                                    int n_slabs, int64_t wsize,
                                    float* __restrict__ dw, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < wsize;
       i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < n_slabs; ++s) t += partial[(int64_t)s * wsize + i];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// first level of a long reduction: row j of out = sum of the slabs [32 j, 32 j + 32)
__global__ void wgrad_reduce_group_kernel(const float* __restrict__ partial, int n_slabs, int64_t wsize,
                                          float* __restrict__ out) {
  const int s0 = blockIdx.y * 32, s1 = s0 + 32 < n_slabs ? s0 + 32 : n_slabs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < wsize;
       i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = s0; s < s1; ++s) t += partial[(int64_t)s * wsize + i];
    out[(int64_t)blockIdx.y * wsize + i] = t;
  }
}

int wgrad_slabs(const ConvGeom& g) {
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  int64_t s = (P + 2047) / 2048;
  if (s > 64) s = 64;
  // Few filter elements over many positions (the 2 -> 64 / 64 -> 2 head and
  // output convs of the 2-D generators: 1 152 elements, 270 000 – 1 080 000
  // positions): 64 slabs x 9 taps are 576 workgroups walking 4 200 – 17 000
  // positions each, 830 us.  More slabs while the partials stay small (32 MB).
  const int64_t wbytes = (int64_t)g.k[0] * g.k[1] * g.k[2] * g.Cin * g.Cout * 4;
  int64_t more = (P + 511) / 512;
  if (more > 2048) more = 2048;
  if (more * wbytes > ((int64_t)32 << 20)) more = ((int64_t)32 << 20) / wbytes;
  if (more > s) s = more;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

// the sliding-window kernel covers 3x3x3 stride-1 convs with C_in in {4, 8}
// and C_out == 2 (generator tails); C_in == 8 may also be read as bf16
bool conv_small_supported(const ConvGeom& g, int in_bf16) {
  if (!(g.d2s == 1 && g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3 && g.s[0] == 1 &&
        g.s[1] == 1 && g.s[2] == 1 && g.Cout == 2 && g.O[2] >= 8))
    return false;
  return in_bf16 ? g.Cin == 8 : (g.Cin == 8 || g.Cin == 4);
}

int launch_conv_generic_fwd(s3_ctx* ctx, const ConvGeom& g, const void* x,
                            const float* w, const float* bias,
                            const float* res, void* y, int out_bf16,
                            int in_bf16) {
  if (in_bf16 && !out_bf16 && !res && conv_tail_mfma_supported(g) &&
      !s3_opt_has(S3O_NO_TAIL_MFMA))
    return launch_conv_tail_mfma(ctx, g, x, w, bias, (float*)y);
  if (!out_bf16 && !res && conv_small_supported(g, in_bf16)) {
    constexpr int TT = 4;
    const int64_t total = (int64_t)g.N * g.O[0] * g.O[1] * ((g.O[2] + TT - 1) / TT);
    dim3 gridS((unsigned)((total + 255) / 256)), blockS(256);
    if (in_bf16)
      hipLaunchKernelGGL((conv_small_kernel<8, 2, TT, true>), gridS, blockS, 0, ctx->stream, x, w, bias, (float*)y, g);
    else if (g.Cin == 8)
      hipLaunchKernelGGL((conv_small_kernel<8, 2, TT, false>), gridS, blockS, 0, ctx->stream, x, w, bias, (float*)y, g);
    else
      hipLaunchKernelGGL((conv_small_kernel<4, 2, TT, false>), gridS, blockS, 0, ctx->stream, x, w, bias, (float*)y, g);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (in_bf16) S3_FAIL(ctx, S3_EINVAL, "direct conv: bf16 input not supported for this geometry");
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  int co_t = g.Cout >= 16 ? 16 : (g.Cout >= 8 ? 8 : (g.Cout >= 4 ? 4 : (g.Cout >= 2 ? 2 : 1)));
  int n_cg = (g.Cout + co_t - 1) / co_t;
  int64_t tasks = ((P + 63) / 64) * n_cg;
  dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
  int civ = (g.Cin % 4 == 0) ? 4 : ((g.Cin % 2 == 0) ? 2 : 1);
#define S3_LAUNCH_FWD(CO, CI)                                                 \
  hipLaunchKernelGGL((conv_fwd_kernel<CO, CI>), grid, block, 0, ctx->stream, \
                     (const float*)x, w, bias, res, y, g, n_cg, out_bf16)
#define S3_FWD_CI(CO)                                  \
  if (civ == 4) S3_LAUNCH_FWD(CO, 4);                  \
  else if (civ == 2) S3_LAUNCH_FWD(CO, 2);             \
  else S3_LAUNCH_FWD(CO, 1)
  switch (co_t) {
    case 16: S3_FWD_CI(16); break;
    case 8: S3_FWD_CI(8); break;
    case 4: S3_FWD_CI(4); break;
    case 2: S3_FWD_CI(2); break;
    default: S3_FWD_CI(1); break;
  }
#undef S3_FWD_CI
#undef S3_LAUNCH_FWD
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_generic_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy,
                              const float* w, float* dx) {
  const int64_t P = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  if (s3_opt_has(S3O_TRACE))
    fprintf(stderr, "[dgrad] N=%d D=%dx%dx%d O=%dx%dx%d Cin=%d Cout=%d k=%d%d%d s=%d%d%d\n", g.N, g.D[0], g.D[1], g.D[2], g.O[0], g.O[1], g.O[2], g.Cin, g.Cout, g.k[0], g.k[1], g.k[2], g.s[0], g.s[1], g.s[2]);
  int ci_t = g.Cin >= 8 ? 8 : (g.Cin >= 4 ? 4 : (g.Cin >= 2 ? 2 : 1));
  int n_cg = (g.Cin + ci_t - 1) / ci_t;
  int64_t tasks = ((P + 63) / 64) * n_cg;
  dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
  switch (ci_t) {
    case 8: hipLaunchKernelGGL(conv_dgrad_kernel<8>, grid, block, 0, ctx->stream, dy, w, dx, g, n_cg); break;
    case 4: hipLaunchKernelGGL(conv_dgrad_kernel<4>, grid, block, 0, ctx->stream, dy, w, dx, g, n_cg); break;
    case 2: hipLaunchKernelGGL(conv_dgrad_kernel<2>, grid, block, 0, ctx->stream, dy, w, dx, g, n_cg); break;
    default: hipLaunchKernelGGL(conv_dgrad_kernel<1>, grid, block, 0, ctx->stream, dy, w, dx, g, n_cg); break;
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

size_t conv_generic_wgrad_partial_bytes(const ConvGeom& g) {
  return (size_t)wgrad_slabs(g) * g.k[0] * g.k[1] * g.k[2] * g.Cin * g.Cout * sizeof(float);
}

int launch_conv_generic_wgrad(s3_ctx* ctx, const ConvGeom& g, const float* x,
                              const float* dy, float* dw, float* partial,
                              size_t partial_bytes, int accumulate) {
  const int n_slabs = wgrad_slabs(g);
  if (s3_opt_has(S3O_TRACE))
    fprintf(stderr, "[wgrad] N=%d D=%dx%dx%d O=%dx%dx%d Cin=%d Cout=%d k=%d%d%d s=%d%d%d\n", g.N, g.D[0], g.D[1], g.D[2], g.O[0], g.O[1], g.O[2], g.Cin, g.Cout, g.k[0], g.k[1], g.k[2], g.s[0], g.s[1], g.s[2]);
  if (partial_bytes < conv_generic_wgrad_partial_bytes(g))
    S3_FAIL(ctx, S3_EINVAL, "wgrad: partial buffer too small");
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int tci = (g.Cin + WG_TILE - 1) / WG_TILE, tco = (g.Cout + WG_TILE - 1) / WG_TILE;
  dim3 grid(n_slabs, taps, tci * tco), block(256);
  hipLaunchKernelGGL(conv_wgrad_kernel, grid, block, 0, ctx->stream, x, dy, partial, g, n_slabs, tci, tco);
  const int64_t wsize = (int64_t)taps * g.Cin * g.Cout;
  int rg = (int)((wsize + 255) / 256);
  if (rg > 2048) rg = 2048;
  if (n_slabs > 128) {
    // (hundreds of slabs of a few-element filter: one thread per element
    // walking them all was 77 us; two levels of 32)
    const int ng = (n_slabs + 31) / 32;
    int rc = ensure_scratch(ctx, (size_t)ng * wsize * sizeof(float));
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(rg, ng), dim3(256), 0, ctx->stream, partial, n_slabs, wsize,
                       ctx->scratch);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rg), dim3(256), 0, ctx->stream, (const float*)ctx->scratch, ng, wsize,
                       dw, accumulate);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rg), dim3(256), 0, ctx->stream, partial, n_slabs, wsize, dw, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
