// Few-positions convolutions on the fp32 matrix instruction: ONE launch per
// forward conv / data gradient / weight gradient of the tiny-sample training
// configs (C1: 64 -> 64 3x3 convs over 15 x 5 x 5 = 375 positions).
//
// A mini-batch of that size is a chain of dependent launches, each of which
// costs >= 4.6 us on this part whatever it does, so the
// split-K weight-streaming GEMM of kernels_conv_fewpos.hip (GEMM + epilogue,
// weight gradient + reduce, a filter transpose in front of every data
// gradient: 2 + 2 + 1 launches) pays mostly for its launch count.  Here:
//
//   * fewpos_mfma_kernel<MODE>: a workgroup owns 16 positions x 64 output
//     channels; its eight waves split the K = taps x C reduction in 16-channel
//     chunks (v_mfma_f32_16x16x4_f32: exact fp32 products and sums), sum their
//     partials through LDS in fixed order and apply bias / activation /
//     residual / depth-to-space on the way out.  A operands are 16-B reads of
//     the gathered source cell, B operands 16-B reads of the filter: a lane
//     owns four consecutive output channels (MODE 0) — or, for the data
//     gradient (MODE 1), reads the UNtransposed [tap][ci][co] filter along co,
//     which is the K axis there: no transpose_taps launch.
//   * fewpos_wgrad_mfma_kernel: a workgroup owns (tap, 64 ci, 16 co); its eight
//     waves interleave over the positions (M = ci, N = co, K = positions),
//     LDS sum in fixed order, result (+)= straight into dW: no partial buffer,
//     no reduce launch.  The workgroups of tap 0 / ci tile 0 also leave the
//     bias gradient (column sums of dPre).
//
// Every output element is a fixed-order sum that does not depend on the batch
// size or on which other positions share its workgroup.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_TAPS = 27;
constexpr int NW = 8;          // waves per workgroup (they split the reduction axis)
constexpr int NT = NW * 64;

__device__ inline float actf(float v, int act, float alpha) {
  if (act == S3_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == S3_ACT_LEAKY) return v > 0.f ? v : alpha * v;
  return v;
}

// source cell of row `row` under tap `tap` (-1: padding / no contribution)
// MODE 0: row = output position, source = input cell
// MODE 1: row = input position,  source = output cell that feeds it through the tap
template <int MODE>
__device__ inline int src_cell(const ConvGeom& g, unsigned row, int tap) {
  const int kk[3] = {tap / (g.k[1] * g.k[2]), (tap / g.k[2]) % g.k[1], tap % g.k[2]};
  int p[3], q[3];
  unsigned r = row;
  if (MODE == 0) {
    p[2] = (int)(r % (unsigned)g.O[2]); r /= (unsigned)g.O[2];
    p[1] = (int)(r % (unsigned)g.O[1]); r /= (unsigned)g.O[1];
    p[0] = (int)(r % (unsigned)g.O[0]); r /= (unsigned)g.O[0];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      int i = p[d] * g.s[d] + kk[d] - g.lo[d];
      if (g.pad_mode == S3_PAD_REFLECT) i = s3_reflect(i, g.D[d]);
      if (i < 0 || i >= g.D[d]) return -1;
      q[d] = i;
    }
    return (((int)r * g.D[0] + q[0]) * g.D[1] + q[1]) * g.D[2] + q[2];
  }
  p[2] = (int)(r % (unsigned)g.D[2]); r /= (unsigned)g.D[2];
  p[1] = (int)(r % (unsigned)g.D[1]); r /= (unsigned)g.D[1];
  p[0] = (int)(r % (unsigned)g.D[0]); r /= (unsigned)g.D[0];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int t = p[d] + g.lo[d] - kk[d];
    if (t < 0 || t % g.s[d] != 0) return -1;
    t /= g.s[d];
    if (t >= g.O[d]) return -1;
    q[d] = t;
  }
  return (((int)r * g.O[0] + q[0]) * g.O[1] + q[1]) * g.O[2] + q[2];
}

// y[row][n] = sum_tap sum_k src[cell(row, tap)][k] * B_tap[k][n]
//   MODE 0: B_tap[k][n] = w[tap][k][n]   (K = C_in,  Nc = C_out), epilogue applied
//   MODE 1: B_tap[k][n] = w[tap][n][k]   (K = C_out, Nc = C_in),  plain store
template <int MODE>
__global__ __launch_bounds__(NT) void fewpos_mfma_kernel(
    const float* __restrict__ src, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ res,
    float* __restrict__ y, ConvGeom g, int rows, int K, int Nc,
    const float* __restrict__ mask_y, float slope) {
  // mask_y (MODE 1, nullable): the conv's own fp32 output — src is dL/dy and
  // the activation's adjoint (x 1 where y > 0, x slope elsewhere) is applied
  // to the operand as it is read: no mask pass in front of this kernel
  __shared__ int sidx[MAX_TAPS * 16];
  __shared__ float red[NW][16][64];
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int row0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < taps * 16; i += NT) {
    const int row = row0 + (i & 15);
    sidx[i] = row < rows ? src_cell<MODE>(g, (unsigned)row, i >> 4) : -1;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int kch = K >> 4, nchunks = taps * kch;
  f32x4 acc[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nl = n0 + c * 4;              // this lane's four output channels
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load = [&](int ch, f32x4& a, f32x4 (&b)[4]) {
    const int tap = ch / kch, kb = ch - tap * kch;
    const int s = sidx[tap * 16 + c];
    const int k0 = kb * 16 + q * 4;
    a = s >= 0 ? *reinterpret_cast<const f32x4*>(src + (int64_t)s * K + k0) : zero4;
    if (MODE == 1 && mask_y && s >= 0) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(mask_y + (int64_t)s * K + k0);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= m[e] > 0.f ? 1.f : slope;
    }
    if (MODE == 0) {
      // b[j][nf] = w[tap][k0 + j][nl + nf]
      const float* wp = w + ((int64_t)tap * K + k0) * Nc + nl;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b[j] = nl < Nc ? *reinterpret_cast<const f32x4*>(wp + (int64_t)j * Nc) : zero4;
    } else {
      // b[nf][j] = w[tap][nl + nf][k0 + j]
      const float* wp = w + ((int64_t)tap * Nc + nl) * K + k0;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        b[nf] = nl + nf < Nc ? *reinterpret_cast<const f32x4*>(wp + (int64_t)nf * K) : zero4;
    }
  };
  auto fma16 = [&](const f32x4& a, const f32x4 (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], MODE == 0 ? b[j][nf] : b[nf][j], acc[nf], 0, 0, 0);
  };

  // chunks wave, wave + NW, ...: a ring of three operand sets, so that two
  // chunks' loads are in flight under the sixteen MFMAs of the current one
  // (the kernel is a chain of L2 / HBM round trips, not of MFMAs)
  f32x4 ra[3], rb[3][4];
  int ch = wave;
#pragma unroll
  for (int u = 0; u < 3; ++u)
    if (ch + u * NW < nchunks) load(ch + u * NW, ra[u], rb[u]);
  while (ch < nchunks) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (ch < nchunks) {
        fma16(ra[u], rb[u]);
        const int nx = ch + 3 * NW;
        if (nx < nchunks) load(nx, ra[u], rb[u]);
        ch += NW;
      }
    }
  }

  // acc[nf][i] = C[row q*4+i][channel c*4+nf]
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(&red[wave][q * 4 + i][c * 4]) =
        (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int orow = threadIdx.x >> 4, og = threadIdx.x & 15;
  const int grow = row0 + orow, n = n0 + og * 4;
  if (grow >= rows || n >= Nc) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][orow][og * 4]);
#pragma unroll
  for (int wv = 1; wv < NW; ++wv) v += *reinterpret_cast<const f32x4*>(&red[wv][orow][og * 4]);
  if (MODE == 1) {
    *reinterpret_cast<f32x4*>(y + (int64_t)grow * Nc + n) = v;
    return;
  }
  if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = actf(v[e], g.act, g.alpha);
  const int b = g.d2s;
  if (b <= 1) {
    const int64_t dst = (int64_t)grow * Nc + n;
    if (res) v += *reinterpret_cast<const f32x4*>(res + dst);
    *reinterpret_cast<f32x4*>(y + dst) = v;
    return;
  }
  unsigned r = (unsigned)grow;
  const int o2 = (int)(r % (unsigned)g.O[2]); r /= (unsigned)g.O[2];
  const int o1 = (int)(r % (unsigned)g.O[1]); r /= (unsigned)g.O[1];
  const int o0 = (int)(r % (unsigned)g.O[0]); r /= (unsigned)g.O[0];
  const int cpo = Nc / (b * b);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int co = n + e, blk = co / cpo, cc = co - blk * cpo;
    const int64_t dst = ((((int64_t)r * g.O[0] * b + o0 * b + blk / b) * (g.O[1] * b) +
                          o1 * b + blk % b) * g.O[2] + o2) * cpo + cc;
    float t = v[e];
    if (res) t += res[dst];
    y[dst] = t;
  }
}

// The same with a 16-channel tile: a 16 x 64 x (taps x C) tile is ~1.2 MFLOP =
// 4 600 cycles of one CU's fp32 matrix pipes (256 FLOP / clk), and a C1 layer
// has 24 such tiles for 256 CUs.  Four times the workgroups, a quarter of the
// MFMAs each, and with 8 B of operands per lane and chunk a ring deep enough
// that every load of a wave is in flight at once.
// VK: K is a multiple of 16 (16-B operand reads, no guards); otherwise the last
// chunk of a tap is ragged and the operands are read element by element — the
// few-channel head / tail convs of a network (C_in or C_out 2 ... 8)
template <int MODE, bool VK>
__device__ __forceinline__ void mfma16_body(
    const int bx, const int by, int* __restrict__ sidx /* [MAX_TAPS * 16] */,
    float (*__restrict__ red)[16][16] /* [NW] */,
    const float* __restrict__ src, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ res,
    float* __restrict__ y, const ConvGeom& g, int rows, int K, int Nc,
    const float* __restrict__ mask_y, float slope) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int row0 = bx * 16, n0 = by * 16;
  for (int i = threadIdx.x; i < taps * 16; i += NT) {
    const int row = row0 + (i & 15);
    sidx[i] = row < rows ? src_cell<MODE>(g, (unsigned)row, i >> 4) : -1;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int kch = (K + 15) >> 4, nchunks = taps * kch;
  const int nl = n0 + c;                  // this lane's output channel
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 acc = zero4;

  auto load = [&](int ch, f32x4& a, f32x4& b) {
    const int tap = ch / kch, kb = ch - tap * kch;
    const int s = sidx[tap * 16 + c];
    const int k0 = kb * 16 + q * 4;
    if (VK) {
      a = s >= 0 ? *reinterpret_cast<const f32x4*>(src + (int64_t)s * K + k0) : zero4;
      if (MODE == 1 && mask_y && s >= 0) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mask_y + (int64_t)s * K + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] *= m[e] > 0.f ? 1.f : slope;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = s >= 0 && k0 + e < K;
        float v = in ? src[(int64_t)s * K + k0 + e] : 0.f;
        if (MODE == 1 && mask_y && in) v *= mask_y[(int64_t)s * K + k0 + e] > 0.f ? 1.f : slope;
        a[e] = v;
      }
    }
    if (MODE == 0) {          // b[j] = w[tap][k0 + j][nl]
      const float* wp = w + ((int64_t)tap * K + k0) * Nc + nl;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = (nl < Nc && (VK || k0 + j < K)) ? wp[(int64_t)j * Nc] : 0.f;
    } else if (VK) {          // b[j] = w[tap][nl][k0 + j]
      b = nl < Nc ? *reinterpret_cast<const f32x4*>(w + ((int64_t)tap * Nc + nl) * K + k0) : zero4;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = (nl < Nc && k0 + j < K) ? w[((int64_t)tap * Nc + nl) * K + k0 + j] : 0.f;
    }
  };
  constexpr int R = 5;
  f32x4 ra[R], rb[R];
  int ch = wave;
#pragma unroll
  for (int u = 0; u < R; ++u)
    if (ch + u * NW < nchunks) load(ch + u * NW, ra[u], rb[u]);
  while (ch < nchunks) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (ch < nchunks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][j], rb[u][j], acc, 0, 0, 0);
        const int nx = ch + R * NW;
        if (nx < nchunks) load(nx, ra[u], rb[u]);
        ch += NW;
      }
    }
  }
  // acc[i] = C[row q*4+i][channel c]
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][q * 4 + i][c] = acc[i];
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int orow = threadIdx.x >> 4, oc = threadIdx.x & 15;
  const int grow = row0 + orow, n = n0 + oc;
  if (grow >= rows || n >= Nc) return;
  float v = red[0][orow][oc];
#pragma unroll
  for (int wv = 1; wv < NW; ++wv) v += red[wv][orow][oc];
  if (MODE == 1) {
    y[(int64_t)grow * Nc + n] = v;
    return;
  }
  if (bias) v += bias[n];
  v = actf(v, g.act, g.alpha);
  int64_t dst = (int64_t)grow * Nc + n;
  const int b = g.d2s;
  if (b > 1) {
    unsigned r = (unsigned)grow;
    const int o2 = (int)(r % (unsigned)g.O[2]); r /= (unsigned)g.O[2];
    const int o1 = (int)(r % (unsigned)g.O[1]); r /= (unsigned)g.O[1];
    const int o0 = (int)(r % (unsigned)g.O[0]); r /= (unsigned)g.O[0];
    const int cpo = Nc / (b * b);
    const int blk = n / cpo, cc = n - blk * cpo;
    dst = ((((int64_t)r * g.O[0] * b + o0 * b + blk / b) * (g.O[1] * b) + o1 * b + blk % b) * g.O[2] + o2) * cpo + cc;
  }
  if (res) v += res[dst];
  y[dst] = v;
}

template <int MODE, bool VK>
__global__ __launch_bounds__(NT) void fewpos_mfma16_kernel(
    const float* __restrict__ src, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ res,
    float* __restrict__ y, ConvGeom g, int rows, int K, int Nc,
    const float* __restrict__ mask_y, float slope) {
  __shared__ int sidx[MAX_TAPS * 16];
  __shared__ float red[NW][16][16];
  mfma16_body<MODE, VK>(blockIdx.x, blockIdx.y, sidx, red, src, w, bias, res, y, g, rows, K, Nc, mask_y, slope);
}

// dW[tap][ci][co] (+)= sum_p x[cell(p, tap)][ci] * dPre[p][co]
// grid (taps, C_in / 64 tiles, C_out / 16 tiles); db: column sums of dPre
// (written by the workgroups of tap 0, ci tile 0) or nullptr
template <bool VEC4>     // C_in and C_out multiples of 4: 16-B operand reads / stores
__device__ __forceinline__ void wgrad_body(
    const int bx, const int by, const int bz,
    int* __restrict__ sdyn /* [rows]: source cell of every position under this tap */,
    float (*__restrict__ red)[64][16] /* [NW] */, float (*__restrict__ bred)[4][16] /* [NW] */,
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ dw, float* __restrict__ db, const ConvGeom& g, int rows,
    int accumulate, const float* __restrict__ mask_y, float slope) {
  const int tap = bx, ci0 = by * 64, co0 = bz * 16;
  for (int p = threadIdx.x; p < rows; p += NT) sdyn[p] = src_cell<0>(g, (unsigned)p, tap);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int Cin = g.Cin, Cout = g.Cout;
  const int ci = ci0 + c * 4, co = co0 + c;
  const bool want_b = db != nullptr && bx == 0 && by == 0;
  f32x4 acc[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) acc[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int steps = (rows + 3) >> 2;
  // a step = four positions (one MFMA per ci fragment); the waves interleave
  // over the steps, four steps per trip, and the next trip's operands are in
  // flight under the sixteen MFMAs of the current one
  constexpr int U = 4;
  f32x4 xa[2][U];
  float da[2][U];
  auto load = [&](int st, f32x4 (&xv)[U], float (&dv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = (st + u * NW) * 4 + q;
      const bool in = p < rows;
      const int sc = in ? sdyn[p] : -1;
      if (VEC4) {
        xv[u] = (sc >= 0 && ci < Cin) ? *reinterpret_cast<const f32x4*>(x + (int64_t)sc * Cin + ci) : zero4;
      } else {          // (few input channels: the first layer of a network)
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[u][e] = (sc >= 0 && ci + e < Cin) ? x[(int64_t)sc * Cin + ci + e] : 0.f;
      }
      float d = (in && co < Cout) ? dy[(int64_t)p * Cout + co] : 0.f;
      if (mask_y) {     // dy = dL/dy of an activated conv: its adjoint on the fly
        const float m = (in && co < Cout) ? mask_y[(int64_t)p * Cout + co] : 0.f;
        d *= m > 0.f ? 1.f : slope;
      }
      dv[u] = d;
    }
  };
  auto fma = [&](const f32x4 (&xv)[U], const float (&dv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][mf], dv[u], acc[mf], 0, 0, 0);
      bsum += dv[u];
    }
  };
  int st = wave;
  if (st < steps) load(st, xa[0], da[0]);
  while (st < steps) {
    int nx = st + U * NW;
    if (nx < steps) load(nx, xa[1], da[1]);
    fma(xa[0], da[0]);
    st = nx;
    if (st >= steps) break;
    nx = st + U * NW;
    if (nx < steps) load(nx, xa[0], da[0]);
    fma(xa[1], da[1]);
    st = nx;
  }
  // acc[mf][i] = C[m = q*4+i][n = c]  with m <-> ci0 + m*4 + mf
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(q * 4 + i) * 4 + mf][c] = acc[mf][i];
  if (want_b) bred[wave][q][c] = bsum;
  __syncthreads();
  if (threadIdx.x < 256) {
    const int m = threadIdx.x >> 2, cg = (threadIdx.x & 3) * 4;
    const int oci = ci0 + m, oco = co0 + cg;
    if (oci < Cin && oco < Cout) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][m][cg]);
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) v += *reinterpret_cast<const f32x4*>(&red[wv][m][cg]);
      float* dst = dw + ((int64_t)tap * Cin + oci) * Cout + oco;
      if (VEC4) {
        if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {            // (a tail conv: 2 or 3 output channels)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (oco + e < Cout) dst[e] = accumulate ? dst[e] + v[e] : v[e];
      }
    }
  }
  if (want_b && threadIdx.x >= 256 && threadIdx.x < 272 && co0 + (int)threadIdx.x - 256 < Cout) {
    const int bc = threadIdx.x - 256;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) t += bred[wv][qq][bc];
    float* dst = db + co0 + bc;
    *dst = accumulate ? *dst + t : t;
  }
}

template <bool VEC4>
__global__ __launch_bounds__(NT) void fewpos_wgrad_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ dw, float* __restrict__ db, ConvGeom g, int rows,
    int accumulate, const float* __restrict__ mask_y, float slope) {
  extern __shared__ int sdyn[];
  __shared__ float red[NW][64][16];
  __shared__ float bred[NW][4][16];
  wgrad_body<VEC4>(blockIdx.x, blockIdx.y, blockIdx.z, sdyn, red, bred, x, dy, dw, db, g, rows, accumulate, mask_y, slope);
}

// Data gradient AND weight (+ bias) gradient of a conv in one launch: both read
// the same dPre and neither reads the other, so their workgroups share a grid —
// the first nd_x * nd_y are 16 x 16 tiles of dX (over gd: the conv's geometry,
// or its padded frame for reflect padding), the rest the (tap, ci tile, co
// tile) items of dW.  One dependent launch (>= 4.6 us) less per conv and pass.
template <bool V>     // C_in and C_out multiples of 16
__global__ __launch_bounds__(NT) void fewpos_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
    float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
    ConvGeom g, ConvGeom gd, int rows_d, int rows_w, int nd_x, int nd_y, int nw_y, int nw_z,
    int accumulate, const float* __restrict__ mask_y, float slope) {
  extern __shared__ int smem[];
  const int nd = nd_x * nd_y;
  if ((int)blockIdx.x < nd) {
    int* sidx = smem;
    float (*red)[16][16] = reinterpret_cast<float (*)[16][16]>(smem + MAX_TAPS * 16);
    mfma16_body<1, V>(blockIdx.x % nd_x, blockIdx.x / nd_x, sidx, red, dy, w, nullptr, nullptr, dx, gd, rows_d,
                   gd.Cout, gd.Cin, mask_y, slope);
    return;
  }
  int item = blockIdx.x - nd;
  const int bz = item % nw_z; item /= nw_z;
  const int by = item % nw_y; item /= nw_y;
  float (*red)[64][16] = reinterpret_cast<float (*)[64][16]>(smem);
  float (*bred)[4][16] = reinterpret_cast<float (*)[4][16]>(smem + NW * 64 * 16);
  int* sdyn = smem + NW * 64 * 16 + NW * 4 * 16;
  wgrad_body<V>(item, by, bz, sdyn, red, bred, x, dy, dw, db, g, rows_w, accumulate, mask_y, slope);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// geometry the one-launch kernels take (forward, data gradient and weight
// gradient alike: the plan skips the filter transpose on this predicate)
bool conv_fewpos_mfma_ok(const ConvGeom& g) {
  if (s3_opt_has(S3O_NO_FEWPOS_MFMA)) return false;
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t Pin = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  int64_t Pf = g.N;                       // padded frame of the reflect dgrad
  for (int d = 0; d < 3; ++d) Pf *= g.D[d] + 2 * g.lo[d];
  return taps <= MAX_TAPS && (g.Cin & 15) == 0 && (g.Cout & 15) == 0 &&
         P <= 8192 && Pin <= 32768 && Pf <= 65536 &&
         (g.d2s <= 1 || g.Cout % (g.d2s * g.d2s) == 0);
}

// the weight-gradient kernel alone takes any C_in / C_out (scalar operand reads
// / stores when they are not multiples of 4) and any number of taps
bool conv_fewpos_wgrad_mfma_ok(const ConvGeom& g) {
  if (s3_opt_has(S3O_NO_FEWPOS_MFMA)) return false;
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t Pin = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  return P <= 8192 && Pin < (1 << 24);
}

// ... and the convs the split-K family never took (C_in or C_out below 16, small
// filters: the head / tail convs of a network, the first discriminator layer)
// while their 16 x 16 tiles still find the chip mostly idle: the 16-channel
// kernel with ragged K.  In bf16 plans these leave the gather-MFMA kernel + its
// per-step filter pack for exact fp32 products.
bool conv_fewpos_mfma_small_ok(const s3_ctx* ctx, const ConvGeom& g) {
  if (s3_opt_has(S3O_NO_FEWPOS_MFMA) || s3_opt_has(S3O_NO_FEWPOS_SMALL)) return false;
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t Pin = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  int64_t Pf = g.N;
  for (int d = 0; d < 3; ++d) Pf *= g.D[d] + 2 * g.lo[d];
  const int64_t rows_d = g.pad_mode == S3_PAD_REFLECT ? Pf : Pin;
  const int64_t lim = 4 * (int64_t)ctx->num_cu;
  return taps <= MAX_TAPS && P <= 4096 && Pin <= 32768 && Pf <= 65536 &&
         (g.d2s <= 1 || g.Cout % (g.d2s * g.d2s) == 0) &&
         ((P + 15) / 16) * ((g.Cout + 15) / 16) <= lim && ((rows_d + 15) / 16) * ((g.Cin + 15) / 16) <= lim;
}

// mode 0: y = act(conv(x) + bias) (+ res), depth-to-space store; mode 1: dx = adjoint(dy)
int launch_conv_fewpos_mfma(s3_ctx* ctx, const ConvGeom& g, int mode, const float* src,
                            const float* w, const float* bias, const float* res, float* y,
                            const float* mask_y, float slope) {
  if (mask_y && mode != 1) S3_FAIL(ctx, S3_EINVAL, "fewpos mfma: mask operand");
  const int64_t rows = mode == 0 ? (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] : (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  const int K = mode == 0 ? g.Cin : g.Cout, Nc = mode == 0 ? g.Cout : g.Cin;
  const bool vk = (K & 15) == 0;
  const unsigned rt = (unsigned)((rows + 15) / 16);
  // 16-channel tiles while that still leaves the workgroups CUs of their own
  const bool narrow = !vk || (Nc & 15) != 0 || (int64_t)rt * ((Nc + 15) / 16) <= 4 * ctx->num_cu;
  if ((vk || !narrow) &&
      (!aligned16(src) || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias)) ||
       (res && !aligned16(res)) || (mask_y && !aligned16(mask_y))))
    S3_FAIL(ctx, S3_EINVAL, "fewpos mfma: operand not 16-B aligned");
  const dim3 grid(rt, narrow ? (Nc + 15) / 16 : (Nc + 63) / 64);
  const float* nof = nullptr;
#define S3_FP16(M, V, B, R, MY, SL)                                                                          \
  hipLaunchKernelGGL((fewpos_mfma16_kernel<M, V>), grid, dim3(NT), 0, ctx->stream, src, w, B, R, y, g,      \
                     (int)rows, K, Nc, MY, SL)
  if (mode == 0) {
    if (narrow && vk) S3_FP16(0, true, bias, res, nof, 0.f);
    else if (narrow) S3_FP16(0, false, bias, res, nof, 0.f);
    else
      hipLaunchKernelGGL(fewpos_mfma_kernel<0>, grid, dim3(NT), 0, ctx->stream, src, w, bias, res, y, g,
                         (int)rows, K, Nc, nof, 0.f);
  } else {
    if (narrow && vk) S3_FP16(1, true, nof, nof, mask_y, slope);
    else if (narrow) S3_FP16(1, false, nof, nof, mask_y, slope);
    else
      hipLaunchKernelGGL(fewpos_mfma_kernel<1>, grid, dim3(NT), 0, ctx->stream, src, w, nof, nof, y, g,
                         (int)rows, K, Nc, mask_y, slope);
  }
#undef S3_FP16
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_fewpos_wgrad_mfma(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                                  float* dw, float* db, int accumulate, const float* mask_y, float slope) {
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dw))
    S3_FAIL(ctx, S3_EINVAL, "fewpos wgrad mfma: operand not 16-B aligned");
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t rows = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  dim3 grid(taps, (g.Cin + 63) / 64, (g.Cout + 15) / 16);
  if (((g.Cin | g.Cout) & 3) == 0)
    hipLaunchKernelGGL(fewpos_wgrad_mfma_kernel<true>, grid, dim3(NT), (size_t)rows * sizeof(int), ctx->stream,
                       x, dy, dw, db, g, (int)rows, accumulate, mask_y, slope);
  else
    hipLaunchKernelGGL(fewpos_wgrad_mfma_kernel<false>, grid, dim3(NT), (size_t)rows * sizeof(int), ctx->stream,
                       x, dy, dw, db, g, (int)rows, accumulate, mask_y, slope);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// whether launch_conv_fewpos_bwd_mfma takes this conv (16-channel dX tiles only)
bool conv_fewpos_bwd_mfma_ok(const s3_ctx* ctx, const ConvGeom& g, const ConvGeom& gd) {
  if (!conv_fewpos_mfma_ok(g) && !conv_fewpos_mfma_small_ok(ctx, g)) return false;
  const int64_t rows_d = (int64_t)gd.N * gd.D[0] * gd.D[1] * gd.D[2];
  const int64_t rows_w = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  return ((rows_d + 15) / 16) * ((gd.Cin + 15) / 16) <= 4 * ctx->num_cu && rows_w <= 8192;
}

// dx = adjoint(dy) over gd (g or its padded frame) and dw (+)= x^T dy, db (+)= sum dy
int launch_conv_fewpos_bwd_mfma(s3_ctx* ctx, const ConvGeom& g, const ConvGeom& gd, const float* x,
                                const float* dy, const float* w, float* dx, float* dw, float* db,
                                int accumulate, const float* mask_y, float slope) {
  const bool v = ((g.Cin | g.Cout) & 15) == 0;
  if (v && (!aligned16(x) || !aligned16(dy) || !aligned16(w) || !aligned16(dx) || !aligned16(dw) ||
            (mask_y && !aligned16(mask_y))))
    S3_FAIL(ctx, S3_EINVAL, "fewpos bwd mfma: operand not 16-B aligned");
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t rows_w = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t rows_d = (int64_t)gd.N * gd.D[0] * gd.D[1] * gd.D[2];
  const int nd_x = (int)((rows_d + 15) / 16), nd_y = (gd.Cin + 15) / 16;
  const int nw_y = (g.Cin + 63) / 64, nw_z = (g.Cout + 15) / 16;
  const size_t lds_d = (size_t)(MAX_TAPS * 16 + NW * 16 * 16) * 4;
  const size_t lds_w = (size_t)(NW * 64 * 16 + NW * 4 * 16 + rows_w) * 4;
  const size_t lds = lds_d > lds_w ? lds_d : lds_w;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fewpos_bwd_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fewpos_bwd_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set.mark(ctx->device);
  }
  const dim3 grid((unsigned)(nd_x * nd_y + taps * nw_y * nw_z));
  if (v)
    hipLaunchKernelGGL(fewpos_bwd_kernel<true>, grid, dim3(NT), lds, ctx->stream, x, dy, w, dx, dw, db, g, gd,
                       (int)rows_d, (int)rows_w, nd_x, nd_y, nw_y, nw_z, accumulate, mask_y, slope);
  else
    hipLaunchKernelGGL(fewpos_bwd_kernel<false>, grid, dim3(NT), lds, ctx->stream, x, dy, w, dx, dw, db, g, gd,
                       (int)rows_d, (int)rows_w, nd_x, nd_y, nw_y, nw_z, accumulate, mask_y, slope);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
