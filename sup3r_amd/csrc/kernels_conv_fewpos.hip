// "Few positions, many channels" convolutions: the deep layers of the patch
// discriminator (128-256 channels on <= a few thousand positions) and every
// conv of the tiny-sample training configs.  Here the filter (up to 27 x 256 x
// 256 fp32 = 7 MB) dwarfs the activations, so the work is organised as a
// weight-streaming GEMM instead of a position-parallel direct conv:
//
//   * lanes run along the output channel -> filter rows w[tap][k][n0..n0+63]
//     are coalesced 256-B reads, each filter element is read once per batch
//     of PB positions;
//   * activations are wave-uniform (scalar) reads through a per-block table of
//     source indices, so ONE kernel serves the forward conv (table = input
//     cell of output o under tap t, any stride / padding) and the data
//     gradient (table = output cell that input i feeds through tap t, with the
//     [tap][co][ci] transposed filter);
//   * K = taps x C is split over the grid (one tap per block, 4 sub-slabs of
//     channels per block); partials are reduced in fixed order by the epilogue
//     kernel which also applies bias / activation / residual / depth-to-space.
//
// The weight gradient has no reduction over channels: one thread per
// (tap, ci, co) walks the positions (dPre rows coalesced along co).
#include "common.h"

namespace {

constexpr int PB = 8;   // positions per block

__device__ inline float act_f(float v, int act, float alpha) {
  if (act == S3_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == S3_ACT_LEAKY) return v > 0.f ? v : alpha * v;
  return v;
}

// mode 0: forward  (rows = output positions of g, src = input cells)
// mode 1: dgrad    (rows = input positions of g,  src = output cells)
__device__ inline int64_t fewpos_src(const ConvGeom& g, int mode, int64_t row,
                                     int tap) {
  const int a = tap / (g.k[1] * g.k[2]), b = (tap / g.k[2]) % g.k[1], c = tap % g.k[2];
  const int kk[3] = {a, b, c};
  // (fewpos geometries have < 2^16 positions: 32-bit divisions — the 64-bit
  // ones were most of the weight-gradient kernel's time)
  unsigned r = (unsigned)row;
  int p[3], n;
  if (mode == 0) {
    p[2] = (int)(r % g.O[2]); r /= g.O[2];
    p[1] = (int)(r % g.O[1]); r /= g.O[1];
    p[0] = (int)(r % g.O[0]); r /= g.O[0];
    n = (int)r;
    int i[3];
    for (int d = 0; d < 3; ++d) {
      i[d] = p[d] * g.s[d] + kk[d] - g.lo[d];
      if (g.pad_mode == S3_PAD_REFLECT) i[d] = s3_reflect(i[d], g.D[d]);
      if (i[d] < 0 || i[d] >= g.D[d]) return -1;
    }
    return (((int64_t)n * g.D[0] + i[0]) * g.D[1] + i[1]) * g.D[2] + i[2];
  }
  p[2] = (int)(r % g.D[2]); r /= g.D[2];
  p[1] = (int)(r % g.D[1]); r /= g.D[1];
  p[0] = (int)(r % g.D[0]); r /= g.D[0];
  n = (int)r;
  int o[3];
  for (int d = 0; d < 3; ++d) {
    int t = p[d] + g.lo[d] - kk[d];
    if (t < 0 || t % g.s[d] != 0) return -1;
    t /= g.s[d];
    if (t >= g.O[d]) return -1;
    o[d] = t;
  }
  return (((int64_t)n * g.O[0] + o[0]) * g.O[1] + o[1]) * g.O[2] + o[2];
}

// partial[tap][row][n] = sum_k src[srcidx(row, tap)][k] * w[tap][k][n]
__global__ __launch_bounds__(256) void fewpos_gemm_kernel(
    const float* __restrict__ src, const float* __restrict__ w,
    float* __restrict__ partial, ConvGeom g, int mode, int64_t rows, int K,
    int Nc) {
  __shared__ int64_t sidx[PB];
  __shared__ float red[4][PB][64];
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x * 64 + tx;
  const int64_t row0 = (int64_t)blockIdx.y * PB;
  const int tap = blockIdx.z;
  if (threadIdx.x < PB) {
    const int64_t row = row0 + threadIdx.x;
    sidx[threadIdx.x] = row < rows ? fewpos_src(g, mode, row, tap) : -1;
  }
  __syncthreads();
  const float* sp[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) sp[b] = sidx[b] >= 0 ? src + sidx[b] * K : nullptr;
  float acc[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) acc[b] = 0.f;
  const bool live = n < Nc;
  const float* wt = w + (int64_t)tap * K * Nc;
  for (int k = ty; k < K; k += 4) {
    const float wv = live ? wt[(int64_t)k * Nc + n] : 0.f;
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const float xv = sp[b] ? sp[b][k] : 0.f;
      acc[b] = fmaf(xv, wv, acc[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < PB; ++b) red[ty][b][tx] = acc[b];
  __syncthreads();
  if (ty == 0 && live) {
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if (row0 + b >= rows) break;
      const float t = red[0][b][tx] + red[1][b][tx] + red[2][b][tx] + red[3][b][tx];
      partial[((int64_t)tap * rows + row0 + b) * Nc + n] = t;
    }
  }
}

// y = act(sum_taps partial + bias) (+ res), optional depth-to-space store
__global__ void fewpos_epilogue_kernel(const float* __restrict__ partial,
                                       const float* __restrict__ bias,
                                       const float* __restrict__ res,
                                       float* __restrict__ y, ConvGeom g,
                                       int taps, int64_t rows, int Nc,
                                       int fwd) {
  const int64_t total = rows * Nc;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < taps; ++s) t += partial[(int64_t)s * total + idx];
    if (!fwd) { y[idx] = t; continue; }
    const int co = (int)(idx % Nc);
    int64_t r = idx / Nc;
    t += bias ? bias[co] : 0.f;
    int64_t dst = idx;
    const int b = g.d2s;
    if (b > 1) {
      const int o2 = (int)(r % g.O[2]); r /= g.O[2];
      const int o1 = (int)(r % g.O[1]); r /= g.O[1];
      const int o0 = (int)(r % g.O[0]); r /= g.O[0];
      const int n = (int)r;
      const int cpo = Nc / (b * b);
      const int blk = co / cpo, cc = co % cpo;
      dst = ((((int64_t)n * g.O[0] * b + o0 * b + blk / b) * (g.O[1] * b) +
              o1 * b + blk % b) * g.O[2] + o2) * cpo + cc;
    }
    t = act_f(t, g.act, g.alpha);
    if (res) t += res[dst];
    y[dst] = t;
  }
}

// dW[tap][ci][co] (+)= sum_p x[src(p, tap)][ci] * dPre[p][co]
// One thread per (tap, ci block of 4, co) walks a SLAB of the positions
// (blockIdx.y / ci_blocks = slab; a serial walk over all of them is a chain of
// dependent loads: 174 us for 1500 positions); slabs are summed in fixed
// order by fewpos_wgrad_reduce.
__global__ __launch_bounds__(256) void fewpos_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ part, ConvGeom g, int64_t rows, int ci_blocks, int slabs) {
  constexpr int CI_T = 4;
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int co = blockIdx.x * 64 + tx;
  const int cib = blockIdx.y % ci_blocks, slab = blockIdx.y / ci_blocks;
  const int ci0 = (cib * 4 + ty) * CI_T;
  const int tap = blockIdx.z;
  if (ci0 >= g.Cin) return;
  const bool live = co < g.Cout;
  float acc[CI_T];
#pragma unroll
  for (int j = 0; j < CI_T; ++j) acc[j] = 0.f;
  const int64_t p0 = rows * slab / slabs, p1 = rows * (slab + 1) / slabs;
  // four positions per trip, every load issued before the first fma (one
  // position per trip was a chain of exposed memory latencies: 28 us for 47
  // positions); same summation order, padding taps add an exact zero
  constexpr int UP = 4;
  // output coordinates of the walk: one set of divisions per slab, then carries
  const int ka = tap / (g.k[1] * g.k[2]), kb = (tap / g.k[2]) % g.k[1], kc = tap % g.k[2];
  int wn, w0, w1, w2;
  {
    unsigned r = (unsigned)p0;
    w2 = (int)(r % (unsigned)g.O[2]); r /= (unsigned)g.O[2];
    w1 = (int)(r % (unsigned)g.O[1]); r /= (unsigned)g.O[1];
    w0 = (int)(r % (unsigned)g.O[0]); wn = (int)(r / (unsigned)g.O[0]);
  }
  auto src_here = [&]() -> int64_t {
    int i0 = w0 * g.s[0] + ka - g.lo[0], i1 = w1 * g.s[1] + kb - g.lo[1], i2 = w2 * g.s[2] + kc - g.lo[2];
    if (g.pad_mode == S3_PAD_REFLECT) {
      i0 = s3_reflect(i0, g.D[0]); i1 = s3_reflect(i1, g.D[1]); i2 = s3_reflect(i2, g.D[2]);
    }
    if (i0 < 0 || i0 >= g.D[0] || i1 < 0 || i1 >= g.D[1] || i2 < 0 || i2 >= g.D[2]) return -1;
    return (((int64_t)wn * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + i2;
  };
  auto step = [&]() {
    if (++w2 == g.O[2]) { w2 = 0; if (++w1 == g.O[1]) { w1 = 0; if (++w0 == g.O[0]) { w0 = 0; ++wn; } } }
  };
  for (int64_t p = p0; p < p1; p += UP) {
    int64_t s[UP];
    float d[UP], xv[UP][CI_T];
#pragma unroll
    for (int u = 0; u < UP; ++u) {          // wave-uniform
      s[u] = p + u < p1 ? src_here() : -1;
      step();
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      d[u] = (s[u] >= 0 && live) ? dy[(p + u) * g.Cout + co] : 0.f;
#pragma unroll
      for (int j = 0; j < CI_T; ++j)
        xv[u][j] = (s[u] >= 0 && ci0 + j < g.Cin) ? x[s[u] * g.Cin + ci0 + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UP; ++u)
#pragma unroll
      for (int j = 0; j < CI_T; ++j) acc[j] = fmaf(xv[u][j], d[u], acc[j]);
  }
  if (!live) return;
  const int64_t wsize = (int64_t)gridDim.z * g.Cin * g.Cout;
#pragma unroll
  for (int j = 0; j < CI_T; ++j) {
    if (ci0 + j >= g.Cin) break;
    part[slab * wsize + ((int64_t)tap * g.Cin + ci0 + j) * g.Cout + co] = acc[j];
  }
}

__global__ void fewpos_wgrad_reduce(const float* __restrict__ part, int slabs, int64_t wsize,
                                    float* __restrict__ dw, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < wsize;
       i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < slabs; ++s) t += part[(int64_t)s * wsize + i];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// wt[tap][co][ci] = w[tap][ci][co]
__global__ void transpose_taps_kernel(const float* __restrict__ w,
                                      float* __restrict__ wt, int taps, int cin,
                                      int cout) {
  const int64_t total = (int64_t)taps * cin * cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int ci = (int)(r % cin); r /= cin;
    const int co = (int)(r % cout); r /= cout;
    const int tp = (int)r;
    wt[idx] = w[((int64_t)tp * cin + ci) * cout + co];
  }
}

}  // namespace

bool conv_fewpos_supported(const ConvGeom& g) {
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t Pin = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  const int64_t wsize = (int64_t)g.k[0] * g.k[1] * g.k[2] * g.Cin * g.Cout;
  return P <= 4096 && Pin <= 32768 && wsize >= 16384 && g.Cin >= 16 && g.Cout >= 16;
}

// the weight-gradient kernel alone has no lower bounds on the filter size
bool conv_fewpos_wgrad_ok(const ConvGeom& g) {
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t Pin = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  return P <= 4096 && Pin <= 32768;
}

size_t conv_fewpos_partial_bytes(const ConvGeom& g) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int64_t Pin = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  int64_t Pf = g.N;                       // padded frame of the reflect dgrad
  for (int d = 0; d < 3; ++d) Pf *= g.D[d] + 2 * g.lo[d];
  const int64_t a = P * g.Cout, b = (Pin > Pf ? Pin : Pf) * g.Cin;
  return (size_t)taps * (a > b ? a : b) * sizeof(float);
}

int launch_conv_fewpos_fwd(s3_ctx* ctx, const ConvGeom& g, const float* x,
                           const float* w, const float* bias, const float* res,
                           float* y, float* partial, size_t partial_bytes) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t rows = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  if (partial_bytes < (size_t)taps * rows * g.Cout * sizeof(float))
    S3_FAIL(ctx, S3_EINVAL, "fewpos fwd: partial buffer too small");
  dim3 grid((g.Cout + 63) / 64, (unsigned)((rows + PB - 1) / PB), taps);
  hipLaunchKernelGGL(fewpos_gemm_kernel, grid, dim3(256), 0, ctx->stream, x, w, partial, g, 0, rows, g.Cin, g.Cout);
  int eg = (int)((rows * g.Cout + 255) / 256);
  if (eg > 2048) eg = 2048;
  hipLaunchKernelGGL(fewpos_epilogue_kernel, dim3(eg), dim3(256), 0, ctx->stream, partial, bias, res, y, g, taps, rows, g.Cout, 1);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// wt: [tap][co][ci] transposed filter (launch_conv_fewpos_transpose)
int launch_conv_fewpos_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy,
                             const float* wt, float* dx, float* partial,
                             size_t partial_bytes) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t rows = (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  if (partial_bytes < (size_t)taps * rows * g.Cin * sizeof(float))
    S3_FAIL(ctx, S3_EINVAL, "fewpos dgrad: partial buffer too small");
  dim3 grid((g.Cin + 63) / 64, (unsigned)((rows + PB - 1) / PB), taps);
  hipLaunchKernelGGL(fewpos_gemm_kernel, grid, dim3(256), 0, ctx->stream, dy, wt, partial, g, 1, rows, g.Cout, g.Cin);
  int eg = (int)((rows * g.Cin + 255) / 256);
  if (eg > 2048) eg = 2048;
  hipLaunchKernelGGL(fewpos_epilogue_kernel, dim3(eg), dim3(256), 0, ctx->stream, partial, nullptr, nullptr, dx, g, taps, rows, g.Cin, 0);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

static int fewpos_wgrad_slabs(const ConvGeom& g) {
  const int64_t rows = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  int64_t s = rows / 32;              // >= 32 positions per slab
  if (s > 32) s = 32;
  return (int)(s < 1 ? 1 : s);
}

size_t conv_fewpos_wgrad_partial_bytes(const ConvGeom& g) {
  return (size_t)fewpos_wgrad_slabs(g) * g.k[0] * g.k[1] * g.k[2] * g.Cin * g.Cout * sizeof(float);
}

int launch_conv_fewpos_wgrad(s3_ctx* ctx, const ConvGeom& g, const float* x,
                             const float* dy, float* dw, float* partial, size_t partial_bytes,
                             int accumulate) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t rows = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  const int slabs = fewpos_wgrad_slabs(g);
  if (partial_bytes < conv_fewpos_wgrad_partial_bytes(g))
    S3_FAIL(ctx, S3_EINVAL, "fewpos wgrad: partial buffer too small");
  const int ci_blocks = (g.Cin + 15) / 16;
  dim3 grid((g.Cout + 63) / 64, ci_blocks * slabs, taps);
  hipLaunchKernelGGL(fewpos_wgrad_kernel, grid, dim3(256), 0, ctx->stream, x, dy, partial, g, rows,
                     ci_blocks, slabs);
  const int64_t wsize = (int64_t)taps * g.Cin * g.Cout;
  int rg = (int)((wsize + 255) / 256);
  if (rg > 2048) rg = 2048;
  hipLaunchKernelGGL(fewpos_wgrad_reduce, dim3(rg), dim3(256), 0, ctx->stream, partial, slabs, wsize,
                     dw, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// geometry of the data gradient over the virtually padded frame of a
// reflect-padded conv: positions run over D + 2 lo per axis with lo' = 0; the
// caller folds the frame back onto x (adjoint of the reflect padding)
ConvGeom conv_fewpos_frame_geom(const ConvGeom& g) {
  ConvGeom f = g;
  for (int d = 0; d < 3; ++d) { f.D[d] = g.D[d] + 2 * g.lo[d]; f.lo[d] = 0; }
  f.pad_mode = S3_PAD_ZERO;
  return f;
}

int launch_conv_fewpos_transpose(s3_ctx* ctx, const ConvGeom& g, const float* w,
                                 float* wt) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t total = (int64_t)taps * g.Cin * g.Cout;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(transpose_taps_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, wt, taps, g.Cin, g.Cout);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
