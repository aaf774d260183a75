// Feature maps of the structured content losses of
// sup3r/utilities/loss_metrics.py and their adjoints.
//
// Every one of those losses is  M(F(gen), F(true))  with M = keras
// MeanAbsoluteError / MeanSquaredError (s3_loss_content) and F one of the maps
// below; the host composes  F -> M -> F^T  so that the gradient lands in the
// generator-output gradient buffer.  x is (n, s1, s2, t, c) fp32 (t = 1 for 4-D
// batches); only the first c_used channels take part (calc_loss_gen_content
// drops the trailing exo channels, sup3r/models/base.py:478-503).
//
//   S3_LMAP_DERIV_S  d/ds1 + d/ds2 (SpatialDerivativeLoss, loss_metrics.py:228-260)
//   S3_LMAP_DERIV_T  d/dt          (TemporalDerivativeLoss, :263-294)
//   S3_LMAP_MATERIAL du/dt + u du/ds1 + v du/ds2 of every (u, v) pair
//                                  (MaterialDerivativeLoss, :150-225)
//   S3_LMAP_MEAN_S   mean over (s1, s2)  (CoarseMseLoss, :297-322)
//   S3_LMAP_EXT_S    min | max over (s1, s2) (SpatialExtremesLoss, :325-357)
//   S3_LMAP_EXT_T    min | max over t        (TemporalExtremesLoss, :360-392)
//   S3_LMAP_COARSEN  block mean over s x s x t_enhance (LowResLoss, :488-638)
//
// _derivative (:12-59) is np.gradient's first-order scheme: one-sided at the
// two ends, central inside.  All kernels are HBM-bound element passes.
#include "common.h"

namespace {

constexpr int kB = 256;

inline int grid_of(int64_t n, int num_cu) {
  int64_t b = (n + kB - 1) / kB;
  const int64_t cap = (int64_t)num_cu * 8;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

struct LGeom {
  int n, s1, s2, t, c, cu;   // cu = channels used
};

// derivative of v along an axis of length L at index i (stride st elements)
__device__ __forceinline__ float deriv_at(const float* __restrict__ p, int i, int L, int64_t st) {
  if (i == 0) return p[st] - p[0];
  if (i == L - 1) return p[0] - p[-st];
  return 0.5f * (p[st] - p[-st]);
}

// coefficient of x[j] in derivative row i
__device__ __forceinline__ float dcoef(int i, int j, int L) {
  if (i < 0 || i >= L) return 0.f;
  if (i == 0) return j == 0 ? -1.f : (j == 1 ? 1.f : 0.f);
  if (i == L - 1) return j == L - 1 ? 1.f : (j == L - 2 ? -1.f : 0.f);
  return j == i + 1 ? 0.5f : (j == i - 1 ? -0.5f : 0.f);
}

__device__ __forceinline__ void decode(int64_t idx, const LGeom& g, int cdim, int& n, int& i1,
                                       int& i2, int& it, int& ch) {
  ch = (int)(idx % cdim); idx /= cdim;
  it = (int)(idx % g.t); idx /= g.t;
  i2 = (int)(idx % g.s2); idx /= g.s2;
  i1 = (int)(idx % g.s1); idx /= g.s1;
  n = (int)idx;
}

// mode 0: d/ds1 + d/ds2, mode 1: d/dt; out (n, s1, s2, t, cu)
__global__ void deriv_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, LGeom g,
                                 int mode) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.cu;
  const int64_t st2 = (int64_t)g.t * g.c, st1 = st2 * g.s2, stt = g.c;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.cu, n, i1, i2, it, ch);
    const float* p = x + ((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + ch;
    out[idx] = mode == 0 ? deriv_at(p, i1, g.s1, st1) + deriv_at(p, i2, g.s2, st2)
                         : deriv_at(p, it, g.t, stt);
  }
}

// d_x[.., ch < cu] += D^T gout
__global__ void deriv_bwd_kernel(const float* __restrict__ gout, float* __restrict__ dx, LGeom g,
                                 int mode) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.cu;
  const int64_t gt = g.cu, g2 = (int64_t)g.t * g.cu, g1 = g2 * g.s2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.cu, n, i1, i2, it, ch);
    const float* q = gout + idx;
    float acc = 0.f;
    if (mode == 0) {
      for (int d = -1; d <= 1; ++d) {
        if (i1 + d >= 0 && i1 + d < g.s1) acc += dcoef(i1 + d, i1, g.s1) * q[d * g1];
        if (i2 + d >= 0 && i2 + d < g.s2) acc += dcoef(i2 + d, i2, g.s2) * q[d * g2];
      }
    } else {
      for (int d = -1; d <= 1; ++d)
        if (it + d >= 0 && it + d < g.t) acc += dcoef(it + d, it, g.t) * q[d * gt];
    }
    dx[((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + ch] += acc;
  }
}

// material derivative of the u component of every (u, v) pair; out (n, s1, s2, t, hub)
__global__ void material_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, LGeom g,
                                    int hub) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * hub;
  const int64_t st2 = (int64_t)g.t * g.c, st1 = st2 * g.s2, stt = g.c;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, k;
    decode(idx, g, hub, n, i1, i2, it, k);
    const float* p = x + ((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + 2 * k;
    out[idx] = deriv_at(p, it, g.t, stt) + p[0] * deriv_at(p, i1, g.s1, st1) +
               p[1] * deriv_at(p, i2, g.s2, st2);
  }
}

__global__ void material_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                    float* __restrict__ dx, LGeom g, int hub) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * hub;
  const int64_t st2 = (int64_t)g.t * g.c, st1 = st2 * g.s2;
  const int64_t gt = hub, g2 = (int64_t)g.t * hub, g1 = g2 * g.s2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, k;
    decode(idx, g, hub, n, i1, i2, it, k);
    const int64_t xo = ((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + 2 * k;
    const float* p = x + xo;
    const float* q = gout + idx;
    // u enters as f (three stencils) and as the advecting factor; v only as a factor
    float du = q[0] * deriv_at(p, i1, g.s1, st1);
    const float dv = q[0] * deriv_at(p, i2, g.s2, st2);
    for (int d = -1; d <= 1; ++d) {
      if (it + d >= 0 && it + d < g.t) du += dcoef(it + d, it, g.t) * q[d * gt];
      if (i1 + d >= 0 && i1 + d < g.s1) du += dcoef(i1 + d, i1, g.s1) * p[d * st1] * q[d * g1];
      if (i2 + d >= 0 && i2 + d < g.s2) du += dcoef(i2 + d, i2, g.s2) * p[d * st2 + 1] * q[d * g2];
    }
    dx[xo] += du;
    dx[xo + 1] += dv;
  }
}

// ---- reductions.  Spatial: out[(n, t, ch)] over the s1 * s2 positions, two
// stages (slabs of positions -> partial[n][slab][t * cu] -> out).  what: 0 sum,
// 1 min, 2 max, 3 count of x == ref[(n, t, ch)]
constexpr int kSlabs = 64;

__device__ __forceinline__ float red_init(int what) {
  return what == 1 ? 3.402823466e38f : (what == 2 ? -3.402823466e38f : 0.f);
}
__device__ __forceinline__ float red_op(int what, float a, float v, float ref) {
  if (what == 1) return fminf(a, v);
  if (what == 2) return fmaxf(a, v);
  if (what == 3) return a + (v == ref ? 1.f : 0.f);
  return a + v;
}
__device__ __forceinline__ float red_merge(int what, float a, float b) {
  if (what == 1) return fminf(a, b);
  if (what == 2) return fmaxf(a, b);
  return a + b;
}

__global__ void reduce_s_stage1(const float* __restrict__ x, const float* __restrict__ ref,
                                float* __restrict__ partial, LGeom g, int what) {
  const int n = blockIdx.y, slab = blockIdx.x;
  const int npos = g.s1 * g.s2, tc = g.t * g.cu;
  const int p0 = (int)((int64_t)npos * slab / kSlabs), p1 = (int)((int64_t)npos * (slab + 1) / kSlabs);
  for (int j = threadIdx.x; j < tc; j += blockDim.x) {
    const int it = j / g.cu, ch = j % g.cu;
    const float r = ref ? ref[(int64_t)n * tc + j] : 0.f;
    float a = red_init(what);
    for (int p = p0; p < p1; ++p)
      a = red_op(what, a, x[(((int64_t)n * npos + p) * g.t + it) * g.c + ch], r);
    partial[((int64_t)n * kSlabs + slab) * tc + j] = a;
  }
}

__global__ void reduce_stage2(const float* __restrict__ partial, float* __restrict__ out, int n,
                              int slabs, int width, int what, float scale) {
  const int64_t total = (int64_t)n * width;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int nn = (int)(idx / width), j = (int)(idx % width);
    float a = red_init(what);
    for (int s = 0; s < slabs; ++s) a = red_merge(what, a, partial[((int64_t)nn * slabs + s) * width + j]);
    out[idx] = a * scale;
  }
}

// temporal: out[(n, s1, s2, ch)] over t
__global__ void reduce_t_kernel(const float* __restrict__ x, const float* __restrict__ ref,
                                float* __restrict__ out, LGeom g, int what) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.cu;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % g.cu);
    const int64_t pos = idx / g.cu;
    const float r = ref ? ref[idx] : 0.f;
    float a = red_init(what);
    for (int it = 0; it < g.t; ++it) a = red_op(what, a, x[(pos * g.t + it) * g.c + ch], r);
    out[idx] = a;
  }
}

// adjoint of the mean over (s1, s2): dx += gout[(n, t, ch)] / (s1 s2)
__global__ void mean_s_bwd_kernel(const float* __restrict__ gout, float* __restrict__ dx, LGeom g) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.cu;
  const float inv = 1.f / (float)(g.s1 * g.s2);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.cu, n, i1, i2, it, ch);
    dx[((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + ch] +=
        gout[((int64_t)n * g.t + it) * g.cu + ch] * inv;
  }
}

// adjoint of min | max (tf.reduce_min / reduce_max gradient: shared equally by
// the tied extrema).  spatial != 0: extrema indexed (n, t, ch); else (n, s1, s2, ch)
__global__ void ext_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mn,
                               const float* __restrict__ mx, const float* __restrict__ cmn,
                               const float* __restrict__ cmx, const float* __restrict__ gmn,
                               const float* __restrict__ gmx, float* __restrict__ dx, LGeom g,
                               int spatial) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.cu;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.cu, n, i1, i2, it, ch);
    const int64_t xo = ((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + ch;
    const int64_t e = spatial ? ((int64_t)n * g.t + it) * g.cu + ch
                              : (((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.cu + ch;
    const float v = x[xo];
    float d = 0.f;
    if (v == mn[e]) d += gmn[e] / cmn[e];
    if (v == mx[e]) d += gmx[e] / cmx[e];
    if (d != 0.f) dx[xo] += d;
  }
}

// adjoint of s3_coarsen (average | subsample over t_enhance, block mean over s x s)
__global__ void coarsen_bwd_kernel(const float* __restrict__ gout, float* __restrict__ dx, LGeom g,
                                   int s, int te, int method) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.cu;
  const int o1 = g.s1 / s, o2 = g.s2 / s, ot = te > 1 ? g.t / te : g.t;
  const float w = 1.f / (float)(s * s) / (method == S3_TC_AVERAGE && te > 1 ? (float)te : 1.f);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.cu, n, i1, i2, it, ch);
    int jt = it;
    if (te > 1) {
      if (method == S3_TC_SUBSAMPLE && it % te != 0) continue;
      jt = it / te;
      if (jt >= ot) continue;
    }
    dx[((((int64_t)n * g.s1 + i1) * g.s2 + i2) * g.t + it) * g.c + ch] +=
        w * gout[((((int64_t)n * o1 + i1 / s) * o2 + i2 / s) * ot + jt) * g.c + ch];
  }
}

// ---- MmdLoss (loss_metrics.py:62-147): per position, all pairs of observations
__global__ void mmd_kernel(const float* __restrict__ a, int c_a, const float* __restrict__ b, int c_b,
                           int n, int64_t npos, int cu, float inv_s2, float gscale,
                           float* __restrict__ partial, float* __restrict__ d_a) {
  __shared__ float sm[8];
  float acc = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npos;
       p += (int64_t)gridDim.x * blockDim.x) {
    for (int i = 0; i < n; ++i) {
      const float* ai = a + ((int64_t)i * npos + p) * c_a;
      float gi[8];
      for (int c = 0; c < cu; ++c) gi[c] = 0.f;
      for (int j = 0; j < n; ++j) {
        const float* aj = a + ((int64_t)j * npos + p) * c_a;
        const float* bj = b + ((int64_t)j * npos + p) * c_b;
        const float* bi = b + ((int64_t)i * npos + p) * c_b;
        float daa = 0.f, dab = 0.f, dbb = 0.f;
        for (int c = 0; c < cu; ++c) {
          const float u = ai[c] - aj[c], v = ai[c] - bj[c], w = bi[c] - bj[c];
          daa += u * u; dab += v * v; dbb += w * w;
        }
        const float kaa = __expf(-0.5f * daa * inv_s2), kab = __expf(-0.5f * dab * inv_s2),
                    kbb = __expf(-0.5f * dbb * inv_s2);
        acc += kaa + kbb - 2.f * kab;
        if (d_a)
          for (int c = 0; c < cu; ++c)
            gi[c] += -2.f * kaa * (ai[c] - aj[c]) * inv_s2 + 2.f * kab * (ai[c] - bj[c]) * inv_s2;
      }
      if (d_a)
        for (int c = 0; c < cu; ++c) d_a[((int64_t)i * npos + p) * c_a + c] += gi[c] * gscale;
    }
  }
  // block sum (thread 0 holds the result)
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    partial[blockIdx.x] = t;
  }
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int nblk, float scale,
                                    float* __restrict__ out) {
  __shared__ float sm[kB];
  float a = 0.f;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += partial[i];
  sm[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) t += sm[i];
    out[0] = t * scale;
  }
}

bool geom_ok(s3_ctx* ctx, const LGeom& g) {
  return ctx && g.n > 0 && g.s1 > 0 && g.s2 > 0 && g.t > 0 && g.cu > 0 && g.cu <= g.c;
}

}  // namespace

extern "C" int s3_lossmap_fwd(s3_ctx* ctx, int kind, const float* x, int n, int s1, int s2,
                              int t, int c, int c_used, int p0, int p1, int p2, float* out,
                              float* work) {
  LGeom g{n, s1, s2, t, c, c_used};
  if (!geom_ok(ctx, g)) return S3_EINVAL;
  const int64_t nel = (int64_t)n * s1 * s2 * t * c_used;
  switch (kind) {
    case S3_LMAP_DERIV_S:
    case S3_LMAP_DERIV_T:
      if (kind == S3_LMAP_DERIV_S ? (s1 < 2 || s2 < 2) : t < 2)
        S3_FAIL(ctx, S3_EINVAL, "lossmap: derivative axis shorter than 2");
      hipLaunchKernelGGL(deriv_fwd_kernel, dim3(grid_of(nel, ctx->num_cu)), dim3(kB), 0, ctx->stream,
                         x, out, g, kind == S3_LMAP_DERIV_T);
      break;
    case S3_LMAP_MATERIAL: {
      const int hub = c_used / 2;
      if (hub < 1 || s1 < 2 || s2 < 2 || t < 2)
        S3_FAIL(ctx, S3_EINVAL, "lossmap: material derivative needs a (u, v) pair and axes >= 2");
      hipLaunchKernelGGL(material_fwd_kernel, dim3(grid_of(nel / c_used * hub, ctx->num_cu)), dim3(kB),
                         0, ctx->stream, x, out, g, hub);
    } break;
    case S3_LMAP_MEAN_S:
    case S3_LMAP_EXT_S: {
      // work: n * kSlabs * t * cu floats
      if (!work) S3_FAIL(ctx, S3_EINVAL, "lossmap: spatial reductions need a work buffer");
      const int tc = t * c_used;
      const int nwhat = kind == S3_LMAP_MEAN_S ? 1 : 2;
      for (int q = 0; q < nwhat; ++q) {
        const int what = kind == S3_LMAP_MEAN_S ? 0 : 1 + q;
        hipLaunchKernelGGL(reduce_s_stage1, dim3(kSlabs, n), dim3(kB), 0, ctx->stream, x,
                           (const float*)nullptr, work, g, what);
        hipLaunchKernelGGL(reduce_stage2, dim3(grid_of((int64_t)n * tc, ctx->num_cu)), dim3(kB), 0,
                           ctx->stream, work, out + (int64_t)q * n * tc, n, kSlabs, tc, what,
                           what == 0 ? 1.f / (float)(s1 * s2) : 1.f);
      }
    } break;
    case S3_LMAP_EXT_T: {
      const int64_t ne = (int64_t)n * s1 * s2 * c_used;
      for (int q = 0; q < 2; ++q)
        hipLaunchKernelGGL(reduce_t_kernel, dim3(grid_of(ne, ctx->num_cu)), dim3(kB), 0, ctx->stream,
                           x, (const float*)nullptr, out + q * ne, g, 1 + q);
    } break;
    default:
      S3_FAIL(ctx, S3_EINVAL, "lossmap_fwd: unknown kind");
  }
  (void)p0; (void)p1; (void)p2;
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_lossmap_bwd(s3_ctx* ctx, int kind, const float* x, const float* fx,
                              const float* g_out, int n, int s1, int s2, int t, int c,
                              int c_used, int p0, int p1, int p2, float* d_x, float* work) {
  LGeom g{n, s1, s2, t, c, c_used};
  if (!geom_ok(ctx, g)) return S3_EINVAL;
  const int64_t nel = (int64_t)n * s1 * s2 * t * c_used;
  const dim3 grid(grid_of(nel, ctx->num_cu));
  switch (kind) {
    case S3_LMAP_DERIV_S:
    case S3_LMAP_DERIV_T:
      hipLaunchKernelGGL(deriv_bwd_kernel, grid, dim3(kB), 0, ctx->stream, g_out, d_x, g,
                         kind == S3_LMAP_DERIV_T);
      break;
    case S3_LMAP_MATERIAL: {
      const int hub = c_used / 2;
      hipLaunchKernelGGL(material_bwd_kernel, dim3(grid_of(nel / c_used * hub, ctx->num_cu)), dim3(kB),
                         0, ctx->stream, x, g_out, d_x, g, hub);
    } break;
    case S3_LMAP_MEAN_S:
      hipLaunchKernelGGL(mean_s_bwd_kernel, grid, dim3(kB), 0, ctx->stream, g_out, d_x, g);
      break;
    case S3_LMAP_EXT_S:
    case S3_LMAP_EXT_T: {
      // fx = [min | max] from the forward map; work: counts [cmin | cmax] (+ slabs)
      if (!work || !fx) S3_FAIL(ctx, S3_EINVAL, "lossmap: extremes adjoint needs fx and a work buffer");
      const bool sp = kind == S3_LMAP_EXT_S;
      const int64_t ne = sp ? (int64_t)n * t * c_used : (int64_t)n * s1 * s2 * c_used;
      float* cnt = work;
      float* slab = work + 2 * ne;
      for (int q = 0; q < 2; ++q) {
        if (sp) {
          const int tc = t * c_used;
          hipLaunchKernelGGL(reduce_s_stage1, dim3(kSlabs, n), dim3(kB), 0, ctx->stream, x,
                             fx + q * ne, slab, g, 3);
          hipLaunchKernelGGL(reduce_stage2, dim3(grid_of(ne, ctx->num_cu)), dim3(kB), 0, ctx->stream,
                             slab, cnt + q * ne, n, kSlabs, tc, 0, 1.f);
        } else {
          hipLaunchKernelGGL(reduce_t_kernel, dim3(grid_of(ne, ctx->num_cu)), dim3(kB), 0, ctx->stream,
                             x, fx + q * ne, cnt + q * ne, g, 3);
        }
      }
      hipLaunchKernelGGL(ext_bwd_kernel, grid, dim3(kB), 0, ctx->stream, x, fx, fx + ne, cnt, cnt + ne,
                         g_out, g_out + ne, d_x, g, sp ? 1 : 0);
    } break;
    case S3_LMAP_COARSEN:
      if (p0 < 1 || s1 % p0 || s2 % p0 || (p1 > 1 && t % p1))
        S3_FAIL(ctx, S3_EINVAL, "lossmap: enhancement factors must divide the grid");
      hipLaunchKernelGGL(coarsen_bwd_kernel, grid, dim3(kB), 0, ctx->stream, g_out, d_x, g, p0, p1, p2);
      break;
    default:
      S3_FAIL(ctx, S3_EINVAL, "lossmap_bwd: unknown kind");
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_loss_mmd(s3_ctx* ctx, const float* a, int c_a, const float* b, int c_b, int n,
                           int64_t n_pos, int c_used, float sigma, float weight, float* loss_out,
                           float* d_a) {
  if (!ctx || n < 1 || n_pos < 1 || c_used < 1 || c_used > 8 || c_used > c_a || c_used > c_b)
    return S3_EINVAL;
  int nblk = grid_of(n_pos, ctx->num_cu);
  if (nblk > 1024) nblk = 1024;
  int rc = ensure_scratch(ctx, (size_t)(nblk + 4) * sizeof(float));
  if (rc) return rc;
  const float norm = 1.f / ((float)n * (float)n * (float)n_pos);
  hipLaunchKernelGGL(mmd_kernel, dim3(nblk), dim3(kB), 0, ctx->stream, a, c_a, b, c_b, n, n_pos,
                     c_used, 1.f / (sigma * sigma), weight * norm, (float*)ctx->scratch, d_a);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kB), 0, ctx->stream, (const float*)ctx->scratch,
                     nblk, norm, loss_out);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// ===========================================================================
// SpatialFftLoss / SpatiotemporalFftLoss (loss_metrics.py:395-485): MAE between
// log(1 + w |FFT(x)|) of generated and true fields, w = product of the squared
// (un-wrapped) frequency indices.  The transform is a separable direct DFT —
// one pass per axis over a (outer, L, inner) view, L <= 512 here (80, 288):
// a workgroup holds a panel of columns and the L twiddles in LDS.  Unnormalised,
// sign = -1 forward (tf.signal.fft2d / fft3d), +1 for the adjoint.
namespace {

constexpr int DFT_COLS = 32;      // columns (outer x inner) per workgroup

__global__ __launch_bounds__(256) void dft_axis_kernel(
    const float* __restrict__ in_re, const float* __restrict__ in_im,
    float* __restrict__ out_re, float* __restrict__ out_im, int64_t outer, int L, int64_t inner,
    float sign) {
  extern __shared__ float dsm[];
  float* pre = dsm;                       // [L][DFT_COLS]
  float* pim = pre + (size_t)L * DFT_COLS;
  float* tc = pim + (size_t)L * DFT_COLS; // cos, sin of 2 pi m / L
  float* ts = tc + L;
  const int64_t ncol = outer * inner;
  const int64_t col0 = (int64_t)blockIdx.x * DFT_COLS;
  for (int m = threadIdx.x; m < L; m += blockDim.x) {
    float s, c;
    sincospif(2.f * (float)m / (float)L, &s, &c);
    tc[m] = c; ts[m] = sign * s;
  }
  for (int e = threadIdx.x; e < L * DFT_COLS; e += blockDim.x) {
    const int j = e / DFT_COLS, cc = e % DFT_COLS;
    const int64_t col = col0 + cc;
    float vr = 0.f, vi = 0.f;
    if (col < ncol) {
      const int64_t o = col / inner, i = col % inner;
      const int64_t a = (o * L + j) * inner + i;
      vr = in_re[a];
      vi = in_im ? in_im[a] : 0.f;
    }
    pre[e] = vr; pim[e] = vi;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L * DFT_COLS; e += blockDim.x) {
    const int k = e / DFT_COLS, cc = e % DFT_COLS;
    const int64_t col = col0 + cc;
    if (col >= ncol) continue;
    float ar = 0.f, ai = 0.f;
    int idx = 0;
    for (int j = 0; j < L; ++j) {
      const float c = tc[idx], s = ts[idx];
      const float xr = pre[j * DFT_COLS + cc], xi = pim[j * DFT_COLS + cc];
      ar += xr * c - xi * s;
      ai += xr * s + xi * c;
      idx += k;
      if (idx >= L) idx -= L;
    }
    const int64_t o = col / inner, i = col % inner;
    const int64_t a = (o * L + k) * inner + i;
    out_re[a] = ar; out_im[a] = ai;
  }
}

// y = log(1 + w |X|); mode3d: w = k1^2 k2^2 kt^2, else k1^2 k2^2
__global__ void specmap_fwd_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                   float* __restrict__ y, LGeom g, int mode3d) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.c;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.c, n, i1, i2, it, ch);
    float w = (float)i1 * (float)i1 * (float)i2 * (float)i2;
    if (mode3d) w *= (float)it * (float)it;
    y[idx] = log1pf(w * sqrtf(re[idx] * re[idx] + im[idx] * im[idx]));
  }
}

// G = g_y * w / (1 + w |X|) * X / |X|  (0 where |X| = 0, as tf.abs does)
__global__ void specmap_bwd_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                   const float* __restrict__ gy, float* __restrict__ gre,
                                   float* __restrict__ gim, LGeom g, int mode3d) {
  const int64_t total = (int64_t)g.n * g.s1 * g.s2 * g.t * g.c;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int n, i1, i2, it, ch;
    decode(idx, g, g.c, n, i1, i2, it, ch);
    float w = (float)i1 * (float)i1 * (float)i2 * (float)i2;
    if (mode3d) w *= (float)it * (float)it;
    const float xr = re[idx], xi = im[idx];
    const float mag = sqrtf(xr * xr + xi * xi);
    const float f = mag > 0.f ? gy[idx] * w / ((1.f + w * mag) * mag) : 0.f;
    gre[idx] = f * xr; gim[idx] = f * xi;
  }
}

}  // namespace

extern "C" int s3_dft_axis(s3_ctx* ctx, const float* in_re, const float* in_im, float* out_re,
                           float* out_im, int64_t outer, int L, int64_t inner, int sign) {
  if (!ctx || L < 1 || outer < 1 || inner < 1) return S3_EINVAL;
  if (L > 1024) S3_FAIL(ctx, S3_EINVAL, "dft_axis: axis longer than 1024");
  const size_t lds = ((size_t)2 * L * DFT_COLS + 2 * L) * sizeof(float);
  const int64_t ncol = outer * inner;
  const void* kern = reinterpret_cast<const void*>(dft_axis_kernel);
  if (lds > 64 * 1024)
    S3_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(dft_axis_kernel, dim3((unsigned)((ncol + DFT_COLS - 1) / DFT_COLS)), dim3(256),
                     lds, ctx->stream, in_re, in_im, out_re, out_im, outer, L, inner,
                     sign < 0 ? -1.f : 1.f);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_specmap(s3_ctx* ctx, int backward, const float* re, const float* im,
                          const float* g_y, int n, int s1, int s2, int t, int c, int mode3d,
                          float* out0, float* out1) {
  LGeom g{n, s1, s2, t, c, c};
  if (!geom_ok(ctx, g)) return S3_EINVAL;
  const int64_t total = (int64_t)n * s1 * s2 * t * c;
  if (!backward)
    hipLaunchKernelGGL(specmap_fwd_kernel, dim3(grid_of(total, ctx->num_cu)), dim3(kB), 0,
                       ctx->stream, re, im, out0, g, mode3d);
  else
    hipLaunchKernelGGL(specmap_bwd_kernel, dim3(grid_of(total, ctx->num_cu)), dim3(kB), 0,
                       ctx->stream, re, im, g_y, out0, out1, g, mode3d);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
