// Batch transform on the device: the step immediately
// before the hot path.  SingleBatchQueue.transform (batch_queues/base.py:
// 32-87) derives the low-res batch from the hi-res samples on the host with
//   spatial_coarsening   (utilities/utilities.py:406-523: s x s block mean),
//   temporal_coarsening  (utilities/utilities.py:345-403: subsample / average
//                         / total / max / min over t_enhance steps),
//   smooth_data          (batch_queues/utilities.py:57-103: scipy
//                         gaussian_filter(sigma, mode='nearest') over the two
//                         spatial axes of every (obs, t, feature) slice).
// All three are HBM-bound streaming ops; coarsening reads the hi-res batch
// once and writes the low-res one (fused spatial + temporal), the filter is
// separable (two passes, weights in a kernel argument).
#include "common.h"

namespace {

constexpr int kBlk = 256;
constexpr int kMaxRadius = 32;

struct GaussW {
  float w[2 * kMaxRadius + 1];
};

// y[n, a, b, q, c] = reduce_{i<s, j<s, r<t} x[n, a s + i, b s + j, q t + r, c]
// summation order = numpy's: the block sum over (i, j) first (/ s^2), then the
// temporal reduction of the spatial means
__global__ void coarsen_kernel(const float* __restrict__ x, float* __restrict__ y,
                               int N, int S1, int S2, int T, int C, int s, int t,
                               int method) {
  const int O1 = S1 / s, O2 = S2 / s, OT = method < 0 ? T : T / t;
  const int64_t total = (int64_t)N * O1 * O2 * OT * C;
  const float inv = 1.f / (float)(s * s);
  for (int64_t idx = (int64_t)blockIdx.x * kBlk + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kBlk) {
    int64_t r = idx;
    const int c = (int)(r % C); r /= C;
    const int q = (int)(r % OT); r /= OT;
    const int b = (int)(r % O2); r /= O2;
    const int a = (int)(r % O1); r /= O1;
    const int n = (int)r;
    const int nt = method < 0 ? 1 : (method == S3_TC_SUBSAMPLE ? 1 : t);
    const int tq = method < 0 ? q : q * t;
    float accv = 0.f;
    for (int rr = 0; rr < nt; ++rr) {
      float sm = 0.f;
      for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j)
          sm += x[((((int64_t)n * S1 + a * s + i) * S2 + b * s + j) * T + tq + rr) * C + c];
      sm *= inv;
      if (rr == 0) accv = sm;
      else if (method == S3_TC_MAX) accv = fmaxf(accv, sm);
      else if (method == S3_TC_MIN) accv = fminf(accv, sm);
      else accv += sm;
    }
    if (method == S3_TC_AVERAGE) accv /= (float)t;
    y[idx] = accv;
  }
}

// one separable pass of scipy's gaussian_filter(mode='nearest') along spatial
// axis `axis` (0: s1, 1: s2) of x (N, S1, S2, T, C); channels whose bit is not
// set in `cmask` are copied
__global__ void gauss_pass_kernel(const float* __restrict__ x, float* __restrict__ y,
                                  int N, int S1, int S2, int T, int C, int axis,
                                  int radius, GaussW gw, unsigned cmask) {
  const int64_t total = (int64_t)N * S1 * S2 * T * C;
  const int L = axis == 0 ? S1 : S2;
  const int64_t stride = axis == 0 ? (int64_t)S2 * T * C : (int64_t)T * C;
  for (int64_t idx = (int64_t)blockIdx.x * kBlk + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(idx % C);
    if (!((cmask >> c) & 1u)) { y[idx] = x[idx]; continue; }
    const int pos = (int)((idx / stride) % L);
    const int64_t base = idx - (int64_t)pos * stride;
    float acc = 0.f;
    for (int k = -radius; k <= radius; ++k) {
      int p = pos + k;
      p = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);       // mode='nearest'
      acc += gw.w[k + radius] * x[base + (int64_t)p * stride];
    }
    y[idx] = acc;
  }
}

// u/v -> windspeed / winddirection in place (writers/base.py:233-302 +
// derivers/utilities.py:204-258): rotate by the grid angle theta(s1, s2) back
// to the meridian frame, ws = hypot, wd = (degrees(atan2(u, v)) + 360) % 360
__global__ void invert_uv_kernel(float* __restrict__ data, int64_t n_sp, int64_t t, int c,
                                 int u_idx, int v_idx, const float* __restrict__ cos_t,
                                 const float* __restrict__ sin_t) {
  const int64_t total = n_sp * t;
  for (int64_t idx = (int64_t)blockIdx.x * kBlk + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kBlk) {
    const int64_t sp = idx / t;
    const float cs = cos_t[sp], sn = sin_t[sp];
    float* cell = data + idx * c;
    const float u = cell[u_idx], v = cell[v_idx];
    const float ur = cs * u - sn * v, vr = sn * u + cs * v;
    cell[u_idx] = hypotf(ur, vr);
    cell[v_idx] = fmodf(atan2f(ur, vr) * 57.29577951308232f + 360.f, 360.f);
  }
}

struct Clip16 { float lo[16]; float hi[16]; };
// enforce_limits with clipping (utilities/utilities.py:155-220): NaNs pass
// through np.maximum / np.minimum unchanged, so they do here
__global__ void clip_channels_kernel(float* __restrict__ data, int c, int64_t n, Clip16 lim) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlk) {
    const int ch = (int)(i % c);
    float v = data[i];
    if (v < lim.lo[ch]) v = lim.lo[ch];
    if (v > lim.hi[ch]) v = lim.hi[ch];
    data[i] = v;
  }
}

int grid_of(int64_t n, int num_cu) {
  int64_t g = (n + kBlk - 1) / kBlk;
  const int64_t cap = (int64_t)num_cu * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int s3_coarsen(s3_ctx* ctx, const float* hr, int n, int s1, int s2,
                          int t, int c, int s_enhance, int t_enhance,
                          int t_method, float* lr) {
  if (!ctx) return S3_EINVAL;
  if (s_enhance < 1 || s1 % s_enhance || s2 % s_enhance)
    S3_FAIL(ctx, S3_EINVAL, "s_enhance must evenly divide grid size");
  int method = t_method;
  if (t_enhance <= 1) method = -1;                   // spatial only
  else if (t % t_enhance)
    S3_FAIL(ctx, S3_EINVAL, "t_enhance must evenly divide the time axis");
  else if (method < S3_TC_SUBSAMPLE || method > S3_TC_MIN)
    S3_FAIL(ctx, S3_EINVAL, "unknown temporal coarsening method");
  const int ot = method < 0 ? t : t / t_enhance;
  const int64_t total = (int64_t)n * (s1 / s_enhance) * (s2 / s_enhance) * ot * c;
  hipLaunchKernelGGL(coarsen_kernel, dim3(grid_of(total, ctx->num_cu)), dim3(kBlk), 0,
                     ctx->stream, hr, lr, n, s1, s2, t, c, s_enhance,
                     t_enhance < 1 ? 1 : t_enhance, method);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_gaussian_smooth(s3_ctx* ctx, const float* x, int n, int s1,
                                  int s2, int t, int c, const float* weights_host,
                                  int radius, unsigned channel_mask, float* tmp,
                                  float* y) {
  if (!ctx) return S3_EINVAL;
  if (radius < 0 || radius > kMaxRadius)
    S3_FAIL(ctx, S3_EINVAL, "gaussian radius out of range (<= 32)");
  if (c > 32) S3_FAIL(ctx, S3_EINVAL, "gaussian_smooth supports at most 32 channels");
  GaussW gw;
  for (int i = 0; i < 2 * radius + 1; ++i) gw.w[i] = weights_host[i];
  const int64_t total = (int64_t)n * s1 * s2 * t * c;
  const int grid = grid_of(total, ctx->num_cu);
  // scipy filters axis 0 first, then axis 1 (ndimage.gaussian_filter)
  hipLaunchKernelGGL(gauss_pass_kernel, dim3(grid), dim3(kBlk), 0, ctx->stream, x, tmp,
                     n, s1, s2, t, c, 0, radius, gw, channel_mask);
  hipLaunchKernelGGL(gauss_pass_kernel, dim3(grid), dim3(kBlk), 0, ctx->stream, tmp, y,
                     n, s1, s2, t, c, 1, radius, gw, channel_mask);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_invert_uv(s3_ctx* ctx, float* data, int64_t n_sp, int64_t t, int c,
                            int u_idx, int v_idx, const float* cos_theta,
                            const float* sin_theta) {
  if (!ctx || !data || !cos_theta || !sin_theta) return S3_EINVAL;
  if (u_idx < 0 || v_idx < 0 || u_idx >= c || v_idx >= c || u_idx == v_idx)
    S3_FAIL(ctx, S3_EINVAL, "invert_uv: bad channel indices");
  hipLaunchKernelGGL(invert_uv_kernel, dim3(grid_of(n_sp * t, ctx->num_cu)), dim3(kBlk), 0,
                     ctx->stream, data, n_sp, t, c, u_idx, v_idx, cos_theta, sin_theta);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// enforce_limits(nn_fill=True): mask of one channel's out-of-range (or NaN)
// values, and the refill of the masked positions from an index map
__global__ void range_mask_kernel(const float* __restrict__ data, int c, int ch, int64_t n_pos,
                                  float lo, float hi, unsigned char* __restrict__ mask) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pos;
       p += (int64_t)gridDim.x * blockDim.x) {
    const float v = data[p * c + ch];
    mask[p] = (v >= lo && v <= hi) ? 0 : 1;          // NaN compares false: masked
  }
}

__global__ void fill_indexed_kernel(float* __restrict__ data, int c, int ch, int64_t n_pos,
                                    const unsigned char* __restrict__ mask,
                                    const int* __restrict__ src) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pos;
       p += (int64_t)gridDim.x * blockDim.x)
    if (mask[p]) data[p * c + ch] = data[(int64_t)src[p] * c + ch];   // sources are never masked
}

extern "C" int s3_range_mask(s3_ctx* ctx, const float* data, int c, int ch, int64_t n_pos,
                             float lo, float hi, unsigned char* mask) {
  if (!ctx || !data || !mask) return S3_EINVAL;
  if (ch < 0 || ch >= c) S3_FAIL(ctx, S3_EINVAL, "range_mask: bad channel index");
  hipLaunchKernelGGL(range_mask_kernel, dim3(grid_of(n_pos, ctx->num_cu)), dim3(kBlk), 0, ctx->stream,
                     data, c, ch, n_pos, lo, hi, mask);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_fill_indexed(s3_ctx* ctx, float* data, int c, int ch, int64_t n_pos,
                               const unsigned char* mask, const int* src) {
  if (!ctx || !data || !mask || !src) return S3_EINVAL;
  if (ch < 0 || ch >= c) S3_FAIL(ctx, S3_EINVAL, "fill_indexed: bad channel index");
  if (n_pos >= ((int64_t)1 << 31)) S3_FAIL(ctx, S3_EINVAL, "fill_indexed: 32-bit position indices");
  hipLaunchKernelGGL(fill_indexed_kernel, dim3(grid_of(n_pos, ctx->num_cu)), dim3(kBlk), 0, ctx->stream,
                     data, c, ch, n_pos, mask, src);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_clip_channels(s3_ctx* ctx, float* data, int c, int64_t n_pos,
                                const float* min_host, const float* max_host) {
  if (!ctx || !data) return S3_EINVAL;
  if (c > 16) S3_FAIL(ctx, S3_EINVAL, "clip_channels supports at most 16 channels");
  Clip16 lim;
  for (int i = 0; i < c; ++i) { lim.lo[i] = min_host[i]; lim.hi[i] = max_host[i]; }
  hipLaunchKernelGGL(clip_channels_kernel, dim3(grid_of(n_pos * c, ctx->num_cu)), dim3(kBlk), 0,
                     ctx->stream, data, c, n_pos * c, lim);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
