// Time windows of a (outer = n * s1 * s2, T, c) field and their adjoints — the
// slicing SolarCC.calc_loss does on the hi-res tensors (sup3r/models/
// solar_cc.py:155-232): hi_res[:, :, :, t0:t0+len, :] as a contiguous tensor,
// tf.reduce_mean(hi_res[:, :, :, t0:t0+len, :], axis=3), and the gradient of
// either scattered back into the full field.
#include "common.h"

namespace {

// adjoint == 0: window[o][j][c] = full[o][t0 + j][c]
// adjoint != 0: full[o][t0 + j][c] += scale * window[o][j][c]
__global__ void time_window_kernel(float* __restrict__ full, float* __restrict__ window, int64_t outer,
                                   int T, int c, int t0, int len, int adjoint, float scale) {
  const int64_t row = (int64_t)len * c;
  const int64_t total = outer * row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = idx / row, r = idx - o * row;
    const int64_t f = (o * T + t0) * c + r;
    if (adjoint) full[f] += scale * window[idx];
    else window[idx] = full[f];
  }
}

// adjoint == 0: mean[o][c] = (1 / len) sum_j full[o][t0 + j][c]
// adjoint != 0: full[o][t0 + j][c] += (scale / len) * mean[o][c]
__global__ void time_mean_kernel(float* __restrict__ full, float* __restrict__ mean, int64_t outer, int T,
                                 int c, int t0, int len, int adjoint, float scale) {
  const int64_t total = outer * c;
  const float inv = 1.f / (float)len;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = idx / c;
    const int ch = (int)(idx - o * c);
    float* base = full + (o * T + t0) * c + ch;
    if (adjoint) {
      const float g = scale * inv * mean[idx];
      for (int j = 0; j < len; ++j) base[(int64_t)j * c] += g;
    } else {
      float s = 0.f;
      for (int j = 0; j < len; ++j) s += base[(int64_t)j * c];
      mean[idx] = s * inv;
    }
  }
}

int grid_tw(int64_t n, int num_cu) {
  int64_t b = (n + 255) / 256;
  const int64_t cap = (int64_t)num_cu * 16;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int s3_time_window(s3_ctx* ctx, float* full, int64_t outer, int t, int c, int t0, int len,
                              float* window, int adjoint, float scale) {
  if (!ctx || !full || !window || outer < 1 || t < 1 || c < 1) return S3_EINVAL;
  if (t0 < 0 || len < 1 || t0 + len > t) S3_FAIL(ctx, S3_EINVAL, "time_window: slice outside the time axis");
  hipLaunchKernelGGL(time_window_kernel, dim3(grid_tw(outer * len * c, ctx->num_cu)), dim3(256), 0, ctx->stream,
                     full, window, outer, t, c, t0, len, adjoint, scale);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_time_mean(s3_ctx* ctx, float* full, int64_t outer, int t, int c, int t0, int len,
                            float* mean, int adjoint, float scale) {
  if (!ctx || !full || !mean || outer < 1 || t < 1 || c < 1) return S3_EINVAL;
  if (t0 < 0 || len < 1 || t0 + len > t) S3_FAIL(ctx, S3_EINVAL, "time_mean: slice outside the time axis");
  hipLaunchKernelGGL(time_mean_kernel, dim3(grid_tw(outer * c, ctx->num_cu)), dim3(256), 0, ctx->stream, full,
                     mean, outer, t, c, t0, len, adjoint, scale);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
