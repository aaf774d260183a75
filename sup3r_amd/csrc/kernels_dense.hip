// Flatten + Dense of the patch discriminator (K10): y = x W + b with
// W:(in, out) row-major, batch n small (<= 64).  Every pass is bound by
// streaming W (15360 x 2048 fp32 = 126 MB at the production shape) once from
// HBM, so: lanes run along `out` (coalesced 256-B rows of W), x / dy values
// are wave-uniform (scalar loads), the batch lives in registers (NB = 8 rows
// per pass), and K is split over the grid with a deterministic two-stage sum.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int NB = 8;

__device__ inline float act_f(float v, int act, float alpha) {
  if (act == S3_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == S3_ACT_LEAKY) return v > 0.f ? v : alpha * v;
  return v;
}

// partial[kslab][n][u]
__global__ __launch_bounds__(256) void dense_fwd_stage1(
    const float* __restrict__ x, const float* __restrict__ w,
    float* __restrict__ partial, int n, int cin, int cout, int n0,
    int rows_per_slab) {
  __shared__ float red[4][NB][64];
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = blockIdx.x * 64 + tx;
  const int slab = blockIdx.y;
  const int i_begin = slab * rows_per_slab;
  int i_end = i_begin + rows_per_slab;
  if (i_end > cin) i_end = cin;
  float acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = 0.f;
  const bool live = u < cout;
  for (int i = i_begin + ty; i < i_end; i += 4) {
    const float wv = live ? w[(int64_t)i * cout + u] : 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float xv = (n0 + b < n) ? x[(int64_t)(n0 + b) * cin + i] : 0.f;
      acc[b] = fmaf(xv, wv, acc[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) red[ty][b][tx] = acc[b];
  __syncthreads();
  if (ty == 0 && live) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (n0 + b >= n) break;
      float t = red[0][b][tx] + red[1][b][tx] + red[2][b][tx] + red[3][b][tx];
      partial[((int64_t)slab * n + n0 + b) * cout + u] = t;
    }
  }
}

__global__ void dense_fwd_stage2(const float* __restrict__ partial,
                                 const float* __restrict__ bias,
                                 float* __restrict__ y, int n, int cout,
                                 int n_slabs, int act, float alpha) {
  const int64_t total = (int64_t)n * cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < n_slabs; ++s) t += partial[(int64_t)s * total + idx];
    if (bias) t += bias[idx % cout];
    y[idx] = act_f(t, act, alpha);
  }
}

// dx[n][i] = sum_u dy[n][u] * W[i][u]; one wave per row i
__global__ __launch_bounds__(256) void dense_dgrad_kernel(
    const float* __restrict__ dy, const float* __restrict__ w,
    float* __restrict__ dx, int n, int cin, int cout, int n0) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= cin) return;
  float acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = 0.f;
  for (int u = lane; u < cout; u += 64) {
    const float wv = w[(int64_t)i * cout + u];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float d = (n0 + b < n) ? dy[(int64_t)(n0 + b) * cout + u] : 0.f;
      acc[b] = fmaf(d, wv, acc[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float v = acc[b];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0 && n0 + b < n) dx[(int64_t)(n0 + b) * cin + i] = v;
  }
}

// dW[i][u] (+)= sum_n x[n][i] * dy[n][u]
__global__ __launch_bounds__(256) void dense_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ dw, int n, int cin, int cout, int rows_per_block,
    int accumulate) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= cout) return;
  const int i_begin = blockIdx.y * rows_per_block;
  int i_end = i_begin + rows_per_block;
  if (i_end > cin) i_end = cin;
  for (int n0 = 0; n0 < n; n0 += NB) {
    float d[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      d[b] = (n0 + b < n) ? dy[(int64_t)(n0 + b) * cout + u] : 0.f;
    const bool acc_mode = accumulate || n0 > 0;
    for (int i = i_begin; i < i_end; ++i) {
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float xv = (n0 + b < n) ? x[(int64_t)(n0 + b) * cin + i] : 0.f;
        t = fmaf(xv, d[b], t);
      }
      float* o = dw + (int64_t)i * cout + u;
      *o = acc_mode ? *o + t : t;
    }
  }
}

}  // namespace

int launch_dense_fwd(s3_ctx* ctx, const float* x, const float* w,
                     const float* bias, float* y, int n, int cin, int cout,
                     int act, float alpha) {
  const int col_tiles = (cout + 63) / 64;
  // ~8 waves per SIMD in flight: the pass is one stream over W and needs the
  // memory parallelism (512 workgroups ran it at 1.1 TB/s)
  const int wg_target = (int)s3_opt_int(S3O_DENSE_WGS, 2048);
  int n_slabs = (wg_target + col_tiles - 1) / col_tiles;
  int max_slabs = (cin + 63) / 64;
  if (n_slabs > max_slabs) n_slabs = max_slabs;
  if (n_slabs < 1) n_slabs = 1;
  const int rows = (cin + n_slabs - 1) / n_slabs;
  n_slabs = (cin + rows - 1) / rows;
  int rc = ensure_scratch(ctx, (size_t)n_slabs * n * cout * sizeof(float));
  if (rc) return rc;
  for (int n0 = 0; n0 < n; n0 += NB)
    hipLaunchKernelGGL(dense_fwd_stage1, dim3(col_tiles, n_slabs), dim3(256), 0, ctx->stream, x, w, ctx->scratch, n, cin, cout, n0, rows);
  int64_t total = (int64_t)n * cout;
  int g2 = (int)((total + 255) / 256);
  if (g2 > 1024) g2 = 1024;
  hipLaunchKernelGGL(dense_fwd_stage2, dim3(g2), dim3(256), 0, ctx->stream, ctx->scratch, bias, y, n, cout, n_slabs, act, alpha);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_dense_dgrad(s3_ctx* ctx, const float* dy, const float* w, float* dx,
                       int n, int cin, int cout) {
  for (int n0 = 0; n0 < n; n0 += NB)
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3((cin + 3) / 4), dim3(256), 0, ctx->stream, dy, w, dx, n, cin, cout, n0);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_dense_wgrad(s3_ctx* ctx, const float* x, const float* dy, float* dw,
                       int n, int cin, int cout, int accumulate) {
  const int col_blocks = (cout + 255) / 256;
  int row_blocks = (1024 + col_blocks - 1) / col_blocks;
  if (row_blocks > cin) row_blocks = cin;
  const int rows = (cin + row_blocks - 1) / row_blocks;
  row_blocks = (cin + rows - 1) / rows;
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, ctx->stream, x, dy, dw, n, cin, cout, rows, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
