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
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int s = 0;
    for (; s + 4 <= n_slabs; s += 4) {
      t0 += partial[(int64_t)s * total + idx]; t1 += partial[(int64_t)(s + 1) * total + idx];
      t2 += partial[(int64_t)(s + 2) * total + idx]; t3 += partial[(int64_t)(s + 3) * total + idx];
    }
    for (; s < n_slabs; ++s) t0 += partial[(int64_t)s * total + idx];
    float t = (t0 + t1) + (t2 + t3);
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

// ---- 16-B walks of W (cin % 4 == 0, cout % 4 == 0): with one dword per lane
// and row the three passes over the 126 MB of the first discriminator dense
// layer ran at 1.9 - 2.5 TB/s (65 / 59 / 51 us); a lane owns four consecutive
// outputs, a wave four consecutive rows per step (their x values are one
// 16-B scalar load per sample), four independent W rows in flight per lane.
__global__ __launch_bounds__(256) void dense_fwd4_stage1(
    const float* __restrict__ x, const float* __restrict__ w,
    float* __restrict__ partial, int n, int cin, int cout, int n0,
    int rows_per_slab) {
  __shared__ float4 red[4][NB][64];
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u4 = blockIdx.x * 64 + tx;               // float4 column
  const int c4 = cout >> 2;
  const int slab = blockIdx.y;
  const int i_begin = slab * rows_per_slab;          // (a multiple of 4)
  int i_end = i_begin + rows_per_slab;
  if (i_end > cin) i_end = cin;
  float4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool live = u4 < c4;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int i = i_begin + 4 * ty; i < i_end; i += 16) {
    float4 wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      wv[r] = live ? w4[(int64_t)(i + r) * c4 + u4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + b < n) xv = *reinterpret_cast<const float4*>(x + (int64_t)(n0 + b) * cin + i);
      const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[b].x = fmaf(xr[r], wv[r].x, acc[b].x); acc[b].y = fmaf(xr[r], wv[r].y, acc[b].y);
        acc[b].z = fmaf(xr[r], wv[r].z, acc[b].z); acc[b].w = fmaf(xr[r], wv[r].w, acc[b].w);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) red[ty][b][tx] = acc[b];
  __syncthreads();
  if (ty == 0 && live) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (n0 + b >= n) break;
      const float4 a0 = red[0][b][tx], a1 = red[1][b][tx], a2 = red[2][b][tx], a3 = red[3][b][tx];
      reinterpret_cast<float4*>(partial)[((int64_t)slab * n + n0 + b) * c4 + u4] =
          make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z,
                      a0.w + a1.w + a2.w + a3.w);
    }
  }
}

// dx[n][i] = sum_u dy[n][u] * W[i][u]; a wave owns FOUR rows i (one dy load
// per sample and step serves all four), 16 B of each W row per lane and step
constexpr int DGR = 4;
__global__ __launch_bounds__(256) void dense_dgrad4_kernel(
    const float* __restrict__ dy, const float* __restrict__ w,
    float* __restrict__ dx, int n, int cin, int cout, int n0) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i0 = (blockIdx.x * 4 + wave) * DGR;
  if (i0 >= cin) return;
  const int c4 = cout >> 2;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float acc[DGR][NB];
#pragma unroll
  for (int r = 0; r < DGR; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  for (int u = lane; u < c4; u += 64) {
    float4 wv[DGR];
#pragma unroll
    for (int r = 0; r < DGR; ++r)
      wv[r] = i0 + r < cin ? w4[(int64_t)(i0 + r) * c4 + u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (n0 + b < n) {
        const float4 d = reinterpret_cast<const float4*>(dy + (int64_t)(n0 + b) * cout)[u];
#pragma unroll
        for (int r = 0; r < DGR; ++r) {
          acc[r][b] = fmaf(d.x, wv[r].x, acc[r][b]); acc[r][b] = fmaf(d.y, wv[r].y, acc[r][b]);
          acc[r][b] = fmaf(d.z, wv[r].z, acc[r][b]); acc[r][b] = fmaf(d.w, wv[r].w, acc[r][b]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < DGR; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float v = acc[r][b];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0 && n0 + b < n && i0 + r < cin) dx[(int64_t)(n0 + b) * cin + i0 + r] = v;
    }
}

// dW[i][u] (+)= sum_n x[n][i] * dy[n][u], four outputs per lane, four rows per step
__global__ __launch_bounds__(256) void dense_wgrad4_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ dw, int n, int cin, int cout, int rows_per_block,
    int accumulate) {
  const int c4 = cout >> 2;
  const int u4 = blockIdx.x * 256 + threadIdx.x;
  if (u4 >= c4) return;
  const int i_begin = blockIdx.y * rows_per_block;   // (a multiple of 4)
  int i_end = i_begin + rows_per_block;
  if (i_end > cin) i_end = cin;
  float4* dw4 = reinterpret_cast<float4*>(dw);
  for (int n0 = 0; n0 < n; n0 += NB) {
    float4 d[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      d[b] = (n0 + b < n) ? reinterpret_cast<const float4*>(dy + (int64_t)(n0 + b) * cout)[u4]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool acc_mode = accumulate || n0 > 0;
    for (int i = i_begin; i < i_end; i += 4) {
      float4 t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        t[r] = acc_mode ? dw4[(int64_t)(i + r) * c4 + u4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + b < n) xv = *reinterpret_cast<const float4*>(x + (int64_t)(n0 + b) * cin + i);
        const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r].x = fmaf(xr[r], d[b].x, t[r].x); t[r].y = fmaf(xr[r], d[b].y, t[r].y);
          t[r].z = fmaf(xr[r], d[b].z, t[r].z); t[r].w = fmaf(xr[r], d[b].w, t[r].w);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dw4[(int64_t)(i + r) * c4 + u4] = t[r];
    }
  }
}

static bool dense4_ok(const void* a, const void* b, const void* c, int cin, int cout) {
  return (cin & 3) == 0 && (cout & 3) == 0 && cout >= 256 &&
         (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

}  // namespace

int launch_dense_fwd(s3_ctx* ctx, const float* x, const float* w,
                     const float* bias, float* y, int n, int cin, int cout,
                     int act, float alpha) {
  if (dense4_ok(x, w, ctx->scratch, cin, cout)) {
    const int col_tiles = (cout / 4 + 63) / 64;
    // (a workgroup moves 4 x the bytes per step of the dword walk: half its
    // workgroups keep stage 2's partial list short)
    const int wg_target = (int)s3_opt_int(S3O_DENSE_WGS, 1024);
    int n_slabs = (wg_target + col_tiles - 1) / col_tiles;
    const int max_slabs = (cin + 63) / 64;
    if (n_slabs > max_slabs) n_slabs = max_slabs;
    if (n_slabs < 1) n_slabs = 1;
    const int rows = ((cin + n_slabs - 1) / n_slabs + 15) / 16 * 16;   // whole 4-wave steps
    n_slabs = (cin + rows - 1) / rows;
    int rc = ensure_scratch(ctx, (size_t)n_slabs * n * cout * sizeof(float));
    if (rc) return rc;
    if (((uintptr_t)ctx->scratch & 15) == 0) {
      for (int n0 = 0; n0 < n; n0 += NB)
        hipLaunchKernelGGL(dense_fwd4_stage1, dim3(col_tiles, n_slabs), dim3(256), 0, ctx->stream, x, w,
                           ctx->scratch, n, cin, cout, n0, rows);
      const int64_t total = (int64_t)n * cout;
      int g2 = (int)((total + 255) / 256);
      if (g2 > 1024) g2 = 1024;
      hipLaunchKernelGGL(dense_fwd_stage2, dim3(g2), dim3(256), 0, ctx->stream, ctx->scratch, bias, y, n, cout,
                         n_slabs, act, alpha);
      S3_HIP(ctx, hipGetLastError());
      return S3_OK;
    }
  }
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
  if (dense4_ok(dy, w, dx, cin, cout)) {
    for (int n0 = 0; n0 < n; n0 += NB)
      hipLaunchKernelGGL(dense_dgrad4_kernel, dim3((cin + 4 * DGR - 1) / (4 * DGR)), dim3(256), 0, ctx->stream, dy,
                         w, dx, n, cin, cout, n0);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  for (int n0 = 0; n0 < n; n0 += NB)
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3((cin + 3) / 4), dim3(256), 0, ctx->stream, dy, w, dx, n, cin, cout, n0);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_dense_wgrad(s3_ctx* ctx, const float* x, const float* dy, float* dw,
                       int n, int cin, int cout, int accumulate) {
  if (dense4_ok(x, dy, dw, cin, cout)) {
    const int col_blocks = (cout / 4 + 255) / 256;
    int row_blocks = (2048 + col_blocks - 1) / col_blocks;
    if (row_blocks > cin / 4) row_blocks = cin / 4;
    const int rows = ((cin + row_blocks - 1) / row_blocks + 3) / 4 * 4;
    row_blocks = (cin + rows - 1) / rows;
    hipLaunchKernelGGL(dense_wgrad4_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, ctx->stream, x, dy, dw,
                       n, cin, cout, rows, accumulate);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  const int col_blocks = (cout + 255) / 256;
  int row_blocks = (1024 + col_blocks - 1) / col_blocks;
  if (row_blocks > cin) row_blocks = cin;
  const int rows = (cin + row_blocks - 1) / row_blocks;
  row_blocks = (cin + rows - 1) / rows;
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, ctx->stream, x, dy, dw, n, cin, cout, rows, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
