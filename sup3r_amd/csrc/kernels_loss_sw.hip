// SlicedWassersteinLoss (sup3r/utilities/loss_metrics.py:724-789): n_proj
// random unit directions over the H*W*T positions, every (observation, feature)
// column projected onto them, the n_proj projections of a column sorted, mean
// squared difference of the sorted generated / true projections.
//
// At the C2 hi-res batch the direction matrix would be 1024 x 1.84 M floats
// (7.5 GB): it is never materialised.  Direction p at position l is a
// counter-based draw — Philox4x32-10 keyed by the call's seed, counter
// (l >> 2, p), Box-Muller on the four words -> the normals of positions
// 4 (l >> 2) .. + 3 — so the forward (lane = projection, positions streamed
// through LDS) and the backward (lane = four positions, projections streamed
// through LDS) regenerate identical directions.  The reference draws new
// directions per call (tf.random.normal); so does the caller here, by seed.
// Everything is deterministic for a given seed: partial sums per position
// range reduced in fixed order, bitonic sort, tree reductions, no atomics.
#include <cstdint>

#include "common.h"

namespace {

constexpr int SWB = 256;   // projections per workgroup (forward) / lanes
constexpr int SWC = 16;    // (observation, feature) columns per pass
constexpr int SWT = 256;   // positions per LDS tile
constexpr int SWROW = 2 * SWC + 1;  // partial row: 16 + 16 sums, |dir|^2
constexpr int SWPC = 512;  // projections per LDS chunk (backward)

__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the four standard normals of projection p at positions 4 l4 .. 4 l4 + 3
__device__ __forceinline__ void sw_normals(uint64_t seed, uint32_t p, uint64_t l4, float z[4]) {
  uint32_t r[4];
  philox4((uint32_t)l4, (uint32_t)(l4 >> 32), p, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float s = 1.f / 16777216.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * s;   // (0, 1)
    const float u2 = (float)(r[2 * h + 1] >> 8) * s;        // [0, 1)
    const float rad = sqrtf(-2.f * __logf(u1));
    const float ang = 6.28318530718f * u2;
    z[2 * h] = rad * __cosf(ang);
    z[2 * h + 1] = rad * __sinf(ang);
  }
}

// raw (un-normalised) direction matrix [n_proj][L] — test / debugging aid
__global__ void sw_directions_kernel(uint64_t seed, int P, int64_t L, float* __restrict__ out) {
  const int64_t g4 = (L + 3) >> 2;
  const int64_t total = (int64_t)P * g4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(idx / g4);
    const int64_t l4 = idx - (int64_t)p * g4;
    float z[4];
    sw_normals(seed, (uint32_t)p, (uint64_t)l4, z);
    for (int k = 0; k < 4; ++k)
      if (4 * l4 + k < L) out[(int64_t)p * L + 4 * l4 + k] = z[k];
  }
}

// partial[r][p][0..31] = sum over the positions of range r of dir[p][l] * column value
// (0..15: a, 16..31: b), [32] = sum dir^2
__global__ __launch_bounds__(SWB) void sw_project_kernel(
    const float* __restrict__ a, int c_a, const float* __restrict__ b, int c_b, int64_t L,
    int c_used, int col0, int ncol, int P, uint64_t seed, int64_t ntiles, int64_t tiles_per_range,
    float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float xs[SWT][2 * SWC];
  const int tid = threadIdx.x;
  const int p = blockIdx.x * SWB + tid;
  const int64_t t_begin = (int64_t)blockIdx.y * tiles_per_range;
  const int64_t t_end = t_begin + tiles_per_range < ntiles ? t_begin + tiles_per_range : ntiles;
  float acc[2 * SWC];
#pragma unroll
  for (int j = 0; j < 2 * SWC; ++j) acc[j] = 0.f;
  float nrm = 0.f;
  for (int64_t tile = t_begin; tile < t_end; ++tile) {
    const int64_t l0 = tile * SWT;
    __syncthreads();
    {
      const int64_t l = l0 + tid;
      for (int j = 0; j < SWC; ++j) {
        float va = 0.f, vb = 0.f;
        if (j < ncol && l < L) {
          const int col = col0 + j, bi = col / c_used, ci = col - bi * c_used;
          va = a[((int64_t)bi * L + l) * c_a + ci];
          vb = b[((int64_t)bi * L + l) * c_b + ci];
        }
        xs[tid][j] = va;
        xs[tid][SWC + j] = vb;
      }
    }
    __syncthreads();
    if (p < P) {
#pragma unroll 1
      for (int q = 0; q < SWT / 4; ++q) {
        float z[4];
        sw_normals(seed, (uint32_t)p, (uint64_t)((l0 >> 2) + q), z);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (l0 + 4 * q + k >= L) z[k] = 0.f;
          nrm += z[k] * z[k];
          const float4* row = reinterpret_cast<const float4*>(xs[4 * q + k]);
#pragma unroll
          for (int j4 = 0; j4 < 2 * SWC / 4; ++j4) {
            const float4 v = row[j4];   // one address per wave: LDS broadcast
            acc[4 * j4 + 0] += z[k] * v.x; acc[4 * j4 + 1] += z[k] * v.y;
            acc[4 * j4 + 2] += z[k] * v.z; acc[4 * j4 + 3] += z[k] * v.w;
          }
        }
      }
    }
  }
  if (p < P) {
    float* dst = partial + ((int64_t)blockIdx.y * P + p) * SWROW;
#pragma unroll
    for (int j = 0; j < 2 * SWC; ++j) dst[j] = acc[j];
    dst[2 * SWC] = nrm;
  }
}

// fixed-order sum over the ranges: raw[j][p] (j < 32), inv_norm[p] = 1 / |dir p|
__global__ void sw_reduce_kernel(const float* __restrict__ partial, int R, int P,
                                 float* __restrict__ raw, float* __restrict__ inv_norm) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * SWROW) return;
  const int p = idx / SWROW, j = idx - p * SWROW;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += partial[(int64_t)r * P * SWROW + idx];
  if (j == 2 * SWC) inv_norm[p] = s > 0.f ? rsqrtf(s) : 0.f;
  else raw[(int64_t)j * P + p] = s;
}

// one workgroup per column: sort the P normalised projections of a and b,
// column loss = sum (a_sorted - b_sorted)^2, g[p][j] = scale * (a_sorted[rank of p]
// - b_sorted[rank of p]) * inv_norm[p]  (d loss / d raw projection)
__global__ __launch_bounds__(256) void sw_sort_kernel(const float* __restrict__ raw,
                                                      const float* __restrict__ inv_norm, int P,
                                                      int P2, int ncol, float scale,
                                                      float* __restrict__ g,
                                                      float* __restrict__ colloss) {
  extern __shared__ float sm[];
  float* v1 = sm;
  float* v2 = sm + P2;
  int* i1 = reinterpret_cast<int*>(sm + 2 * P2);
  __shared__ float red[256];
  const int j = blockIdx.x, tid = threadIdx.x;
  if (j >= ncol) {
    for (int p = tid; p < P; p += 256) g[(int64_t)p * SWC + j] = 0.f;
    return;
  }
  for (int p = tid; p < P2; p += 256) {
    const float in = p < P ? inv_norm[p] : 0.f;
    v1[p] = p < P ? raw[(int64_t)j * P + p] * in : INFINITY;
    v2[p] = p < P ? raw[(int64_t)(SWC + j) * P + p] * in : INFINITY;
    i1[p] = p;
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)
    for (int s = k >> 1; s > 0; s >>= 1) {
      for (int i = tid; i < P2; i += 256) {
        const int o = i ^ s;
        if (o > i) {
          const bool up = (i & k) == 0;
          const float x = v1[i], y = v1[o];
          if ((x > y) == up && x != y) {
            v1[i] = y; v1[o] = x;
            const int t = i1[i]; i1[i] = i1[o]; i1[o] = t;
          }
          const float x2 = v2[i], y2 = v2[o];
          if ((x2 > y2) == up && x2 != y2) { v2[i] = y2; v2[o] = x2; }
        }
      }
      __syncthreads();
    }
  float local = 0.f;
  for (int r = tid; r < P; r += 256) {
    const float d = v1[r] - v2[r];
    local += d * d;
    const int p = i1[r];
    g[(int64_t)p * SWC + j] = scale * d * inv_norm[p];
  }
  red[tid] = local;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) colloss[j] = red[0];
}

// d_a[b, l, c] += sum_p dir[p][l] * g[p][col]   (lane = four positions)
__global__ __launch_bounds__(256) void sw_backproject_kernel(const float* __restrict__ g, int P,
                                                             uint64_t seed, int64_t L, int c_used,
                                                             int col0, int ncol,
                                                             float* __restrict__ d_a, int c_a) {
  __shared__ __attribute__((aligned(16))) float gs[SWPC][SWC];
  const int tid = threadIdx.x;
  const int64_t l4 = (int64_t)blockIdx.x * 256 + tid;
  float acc[4][SWC];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < SWC; ++j) acc[k][j] = 0.f;
  for (int p0 = 0; p0 < P; p0 += SWPC) {
    const int pc = P - p0 < SWPC ? P - p0 : SWPC;
    __syncthreads();
    for (int e = tid; e < pc * SWC; e += 256) (&gs[0][0])[e] = g[(int64_t)p0 * SWC + e];
    __syncthreads();
    if (4 * l4 < L) {
#pragma unroll 1
      for (int pp = 0; pp < pc; ++pp) {
        float z[4];
        sw_normals(seed, (uint32_t)(p0 + pp), (uint64_t)l4, z);
        const float4* row = reinterpret_cast<const float4*>(gs[pp]);
#pragma unroll
        for (int j4 = 0; j4 < SWC / 4; ++j4) {
          const float4 v = row[j4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc[k][4 * j4 + 0] += z[k] * v.x; acc[k][4 * j4 + 1] += z[k] * v.y;
            acc[k][4 * j4 + 2] += z[k] * v.z; acc[k][4 * j4 + 3] += z[k] * v.w;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t l = 4 * l4 + k;
    if (l >= L) continue;
#pragma unroll
    for (int j = 0; j < SWC; ++j) {
      if (j >= ncol) continue;
      const int col = col0 + j, bi = col / c_used, ci = col - bi * c_used;
      d_a[((int64_t)bi * L + l) * c_a + ci] += acc[k][j];
    }
  }
}

__global__ void sw_final_kernel(const float* __restrict__ colloss, int nv, float norm,
                                float* __restrict__ loss_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < nv; ++j) s += colloss[j];
    *loss_out = s * norm;
  }
}

}  // namespace

extern "C" int s3_sw_directions(s3_ctx* ctx, uint64_t seed, int n_proj, int64_t n_pos, float* out) {
  if (!ctx || !out || n_proj < 1 || n_pos < 1) return S3_EINVAL;
  const int64_t total = (int64_t)n_proj * ((n_pos + 3) >> 2);
  int64_t nblk = (total + 255) / 256;
  if (nblk > 65536) nblk = 65536;
  hipLaunchKernelGGL(sw_directions_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, seed, n_proj,
                     n_pos, out);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_loss_sliced_wasserstein(s3_ctx* ctx, const float* a, int c_a, const float* b, int c_b,
                                          int n, int64_t n_pos, int c_used, int n_proj, uint64_t seed,
                                          float weight, float* loss_out, float* d_a) {
  if (!ctx || !a || !b || !loss_out || n < 1 || n_pos < 1 || c_used < 1 || c_used > c_a || c_used > c_b)
    return S3_EINVAL;
  if (n_proj < 1 || n_proj > 4096) S3_FAIL(ctx, S3_EINVAL, "sliced wasserstein: 1 <= n_projections <= 4096");
  const int P = n_proj;
  int P2 = 2;
  while (P2 < P) P2 <<= 1;
  const int nv = n * c_used;
  const int64_t ntiles = (n_pos + SWT - 1) / SWT;
  int64_t R = ntiles < 128 ? ntiles : 128;
  const int64_t tpr = (ntiles + R - 1) / R;
  R = (ntiles + tpr - 1) / tpr;
  const size_t f_partial = (size_t)R * P * SWROW, f_raw = (size_t)2 * SWC * P, f_inv = (size_t)P;
  const size_t f_g = (size_t)P * SWC, f_col = (size_t)((nv + SWC - 1) / SWC) * SWC;
  int rc = ensure_scratch(ctx, (f_partial + f_raw + f_inv + f_g + f_col + 16) * sizeof(float));
  if (rc) return rc;
  float* partial = (float*)ctx->scratch;
  float* raw = partial + f_partial;
  float* inv_norm = raw + f_raw;
  float* g = inv_norm + f_inv;
  float* colloss = g + f_g;
  const float scale = weight * 2.f / ((float)nv * (float)P);
  for (int col0 = 0; col0 < nv; col0 += SWC) {
    const int ncol = nv - col0 < SWC ? nv - col0 : SWC;
    hipLaunchKernelGGL(sw_project_kernel, dim3((P + SWB - 1) / SWB, (unsigned)R), dim3(SWB), 0, ctx->stream, a,
                       c_a, b, c_b, n_pos, c_used, col0, ncol, P, seed, ntiles, tpr, partial);
    hipLaunchKernelGGL(sw_reduce_kernel, dim3((P * SWROW + 255) / 256), dim3(256), 0, ctx->stream, partial,
                       (int)R, P, raw, inv_norm);
    hipLaunchKernelGGL(sw_sort_kernel, dim3(SWC), dim3(256), (size_t)3 * P2 * sizeof(float), ctx->stream, raw,
                       inv_norm, P, P2, ncol, scale, g, colloss + col0);
    if (d_a) {
      const int64_t g4 = (n_pos + 3) >> 2;
      hipLaunchKernelGGL(sw_backproject_kernel, dim3((unsigned)((g4 + 255) / 256)), dim3(256), 0, ctx->stream,
                         g, P, seed, n_pos, c_used, col0, ncol, d_a, c_a);
    }
  }
  hipLaunchKernelGGL(sw_final_kernel, dim3(1), dim3(64), 0, ctx->stream, colloss, nv,
                     1.f / ((float)nv * (float)P), loss_out);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
