// HBM-bound helper kernels of the Sup3rGan path: index-permutation ops
// (temporal nearest repeat, depth-to-space, pad, crop, roll, concat),
// activations, residual adds, bias gradients, losses, Adam, reductions.
// All are one-pass streaming kernels: coalesced 16-B accesses where the
// channel count allows, grid-stride loops capped at ~8 blocks per CU.
#include "common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n_threads, int num_cu) {
  int64_t b = (n_threads + kBlock - 1) / kBlock;
  int64_t cap = (int64_t)num_cu * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ gather
// out[n, o0, o1, o2, c] = in[map(...)]; V = channels per thread (1 or 4)
template <int V, typename T>
__global__ void gather_kernel(const T* __restrict__ in, T* __restrict__ out,
                              GatherGeom g) {
  const int cg_out = g.Co / V;
  const int64_t total = (int64_t)g.N * g.Do[0] * g.Do[1] * g.Do[2] * cg_out;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    int cg = (int)(r % cg_out); r /= cg_out;
    int o2 = (int)(r % g.Do[2]); r /= g.Do[2];
    int o1 = (int)(r % g.Do[1]); r /= g.Do[1];
    int o0 = (int)(r % g.Do[0]); r /= g.Do[0];
    int n = (int)r;
    int c = cg * V;
    int i0 = o0, i1 = o1, i2 = o2, ci = c;
    bool zero = false;
    int64_t out_c = c;
    switch (g.kind) {
      case S3_OP_REPEAT_T: i2 = o2 / g.rep; break;
      case S3_OP_ROLL_T: {
        int s = g.rep % g.Do[2]; if (s < 0) s += g.Do[2];   // tf.roll: any sign
        i2 = o2 - s; if (i2 < 0) i2 += g.Do[2];
      } break;
      case S3_OP_D2S: {
        int b = g.d2s;
        i0 = o0 / b; i1 = o1 / b;
        ci = ((o0 % b) * b + (o1 % b)) * g.Co + c;
      } break;
      case S3_OP_CROP: i0 = o0 + g.lo[0]; i1 = o1 + g.lo[1]; i2 = o2 + g.lo[2]; break;
      case S3_OP_DILATE:   // lo[] = stride: out[i s] = in[i], zeros in between
        zero = (o0 % g.lo[0]) || (o1 % g.lo[1]) || (o2 % g.lo[2]);
        i0 = o0 / g.lo[0]; i1 = o1 / g.lo[1]; i2 = o2 / g.lo[2];
        break;
      case S3_OP_PAD: {
        i0 = o0 - g.lo[0]; i1 = o1 - g.lo[1]; i2 = o2 - g.lo[2];
        if (g.pad_mode == S3_PAD_REFLECT) {
          i0 = s3_reflect(i0, g.Di[0]); i1 = s3_reflect(i1, g.Di[1]);
          i2 = s3_reflect(i2, g.Di[2]);
        } else {
          zero = i0 < 0 || i0 >= g.Di[0] || i1 < 0 || i1 >= g.Di[1] ||
                 i2 < 0 || i2 >= g.Di[2];
        }
      } break;
      case S3_OP_CONCAT: {
        // thread indexes the INPUT channel range; output channel is offset
        // (Do == Di, Co here is the number of channels copied)
        out_c = c + g.c_off;
      } break;
      default: break;
    }
    int64_t src = ((((int64_t)n * g.Di[0] + i0) * g.Di[1] + i1) * g.Di[2] + i2) *
                      g.Ci + ci;
    int co_total = (g.kind == S3_OP_CONCAT) ? g.rep : g.Co;  // rep = C of out
    int64_t dst = ((((int64_t)n * g.Do[0] + o0) * g.Do[1] + o1) * g.Do[2] + o2) *
                      co_total + out_c;
    if (V * sizeof(T) == 16) {
      uint4 v = zero ? make_uint4(0, 0, 0, 0)
                     : *reinterpret_cast<const uint4*>(in + src);
      *reinterpret_cast<uint4*>(out + dst) = v;
    } else {
      out[dst] = zero ? (T)0 : in[src];
    }
  }
}

// backward of the gather ops: one thread per din element (gathers its
// pre-images from dout; no atomics, deterministic)
__global__ void gather_bwd_kernel(const float* __restrict__ dout,
                                  float* __restrict__ din, GatherGeom g) {
  const int64_t total = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * g.Ci;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    int c = (int)(r % g.Ci); r /= g.Ci;
    int i2 = (int)(r % g.Di[2]); r /= g.Di[2];
    int i1 = (int)(r % g.Di[1]); r /= g.Di[1];
    int i0 = (int)(r % g.Di[0]); r /= g.Di[0];
    int n = (int)r;
    auto at = [&](int o0, int o1, int o2, int co, int ctot) -> float {
      return dout[((((int64_t)n * g.Do[0] + o0) * g.Do[1] + o1) * g.Do[2] + o2) *
                      ctot + co];
    };
    float acc = 0.f;
    switch (g.kind) {
      case S3_OP_REPEAT_T:
        for (int j = 0; j < g.rep; ++j) acc += at(i0, i1, i2 * g.rep + j, c, g.Co);
        break;
      case S3_OP_ROLL_T: {
        int s = g.rep % g.Do[2]; if (s < 0) s += g.Do[2];
        int o2 = i2 + s; if (o2 >= g.Do[2]) o2 -= g.Do[2];
        acc = at(i0, i1, o2, c, g.Co);
      } break;
      case S3_OP_D2S: {
        int b = g.d2s;
        int blk = c / g.Co, co = c % g.Co;
        acc = at(i0 * b + blk / b, i1 * b + blk % b, i2, co, g.Co);
      } break;
      case S3_OP_CROP: {
        int o0 = i0 - g.lo[0], o1 = i1 - g.lo[1], o2 = i2 - g.lo[2];
        if (o0 >= 0 && o0 < g.Do[0] && o1 >= 0 && o1 < g.Do[1] && o2 >= 0 &&
            o2 < g.Do[2])
          acc = at(o0, o1, o2, c, g.Co);
      } break;
      case S3_OP_PAD: {
        // pre-images of i under reflect: i+lo, lo-i (1<=i<=lo), and the
        // mirror about the far edge
        int cand[3][3], cnt[3];
        const int ii[3] = {i0, i1, i2};
        for (int d = 0; d < 3; ++d) {
          int nI = g.Di[d], lo = g.lo[d], nO = g.Do[d];
          cnt[d] = 0;
          cand[d][cnt[d]++] = ii[d] + lo;
          if (g.pad_mode == S3_PAD_REFLECT) {
            if (ii[d] >= 1 && ii[d] <= lo) cand[d][cnt[d]++] = lo - ii[d];
            int m = 2 * (nI - 1) - ii[d] + lo;  // mirrored padded index
            if (ii[d] <= nI - 2 && m < nO && m >= nI + lo) cand[d][cnt[d]++] = m;
          }
        }
        for (int a = 0; a < cnt[0]; ++a)
          for (int b = 0; b < cnt[1]; ++b)
            for (int e = 0; e < cnt[2]; ++e)
              acc += at(cand[0][a], cand[1][b], cand[2][e], c, g.Co);
      } break;
      case S3_OP_DILATE:
        acc = at(i0 * g.lo[0], i1 * g.lo[1], i2 * g.lo[2], c, g.Co);
        break;
      case S3_OP_CONCAT:
        acc = at(i0, i1, i2, c + g.c_off, g.rep);
        break;
      default: break;
    }
    din[idx] = acc;
  }
}

// fold of a reflect / zero padded frame (adjoint of S3_OP_PAD) on float4
// channel groups: the index math of a cell is shared by 4 channels
// MASK: 0 none, 1 fp32 y, 2 bf16 y — multiplies by the activation adjoint of the
// conv that produced the folded tensor (y = act(pre): 1 where y > 0, else slope);
// 3: mask_y is an fp32 tensor ADDED to the fold (an earlier gradient contribution)
// OUT16: the folded tensor is stored as bf16 ONLY (din is then an unsigned
// short buffer): dPre of a conv whose data / weight gradient kernels take bf16
// and whose bias gradient rides along in bsum — nothing reads it as fp32
// SIDE16: fp32 store to din AND a bf16 copy to side16 (a tensor that stays
// fp32 for the skip path but whose producer conv stages bf16)
// FR16: the frame `dout` is stored as bf16 (round 4: the persistent data
// gradient kernel writes its padded frame that way — half the round trip)
template <int MASK, bool OUT16 = false, bool SIDE16 = false, bool FR16 = false>
__global__ void gather_bwd_pad4_kernel(const float* __restrict__ dout,
                                       float* __restrict__ din, GatherGeom g,
                                       const void* __restrict__ mask_y, float slope,
                                       float* __restrict__ bsum,
                                       unsigned short* __restrict__ side16 = nullptr) {
  // bsum (nullable, needs c4n | 256): per-workgroup channel sums of the stored
  // values, partial[block][Ci] — the bias gradient of the conv that produced
  // the folded tensor, for bias_grad_stage2 (a lane keeps one channel group:
  // the grid stride is a multiple of c4n)
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  const int c4n = g.Ci >> 2;
  const int64_t total = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * c4n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    // (32-bit divisions whenever the element count allows: the 64-bit ones
    // cost more than the memory traffic of this kernel)
    int c4, i2, i1, i0, n;
    if (total <= 0x7fffffffLL) {
      unsigned r = (unsigned)idx, q;
      q = r / (unsigned)c4n; c4 = (int)(r - q * (unsigned)c4n); r = q;
      q = r / (unsigned)g.Di[2]; i2 = (int)(r - q * (unsigned)g.Di[2]); r = q;
      q = r / (unsigned)g.Di[1]; i1 = (int)(r - q * (unsigned)g.Di[1]); r = q;
      q = r / (unsigned)g.Di[0]; i0 = (int)(r - q * (unsigned)g.Di[0]); n = (int)q;
    } else {
      int64_t r = idx;
      c4 = (int)(r % c4n); r /= c4n;
      i2 = (int)(r % g.Di[2]); r /= g.Di[2];
      i1 = (int)(r % g.Di[1]); r /= g.Di[1];
      i0 = (int)(r % g.Di[0]); r /= g.Di[0];
      n = (int)r;
    }
    int cand[3][3], cnt[3];
    const int ii[3] = {i0, i1, i2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int nI = g.Di[d], lo = g.lo[d], nO = g.Do[d];
      cnt[d] = 0;
      cand[d][cnt[d]++] = ii[d] + lo;
      if (g.pad_mode == S3_PAD_REFLECT) {
        if (ii[d] >= 1 && ii[d] <= lo) cand[d][cnt[d]++] = lo - ii[d];
        const int m = 2 * (nI - 1) - ii[d] + lo;
        if (ii[d] <= nI - 2 && m < nO && m >= nI + lo) cand[d][cnt[d]++] = m;
      }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < cnt[0]; ++a)
      for (int b = 0; b < cnt[1]; ++b)
        for (int e = 0; e < cnt[2]; ++e) {
          const int64_t fo = ((((int64_t)n * g.Do[0] + cand[0][a]) * g.Do[1] + cand[1][b]) * g.Do[2] +
                              cand[2][e]) * g.Co + c4 * 4;
          float4 v;
          if constexpr (FR16) {
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dout) + fo);
            v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xFFFF0000u),
                            __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xFFFF0000u));
          } else {
            v = *reinterpret_cast<const float4*>(dout + fo);
          }
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    if (MASK == 1) {
      const float4 y = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + idx * 4);
      acc.x *= y.x > 0.f ? 1.f : slope; acc.y *= y.y > 0.f ? 1.f : slope;
      acc.z *= y.z > 0.f ? 1.f : slope; acc.w *= y.w > 0.f ? 1.f : slope;
    } else if (MASK == 2) {
      // bf16: the sign bit is bit 15 of each half word; zero is not > 0
      const uint2 y = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(mask_y) + idx * 4);
      auto pos = [](unsigned h) { return (h & 0x8000u) == 0 && (h & 0x7FFFu) != 0; };
      acc.x *= pos(y.x & 0xFFFFu) ? 1.f : slope; acc.y *= pos(y.x >> 16) ? 1.f : slope;
      acc.z *= pos(y.y & 0xFFFFu) ? 1.f : slope; acc.w *= pos(y.y >> 16) ? 1.f : slope;
    }
    if (MASK == 3) {
      const float4 y = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + idx * 4);
      acc.x += y.x; acc.y += y.y; acc.z += y.z; acc.w += y.w;
    }
    if constexpr (OUT16) {
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 lo2 = {acc.x, acc.y}, hi2 = {acc.z, acc.w};
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(din) + idx * 4) =
          make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf2)));
    } else {
      *reinterpret_cast<float4*>(din + idx * 4) = acc;
      if constexpr (SIDE16) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 lo2 = {acc.x, acc.y}, hi2 = {acc.z, acc.w};
        *reinterpret_cast<uint2*>(side16 + idx * 4) =
            make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf2)),
                       __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf2)));
      }
    }
    bs.x += acc.x; bs.y += acc.y; bs.z += acc.z; bs.w += acc.w;
  }
  if (bsum) {
    __shared__ float4 bred[256];
    bred[threadIdx.x] = bs;
    __syncthreads();
    if ((int)threadIdx.x < c4n) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = threadIdx.x; q < 256; q += c4n) {
        const float4 v = bred[q];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      reinterpret_cast<float4*>(bsum)[(int64_t)blockIdx.x * c4n + threadIdx.x] = t;
    }
  }
}

// The bf16-frame fold on EIGHT channels per lane (C % 8 == 0): with the 4-wide
// walk above a lane moved 8 B per frame cell and the index arithmetic (four
// divisions, up to eight candidate cells) set the pace — the masked fold ran
// 246 MB in 69 us where the fp32 frame's 342 MB had taken 70.  16-B loads of
// the frame / the bf16 mask, 16-B bf16 stores, half the index math per byte.
// Same MASK / OUT16 / SIDE16 meaning and the same bsum layout
// (partial[block][C]); launched with the 4-wide walk's grid so that the
// consumers of bsum see the block count they expect.
template <int MASK, bool OUT16, bool SIDE16>
__global__ void fold16x8_kernel(const unsigned short* __restrict__ frame, float* __restrict__ din,
                                GatherGeom g, const void* __restrict__ mask_y, float slope,
                                float* __restrict__ bsum, unsigned short* __restrict__ side16) {
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int c8n = g.Ci >> 3;
  const int64_t total = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    unsigned r = (unsigned)idx, q;           // (launcher: total < 2^31)
    q = r / (unsigned)c8n; const int c8 = (int)(r - q * (unsigned)c8n); r = q;
    q = r / (unsigned)g.Di[2]; const int i2 = (int)(r - q * (unsigned)g.Di[2]); r = q;
    q = r / (unsigned)g.Di[1]; const int i1 = (int)(r - q * (unsigned)g.Di[1]); r = q;
    q = r / (unsigned)g.Di[0]; const int i0 = (int)(r - q * (unsigned)g.Di[0]);
    const int n = (int)q;
    int cand[3][3], cnt[3];
    const int ii[3] = {i0, i1, i2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int nI = g.Di[d], lo = g.lo[d], nO = g.Do[d];
      cnt[d] = 0;
      cand[d][cnt[d]++] = ii[d] + lo;
      if (g.pad_mode == S3_PAD_REFLECT) {
        if (ii[d] >= 1 && ii[d] <= lo) cand[d][cnt[d]++] = lo - ii[d];
        const int m = 2 * (nI - 1) - ii[d] + lo;
        if (ii[d] <= nI - 2 && m < nO && m >= nI + lo) cand[d][cnt[d]++] = m;
      }
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < cnt[0]; ++a)
      for (int b = 0; b < cnt[1]; ++b)
        for (int e = 0; e < cnt[2]; ++e) {
          const int64_t fo = ((((int64_t)n * g.Do[0] + cand[0][a]) * g.Do[1] + cand[1][b]) * g.Do[2] +
                              cand[2][e]) * g.Co + c8 * 8;
          const uint4 h = *reinterpret_cast<const uint4*>(frame + fo);
          acc[0] += __uint_as_float(h.x << 16); acc[1] += __uint_as_float(h.x & 0xFFFF0000u);
          acc[2] += __uint_as_float(h.y << 16); acc[3] += __uint_as_float(h.y & 0xFFFF0000u);
          acc[4] += __uint_as_float(h.z << 16); acc[5] += __uint_as_float(h.z & 0xFFFF0000u);
          acc[6] += __uint_as_float(h.w << 16); acc[7] += __uint_as_float(h.w & 0xFFFF0000u);
        }
    if (MASK == 1) {
      const float4* yp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + idx * 8);
      const float4 y0 = yp[0], y1 = yp[1];
      const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] *= yv[k] > 0.f ? 1.f : slope;
    } else if (MASK == 2) {
      const uint4 y = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(mask_y) + idx * 8);
      const unsigned yw[4] = {y.x, y.y, y.z, y.w};
      auto pos = [](unsigned h) { return (h & 0x8000u) == 0 && (h & 0x7FFFu) != 0; };
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] *= pos(yw[k] & 0xFFFFu) ? 1.f : slope;
        acc[2 * k + 1] *= pos(yw[k] >> 16) ? 1.f : slope;
      }
    } else if (MASK == 3) {
      const float4* yp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + idx * 8);
      const float4 y0 = yp[0], y1 = yp[1];
      acc[0] += y0.x; acc[1] += y0.y; acc[2] += y0.z; acc[3] += y0.w;
      acc[4] += y1.x; acc[5] += y1.y; acc[6] += y1.z; acc[7] += y1.w;
    }
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto pk = [](float lo, float hi) {
      const f2 v = {lo, hi};
      return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    };
    const uint4 o16 = make_uint4(pk(acc[0], acc[1]), pk(acc[2], acc[3]), pk(acc[4], acc[5]), pk(acc[6], acc[7]));
    if constexpr (OUT16) {
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(din) + idx * 8) = o16;
    } else {
      float4* dp = reinterpret_cast<float4*>(din + idx * 8);
      dp[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      dp[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
      if constexpr (SIDE16) *reinterpret_cast<uint4*>(side16 + idx * 8) = o16;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) bs[k] += acc[k];
  }
  if (bsum) {
    __shared__ float bred[256 * 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bred[threadIdx.x * 8 + k] = bs[k];
    __syncthreads();
    // (a lane keeps one channel group: the grid stride is a multiple of c8n)
    for (int item = threadIdx.x; item < c8n * 8; item += 256) {
      const int grp = item >> 3, k = item & 7;
      float t = 0.f;
      for (int q = grp; q < 256; q += c8n) t += bred[q * 8 + k];
      bsum[(int64_t)blockIdx.x * g.Ci + grp * 8 + k] = t;
    }
  }
}

// ------------------------------------------------------------- elementwise
__device__ inline float act_f(float v, int act, float alpha) {
  // one select for every kind (slope 1 = identity, 0 = ReLU, alpha = Leaky):
  // testing the kind per element compiles to two scalar branches per value
  const float s = act == S3_ACT_LEAKY ? alpha : (act == S3_ACT_RELU ? 0.f : 1.f);
  return v > 0.f ? v : s * v;
}
__device__ inline float act_d(float y, int act, float alpha) {
  if (act == S3_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == S3_ACT_LEAKY) return y > 0.f ? 1.f : alpha;
  return 1.f;
}

__global__ void act_kernel(const float* __restrict__ x, float* __restrict__ y,
                           int64_t n, int act, float alpha) {
  int64_t n4 = n / 4;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = act_f(v.x, act, alpha); v.y = act_f(v.y, act, alpha);
    v.z = act_f(v.z, act, alpha); v.w = act_f(v.w, act, alpha);
    reinterpret_cast<float4*>(y)[i] = v;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < n; i += stride)
    y[i] = act_f(x[i], act, alpha);
}

__global__ void act_bwd_kernel(const float* __restrict__ y,
                               const float* __restrict__ dy,
                               float* __restrict__ dx, int64_t n, int act,
                               float alpha) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    dx[i] = dy[i] * act_d(y[i], act, alpha);
}

// dpre[n,o0,o1,o2,c] = dy[perm] * act'(y[perm]) with the d2s store permutation
template <bool Y16>
__global__ void conv_epilogue_bwd_kernel(const float* __restrict__ y,
                                         const float* __restrict__ dy,
                                         float* __restrict__ dpre, ConvGeom g) {
  const int64_t total = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] * g.Cout;
  const int b = g.d2s;
  const int co = g.Cout / (b * b);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t src = idx;
    if (b > 1) {
      int64_t r = idx;
      int c = (int)(r % g.Cout); r /= g.Cout;
      int o2 = (int)(r % g.O[2]); r /= g.O[2];
      int o1 = (int)(r % g.O[1]); r /= g.O[1];
      int o0 = (int)(r % g.O[0]); r /= g.O[0];
      int n = (int)r;
      int blk = c / co, cc = c % co;
      src = ((((int64_t)n * g.O[0] * b + o0 * b + blk / b) * (g.O[1] * b) +
              o1 * b + blk % b) * g.O[2] + o2) * co + cc;
    }
    // (Y16: the saved activation is a bf16 tensor; only its sign matters
    // for ReLU / LeakyReLU)
    const float yv = Y16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(y)[src] << 16)
                         : y[src];
    dpre[idx] = dy[src] * act_d(yv, g.act, g.alpha);
  }
}

// depth-to-space variant, four channels per lane, walking dPre (the
// DESTINATION) in order: the generic kernel above spends 89 % of the SIMD
// cycles on five 64-bit divisions per ELEMENT (PMC, 64 -> 200 + d2s 5 conv of
// C2: 0.59 ms).  Here a lane owns 4 consecutive channels of one (cell, block):
// 32-bit index math when the tensor allows, one division chain per four
// elements, fully coalesced 16-B stores (whole 128-B lines per wave) and 16-B /
// 8-B gathers of dy / y from the hi-res layout (32-B sectors, nothing wasted).
// (Walking the hi-res layout instead scatters 16-B pieces of every dPre line
// over 25 far-apart moments: 0.52 ms.)
template <bool Y16, bool D16 = false>
__global__ void conv_epilogue_bwd_d2s4_kernel(const void* __restrict__ y, const float4* __restrict__ dy,
                                              float4* __restrict__ dpre, ConvGeom g, float slope,
                                              unsigned short* __restrict__ d16 = nullptr,
                                              float* __restrict__ bsum = nullptr) {
  // dpre (nullable with D16): every reader of this dPre takes the bf16 copy.
  // bsum (nullable): per-workgroup channel sums of dpre = the conv's bias
  // gradient for bias_grad_stage2; the launch then uses a block size that is
  // a multiple of C_out / 4, so that a lane keeps one channel group
  // (conv_epilogue_bwd_d2s4_block) — with the fp32 store gone too the 64 ->
  // 200 conv of C2 (118 M elements) saves 0.47 GB of stores and the 0.47 GB
  // bias_grad_stage1 read them back with.
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned b = (unsigned)g.d2s, co4 = ((unsigned)g.Cout / (b * b)) >> 2, C4 = (unsigned)g.Cout >> 2;
  const unsigned O0 = (unsigned)g.O[0], O1 = (unsigned)g.O[1], O2 = (unsigned)g.O[2];
  const int64_t total = (int64_t)g.N * O0 * O1 * O2 * C4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    unsigned c4, o2, o1, o0, n;
    if (total <= 0xffffffffLL) {
      unsigned r = (unsigned)idx, q;
      q = r / C4; c4 = r - q * C4; r = q;
      q = r / O2; o2 = r - q * O2; r = q;
      q = r / O1; o1 = r - q * O1; r = q;
      q = r / O0; o0 = r - q * O0; n = q;
    } else {
      int64_t r = idx;
      c4 = (unsigned)(r % C4); r /= C4;
      o2 = (unsigned)(r % O2); r /= O2;
      o1 = (unsigned)(r % O1); r /= O1;
      o0 = (unsigned)(r % O0); r /= O0;
      n = (unsigned)r;
    }
    const unsigned blk = c4 / co4, cc4 = c4 - blk * co4, p0 = blk / b, p1 = blk - p0 * b;
    // float4 index of the hi-res cell (n, o0 b + p0, o1 b + p1, o2), channels 4 cc4 ..
    const int64_t src = ((((int64_t)n * O0 * b + o0 * b + p0) * (O1 * b) + o1 * b + p1) * O2 + o2) * co4 + cc4;
    float4 d = dy[src];
    if (Y16) {
      const uint2 h = reinterpret_cast<const uint2*>(y)[src];
      auto pos = [](unsigned v) { return (v & 0x8000u) == 0 && (v & 0x7FFFu) != 0; };
      d.x *= pos(h.x & 0xFFFFu) ? 1.f : slope; d.y *= pos(h.x >> 16) ? 1.f : slope;
      d.z *= pos(h.y & 0xFFFFu) ? 1.f : slope; d.w *= pos(h.y >> 16) ? 1.f : slope;
    } else {
      const float4 v = reinterpret_cast<const float4*>(y)[src];
      d.x *= v.x > 0.f ? 1.f : slope; d.y *= v.y > 0.f ? 1.f : slope;
      d.z *= v.z > 0.f ? 1.f : slope; d.w *= v.w > 0.f ? 1.f : slope;
    }
    if (dpre) dpre[idx] = d;
    bs.x += d.x; bs.y += d.y; bs.z += d.z; bs.w += d.w;
    if constexpr (D16) {   // bf16 copy for the MFMA gradient kernels
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 lo2 = {d.x, d.y}, hi2 = {d.z, d.w};
      reinterpret_cast<uint2*>(d16)[idx] =
          make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf2)));
    }
  }
  if (bsum) {
    __shared__ float4 bred[256];
    bred[threadIdx.x] = bs;
    __syncthreads();
    if (threadIdx.x < C4) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (unsigned q = threadIdx.x; q < blockDim.x; q += C4) {
        const float4 v = bred[q];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      reinterpret_cast<float4*>(bsum)[(int64_t)blockIdx.x * C4 + threadIdx.x] = t;
    }
  }
}

// the same without a store permutation, four channels per lane
template <bool Y16>
__global__ void conv_epilogue_bwd4_kernel(const void* __restrict__ y, const float4* __restrict__ dy,
                                          float4* __restrict__ dpre, int64_t n4, float slope,
                                          unsigned short* __restrict__ d16, float* __restrict__ bsum,
                                          int c4n) {
  // bsum (nullable, needs c4n | 256): per-workgroup channel sums of dpre — the
  // conv's bias gradient — for bias_grad_stage2 (a lane keeps one channel
  // group: the grid stride is a multiple of c4n), as in gather_bwd_pad4_kernel
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 d = dy[i];
    if (Y16) {
      const uint2 h = reinterpret_cast<const uint2*>(y)[i];
      auto pos = [](unsigned v) { return (v & 0x8000u) == 0 && (v & 0x7FFFu) != 0; };
      d.x *= pos(h.x & 0xFFFFu) ? 1.f : slope; d.y *= pos(h.x >> 16) ? 1.f : slope;
      d.z *= pos(h.y & 0xFFFFu) ? 1.f : slope; d.w *= pos(h.y >> 16) ? 1.f : slope;
    } else {
      const float4 v = reinterpret_cast<const float4*>(y)[i];
      d.x *= v.x > 0.f ? 1.f : slope; d.y *= v.y > 0.f ? 1.f : slope;
      d.z *= v.z > 0.f ? 1.f : slope; d.w *= v.w > 0.f ? 1.f : slope;
    }
    // (dpre == nullptr: every reader of this dPre takes the bf16 copy and the
    // bias gradient rides along in bsum — the 151 MB fp32 store is skipped)
    if (dpre) dpre[i] = d;
    bs.x += d.x; bs.y += d.y; bs.z += d.z; bs.w += d.w;
    if (d16) {   // bf16 copy for the MFMA gradient kernels (see gather_bwd_pad4_kernel)
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 lo2 = {d.x, d.y}, hi2 = {d.z, d.w};
      reinterpret_cast<uint2*>(d16)[i] =
          make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf2)));
    }
  }
  if (bsum) {
    __shared__ float4 bred[256];
    bred[threadIdx.x] = bs;
    __syncthreads();
    if ((int)threadIdx.x < c4n) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = threadIdx.x; q < 256; q += c4n) {
        const float4 v = bred[q];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      reinterpret_cast<float4*>(bsum)[(int64_t)blockIdx.x * c4n + threadIdx.x] = t;
    }
  }
}

__global__ void add_kernel(const float* __restrict__ a,
                           const float* __restrict__ b, float* __restrict__ y,
                           int64_t n, int c, int bcast_c) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    y[i] = a[i] + (bcast_c ? b[i / c] : b[i]);
}

// bf16 cells: y = bf16(a + b), eight channels per lane (inference plans: a
// SkipConnection add that no conv epilogue absorbed — the second of two adds
// behind one conv in sup3rcc/gen_*_5x_1x_* at hi-res — stays in bf16 so that
// the convs on either side keep their bf16 kernels)
__global__ void add16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y,
                             int64_t n8) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  auto add2 = [](unsigned u, unsigned v) {
    const f2 s = {__uint_as_float(u << 16) + __uint_as_float(v << 16),
                  __uint_as_float(u & 0xFFFF0000u) + __uint_as_float(v & 0xFFFF0000u)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf2));
  };
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const uint4 p = a[i], q = b[i];
    y[i] = make_uint4(add2(p.x, q.x), add2(p.y, q.y), add2(p.z, q.z), add2(p.w, q.w));
  }
}

__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y,
                            int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    y[i] += x[i];
}
__global__ void axpy4_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = x[i];
    float4 b = y[i];
    b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
    y[i] = b;
  }
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    p[i] = v;
}

// ------------------------------------------------------------- reductions
__device__ inline float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ inline float block_sum(float v, float* sm) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
  }
  return t;  // valid on thread 0
}

// bias gradient: db[c] = sum over positions of dy[pos][c].  Two stages:
// stage 1: grid of blocks, each reduces a slab of positions to partial[blk][c]
// stage 2: one block sums the partials in fixed order (deterministic)
__global__ void bias_grad_stage1(const float* __restrict__ dy, int64_t n_pos,
                                 int c, float* __restrict__ partial) {
  // thread t handles channel (t % c_pad) for positions strided by rows
  extern __shared__ float sm[];
  const int rows = blockDim.x / c;           // positions handled per sweep
  const int my_c = threadIdx.x % c, my_r = threadIdx.x / c;
  float acc = 0.f;
  if (my_r < rows) {
    // four loads in flight per lane (fixed order: still deterministic)
    const int64_t step = (int64_t)gridDim.x * rows;
    int64_t p = (int64_t)blockIdx.x * rows + my_r;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; p + 3 * step < n_pos; p += 4 * step) {
      a0 += dy[p * c + my_c];
      a1 += dy[(p + step) * c + my_c];
      a2 += dy[(p + 2 * step) * c + my_c];
      a3 += dy[(p + 3 * step) * c + my_c];
    }
    for (; p < n_pos; p += step) a0 += dy[p * c + my_c];
    acc = (a0 + a1) + (a2 + a3);
  }
  sm[threadIdx.x] = (my_r < rows) ? acc : 0.f;
  __syncthreads();
  if (threadIdx.x < c) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += sm[r * c + threadIdx.x];
    partial[(int64_t)blockIdx.x * c + threadIdx.x] = t;
  }
}

// four channels per lane (c % 4 == 0): a row of c floats is c / 4 lanes wide
__global__ void bias_grad_stage1_v4(const float4* __restrict__ dy, int64_t n_pos, int c4,
                                    float* __restrict__ partial) {
  extern __shared__ float4 sm4[];
  const int rows = blockDim.x / c4;
  const int my_c = threadIdx.x % c4, my_r = threadIdx.x / c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (my_r < rows) {
    const int64_t step = (int64_t)gridDim.x * rows;
    int64_t p = (int64_t)blockIdx.x * rows + my_r;
    float4 a0 = acc, a1 = acc;
    auto add4 = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    for (; p + step < n_pos; p += 2 * step) {
      const float4 v0 = dy[p * c4 + my_c];
      const float4 v1 = dy[(p + step) * c4 + my_c];
      add4(a0, v0); add4(a1, v1);
    }
    if (p < n_pos) add4(a0, dy[p * c4 + my_c]);
    acc = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  }
  sm4[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < c4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rows; ++r) {
      const float4 v = sm4[r * c4 + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(partial)[(int64_t)blockIdx.x * c4 + threadIdx.x] = t;
  }
}

// one block per channel: 256 lanes stride over the slabs, fixed-shape tree
// (deterministic)
__global__ void bias_grad_stage2(const float* __restrict__ partial, int nblk,
                                 int c, float* __restrict__ db, int accumulate) {
  __shared__ float sm[256];
  s3_bias_stage2_body(partial, nblk, c, blockIdx.x, db, accumulate, sm);
}

// wide-channel variant (dense layers: few rows, thousands of channels): one
// thread per channel walks the rows, lanes coalesce along channels
__global__ void bias_grad_cols(const float* __restrict__ dy, int64_t n_pos,
                               int c, float* __restrict__ db, int accumulate) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float t = 0.f;
  for (int64_t p = 0; p < n_pos; ++p) t += dy[p * c + ch];
  db[ch] = accumulate ? db[ch] + t : t;
}

// ... the same for MANY rows and > 256 channels (the 64 -> 512 / 576 / 768 /
// 1600 expansion convs of the shipped generators: one serial walk per channel
// took 4 ms at 46 000 positions): blockIdx.y owns a slice of the rows and
// writes one partial row; bias_grad_stage2 sums the slices
__global__ void bias_grad_cols_split(const float* __restrict__ dy, int64_t n_pos, int c,
                                     float* __restrict__ partial) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const int64_t per = (n_pos + gridDim.y - 1) / gridDim.y;
  const int64_t p0 = (int64_t)blockIdx.y * per;
  const int64_t p1 = p0 + per < n_pos ? p0 + per : n_pos;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int64_t p = p0;
  for (; p + 4 <= p1; p += 4) {
    t0 += dy[p * c + ch]; t1 += dy[(p + 1) * c + ch];
    t2 += dy[(p + 2) * c + ch]; t3 += dy[(p + 3) * c + ch];
  }
  for (; p < p1; ++p) t0 += dy[p * c + ch];
  partial[(int64_t)blockIdx.y * c + ch] = (t0 + t1) + (t2 + t3);
}

__global__ void mean_abs_stage1(const float* __restrict__ p, int64_t n,
                                float* __restrict__ partial) {
  __shared__ float sm[8];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    acc += fabsf(p[i]);
  float t = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void sum_stage2(const float* __restrict__ partial, int nblk,
                           float scale, float* __restrict__ out,
                           int accumulate) {
  __shared__ float sm[8];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[i];
  float t = block_sum(acc, sm);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + t * scale : t * scale;
}

// ------------------------------------------------------------------ losses
// content loss over the first c_used channels; grad wrt a (c_a channels)
__global__ void loss_content_kernel(int kind, const float* __restrict__ a,
                                    int c_a, const float* __restrict__ b,
                                    int c_b, const float* __restrict__ mask,
                                    int c_m, int c_used, int64_t n_pos,
                                    float gscale, float* __restrict__ partial,
                                    float* __restrict__ d_a, int accumulate) {
  __shared__ float sm[8];
  const int64_t total = n_pos * c_used;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i / c_used;
    int c = (int)(i % c_used);
    const float mk = mask ? mask[p * c_m + c] : 1.f;
    float d = (a[p * c_a + c] - b[p * c_b + c]) * mk;
    float g;
    if (kind == S3_LOSS_MAE) {
      acc += fabsf(d);
      g = (d > 0.f) ? 1.f : (d < 0.f ? -1.f : 0.f);
    } else if (kind == S3_LOSS_EXP) {
      // ExpLoss: mean(1 - exp(-(x1 - x2)^2)) (loss_metrics.py:98-118)
      const float e = __expf(-d * d);
      acc += 1.f - e;
      g = 2.f * d * e;
    } else {
      acc += d * d;
      g = 2.f * d;
    }
    if (d_a) {
      float v = g * gscale * mk;
      d_a[p * c_a + c] = accumulate ? d_a[p * c_a + c] + v : v;
    }
  }
  float t = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// the dense case (every channel used, no mask, n % 4 == 0): 16-B loads and
// stores, no per-element division (the C2 hi-res batch — 29.5 M elements —
// took 146 us on the walk above: 2 x what its 354 MB cost at 5 TB/s)
__global__ void loss_content4_kernel(int kind, const float4* __restrict__ a, const float4* __restrict__ b,
                                     int64_t n4, float gscale, float* __restrict__ partial,
                                     float4* __restrict__ d_a, int accumulate) {
  __shared__ float sm[8];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 va = a[i], vb = b[i];
    const float d[4] = {va.x - vb.x, va.y - vb.y, va.z - vb.z, va.w - vb.w};
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (kind == S3_LOSS_MAE) {
        acc += fabsf(d[q]);
        g[q] = (d[q] > 0.f) ? 1.f : (d[q] < 0.f ? -1.f : 0.f);
      } else if (kind == S3_LOSS_EXP) {
        const float e = __expf(-d[q] * d[q]);
        acc += 1.f - e;
        g[q] = 2.f * d[q] * e;
      } else {
        acc += d[q] * d[q];
        g[q] = 2.f * d[q];
      }
      g[q] *= gscale;
    }
    if (d_a) {
      float4 v = make_float4(g[0], g[1], g[2], g[3]);
      if (accumulate) {
        const float4 o = d_a[i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      d_a[i] = v;
    }
  }
  float t = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// relativistic BCE (single block; n is the batch size, small)
__global__ void rel_bce_kernel(const float* __restrict__ dt,
                               const float* __restrict__ dg, int n, float scale,
                               float* __restrict__ loss_out,
                               float* __restrict__ d_true,
                               float* __restrict__ d_gen) {
  __shared__ float sm[8];
  __shared__ float s_mt, s_mg, s_gt, s_gf;
  float at = 0.f, ag = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { at += dt[i]; ag += dg[i]; }
  float t = block_sum(at, sm);
  if (threadIdx.x == 0) s_mt = t / n;
  t = block_sum(ag, sm);
  if (threadIdx.x == 0) s_mg = t / n;
  __syncthreads();
  const float mt = s_mt, mg = s_mg;
  float loss = 0.f, sgt = 0.f, sgf = 0.f;
  const float inv = 1.f / (2.f * n);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float xt = dt[i] - mg;  // label 1
    float xf = dg[i] - mt;  // label 0
    float et = expf(-fabsf(xt)), ef = expf(-fabsf(xf));
    loss += fmaxf(xt, 0.f) - xt + log1pf(et);
    loss += fmaxf(xf, 0.f) + log1pf(ef);
    float st = xt >= 0.f ? 1.f / (1.f + et) : et / (1.f + et);
    float sf = xf >= 0.f ? 1.f / (1.f + ef) : ef / (1.f + ef);
    sgt += (st - 1.f) * inv;
    sgf += sf * inv;
  }
  t = block_sum(loss, sm);
  if (threadIdx.x == 0) loss_out[0] = t * inv;
  t = block_sum(sgt, sm);
  if (threadIdx.x == 0) s_gt = t;
  t = block_sum(sgf, sm);
  if (threadIdx.x == 0) s_gf = t;
  __syncthreads();
  if (d_true || d_gen) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      float xt = dt[i] - mg, xf = dg[i] - mt;
      float et = expf(-fabsf(xt)), ef = expf(-fabsf(xf));
      float st = xt >= 0.f ? 1.f / (1.f + et) : et / (1.f + et);
      float sf = xf >= 0.f ? 1.f / (1.f + ef) : ef / (1.f + ef);
      float gt = (st - 1.f) * inv, gf = sf * inv;
      if (d_true) d_true[i] = scale * (gt - s_gf / n);
      if (d_gen) d_gen[i] = scale * (gf - s_gt / n);
    }
  }
}

// -------------------------------------------------------------------- Adam
__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float alpha, float omb1, float omb2,
                            float eps, const float* __restrict__ hd) {
  // hd: the step's scalars staged on the device (s3_optimizer_stage) — the
  // launch is then the same every step and can live in a captured graph
  if (hd) { alpha = hd[0]; omb1 = hd[1]; omb2 = hd[2]; eps = hd[3]; }
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += stride) {
    float4 gw = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float4 ww = reinterpret_cast<float4*>(w)[i];
#define S3_ADAM1(q)                                   \
    mm.q += (gw.q - mm.q) * omb1;                     \
    vv.q += (gw.q * gw.q - vv.q) * omb2;              \
    ww.q -= (mm.q * alpha) / (sqrtf(vv.q) + eps);
    S3_ADAM1(x) S3_ADAM1(y) S3_ADAM1(z) S3_ADAM1(w)
#undef S3_ADAM1
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(w)[i] = ww;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < n; i += stride) {
    float gi = g[i];
    float mi = m[i] + (gi - m[i]) * omb1;
    float vi = v[i] + (gi * gi - v[i]) * omb2;
    m[i] = mi; v[i] = vi;
    w[i] -= (mi * alpha) / (sqrtf(vi) + eps);
  }
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int c_src,
                                     int c0_src, float* __restrict__ dst,
                                     int c_dst, int c0_dst, int nc,
                                     int64_t n_pos, int accumulate) {
  const int64_t total = n_pos * nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i / nc;
    int c = (int)(i % nc);
    float v = src[p * c_src + c0_src + c];
    float* d = dst + p * c_dst + c0_dst + c;
    *d = accumulate ? *d + v : v;
  }
}

struct Affine8 { float scale[16]; float shift[16]; };
__global__ void affine_channels_kernel(const float* __restrict__ src,
                                       float* __restrict__ dst, int c,
                                       int64_t n, Affine8 a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    // two roundings, as numpy's (x * scale) + shift: the empty asm keeps the
    // compiler from contracting the pair into one fma
    float t = src[i] * a.scale[ch];
    asm volatile("" : "+v"(t));
    dst[i] = t + a.shift[ch];
  }
}

// per-(chunk, slab) min / max / NaN count of every channel: the device half of
// ForwardPass._output_check (forward_pass.py:384-425)
constexpr int kStatSlabs = 64;
__global__ void chunk_stats_kernel(const float* __restrict__ x, int64_t pos_per_chunk,
                                   int c, float* __restrict__ partial) {
  const int chunk = blockIdx.y, slab = blockIdx.x;
  const float* xc = x + (int64_t)chunk * pos_per_chunk * c;
  __shared__ float smin[256], smax[256], snan[256];
  for (int ch = 0; ch < c; ++ch) {
    float mn = INFINITY, mx = -INFINITY, nn = 0.f;
    for (int64_t p = (int64_t)slab * blockDim.x + threadIdx.x; p < pos_per_chunk;
         p += (int64_t)kStatSlabs * blockDim.x) {
      const float v = xc[p * c + ch];
      if (v != v) nn += 1.f;
      else { mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx; snan[threadIdx.x] = nn;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
        smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        snan[threadIdx.x] += snan[threadIdx.x + s];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      float* o = partial + (((int64_t)chunk * kStatSlabs + slab) * c + ch) * 3;
      o[0] = smin[0]; o[1] = smax[0]; o[2] = snan[0];
    }
    __syncthreads();
  }
}

// The same statistics at memory speed for dense, 16-byte aligned chunks with
// 1024 % c == 0 (c = 1, 2, 4, 8, 16: every real output feature count): the
// chunk is walked as float4 with four loads in flight per lane; component k of
// float4 number q = slab * 256 + tid + j * (64 * 256) is channel (4 tid + k) % c
// for every j, so the running triples sit in fixed registers.  (The plain
// kernel above reads the chunk once PER CHANNEL with a stride of c floats:
// 423 us for the 368 MB of a C3 batch of 16; this one 70 - 90 us.)
__global__ __launch_bounds__(256) void chunk_stats4_kernel(const float* __restrict__ x, int64_t n4_per_chunk,
                                                           int c, float* __restrict__ partial) {
  const int chunk = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
  const float4* xc = reinterpret_cast<const float4*>(x) + (int64_t)chunk * n4_per_chunk;
  float mn[4], mx[4], nn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; nn[k] = 0.f; }
  auto fold = [&](const float4& v4) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (v[k] != v[k]) nn[k] += 1.f;
      else { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
    }
  };
  const int64_t step = (int64_t)kStatSlabs * 256;
  int64_t q = (int64_t)slab * 256 + tid;
  for (; q + 3 * step < n4_per_chunk; q += 4 * step) {
    const float4 a = xc[q], b = xc[q + step], d = xc[q + 2 * step], e = xc[q + 3 * step];
    fold(a); fold(b); fold(d); fold(e);
  }
  for (; q < n4_per_chunk; q += step) fold(xc[q]);
  __shared__ float smin[256], smax[256], snan[256];
  for (int ch = 0; ch < c; ++ch) {
    float m0 = INFINITY, m1 = -INFINITY, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((4 * tid + k) % c == ch) { m0 = fminf(m0, mn[k]); m1 = fmaxf(m1, mx[k]); m2 += nn[k]; }
    smin[tid] = m0; smax[tid] = m1; snan[tid] = m2;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
      if (tid < s_) {
        smin[tid] = fminf(smin[tid], smin[tid + s_]);
        smax[tid] = fmaxf(smax[tid], smax[tid + s_]);
        snan[tid] += snan[tid + s_];
      }
      __syncthreads();
    }
    if (tid == 0) {
      float* o = partial + (((int64_t)chunk * kStatSlabs + slab) * c + ch) * 3;
      o[0] = smin[0]; o[1] = smax[0]; o[2] = snan[0];
    }
    __syncthreads();
  }
}

// ---- the chunk executor's output epilogue in ONE pass over the hi-res batch:
// un-normalisation (s3_affine_channels' two roundings), halo crop
// (chunk.hr_crop_slice) and the output check's statistics (chunk_stats_kernel's
// partial[chunk][64][c][3]; min / max / NaN count do not depend on the order
// they are folded in).  Separately these were an in-place pass over the
// un-cropped 483 MB, eight block copies and a re-read of the 368 MB result per
// batch of eight C3 chunks (0.86 ms); fused, the cropped window is read once
// and written once.  Workgroup (slab, chunk) walks the cropped rows slab,
// slab + 64, ...; a row is c3 * c contiguous floats moved as float4.  With
// 1024 % c == 0 (c = 1, 2, 4, 8, ...) the channel of component k of thread t is
// (4 t + k) % c for every row and every stride of 256 float4, so the running
// min / max / NaN count live in four fixed register triples per thread.
struct ChunkEpi {
  int64_t y_chunk, y_s0, y_s1;     // element strides of the un-cropped batch
  int64_t y_org;                   // element offset of the crop origin in a chunk
  int c0, c1;                      // cropped rows: c0 x c1
  int row4;                        // float4 per cropped row (c2 * c / 4)
  int c;
  int affine;
  float scale[16], shift[16];
};
__global__ __launch_bounds__(256) void chunk_epilogue_kernel(const float* __restrict__ y,
                                                             float* __restrict__ yc,
                                                             float* __restrict__ partial, ChunkEpi e) {
  const int chunk = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
  const float* yb = y + (int64_t)chunk * e.y_chunk + e.y_org;
  float* ob = yc + (int64_t)chunk * e.c0 * e.c1 * e.row4 * 4;
  float mn[4], mx[4], nn[4], sc[4], sh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ch = (4 * tid + k) % e.c;
    mn[k] = INFINITY; mx[k] = -INFINITY; nn[k] = 0.f;
    sc[k] = e.scale[ch]; sh[k] = e.shift[ch];
  }
  const int rows = e.c0 * e.c1;
  for (int r = slab; r < rows; r += kStatSlabs) {
    const int a = r / e.c1, b = r - a * e.c1;
    const float4* src = reinterpret_cast<const float4*>(yb + a * e.y_s0 + b * e.y_s1);
    float4* dst = reinterpret_cast<float4*>(ob + (int64_t)r * e.row4 * 4);
    for (int q = tid; q < e.row4; q += 256) {
      const float4 v4 = src[q];
      float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (e.affine) {
          float t = v[k] * sc[k];
          asm volatile("" : "+v"(t));          // (x * scale) + shift, two roundings
          v[k] = t + sh[k];
        }
        if (v[k] != v[k]) nn[k] += 1.f;
        else { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
      }
      dst[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __shared__ float smin[256], smax[256], snan[256];
  for (int ch = 0; ch < e.c; ++ch) {
    float m0 = INFINITY, m1 = -INFINITY, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((4 * tid + k) % e.c == ch) { m0 = fminf(m0, mn[k]); m1 = fmaxf(m1, mx[k]); m2 += nn[k]; }
    smin[tid] = m0; smax[tid] = m1; snan[tid] = m2;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
      if (tid < s_) {
        smin[tid] = fminf(smin[tid], smin[tid + s_]);
        smax[tid] = fmaxf(smax[tid], smax[tid + s_]);
        snan[tid] += snan[tid + s_];
      }
      __syncthreads();
    }
    if (tid == 0) {
      float* o = partial + (((int64_t)chunk * kStatSlabs + slab) * e.c + ch) * 3;
      o[0] = smin[0]; o[1] = smax[0]; o[2] = snan[0];
    }
    __syncthreads();
  }
}

}  // namespace

// ================================================================ launchers
int ensure_scratch(s3_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return S3_OK;
  if (ctx->capturing) S3_FAIL(ctx, S3_EINVAL, "scratch would grow inside a capture (run the step eagerly once first)");
  if (ctx->scratch) {
    if (ctx->graphs_made) {
      ctx->retired.push_back(ctx->scratch);
    } else {
      S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
      S3_HIP(ctx, hipFree(ctx->scratch));
    }
    ctx->scratch = nullptr; ctx->scratch_bytes = 0;
  }
  size_t want = bytes < (size_t)(1 << 20) ? (size_t)(1 << 20) : bytes;
  S3_HIP(ctx, hipMalloc((void**)&ctx->scratch, want));
  S3_HIP(ctx, hipMemsetAsync(ctx->scratch, ctx->opt.has[S3O_POISON_ALLOC] ? 0xFF : 0, want, ctx->stream));
  ctx->scratch_bytes = want;
  return S3_OK;
}

int launch_gather(s3_ctx* ctx, const GatherGeom& g, const void* in, void* out,
                  int esize) {
  const int vw = 16 / esize;   // elements per 16-B access
  bool vec = (g.Co % vw == 0) && (g.Ci % vw == 0) &&
             (g.kind != S3_OP_CONCAT || (g.c_off % vw == 0 && g.rep % vw == 0));
  int64_t n = (int64_t)g.N * g.Do[0] * g.Do[1] * g.Do[2] * (g.Co / (vec ? vw : 1));
  int grid = grid_for(n, ctx->num_cu);
  if (esize == 4) {
    if (vec) hipLaunchKernelGGL((gather_kernel<4, float>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const float*)in, (float*)out, g);
    else hipLaunchKernelGGL((gather_kernel<1, float>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const float*)in, (float*)out, g);
  } else {
    if (vec) hipLaunchKernelGGL((gather_kernel<8, unsigned short>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const unsigned short*)in, (unsigned short*)out, g);
    else hipLaunchKernelGGL((gather_kernel<1, unsigned short>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const unsigned short*)in, (unsigned short*)out, g);
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

bool gather_bwd_mask_ok(const GatherGeom& g) {
  return g.kind == S3_OP_PAD && g.Ci == g.Co && (g.Ci & 3) == 0;
}

// fold of a padded frame with the producer's activation adjoint applied
// (mask_y: the producer's output, fp32 or bf16)
// channel sums ride along when the geometry allows (see the kernel)
bool gather_bwd_bsum_ok(const GatherGeom& g) {
  const int c4n = g.Ci >> 2;
  return gather_bwd_mask_ok(g) && c4n >= 1 && c4n <= 64 && (256 % c4n) == 0 && kBlock == 256;
}
int gather_bwd_bsum_blocks(const s3_ctx* ctx, const GatherGeom& g) {
  int64_t n = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * g.Ci;
  return grid_for(n / 4, ctx->num_cu);
}

// the 8-channel walk of a bf16 frame: C % 8 == 0, 32-bit item count, and the
// bsum contract (one channel group per lane: c8n | 256, grid stride | c8n)
static bool fold16x8_ok(const s3_ctx* ctx, const GatherGeom& g, const float* bsum) {
  const int64_t n = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * g.Ci;
  const int c8n = g.Ci >> 3;
  if ((g.Ci & 7) || c8n < 1 || n / 8 > 0x7fffffffLL) return false;
  if (bsum && (c8n > 64 || 256 % c8n != 0 || kBlock != 256)) return false;
  return true;
}

int launch_gather_bwd_masked(s3_ctx* ctx, const GatherGeom& g, const float* dout, float* din,
                             const void* mask_y, int y_bf16, float slope, float* bsum, int out_bf16,
                             int frame16) {
  if (!gather_bwd_mask_ok(g)) S3_FAIL(ctx, S3_EINVAL, "gather_bwd_masked: unsupported geometry");
  int64_t n = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * g.Ci;
  if (frame16 && fold16x8_ok(ctx, g, bsum)) {
    const dim3 gridf(grid_for(n / 4, ctx->num_cu));       // (the block count bsum's reader expects)
#define S3_FOLD8(M, O16)                                                                                   \
    hipLaunchKernelGGL((fold16x8_kernel<M, O16, false>), gridf, dim3(kBlock), 0, ctx->stream,              \
                       (const unsigned short*)dout, din, g, mask_y, slope, bsum, (unsigned short*)nullptr)
    if (out_bf16) { if (y_bf16) S3_FOLD8(2, true); else S3_FOLD8(1, true); }
    else { if (y_bf16) S3_FOLD8(2, false); else S3_FOLD8(1, false); }
#undef S3_FOLD8
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (frame16) {    // (dout: bf16 frame)
    const dim3 gridf(grid_for(n / 4, ctx->num_cu));
#define S3_FOLD16(M, O16)                                                                                    \
    hipLaunchKernelGGL((gather_bwd_pad4_kernel<M, O16, false, true>), gridf, dim3(kBlock), 0, ctx->stream,   \
                       dout, din, g, mask_y, slope, bsum)
    if (out_bf16) { if (y_bf16) S3_FOLD16(2, true); else S3_FOLD16(1, true); }
    else { if (y_bf16) S3_FOLD16(2, false); else S3_FOLD16(1, false); }
#undef S3_FOLD16
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (out_bf16) {   // (din: bf16 buffer)
    const dim3 grid16(grid_for(n / 4, ctx->num_cu));
    if (y_bf16)
      hipLaunchKernelGGL((gather_bwd_pad4_kernel<2, true>), grid16, dim3(kBlock), 0, ctx->stream, dout, din, g, mask_y,
                         slope, bsum);
    else
      hipLaunchKernelGGL((gather_bwd_pad4_kernel<1, true>), grid16, dim3(kBlock), 0, ctx->stream, dout, din, g, mask_y,
                         slope, bsum);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  const dim3 grid(grid_for(n / 4, ctx->num_cu));
  if (y_bf16)
    hipLaunchKernelGGL(gather_bwd_pad4_kernel<2>, grid, dim3(kBlock), 0, ctx->stream, dout, din, g, mask_y, slope, bsum);
  else
    hipLaunchKernelGGL(gather_bwd_pad4_kernel<1>, grid, dim3(kBlock), 0, ctx->stream, dout, din, g, mask_y, slope, bsum);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// stage 2 of the bias gradient from channel sums a fold kernel left behind
int launch_bias_grad_from_partial(s3_ctx* ctx, const float* partial, int nblk, int c, float* db, int accumulate,
                                  bool defer) {
  if (ctx->pend_bias.partial) {
    int rc = s3_flush_pending_bias(ctx);
    if (rc) return rc;
  }
  if (defer && !s3_opt_has(S3O_NO_SEG_REDUCE)) {
    // (the 5 us launch rides along the weight gradient's reduction: 44 of the
    // 56 per C2 training step)
    ctx->pend_bias.partial = partial; ctx->pend_bias.nblk = nblk; ctx->pend_bias.c = c;
    ctx->pend_bias.db = db; ctx->pend_bias.accumulate = accumulate;
    return S3_OK;
  }
  hipLaunchKernelGGL(bias_grad_stage2, dim3(c), dim3(256), 0, ctx->stream, partial, nblk, c, db, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int s3_flush_pending_bias(s3_ctx* ctx) {
  if (!ctx->pend_bias.partial) return S3_OK;
  const s3_ctx::PendingBias j = ctx->pend_bias;
  ctx->pend_bias.partial = nullptr;
  hipLaunchKernelGGL(bias_grad_stage2, dim3(j.c), dim3(256), 0, ctx->stream, j.partial, j.nblk, j.c, j.db, j.accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// fold of a padded frame plus an earlier contribution: din = fold(dout) + add
int launch_gather_bwd_add(s3_ctx* ctx, const GatherGeom& g, const float* dout, float* din, const float* add,
                          float* bsum, void* side16, int frame16) {
  if (!gather_bwd_mask_ok(g)) S3_FAIL(ctx, S3_EINVAL, "gather_bwd_add: unsupported geometry");
  int64_t n = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * g.Ci;
  if (frame16 && fold16x8_ok(ctx, g, bsum)) {
    const dim3 gridf(grid_for(n / 4, ctx->num_cu));
    if (side16)
      hipLaunchKernelGGL((fold16x8_kernel<3, false, true>), gridf, dim3(kBlock), 0, ctx->stream,
                         (const unsigned short*)dout, din, g, (const void*)add, 0.f, bsum,
                         (unsigned short*)side16);
    else
      hipLaunchKernelGGL((fold16x8_kernel<3, false, false>), gridf, dim3(kBlock), 0, ctx->stream,
                         (const unsigned short*)dout, din, g, (const void*)add, 0.f, bsum,
                         (unsigned short*)nullptr);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (frame16) {
    if (side16)
      hipLaunchKernelGGL((gather_bwd_pad4_kernel<3, false, true, true>), dim3(grid_for(n / 4, ctx->num_cu)),
                         dim3(kBlock), 0, ctx->stream, dout, din, g, (const void*)add, 0.f, bsum,
                         (unsigned short*)side16);
    else
      hipLaunchKernelGGL((gather_bwd_pad4_kernel<3, false, false, true>), dim3(grid_for(n / 4, ctx->num_cu)),
                         dim3(kBlock), 0, ctx->stream, dout, din, g, (const void*)add, 0.f, bsum);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (side16) {
    hipLaunchKernelGGL((gather_bwd_pad4_kernel<3, false, true>), dim3(grid_for(n / 4, ctx->num_cu)), dim3(kBlock), 0,
                       ctx->stream, dout, din, g, (const void*)add, 0.f, bsum, (unsigned short*)side16);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  hipLaunchKernelGGL(gather_bwd_pad4_kernel<3>, dim3(grid_for(n / 4, ctx->num_cu)), dim3(kBlock), 0, ctx->stream,
                     dout, din, g, (const void*)add, 0.f, bsum);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_gather_bwd(s3_ctx* ctx, const GatherGeom& g, const float* dout,
                      float* din, void* side16, int frame16) {
  int64_t n = (int64_t)g.N * g.Di[0] * g.Di[1] * g.Di[2] * g.Ci;
  if (frame16 && gather_bwd_mask_ok(g) && fold16x8_ok(ctx, g, nullptr)) {
    const dim3 gridf(grid_for(n / 4, ctx->num_cu));
    if (side16)
      hipLaunchKernelGGL((fold16x8_kernel<0, false, true>), gridf, dim3(kBlock), 0, ctx->stream,
                         (const unsigned short*)dout, din, g, (const void*)nullptr, 0.f, (float*)nullptr,
                         (unsigned short*)side16);
    else
      hipLaunchKernelGGL((fold16x8_kernel<0, false, false>), gridf, dim3(kBlock), 0, ctx->stream,
                         (const unsigned short*)dout, din, g, (const void*)nullptr, 0.f, (float*)nullptr,
                         (unsigned short*)nullptr);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (frame16) {
    if (!gather_bwd_mask_ok(g)) S3_FAIL(ctx, S3_EINVAL, "gather_bwd: a bf16 frame needs the float4 fold");
    if (side16)
      hipLaunchKernelGGL((gather_bwd_pad4_kernel<0, false, true, true>), dim3(grid_for(n / 4, ctx->num_cu)),
                         dim3(kBlock), 0, ctx->stream, dout, din, g, (const void*)nullptr, 0.f, (float*)nullptr,
                         (unsigned short*)side16);
    else
      hipLaunchKernelGGL((gather_bwd_pad4_kernel<0, false, false, true>), dim3(grid_for(n / 4, ctx->num_cu)),
                         dim3(kBlock), 0, ctx->stream, dout, din, g, (const void*)nullptr, 0.f, (float*)nullptr);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (side16) {   // (callers check gather_bwd_mask_ok: the float4 fold)
    if (!gather_bwd_mask_ok(g)) S3_FAIL(ctx, S3_EINVAL, "gather_bwd: bf16 side copy needs the float4 fold");
    hipLaunchKernelGGL((gather_bwd_pad4_kernel<0, false, true>), dim3(grid_for(n / 4, ctx->num_cu)), dim3(kBlock), 0,
                       ctx->stream, dout, din, g, (const void*)nullptr, 0.f, (float*)nullptr,
                       (unsigned short*)side16);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (g.kind == S3_OP_PAD && g.Ci == g.Co && (g.Ci & 3) == 0) {
    hipLaunchKernelGGL(gather_bwd_pad4_kernel<0>, dim3(grid_for(n / 4, ctx->num_cu)), dim3(kBlock), 0,
                       ctx->stream, dout, din, g, (const void*)nullptr, 0.f, (float*)nullptr);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, dout, din, g);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_act(s3_ctx* ctx, const float* x, float* y, int64_t n, int act,
               float alpha) {
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n / 4 + 1, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, x, y, n, act, alpha);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_act_bwd(s3_ctx* ctx, const float* y, const float* dy, float* dx,
                   int64_t n, int act, float alpha) {
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, y, dy, dx, n, act, alpha);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

static bool conv_epilogue_bwd_d2s4_geom(const ConvGeom& g) {
  return g.d2s > 1 && ((g.Cout / (g.d2s * g.d2s)) & 3) == 0 && (g.Cout & 3) == 0 &&
         (g.act == S3_ACT_LEAKY || g.act == S3_ACT_RELU || g.act == S3_ACT_NONE);
}
bool conv_epilogue_bwd_d16_ok(const ConvGeom& g) {
  const int64_t n = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] * g.Cout;
  if (conv_epilogue_bwd_d2s4_geom(g)) return true;    // (the depth-to-space walk, bf16 y)
  return g.d2s <= 1 && (n & 3) == 0 && (g.act == S3_ACT_LEAKY || g.act == S3_ACT_RELU);
}

// channel sums can ride along the mask pass (bias gradient): C_out / 4 | 256
// (depth-to-space walk: any C_out / 4 <= 256, with a bf16 y and the bf16 copy)
static int conv_epilogue_bwd_d2s4_block(const ConvGeom& g) {
  const int c4n = g.Cout >> 2;
  return c4n >= 1 && c4n <= 256 ? (256 / c4n) * c4n : 0;
}
bool conv_epilogue_bwd_bsum_ok(const ConvGeom& g) {
  const int c4n = g.Cout >> 2;
  if (conv_epilogue_bwd_d2s4_geom(g)) return conv_epilogue_bwd_d2s4_block(g) > 0 && kBlock == 256;
  return g.d2s <= 1 && conv_epilogue_bwd_d16_ok(g) && (g.Cout & 3) == 0 && c4n >= 1 && c4n <= 64 && (256 % c4n) == 0 &&
         kBlock == 256;
}
int conv_epilogue_bwd_blocks(const s3_ctx* ctx, const ConvGeom& g, bool with_bsum) {
  const int64_t n4 = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] * g.Cout / 4;
  const int blk = (with_bsum && conv_epilogue_bwd_d2s4_geom(g)) ? conv_epilogue_bwd_d2s4_block(g) : kBlock;
  const int64_t want = (n4 + blk - 1) / blk;
  const int64_t cap = with_bsum ? 16 * ctx->num_cu : 32 * ctx->num_cu;   // (bsum rows: <= 4096)
  return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

int launch_conv_epilogue_bwd(s3_ctx* ctx, const ConvGeom& g, const float* y,
                             const float* dy, float* dpre, int y_bf16, void* d16, float* bsum) {
  int64_t n = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] * g.Cout;
  if (!dpre && !(d16 && ((g.d2s <= 1 && (n & 3) == 0 && (g.act == S3_ACT_LEAKY || g.act == S3_ACT_RELU)) ||
                         (conv_epilogue_bwd_d2s4_geom(g) && y_bf16))))
    S3_FAIL(ctx, S3_EINVAL, "conv_epilogue_bwd: a bf16-only dPre needs the 4-channel mask pass");
  if (g.d2s <= 1 && (n & 3) == 0 && (g.act == S3_ACT_LEAKY || g.act == S3_ACT_RELU)) {
    const float slope = g.act == S3_ACT_LEAKY ? g.alpha : 0.f;
    const int64_t n4 = n / 4;
    if (bsum && !conv_epilogue_bwd_bsum_ok(g)) S3_FAIL(ctx, S3_EINVAL, "conv_epilogue_bwd: channel sums need C_out / 4 | 256");
    const dim3 grid((unsigned)conv_epilogue_bwd_blocks(ctx, g, bsum != nullptr));
    if (y_bf16)
      hipLaunchKernelGGL(conv_epilogue_bwd4_kernel<true>, grid, dim3(kBlock), 0, ctx->stream, (const void*)y,
                         (const float4*)dy, (float4*)dpre, n4, slope, (unsigned short*)d16, bsum, g.Cout >> 2);
    else
      hipLaunchKernelGGL(conv_epilogue_bwd4_kernel<false>, grid, dim3(kBlock), 0, ctx->stream, (const void*)y,
                         (const float4*)dy, (float4*)dpre, n4, slope, (unsigned short*)d16, bsum, g.Cout >> 2);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if ((bsum || d16) && !(conv_epilogue_bwd_d2s4_geom(g) && y_bf16 && d16 && (!bsum || conv_epilogue_bwd_bsum_ok(g))))
    S3_FAIL(ctx, S3_EINVAL, "conv_epilogue_bwd: side outputs need the 4-channel path");
  if (conv_epilogue_bwd_d2s4_geom(g)) {
    const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
    const dim3 grid(grid_for(n / 4, ctx->num_cu));
    if (y_bf16 && d16 && bsum)
      hipLaunchKernelGGL((conv_epilogue_bwd_d2s4_kernel<true, true>), dim3((unsigned)conv_epilogue_bwd_blocks(ctx, g, true)),
                         dim3(conv_epilogue_bwd_d2s4_block(g)), 0, ctx->stream, (const void*)y, (const float4*)dy,
                         (float4*)dpre, g, slope, (unsigned short*)d16, bsum);
    else if (y_bf16 && d16)
      hipLaunchKernelGGL((conv_epilogue_bwd_d2s4_kernel<true, true>), grid, dim3(kBlock), 0, ctx->stream,
                         (const void*)y, (const float4*)dy, (float4*)dpre, g, slope, (unsigned short*)d16);
    else if (y_bf16)
      hipLaunchKernelGGL(conv_epilogue_bwd_d2s4_kernel<true>, grid, dim3(kBlock), 0, ctx->stream, (const void*)y,
                         (const float4*)dy, (float4*)dpre, g, slope);
    else
      hipLaunchKernelGGL(conv_epilogue_bwd_d2s4_kernel<false>, grid, dim3(kBlock), 0, ctx->stream, (const void*)y,
                         (const float4*)dy, (float4*)dpre, g, slope);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (y_bf16)
    hipLaunchKernelGGL(conv_epilogue_bwd_kernel<true>, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, y, dy, dpre, g);
  else
    hipLaunchKernelGGL(conv_epilogue_bwd_kernel<false>, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, y, dy, dpre, g);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_add(s3_ctx* ctx, const float* a, const float* b, float* y, int64_t n,
               int c, int bcast_c) {
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, a, b, y, n, c, bcast_c);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_add16(s3_ctx* ctx, const void* a, const void* b, void* y, int64_t n) {
  if (n & 7) S3_FAIL(ctx, S3_EINVAL, "add16: element count must be a multiple of 8");
  hipLaunchKernelGGL(add16_kernel, dim3(grid_for(n / 8, ctx->num_cu)), dim3(kBlock), 0, ctx->stream,
                     (const uint4*)a, (const uint4*)b, (uint4*)y, n / 8);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_axpy(s3_ctx* ctx, const float* x, float* y, int64_t n) {
  if ((n & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    hipLaunchKernelGGL(axpy4_kernel, dim3(grid_for(n / 4, ctx->num_cu)), dim3(kBlock), 0, ctx->stream,
                       (const float4*)x, (float4*)y, n / 4);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, x, y, n);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_fill(s3_ctx* ctx, float* p, int64_t n, float v) {
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, p, n, v);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_bias_grad(s3_ctx* ctx, const float* dy, int64_t n_pos, int c,
                     float* db, int accumulate) {
  if (c > 256) {
    if (n_pos >= 256) {
      // conv layers: split the rows over ~4 blocks per CU, then the fixed-shape tree
      const int cb = (c + 255) / 256;
      int ns = (4 * ctx->num_cu + cb - 1) / cb;
      if ((int64_t)ns * 32 > n_pos) ns = (int)((n_pos + 31) / 32);
      int rc = ensure_scratch(ctx, (size_t)ns * c * sizeof(float));
      if (rc) return rc;
      hipLaunchKernelGGL(bias_grad_cols_split, dim3(cb, ns), dim3(256), 0, ctx->stream, dy, n_pos, c, ctx->scratch);
      hipLaunchKernelGGL(bias_grad_stage2, dim3(c), dim3(256), 0, ctx->stream, ctx->scratch, ns, c, db, accumulate);
      S3_HIP(ctx, hipGetLastError());
      return S3_OK;
    }
    hipLaunchKernelGGL(bias_grad_cols, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, dy, n_pos, c, db, accumulate);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  int block = c <= 256 ? 256 : 1024;
  int rows = block / c;
  int64_t want = (n_pos + rows - 1) / rows;
  // stage 2 walks the partial slabs serially per channel (deterministic);
  // enough slabs to put four blocks on every CU
  const int cap = 4 * ctx->num_cu;
  int nblk = (int)(want < cap ? (want < 1 ? 1 : want) : cap);
  int rc = ensure_scratch(ctx, (size_t)nblk * c * sizeof(float));
  if (rc) return rc;
  if ((c & 3) == 0 && (((uintptr_t)dy) & 15) == 0) {
    const int c4 = c / 4, rows4 = block / c4;
    int64_t want4 = (n_pos + rows4 - 1) / rows4;
    if (want4 < nblk) nblk = (int)(want4 < 1 ? 1 : want4);
    hipLaunchKernelGGL(bias_grad_stage1_v4, dim3(nblk), dim3(block), block * sizeof(float4), ctx->stream,
                       (const float4*)dy, n_pos, c4, ctx->scratch);
  } else
  hipLaunchKernelGGL(bias_grad_stage1, dim3(nblk), dim3(block), block * sizeof(float), ctx->stream, dy, n_pos, c, ctx->scratch);
  hipLaunchKernelGGL(bias_grad_stage2, dim3(c), dim3(256), 0, ctx->stream, ctx->scratch, nblk, c, db, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_mean_abs(s3_ctx* ctx, const float* p, int64_t n, float* out_dev) {
  int nblk = grid_for(n, ctx->num_cu);
  if (nblk > 1024) nblk = 1024;
  int rc = ensure_scratch(ctx, (size_t)(nblk + 4) * sizeof(float));
  if (rc) return rc;
  hipLaunchKernelGGL(mean_abs_stage1, dim3(nblk), dim3(kBlock), 0, ctx->stream, p, n, ctx->scratch);
  hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->scratch, nblk, 1.f / (float)n, out_dev, 0);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// keras-2.15 update_step of the other optimizers sup3r may be given by name
// (abstract.py:321-350); one fused pass over the flat store like Adam.
// hp[]: SGD {lr, momentum, nesterov}; RMSprop {lr, rho, momentum, eps} (not
// centered); Adagrad {lr, eps}; Adamax {lr / (1 - b1^t), 1 - b1, b2, eps};
// AdamW = decoupled decay w -= w * wd * lr, then Adam: {alpha, 1-b1, 1-b2,
// eps, wd * lr}.  m / v are the two slot buffers of the store.
template <int KIND>
__global__ void optimizer_kernel(float* __restrict__ w, const float* __restrict__ g,
                                 float* __restrict__ m, float* __restrict__ v, int64_t n,
                                 float h0, float h1, float h2, float h3, float h4,
                                 const float* __restrict__ hd) {
  if (hd) { h0 = hd[0]; h1 = hd[1]; h2 = hd[2]; h3 = hd[3]; h4 = hd[4]; }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i];
    float wi = w[i];
    if (KIND == S3_OPT_SGD) {
      if (h1 != 0.f) {
        const float mi = -gi * h0 + m[i] * h1;
        m[i] = mi;
        wi += h2 != 0.f ? (-gi * h0 + mi * h1) : mi;
      } else {
        wi += -gi * h0;
      }
    } else if (KIND == S3_OPT_RMSPROP) {
      const float vi = h1 * v[i] + h4 * gi * gi;   // h4 = fp32(1 - rho)
      v[i] = vi;
      const float inc = h0 * gi * (1.f / sqrtf(vi + h3));
      if (h2 > 0.f) {
        const float mi = h2 * m[i] + inc;
        m[i] = mi;
        wi -= mi;
      } else {
        wi -= inc;
      }
    } else if (KIND == S3_OPT_ADAGRAD) {
      const float vi = v[i] + gi * gi;
      v[i] = vi;
      wi -= h0 * gi / sqrtf(vi + h1);
    } else if (KIND == S3_OPT_ADAMAX) {
      const float mi = m[i] + (gi - m[i]) * h1;
      const float ui = fmaxf(h2 * v[i], fabsf(gi));
      m[i] = mi;
      v[i] = ui;
      wi -= (h0 * mi) / (ui + h3);
    } else {   // S3_OPT_ADAMW
      wi -= wi * h4;
      const float mi = m[i] + (gi - m[i]) * h1;
      const float vi = v[i] + (gi * gi - v[i]) * h2;
      m[i] = mi;
      v[i] = vi;
      wi -= (mi * h0) / (sqrtf(vi) + h3);
    }
    w[i] = wi;
  }
}

int launch_optimizer(s3_ctx* ctx, int kind, float* w, const float* g, float* m, float* v,
                     int64_t n, const float* h, const float* h_dev) {
  const dim3 grid(grid_for(n, ctx->num_cu)), blk(kBlock);
#define S3_OPT_LAUNCH(K)                                                                   \
  hipLaunchKernelGGL(optimizer_kernel<K>, grid, blk, 0, ctx->stream, w, g, m, v, n, h[0], \
                     h[1], h[2], h[3], h[4], h_dev)
  switch (kind) {
    case S3_OPT_SGD: S3_OPT_LAUNCH(S3_OPT_SGD); break;
    case S3_OPT_RMSPROP: S3_OPT_LAUNCH(S3_OPT_RMSPROP); break;
    case S3_OPT_ADAGRAD: S3_OPT_LAUNCH(S3_OPT_ADAGRAD); break;
    case S3_OPT_ADAMAX: S3_OPT_LAUNCH(S3_OPT_ADAMAX); break;
    case S3_OPT_ADAMW: S3_OPT_LAUNCH(S3_OPT_ADAMW); break;
    default: S3_FAIL(ctx, S3_EINVAL, "optimizer_step: unknown optimizer kind");
  }
#undef S3_OPT_LAUNCH
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_adam(s3_ctx* ctx, float* w, const float* g, float* m, float* v,
                int64_t n, float alpha, float omb1, float omb2, float eps, const float* h_dev) {
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, w, g, m, v, n, alpha, omb1, omb2, eps, h_dev);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

__global__ void stage_hyper_kernel(float* __restrict__ dst, float h0, float h1, float h2, float h3, float h4) {
  dst[0] = h0; dst[1] = h1; dst[2] = h2; dst[3] = h3; dst[4] = h4;
}

int launch_stage_hyper(s3_ctx* ctx, float* dst, const float* h) {
  hipLaunchKernelGGL(stage_hyper_kernel, dim3(1), dim3(1), 0, ctx->stream, dst, h[0], h[1], h[2], h[3], h[4]);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

static int loss_content_impl(s3_ctx* ctx, int kind, const float* a, int c_a,
                             const float* b, int c_b, const float* mask,
                             int c_m, int c_used, int64_t n_pos, float weight,
                             float* loss_out, float* d_a, int accumulate) {
  if (!ctx) return S3_EINVAL;
  if (c_used > c_a || c_used > c_b) S3_FAIL(ctx, S3_EINVAL, "loss_content: c_used exceeds channel counts");
  int64_t total = n_pos * c_used;
  int nblk = grid_for(total, ctx->num_cu);
  if (nblk > 1024) nblk = 1024;
  int rc = ensure_scratch(ctx, (size_t)(nblk + 4) * sizeof(float));
  if (rc) return rc;
  float gscale = weight / (float)total;
  const bool dense = !mask && c_a == c_used && c_b == c_used && (total & 3) == 0 &&
                     (((uintptr_t)a | (uintptr_t)b | (uintptr_t)d_a) & 15) == 0;
  if (dense)
    hipLaunchKernelGGL(loss_content4_kernel, dim3(nblk), dim3(kBlock), 0, ctx->stream, kind, (const float4*)a,
                       (const float4*)b, total / 4, gscale, ctx->scratch, (float4*)d_a, accumulate);
  else
    hipLaunchKernelGGL(loss_content_kernel, dim3(nblk), dim3(kBlock), 0, ctx->stream, kind, a, c_a, b, c_b, mask, c_m, c_used, n_pos, gscale, ctx->scratch, d_a, accumulate);
  hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->scratch, nblk, 1.f / (float)total, loss_out, 0);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_loss_content(s3_ctx* ctx, int kind, const float* a, int c_a,
                               const float* b, int c_b, int c_used,
                               int64_t n_pos, float weight, float* loss_out,
                               float* d_a, int accumulate) {
  return loss_content_impl(ctx, kind, a, c_a, b, c_b, nullptr, 0, c_used, n_pos,
                           weight, loss_out, d_a, accumulate);
}

extern "C" int s3_loss_content_masked(s3_ctx* ctx, int kind, const float* a,
                                      int c_a, const float* b, int c_b,
                                      const float* mask, int c_m, int c_used,
                                      int64_t n_pos, float weight,
                                      float* loss_out, float* d_a,
                                      int accumulate) {
  if (!mask || c_used > c_m) { if (ctx) ctx->err = "loss_content_masked: bad mask"; return S3_EINVAL; }
  return loss_content_impl(ctx, kind, a, c_a, b, c_b, mask, c_m, c_used, n_pos,
                           weight, loss_out, d_a, accumulate);
}

extern "C" int s3_loss_rel_bce(s3_ctx* ctx, const float* disc_true,
                               const float* disc_gen, int n, float scale,
                               float* loss_out, float* d_true, float* d_gen) {
  if (!ctx) return S3_EINVAL;
  hipLaunchKernelGGL(rel_bce_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, disc_true, disc_gen, n, scale, loss_out, d_true, d_gen);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_copy_channels(s3_ctx* ctx, const float* src, int c_src,
                                int c0_src, float* dst, int c_dst, int c0_dst,
                                int nc, int64_t n_pos, int accumulate) {
  if (!ctx) return S3_EINVAL;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(n_pos * nc, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, src, c_src, c0_src, dst, c_dst, c0_dst, nc, n_pos, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_affine_channels(s3_ctx* ctx, const float* src, float* dst,
                                  int c, int64_t n_pos, const float* scale_host,
                                  const float* shift_host) {
  if (!ctx) return S3_EINVAL;
  if (c > 16) S3_FAIL(ctx, S3_EINVAL, "affine_channels supports at most 16 channels");
  Affine8 a;
  for (int i = 0; i < c; ++i) { a.scale[i] = scale_host[i]; a.shift[i] = shift_host[i]; }
  int64_t n = n_pos * c;
  hipLaunchKernelGGL(affine_channels_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(kBlock), 0, ctx->stream, src, dst, c, n, a);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// strided (d0, d1, row) block copy, fp32, device -> device: the chunk windows
// cut out of the resident lo-res domain and the halo crop of the hi-res output
__global__ void copy_block_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                  int64_t d1, int64_t row, int64_t ss0, int64_t ss1,
                                  int64_t ds0, int64_t ds1, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i % row, q = i / row;
    const int64_t b = q % d1, a = q / d1;
    dst[a * ds0 + b * ds1 + r] = src[a * ss0 + b * ss1 + r];
  }
}

extern "C" int s3_copy_block(s3_ctx* ctx, const float* src, float* dst, int64_t d0, int64_t d1,
                             int64_t row_elems, int64_t src_stride0, int64_t src_stride1,
                             int64_t dst_stride0, int64_t dst_stride1) {
  if (!ctx || !src || !dst) return S3_EINVAL;
  if (d0 < 1 || d1 < 1 || row_elems < 1) S3_FAIL(ctx, S3_EINVAL, "copy_block: empty block");
  const int64_t total = d0 * d1 * row_elems;
  hipLaunchKernelGGL(copy_block_kernel, dim3(grid_for(total, ctx->num_cu)), dim3(kBlock), 0, ctx->stream,
                     src, dst, d1, row_elems, src_stride0, src_stride1, dst_stride0, dst_stride1, total);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// 2-D (spatial) models see a chunk's time steps as their batch axis: the
// generated batch is (n_chunks * T, H, W, c); the chunk the executor delivers is
// hi_res[0][hr_crop_slices] of its transpose to (H, W, T, c)
// (sup3r/pipeline/forward_pass.py:274-337,272) — un-normalised on the way
struct ChunkTL { int64_t T, H, W; int64_t lo[3], n[3]; int c, affine; float scale[16], shift[16]; };
__global__ void chunk_time_last_kernel(const float* __restrict__ y, float* __restrict__ yc, ChunkTL e) {
  const int64_t per = e.n[0] * e.n[1] * e.n[2] * e.c;
  const int k = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int ch = (int)(r % e.c); r /= e.c;
    const int64_t t = r % e.n[2]; r /= e.n[2];
    const int64_t w = r % e.n[1]; r /= e.n[1];
    const int64_t h = r;
    float v = y[((((int64_t)k * e.T + e.lo[2] + t) * e.H + e.lo[0] + h) * e.W + e.lo[1] + w) * e.c + ch];
    if (e.affine) {
      // two roundings, as numpy's (x * scale) + shift (affine_channels_kernel)
      float m = v * e.scale[ch];
      asm volatile("" : "+v"(m));
      v = m + e.shift[ch];
    }
    yc[(int64_t)k * per + i] = v;
  }
}

// ... and the way in: n chunks (s1, s2, t, c) -> the (n t, s1, s2, c) batch of a
// 2-D model, normalised (x - mean) / std with numpy's arithmetic: in fp32 when
// the statistics are fp32 arrays, in fp64 (then rounded to fp32) when they are
// fp64 — the two cases of Sup3rGan.norm_input (abstract.py:197-238)
struct ChunkTF { int64_t H, W, T; int c, mode; float mean[16], sd[16]; double dmean[16], dsd[16]; };
__global__ void chunk_time_first_kernel(const float* __restrict__ x, float* __restrict__ out, ChunkTF e) {
  const int64_t per = e.H * e.W * e.T * e.c;
  const int k = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per;
       i += (int64_t)gridDim.x * blockDim.x) {
    // i runs over the OUTPUT (t, h, w, ch) of chunk k
    int64_t r = i;
    const int ch = (int)(r % e.c); r /= e.c;
    const int64_t w = r % e.W; r /= e.W;
    const int64_t h = r % e.H; r /= e.H;
    const int64_t t = r;
    float v = x[(int64_t)k * per + ((h * e.W + w) * e.T + t) * e.c + ch];
    if (e.mode == 1) {
      float d = v - e.mean[ch];
      asm volatile("" : "+v"(d));
      v = __fdiv_rn(d, e.sd[ch]);
    } else if (e.mode == 2) {
      v = (float)(((double)v - e.dmean[ch]) / e.dsd[ch]);
    }
    out[(int64_t)k * per + i] = v;
  }
}

extern "C" int s3_chunk_time_first(s3_ctx* ctx, const float* x, int n_chunks, const int64_t* hwt, int c,
                                   const double* mean_host, const double* std_host, int stats_fp32,
                                   float* out) {
  if (!ctx || !x || !out || !hwt) return S3_EINVAL;
  if (n_chunks < 1 || c < 1 || c > 16) S3_FAIL(ctx, S3_EINVAL, "chunk_time_first: 1 .. 16 channels");
  ChunkTF e;
  e.H = hwt[0]; e.W = hwt[1]; e.T = hwt[2]; e.c = c;
  e.mode = (mean_host && std_host) ? (stats_fp32 ? 1 : 2) : 0;
  for (int i = 0; i < 16; ++i) {
    const double m = e.mode && i < c ? mean_host[i] : 0.0, sd = e.mode && i < c ? std_host[i] : 1.0;
    e.mean[i] = (float)m; e.sd[i] = (float)sd; e.dmean[i] = m; e.dsd[i] = sd;
  }
  const int64_t per = e.H * e.W * e.T * c;
  hipLaunchKernelGGL(chunk_time_first_kernel, dim3(grid_for(per, ctx->num_cu), n_chunks), dim3(kBlock), 0,
                     ctx->stream, x, out, e);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_chunk_time_last(s3_ctx* ctx, const float* y, int n_chunks, const int64_t* thw,
                                  const int64_t* crop_lo, const int64_t* crop_n, int c,
                                  const float* scale_host, const float* shift_host, float* yc) {
  if (!ctx || !y || !yc || !thw || !crop_lo || !crop_n) return S3_EINVAL;
  if (n_chunks < 1 || c < 1 || c > 16) S3_FAIL(ctx, S3_EINVAL, "chunk_time_last: 1 .. 16 channels");
  // crop axes are (H, W, T): the chunk's (s1, s2, t)
  const int64_t ext[3] = {thw[1], thw[2], thw[0]};
  for (int d = 0; d < 3; ++d)
    if (crop_lo[d] < 0 || crop_n[d] < 1 || crop_lo[d] + crop_n[d] > ext[d])
      S3_FAIL(ctx, S3_EINVAL, "chunk_time_last: the crop window leaves the chunk");
  ChunkTL e;
  e.T = thw[0]; e.H = thw[1]; e.W = thw[2];
  for (int d = 0; d < 3; ++d) { e.lo[d] = crop_lo[d]; e.n[d] = crop_n[d]; }
  e.c = c;
  e.affine = scale_host && shift_host;
  for (int i = 0; i < 16; ++i) {
    e.scale[i] = e.affine && i < c ? scale_host[i] : 1.f;
    e.shift[i] = e.affine && i < c ? shift_host[i] : 0.f;
  }
  const int64_t per = crop_n[0] * crop_n[1] * crop_n[2] * c;
  int gx = grid_for(per, ctx->num_cu);
  hipLaunchKernelGGL(chunk_time_last_kernel, dim3(gx, n_chunks), dim3(kBlock), 0, ctx->stream, y, yc, e);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// dst[o][a][r][b] = src[o][a][b], r < reps: a time-invariant exo field
// (topography) uploaded once per chunk and laid over the chunk's time steps —
// (n, H W c) -> (n reps, H W c) for a 2-D model (a = 1), (n, H W, c) ->
// (n, H W, reps, c) for a 3-D one (ForwardPass.pad_source_data,
// sup3r/pipeline/forward_pass.py:160-186, does this with np.repeat on the host)
__global__ void broadcast_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t oa,
                                      int64_t reps, int64_t b) {
  const int64_t total = oa * reps * b;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = i % b;
    const int64_t q = i / (b * reps);
    dst[i] = src[q * b + bi];
  }
}

extern "C" int s3_broadcast_axis(s3_ctx* ctx, const float* src, int64_t outer, int64_t a, int64_t b, int64_t reps,
                                 float* dst) {
  if (!ctx || !src || !dst) return S3_EINVAL;
  if (outer < 1 || a < 1 || b < 1 || reps < 1) S3_FAIL(ctx, S3_EINVAL, "broadcast_axis: empty extent");
  hipLaunchKernelGGL(broadcast_axis_kernel, dim3(grid_for(outer * a * reps * b, ctx->num_cu)), dim3(kBlock), 0,
                     ctx->stream, src, dst, outer * a, reps, b);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// One model step's output -> the next step's input of a MultiStepGan chain
// (sup3r/models/multi_step.py:233-259), position by position:
// un_norm_output of step i (y * std + mean, abstract.py:240-275),
// _match_model_input's channel selection (multi_step.py:148-194), the 'input'
// exo channels _combine_fwp_input appends (interface.py:259-356), norm_input
// of step i + 1 ((x - mean) / std, abstract.py:197-238) — numpy's fp32
// arithmetic: one rounding per operation, no fused multiply-add.
struct StepHO { int c_src, c_sel, n_exo, un, nrm; int map[16]; float scale[16], shift[16], mean[16], sd[16]; };
__global__ void step_handover_kernel(const float* __restrict__ y, const float* __restrict__ exo,
                                     float* __restrict__ x, int64_t n_pos, StepHO e) {
  const int c_dst = e.c_sel + e.n_exo;
  const int64_t total = n_pos * c_dst;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c_dst;
    const int ch = (int)(i - p * c_dst);
    float v;
    if (ch < e.c_sel) {
      const int sc = e.map[ch];
      v = y[p * e.c_src + sc];
      if (e.un) {
        float m = v * e.scale[sc];
        asm volatile("" : "+v"(m));
        v = m + e.shift[sc];
      }
    } else {
      v = exo[p * e.n_exo + (ch - e.c_sel)];
    }
    if (e.nrm) {
      float d = v - e.mean[ch];
      asm volatile("" : "+v"(d));
      v = __fdiv_rn(d, e.sd[ch]);
    }
    x[i] = v;
  }
}

extern "C" int s3_step_handover(s3_ctx* ctx, const float* y, int64_t n_pos, int c_src, const int* map_host,
                                int c_sel, const float* scale_host, const float* shift_host, const float* exo,
                                int n_exo, const float* mean_host, const float* std_host, float* x) {
  if (!ctx || !y || !x || !map_host) return S3_EINVAL;
  if (n_pos < 1 || c_src < 1 || c_src > 16 || c_sel < 1 || n_exo < 0 || c_sel + n_exo > 16)
    S3_FAIL(ctx, S3_EINVAL, "step_handover: 1 .. 16 channels on either side");
  if (n_exo > 0 && !exo) S3_FAIL(ctx, S3_EINVAL, "step_handover: exo channels without an exo tensor");
  StepHO e;
  e.c_src = c_src; e.c_sel = c_sel; e.n_exo = n_exo;
  e.un = scale_host && shift_host;
  e.nrm = mean_host && std_host;
  for (int i = 0; i < 16; ++i) {
    e.map[i] = 0;
    if (i < c_sel) {
      if (map_host[i] < 0 || map_host[i] >= c_src) S3_FAIL(ctx, S3_EINVAL, "step_handover: channel map out of range");
      e.map[i] = map_host[i];
    }
    e.scale[i] = e.un && i < c_src ? scale_host[i] : 1.f;
    e.shift[i] = e.un && i < c_src ? shift_host[i] : 0.f;
    e.mean[i] = e.nrm && i < c_sel + n_exo ? mean_host[i] : 0.f;
    e.sd[i] = e.nrm && i < c_sel + n_exo ? std_host[i] : 1.f;
  }
  hipLaunchKernelGGL(step_handover_kernel, dim3(grid_for(n_pos * (c_sel + n_exo), ctx->num_cu)), dim3(kBlock), 0,
                     ctx->stream, y, exo, x, n_pos, e);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_chunk_stats(s3_ctx* ctx, const float* x, int n_chunks,
                              int64_t pos_per_chunk, int c, float* partial) {
  if (!ctx) return S3_EINVAL;
  if (n_chunks < 1 || c < 1) S3_FAIL(ctx, S3_EINVAL, "chunk_stats: empty input");
  const int64_t per = pos_per_chunk * c;
  if (c <= 16 && 1024 % c == 0 && per % 4 == 0 && !((uintptr_t)x & 15))
    hipLaunchKernelGGL(chunk_stats4_kernel, dim3(kStatSlabs, n_chunks), dim3(256), 0, ctx->stream, x, per / 4, c,
                       partial);
  else
    hipLaunchKernelGGL(chunk_stats_kernel, dim3(kStatSlabs, n_chunks), dim3(256), 0,
                       ctx->stream, x, pos_per_chunk, c, partial);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_chunk_epilogue(s3_ctx* ctx, const float* y, int n_chunks, const int64_t* dims,
                                 const int64_t* crop_lo, const int64_t* crop_n, int c,
                                 const float* scale_host, const float* shift_host, float* yc,
                                 float* partial) {
  if (!ctx || !y || !yc || !partial || !dims || !crop_lo || !crop_n) return S3_EINVAL;
  if (n_chunks < 1 || c < 1 || c > 16 || 1024 % c != 0)
    S3_FAIL(ctx, S3_EINVAL, "chunk_epilogue: 1 .. 16 channels, a divisor of 1024");
  for (int d = 0; d < 3; ++d)
    if (crop_lo[d] < 0 || crop_n[d] < 1 || crop_lo[d] + crop_n[d] > dims[d])
      S3_FAIL(ctx, S3_EINVAL, "chunk_epilogue: the crop window leaves the chunk");
  // float4 rows: 16-byte aligned row starts and lengths (else the caller takes
  // the three-kernel path)
  if ((crop_n[2] * c) % 4 || (crop_lo[2] * c) % 4 || (dims[2] * c) % 4 || ((uintptr_t)y & 15) ||
      ((uintptr_t)yc & 15))
    S3_FAIL(ctx, S3_EINVAL, "chunk_epilogue: rows are not 16-byte aligned");
  ChunkEpi e;
  e.y_s1 = dims[2] * c;
  e.y_s0 = dims[1] * e.y_s1;
  e.y_chunk = dims[0] * e.y_s0;
  e.y_org = crop_lo[0] * e.y_s0 + crop_lo[1] * e.y_s1 + crop_lo[2] * c;
  e.c0 = (int)crop_n[0]; e.c1 = (int)crop_n[1];
  e.row4 = (int)(crop_n[2] * c / 4);
  e.c = c;
  e.affine = scale_host && shift_host;
  for (int i = 0; i < 16; ++i) {
    e.scale[i] = e.affine && i < c ? scale_host[i] : 1.f;
    e.shift[i] = e.affine && i < c ? shift_host[i] : 0.f;
  }
  hipLaunchKernelGGL(chunk_epilogue_kernel, dim3(kStatSlabs, n_chunks), dim3(256), 0, ctx->stream, y, yc,
                     partial, e);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// ---- direct device -> host placement of a cropped chunk -------------------
extern "C" int s3_host_register(s3_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx || !ptr) return S3_EINVAL;
  S3_HIP(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return S3_OK;
}

extern "C" int s3_host_unregister(s3_ctx* ctx, void* ptr) {
  if (!ctx || !ptr) return S3_EINVAL;
  S3_HIP(ctx, hipHostUnregister(ptr));
  return S3_OK;
}

extern "C" int s3_d2h_window(s3_ctx* ctx, const float* src, float* dst_host, int64_t d0,
                             int64_t d1, int64_t row_elems, int64_t dst_stride0,
                             int64_t dst_stride1, void* stream) {
  if (!ctx || !src || !dst_host) return S3_EINVAL;
  if (d0 < 1 || d1 < 1 || row_elems < 1 || dst_stride1 < row_elems ||
      dst_stride0 % dst_stride1 != 0 || dst_stride0 / dst_stride1 < d1)
    S3_FAIL(ctx, S3_EINVAL, "d2h_window: the destination is not a pitched (d0, d1, row) window");
  hipMemcpy3DParms p = {};
  const size_t row_bytes = (size_t)row_elems * sizeof(float);
  p.srcPtr = make_hipPitchedPtr(const_cast<float*>(src), row_bytes, row_bytes, (size_t)d1);
  p.dstPtr = make_hipPitchedPtr(dst_host, (size_t)dst_stride1 * sizeof(float), row_bytes,
                                (size_t)(dst_stride0 / dst_stride1));
  p.extent = make_hipExtent(row_bytes, (size_t)d1, (size_t)d0);
  p.kind = hipMemcpyDeviceToHost;
  S3_HIP(ctx, hipMemcpy3DAsync(&p, stream ? (hipStream_t)stream : ctx->stream));
  return S3_OK;
}

// ---- throttled device -> pinned-host stream (the C3 executor's hi-res chunks)
// hipMemcpyAsync(DeviceToHost) of a large buffer runs as a full-grid blit
// kernel on this runtime (__amd_rocclr_copyBuffer): its waves park on PCIe
// write credit in every wave slot of the chip, and the NEXT batch's first
// kernels — on the compute stream, meant to overlap it — queue behind them
// (the 4 -> 64 head conv of a C3 batch took 3.0 ms instead of 20 us next to
// it).  PCIe Gen5 x16 moves ~55 GB/s: a handful of workgroups with a few
// 16-B stores in flight per lane saturate it, so the copy is a grid of
// `blocks` workgroups (default 16 of the 256 CUs' worth) writing through the
// host-mapped pointer; everything else of the chip stays with the forward pass.
typedef unsigned d2h_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void d2h_stream_kernel(const d2h_u32x4* __restrict__ src,
                                                         d2h_u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    d2h_u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i + q * 256 < n16) v[q] = __builtin_nontemporal_load(src + i + q * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i + q * 256 < n16) dst[i + q * 256] = v[q];
  }
}

extern "C" int s3_d2h_stream(s3_ctx* ctx, const void* src, void* dst_host, size_t bytes,
                             void* stream, int blocks) {
  if (!ctx || !src || !dst_host) return S3_EINVAL;
  if ((bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst_host & 15))
    S3_FAIL(ctx, S3_EINVAL, "d2h_stream: 16-byte aligned buffers of a multiple of 16 bytes");
  if (bytes == 0) return S3_OK;
  void* dptr = nullptr;
  // (pinned host memory: the device-side alias of the host pointer)
  if (hipHostGetDevicePointer(&dptr, dst_host, 0) != hipSuccess) {
    (void)hipGetLastError();
    S3_FAIL(ctx, S3_EINVAL, "d2h_stream: the destination is not pinned (mapped) host memory");
  }
  if (blocks < 1) blocks = 16;
  const size_t n16 = bytes / 16;
  const size_t need = (n16 + 1023) / 1024;
  if ((size_t)blocks > need) blocks = (int)need;
  hipLaunchKernelGGL(d2h_stream_kernel, dim3(blocks), dim3(256), 0,
                     stream ? (hipStream_t)stream : ctx->stream, (const d2h_u32x4*)src, (d2h_u32x4*)dptr, n16);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

extern "C" int s3_host_alloc(s3_ctx* ctx, size_t bytes, int noncoherent, void** out) {
  if (!ctx || !out || bytes == 0) return S3_EINVAL;
  *out = nullptr;
  const unsigned flags = hipHostMallocPortable | hipHostMallocMapped |
                         (noncoherent ? hipHostMallocNonCoherent : hipHostMallocCoherent);
  S3_HIP(ctx, hipHostMalloc(out, bytes, flags));
  return S3_OK;
}

extern "C" int s3_host_free(s3_ctx* ctx, void* ptr) {
  if (!ctx) return S3_EINVAL;
  if (ptr) S3_HIP(ctx, hipHostFree(ptr));
  return S3_OK;
}

extern "C" int s3_d2h_async(s3_ctx* ctx, const void* src, void* dst_host, size_t bytes, void* stream) {
  if (!ctx || !src || !dst_host) return S3_EINVAL;
  S3_HIP(ctx, hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost,
                             stream ? (hipStream_t)stream : ctx->stream));
  return S3_OK;
}

extern "C" int s3_fill(s3_ctx* ctx, float* dst, int64_t n, float value) {
  if (!ctx) return S3_EINVAL;
  return launch_fill(ctx, dst, n, value);
}
