// Split-bf16 (S3_PREC_BF16X3) forms of the two hi-res discriminator data
// gradients that bf16 plans run on LDS-halo kernels — round 4.
//
// A BF16X3 plan keeps fp32 activations / gradients and computes every product
// as hi*hi + hi*lo + lo*hi with hi = bf16(v), lo = bf16(v - hi) on the bf16
// matrix cores.  Through round 3 the data gradients of the
// 2 -> 32 first layer (a 32 -> 2 full correlation over 14.7 M positions) and of
// the 32 -> 32 stride-2 layer fell back to the gather-MFMA adjoint, which
// gathers fp32 dPre cell by cell: 7.8 ms and ~ 3 ms per call at C2 batch 8, 22 %
// of the BF16X3 training step.  Here they get the halo-tile treatment of
// kernels_conv_dgrad_fewch.hip / kernels_conv_dgrad_s2.hip with BOTH halves of
// dPre staged (two bf16 halos, split in the staging loop) and hi / lo filter
// images:
//
//   conv_dgrad_c2_x3_kernel     dx[i][ci] = sum W[tap][ci][co] dPre[i + lo - tap][co]
//                               4 x 8 x 16 output tiles, 6 x 10 x 18 halo x 2
//   conv_dgrad_s2_x3_kernel<NF> the parity-class walk of conv_dgrad_s2_kernel,
//                               4 x 8 x 16 u-tiles, 5 x 9 x 17 halo x 2
//
// Same index math, swizzles and fragment maps as the bf16 kernels (their
// comments explain them); per (tap, fragment) three MFMAs, small terms first.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// eight fp32 values -> their hi and lo bf16 halves (16 B each)
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  hi = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
  lo = make_uint4(pk2(a.x - bf_lo(hi.x), a.y - bf_hi(hi.x)), pk2(a.z - bf_lo(hi.y), a.w - bf_hi(hi.y)),
                  pk2(b.x - bf_lo(hi.z), b.y - bf_hi(hi.z)), pk2(b.z - bf_lo(hi.w), b.w - bf_hi(hi.w)));
}
#define MFMA3(acc, ah, al, bh, bl)                                              \
  do {                                                                          \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);        \
  } while (0)

// ----------------------------------------------------------- 32 -> C_in <= 4
constexpr int DT0 = 4, DT1 = 8, DT2 = 16;
constexpr int DH0 = DT0 + 2, DH1 = DT1 + 2, DH2 = DT2 + 2;
constexpr int DHP = DH0 * DH1 * DH2;         // 1080 halo cells
constexpr int DNT = 256;
constexpr int DLDS1 = DHP * 64;              // one half: 69,120 B
constexpr int C2_IMG = 27 * 16 * 32;         // elements of one filter image

// fp32 w[tap][cin][32] -> hi and lo bf16 images img[half][tap'][16 rows][32],
// tap' = 26 - tap, rows >= cin zero
__global__ void dgrad_c2_x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                        int cin) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < C2_IMG; idx += gridDim.x * blockDim.x) {
    const int co = idx & 31, row = (idx >> 5) & 15, tp = idx >> 9;
    const float v = row < cin ? w[((size_t)(26 - tp) * cin + row) * 32 + co] : 0.f;
    const unsigned h = pk2(v, 0.f) & 0xFFFFu;
    img[idx] = (unsigned short)h;
    img[C2_IMG + idx] = (unsigned short)(pk2(v - bf_lo(h), 0.f) & 0xFFFFu);
  }
}

__global__ __launch_bounds__(DNT) void conv_dgrad_c2_x3_kernel(
    const float* __restrict__ dy, const unsigned short* __restrict__ img, float* __restrict__ dx,
    ConvGeom g, int tiles0, int tiles1, int tiles2) {
  extern __shared__ __attribute__((aligned(16))) char halo[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  int tr = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int org0 = t0i * DT0, org1 = t1i * DT1, org2 = t2i * DT2;
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2];

  // ---- stage both halves of the dPre halo: cell = dPre[org + c + lo - 2], zero outside
  for (int base = tid; base < DHP * 4; base += DNT * 2) {
    float4 va[2], vb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = base + u * DNT;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      if (item < DHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        int h = hp;
        const int c2 = h % DH2; h /= DH2;
        const int c1 = h % DH1; h /= DH1;
        const int c0 = h;
        const int i0 = org0 + c0 + g.lo[0] - 2, i1 = org1 + c1 + g.lo[1] - 2,
                  i2 = org2 + c2 + g.lo[2] - 2;
        if (i0 >= 0 && i0 < O0 && i1 >= 0 && i1 < O1 && i2 >= 0 && i2 < O2) {
          const float* src = dy + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + ch * 8;
          va[u] = *reinterpret_cast<const float4*>(src);
          vb[u] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = base + u * DNT;
      if (item < DHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        const int key = ((hp % DH2) >> 1) & 3;
        uint4 hi, lo;
        split8(va[u], vb[u], hi, lo);
        *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) = hi;
        *reinterpret_cast<uint4*>(halo + DLDS1 + hp * 64 + ((ch ^ key) << 4)) = lo;
      }
    }
  }
  __syncthreads();

  int off_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) off_c[c] = (j + c) * 64 + ((kg ^ (((j + c) >> 1) & 3)) << 4);
  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // filter fragments of a tap: lane (row = j, kg) holds co 8 kg .. 8 kg + 7 (the
  // 27 KB hi + lo images are L1-resident; 54 fragments do not fit in registers
  // next to a second halo's reads)
  const unsigned short* wl = img + j * 32 + kg * 8;
  bf16x8 ah = *reinterpret_cast<const bf16x8*>(wl), al = *reinterpret_cast<const bf16x8*>(wl + C2_IMG);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int tp = (a * 3 + b) * 3 + c;
        bf16x8 nh = ah, nl = al;
        if (tp < 26) {
          nh = *reinterpret_cast<const bf16x8*>(wl + (tp + 1) * 16 * 32);
          nl = *reinterpret_cast<const bf16x8*>(wl + C2_IMG + (tp + 1) * 16 * 32);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const char* p = halo + (((wave + a) * DH1 + (m + b)) * DH2) * 64 + off_c[c];
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(p);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(p + DLDS1);
          MFMA3(acc[m], ah, al, bh, bl);
        }
        ah = nh; al = nl;
      }

  if (kg == 0) {
    const int cin = g.Cin;
    const int o0 = org0 + wave, o2 = org2 + j;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int o1 = org1 + m;
      if (o0 < g.D[0] && o1 < g.D[1] && o2 < g.D[2]) {
        float* dst = dx + ((((size_t)n * g.D[0] + o0) * g.D[1] + o1) * g.D[2] + o2) * cin;
        if (cin == 2) {
          *reinterpret_cast<float2*>(dst) = make_float2(acc[m][0], acc[m][1]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < cin) dst[r] = acc[m][r];
        }
      }
    }
  }
}

// ------------------------------------------------- stride 2, 32 output channels
constexpr int ST0 = 4, ST1 = 8, ST2 = 16;
constexpr int SG0 = ST0 + 1, SG1 = ST1 + 1, SG2 = ST2 + 1;
constexpr int SHP = SG0 * SG1 * SG2;          // 765 halo cells
constexpr int SNT = 256;
constexpr int SLDS1 = SHP * 64;               // one half: 48,960 B

int s2x_rows_pad(int cin) { return (cin + 63) / 64 * 64; }

// fp32 w[tap][cin][32] -> hi and lo images img[half][tap][cin_pad][32]
__global__ void dgrad_s2_x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                        int cin, int rows_pad) {
  const int total = 27 * rows_pad * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int co = idx & 31, row = (idx >> 5) % rows_pad, tp = idx / (32 * rows_pad);
    const float v = row < cin ? w[((size_t)tp * cin + row) * 32 + co] : 0.f;
    const unsigned h = pk2(v, 0.f) & 0xFFFFu;
    img[idx] = (unsigned short)h;
    img[total + idx] = (unsigned short)(pk2(v - bf_lo(h), 0.f) & 0xFFFFu);
  }
}

template <int NF>
__global__ __launch_bounds__(SNT) void conv_dgrad_s2_x3_kernel(
    const float* __restrict__ dy, const unsigned short* __restrict__ img, float* __restrict__ dx,
    ConvGeom g, int rows_pad, int tiles0, int tiles1, int tiles2, const float* __restrict__ mask_y,
    float mask_slope) {
  extern __shared__ __attribute__((aligned(16))) char halo[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int ct = blockIdx.y;
  int tr = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int u0 = t0i * ST0, u1 = t1i * ST1, u2 = t2i * ST2;
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2];
  const size_t img_lo = (size_t)27 * rows_pad * 32;

  // ---- stage both halves of the dPre halo: cell (c0, c1, c2) = dPre[u + c - 1]
  for (int base = tid; base < SHP * 4; base += SNT * 2) {
    float4 va[2], vb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = base + u * SNT;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      if (item < SHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        int h = hp;
        const int c2 = h % SG2; h /= SG2;
        const int c1 = h % SG1; h /= SG1;
        const int c0 = h;
        const int i0 = u0 + c0 - 1, i1 = u1 + c1 - 1, i2 = u2 + c2 - 1;
        if (i0 >= 0 && i0 < O0 && i1 >= 0 && i1 < O1 && i2 >= 0 && i2 < O2) {
          const float* src = dy + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + ch * 8;
          va[u] = *reinterpret_cast<const float4*>(src);
          vb[u] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = base + u * SNT;
      if (item < SHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        const int key = ((hp % SG2) >> 1) & 3;
        uint4 hi, lo;
        split8(va[u], vb[u], hi, lo);
        *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) = hi;
        *reinterpret_cast<uint4*>(halo + SLDS1 + hp * 64 + ((ch ^ key) << 4)) = lo;
      }
    }
  }
  __syncthreads();

  int off_d[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int th = j + 1 - d;
    off_d[d] = th * 64 + ((kg ^ ((th >> 1) & 3)) << 4);
  }
  const unsigned short* wrow = img + ((size_t)ct * 64 + j) * 32 + kg * 8;
  const int R = g.Cin;
#pragma unroll 1
  for (int cls = 0; cls < 8; ++cls) {
    const int p0 = cls >> 2, p1 = (cls >> 1) & 1, p2 = cls & 1;
    const int n0 = p0 ? 1 : 2, n1 = p1 ? 1 : 2, n2 = p2 ? 1 : 2;   // taps per axis
    f32x4 acc[8][NF];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < n0; ++a)
      for (int b = 0; b < n1; ++b)
        for (int c = 0; c < n2; ++c) {
          const int ta = p0 ? 1 : 2 * a, tb = p1 ? 1 : 2 * b, tc = p2 ? 1 : 2 * c;
          const int d0 = p0 ? 0 : a, d1 = p1 ? 0 : b, d2 = p2 ? 0 : c;
          const int tp = (ta * 3 + tb) * 3 + tc;
          bf16x8 ah[NF], al[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const unsigned short* wp = wrow + ((size_t)tp * rows_pad + nf * 16) * 32;
            ah[nf] = *reinterpret_cast<const bf16x8*>(wp);
            al[nf] = *reinterpret_cast<const bf16x8*>(wp + img_lo);
          }
          const char* hb = halo + (((wave + 1 - d0) * SG1 + (1 - d1)) * SG2) * 64 + off_d[d2];
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hb + m * SG2 * 64);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hb + SLDS1 + m * SG2 * 64);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) MFMA3(acc[m][nf], ah[nf], al[nf], bh, bl);
          }
        }
    // ---- store the class: x position i = 2 u + p
    const int i0 = 2 * (u0 + wave) + p0, i2 = 2 * (u2 + j) + p2;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int ch = ct * 64 + nf * 16 + kg * 4;
      if (ch >= R) continue;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int i1 = 2 * (u1 + m) + p1;
        if (i0 >= g.D[0] || i1 >= g.D[1] || i2 >= g.D[2]) continue;
        const size_t e = ((((size_t)n * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + i2) * R + ch;
        float4 v = make_float4(acc[m][nf][0], acc[m][nf][1], acc[m][nf][2], acc[m][nf][3]);
        if (mask_y) {   // fused activation adjoint of the producer of x
          const float4 yv = *reinterpret_cast<const float4*>(mask_y + e);
          v.x *= yv.x > 0.f ? 1.f : mask_slope; v.y *= yv.y > 0.f ? 1.f : mask_slope;
          v.z *= yv.z > 0.f ? 1.f : mask_slope; v.w *= yv.w > 0.f ? 1.f : mask_slope;
        }
        *reinterpret_cast<float4*>(dx + e) = v;
      }
    }
  }
}

}  // namespace

// ---- geometry predicates: those of the bf16 kernels, for BF16X3 plans
bool conv_dgrad_c2_x3_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16X3 || s3_opt_has(S3O_NO_DGRAD_C2) || s3_opt_has(S3O_NO_DGRAD_X3)) return false;
  if (g.Cin < 1 || g.Cin > 4 || g.Cout != 32 || g.d2s != 1) return false;
  if (g.pad_mode == S3_PAD_REFLECT) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] < 0 || g.lo[d] > 2) return false;
    if (g.O[d] != g.D[d] + 2 * g.lo[d] - 2) return false;
  }
  return true;
}
size_t conv_dgrad_c2_x3_packed_bytes() { return (size_t)2 * C2_IMG * 2; }
int launch_conv_dgrad_c2_x3_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  hipLaunchKernelGGL(dgrad_c2_x3_pack_kernel, dim3(54), dim3(256), 0, ctx->stream, w, (unsigned short*)img,
                     g.Cin);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
int launch_conv_dgrad_c2_x3(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img, float* dx) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_c2_x3_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DLDS1));
    attr_set.mark(ctx->device);
  }
  const int tiles0 = (g.D[0] + DT0 - 1) / DT0, tiles1 = (g.D[1] + DT1 - 1) / DT1,
            tiles2 = (g.D[2] + DT2 - 1) / DT2;
  hipLaunchKernelGGL(conv_dgrad_c2_x3_kernel, dim3((unsigned)(g.N * tiles0 * tiles1 * tiles2)), dim3(DNT),
                     2 * DLDS1, ctx->stream, dy, (const unsigned short*)img, dx, g, tiles0, tiles1, tiles2);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

bool conv_dgrad_s2_x3_supported(const s3_ctx* ctx, const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16X3 || s3_opt_has(S3O_NO_DGRAD_S2) || s3_opt_has(S3O_NO_DGRAD_X3)) return false;
  if (g.Cout != 32 || g.Cin % 16 != 0 || g.Cin < 16 || g.d2s != 1) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.k[d] != 3 || g.s[d] != 2 || g.lo[d] != 0) return false;
    if ((g.O[d] - 1) * 2 + 3 > g.D[d] + 1) return false;
  }
  const int64_t min_tiles = s3_opt_has(S3O_DGRAD_S2_MIN_TILES) ? s3_opt_int(S3O_DGRAD_S2_MIN_TILES, 0)
                                                                   : ctx->num_cu;
  int64_t tiles = g.N;
  const int T[3] = {ST0, ST1, ST2};
  for (int d = 0; d < 3; ++d) tiles *= ((g.D[d] + 1) / 2 + T[d] - 1) / T[d];
  return tiles >= min_tiles;
}
size_t conv_dgrad_s2_x3_packed_bytes(const ConvGeom& g) { return (size_t)2 * 27 * s2x_rows_pad(g.Cin) * 32 * 2; }
int launch_conv_dgrad_s2_x3_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  const int rp = s2x_rows_pad(g.Cin);
  hipLaunchKernelGGL(dgrad_s2_x3_pack_kernel, dim3((27 * rp * 32 + 255) / 256), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)img, g.Cin, rp);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
int launch_conv_dgrad_s2_x3(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img, float* dx,
                            const float* mask_y, float mask_slope) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_x3_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLDS1));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_x3_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLDS1));
    attr_set.mark(ctx->device);
  }
  const int U0 = (g.D[0] + 1) / 2, U1 = (g.D[1] + 1) / 2, U2 = (g.D[2] + 1) / 2;
  const int tiles0 = (U0 + ST0 - 1) / ST0, tiles1 = (U1 + ST1 - 1) / ST1, tiles2 = (U2 + ST2 - 1) / ST2;
  const int n_ct = (g.Cin + 63) / 64;
  dim3 grid((unsigned)(g.N * tiles0 * tiles1 * tiles2), (unsigned)n_ct);
  const int rp = s2x_rows_pad(g.Cin);
  if (g.Cin <= 32)
    hipLaunchKernelGGL(conv_dgrad_s2_x3_kernel<2>, grid, dim3(SNT), 2 * SLDS1, ctx->stream, dy,
                       (const unsigned short*)img, dx, g, rp, tiles0, tiles1, tiles2, mask_y, mask_slope);
  else
    hipLaunchKernelGGL(conv_dgrad_s2_x3_kernel<4>, grid, dim3(SNT), 2 * SLDS1, ctx->stream, dy,
                       (const unsigned short*)img, dx, g, rp, tiles0, tiles1, tiles2, mask_y, mask_slope);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
