// Forward conv with 32 input channels on bf16 MFMA with an LDS halo — the
// valid / zero-padded stride-1 discriminator layer 32 -> 64 (S3_PREC_BF16
// plans; fp32 or bf16 activations on either side).
//
// The gather kernel re-reads every input cell 27 times through L1 (0.94 ms at
// C2 batch 8, 170 TFLOP/s).  Here a workgroup stages the (4+2) x (8+2) x (16+2)
// input halo of its 4 x 8 x 16 output tile once into LDS (fp32 -> bf16, 64-B
// cells, 16-B chunks XOR-swizzled by (t >> 1) & 3, the layout of
// conv_dgrad_c2_kernel); every tap reads its shifted window as the MFMA B
// operand (K = 32 = one k-step) and the tap's filter rows (A operand, up to four
// 16-row fragments) come from a packed bf16 image that stays L1-resident.  Four
// waves, eight (s1, s2) rows each; bias + activation in the epilogue; lane
// (t, kg) stores 4 consecutive output channels per fragment.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int HT0 = 4, HT1 = 8, HT2 = 16;
constexpr int HH0 = HT0 + 2, HH1 = HT1 + 2, HH2 = HT2 + 2;
constexpr int HHP = HH0 * HH1 * HH2;         // 1080 halo cells
constexpr int HNW = 4;
constexpr int HNT = HNW * 64;
constexpr int HLDS = HHP * 64;               // 69,120 B

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

// fp32 w[tap][32][cout] -> bf16 img[tap][rows_pad][32]
__global__ void halo32_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                   int cout, int rows_pad) {
  const int total = 27 * rows_pad * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ci = idx & 31, row = (idx >> 5) % rows_pad, tp = idx / (32 * rows_pad);
    const float v = row < cout ? w[((size_t)tp * 32 + ci) * cout + row] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

template <int NF>
__global__ __launch_bounds__(HNT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_halo32_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ img,
    const float* __restrict__ bias, float* __restrict__ y, ConvGeom g, int rows_pad,
    int tiles0, int tiles1, int tiles2, int in16, int out16) {
  extern __shared__ __attribute__((aligned(16))) char halo[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int ct = blockIdx.y;
  int tr = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int org0 = t0i * HT0, org1 = t1i * HT1, org2 = t2i * HT2;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  // ---- stage the input halo: cell (c0, c1, c2) = x[org + c - lo], zero outside
  // (bf16 cells: one 16-B chunk per item, no convert)
  if (in16) {
    const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
    for (int base = tid; base < HHP * 4; base += HNT * 3) {
      uint4 v[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int item = base + u * HNT;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (item < HHP * 4) {
          const int hp = item >> 2, ch = item & 3;
          int h = hp;
          const int c2 = h % HH2; h /= HH2;
          const int c1 = h % HH1; h /= HH1;
          const int c0 = h;
          const int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
          if (i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2)
            v[u] = *reinterpret_cast<const uint4*>(
                x16 + ((((size_t)n * D0 + i0) * D1 + i1) * D2 + i2) * 32 + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int item = base + u * HNT;
        if (item < HHP * 4) {
          const int hp = item >> 2, ch = item & 3;
          const int key = ((hp % HH2) >> 1) & 3;
          *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) = v[u];
        }
      }
    }
  } else
  for (int base = tid; base < HHP * 4; base += HNT * 3) {
    float4 va[3], vb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * HNT;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      if (item < HHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        int h = hp;
        const int c2 = h % HH2; h /= HH2;
        const int c1 = h % HH1; h /= HH1;
        const int c0 = h;
        const int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
        if (i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2) {
          const float* src = x + ((((size_t)n * D0 + i0) * D1 + i1) * D2 + i2) * 32 + ch * 8;
          va[u] = *reinterpret_cast<const float4*>(src);
          vb[u] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * HNT;
      if (item < HHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        const int key = ((hp % HH2) >> 1) & 3;
        *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) =
            make_uint4(pk2(va[u].x, va[u].y), pk2(va[u].z, va[u].w), pk2(vb[u].x, vb[u].y),
                       pk2(vb[u].z, vb[u].w));
      }
    }
  }
  __syncthreads();

  int off_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) off_c[c] = (j + c) * 64 + ((kg ^ (((j + c) >> 1) & 3)) << 4);
  f32x4 acc[8][NF];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned short* wrow = img + ((size_t)ct * 64 + j) * 32 + kg * 8;
  // the next tap's filter fragments (L2-resident image, 110 KB: it does not fit
  // L1) are fetched under this tap's 32 MFMAs — loading them at the top of
  // their own tap left the waves in front of an L2 round trip 27 times per tile
  bf16x8 anext[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
    anext[nf] = *reinterpret_cast<const bf16x8*>(wrow + ((size_t)nf * 16) * 32);
#pragma unroll 1
  for (int tp = 0; tp < 27; ++tp) {
    const int a = tp / 9, b = (tp / 3) % 3, c = tp % 3;
    bf16x8 afr[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) afr[nf] = anext[nf];
    if (tp + 1 < 27) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        anext[nf] = *reinterpret_cast<const bf16x8*>(wrow + ((size_t)(tp + 1) * rows_pad + nf * 16) * 32);
    }
    const char* hb = halo + (((wave + a) * HH1 + b) * HH2) * 64 + off_c[c];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(hb + m * HH2 * 64);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[nf], bfr, acc[m][nf], 0, 0, 0);
    }
  }

  // ---- C/D: col = lane & 15 (t), row = 4 kg + r (channel 16 nf + 4 kg + r)
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  const int o0 = org0 + wave, o2 = org2 + j;
  const int R = g.Cout;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int ch = ct * 64 + nf * 16 + kg * 4;
    if (ch >= R) continue;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (bias && ch + r < R) ? bias[ch + r] : 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int o1 = org1 + m;
      if (o0 >= g.O[0] || o1 >= g.O[1] || o2 >= g.O[2]) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[m][nf][r] + bv[r];
        v[r] = v[r] > 0.f ? v[r] : slope * v[r];
      }
      const size_t oi = ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * R + ch;
      if (out16)
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + oi) =
            make_uint2(pk2(v[0], v[1]), pk2(v[2], v[3]));
      else
        *reinterpret_cast<float4*>(y + oi) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

int halo32_rows_pad(int cout) { return (cout + 63) / 64 * 64; }

}  // namespace

bool conv_halo32_supported(const s3_ctx* ctx, const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_HALO32)) return false;
  if (g.Cin != 32 || g.Cout % 4 != 0 || g.Cout < 16 || g.d2s != 1) return false;
  if (g.pad_mode == S3_PAD_REFLECT) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] < 0 || g.lo[d] > 2) return false;
  // enough tiles to fill the chip (SUP3R_AMD_HALO32_MIN_TILES overrides, for tests)
  const int64_t min_tiles = s3_opt_has(S3O_HALO32_MIN_TILES) ? s3_opt_int(S3O_HALO32_MIN_TILES, 0)
                                                                 : ctx->num_cu;
  return g.O[2] >= 8 &&
         (int64_t)g.N * ((g.O[0] + HT0 - 1) / HT0) * ((g.O[1] + HT1 - 1) / HT1) *
                 ((g.O[2] + HT2 - 1) / HT2) >= min_tiles;
}

size_t conv_halo32_packed_bytes(const ConvGeom& g) {
  return (size_t)27 * halo32_rows_pad(g.Cout) * 32 * 2;
}

int launch_conv_halo32_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  const int rp = halo32_rows_pad(g.Cout);
  hipLaunchKernelGGL(halo32_pack_kernel, dim3((27 * rp * 32 + 255) / 256), dim3(256), 0, ctx->stream,
                     w, (unsigned short*)img, g.Cout, rp);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_halo32_fwd(s3_ctx* ctx, const ConvGeom& g, const void* xv, const void* img,
                           const float* bias, void* yv, int in_bf16, int out_bf16) {
  const float* x = (const float*)xv;
  float* y = (float*)yv;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo32_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, HLDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo32_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, HLDS));
    attr_set.mark(ctx->device);
  }
  const int tiles0 = (g.O[0] + HT0 - 1) / HT0, tiles1 = (g.O[1] + HT1 - 1) / HT1,
            tiles2 = (g.O[2] + HT2 - 1) / HT2;
  const int n_ct = (g.Cout + 63) / 64;
  const int rp = halo32_rows_pad(g.Cout);
  dim3 grid((unsigned)(g.N * tiles0 * tiles1 * tiles2), (unsigned)n_ct);
  if (g.Cout <= 32)
    hipLaunchKernelGGL(conv_halo32_kernel<2>, grid, dim3(HNT), HLDS, ctx->stream, x,
                       (const unsigned short*)img, bias, y, g, rp, tiles0, tiles1, tiles2, in_bf16, out_bf16);
  else
    hipLaunchKernelGGL(conv_halo32_kernel<4>, grid, dim3(HNT), HLDS, ctx->stream, x,
                       (const unsigned short*)img, bias, y, g, rp, tiles0, tiles1, tiles2, in_bf16, out_bf16);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
