// Hi-res tail conv of the generator (Conv3D 8 -> 2, k 3, stride 1, fp32 in /
// fp32 out) for S3_PREC_BF16X3 plans — round 4.
//
// A BF16X3 plan keeps fp32 activations and computes every product as
// hi*hi + hi*lo + lo*hi with hi = bf16(v), lo = bf16(v - hi).
// Through round 3 the tail conv of such a plan ran on the direct fp32 kernel
// (conv_small_kernel): 4.4 ms of the 52.9 ms C2 forward at 32 chunks, the
// largest single op after the trunk, against 0.34 ms for the bf16 plan's
// banded MFMA tail (kernels_conv_tail_mfma.hip).
//
// This kernel is the banded formulation of conv_tail_mfma_kernel<true> with
// both halves of the input staged: MFMA row i = (delta = i >> 1, co = i & 1) is
// output position base + delta, column j a base position, K = 4 cells x 8
// channels; per (a, b) filter row three k-steps cover the 12 cells after the
// base.  One 4 x 8 x 32 tile per workgroup (two workgroups per CU overlap one
// tile's staging with the other's MFMAs): the 6 x 10 x 34 halo is read as fp32
// (32 B per cell), split in registers and written as two 16-B bf16 cells (hi
// plane, lo plane); the filter lives in LDS as 54 (tap, co) entries x {hi, lo}
// from which every lane picks the entry of ITS (delta, co, k-group) — the
// banded fragments are never materialised (27 x 2 fragments would be 216
// VGPRs).  Three MFMAs per fragment, small terms first, as everywhere in X3.
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
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  hi = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
  lo = make_uint4(pk2(a.x - bf_lo(hi.x), a.y - bf_hi(hi.x)), pk2(a.z - bf_lo(hi.y), a.w - bf_hi(hi.y)),
                  pk2(b.x - bf_lo(hi.z), b.y - bf_hi(hi.z)), pk2(b.z - bf_lo(hi.w), b.w - bf_hi(hi.w)));
}
__device__ __forceinline__ float act_sel(float v, float slope) { return v > 0.f ? v : slope * v; }

constexpr int XT0 = 4, XT1 = 8, XT2 = 32;
constexpr int XH0 = XT0 + 2, XH1 = XT1 + 2, XH2 = XT2 + 2;
constexpr int XHP = XH0 * XH1 * XH2;            // 2040 halo cells
constexpr int XCELLS = 2048;                    // + pad read against zero filter entries
constexpr int XHALF = XCELLS * 16;              // one bf16 plane: 32,768 B
constexpr int XTAB = 2 * XHALF;                 // filter table: (tap, co) x {hi, lo} x 16 B
constexpr int XLDS = XTAB + 54 * 32;            // 67,264 B -> two workgroups per CU
constexpr int XNT = 512;                        // 8 waves, one 4 x 32 position set each
constexpr int XPER = (XHP + XNT - 1) / XNT;     // 4 cells staged per thread

__global__ __launch_bounds__(XNT, 2) void conv_tail_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, ConvGeom g, int tiles0, int tiles1, int tiles2, int n_tiles, int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, kq = lane >> 4;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  // block b runs on XCD b % 8: every XCD walks its own contiguous tile range
  const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (tile >= n_tiles) return;
  int n, org0, org1, org2;
  {
    int tr = tile;
    org2 = (tr % tiles2) * XT2; tr /= tiles2;
    org1 = (tr % tiles1) * XT1; tr /= tiles1;
    org0 = (tr % tiles0) * XT0; tr /= tiles0;
    n = tr;
  }

  // ---- stage: fp32 halo -> hi plane / lo plane
  {
    const float* xn = x + (size_t)n * D0 * D1 * D2 * 8;
    const bool refl = g.pad_mode == S3_PAD_REFLECT;
    auto clampi = [](int i, int d) { return i < 0 ? 0 : (i > d - 1 ? d - 1 : i); };
    float4 va[XPER], vb[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
      const int hp = tid + k * XNT;
      va[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      vb[k] = va[k];
      if (hp < XHP) {
        const int c2 = hp % XH2, row = hp / XH2;
        const int c1 = row % XH1, c0 = row / XH1;
        int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
        bool ok = true;
        if (refl) {
          // (ragged tiles: clamped addresses stay legal; results are masked at the store)
          i0 = clampi(s3_reflect(i0, D0), D0);
          i1 = clampi(s3_reflect(i1, D1), D1);
          i2 = clampi(s3_reflect(i2, D2), D2);
        } else {
          ok = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2;
        }
        if (ok) {
          const float4* src = reinterpret_cast<const float4*>(xn + (((size_t)i0 * D1 + i1) * D2 + i2) * 8);
          va[k] = src[0];
          vb[k] = src[1];
        }
      }
    }
    // filter table: entry (tap, co) = w[tap][ci 0..7][co] as hi | lo
    if (tid < 54) {
      const int tap = tid >> 1, co = tid & 1;
      const float* wp = w + (size_t)tap * 16 + co;
      const float4 a = make_float4(wp[0], wp[2], wp[4], wp[6]);
      const float4 b = make_float4(wp[8], wp[10], wp[12], wp[14]);
      uint4 hi, lo;
      split8(a, b, hi, lo);
      *reinterpret_cast<uint4*>(smem + XTAB + tid * 32) = hi;
      *reinterpret_cast<uint4*>(smem + XTAB + tid * 32 + 16) = lo;
    } else if (tid >= 64 && tid < 64 + 2 * (XCELLS - XHP)) {
      // pad cells behind the halo are read against zero filter entries: finite
      const int q = tid - 64;
      *reinterpret_cast<uint4*>(smem + (q & 1) * XHALF + (XHP + (q >> 1)) * 16) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
      const int hp = tid + k * XNT;
      if (hp < XHP) {
        uint4 hi, lo;
        split8(va[k], vb[k], hi, lo);
        *reinterpret_cast<uint4*>(smem + hp * 16) = hi;
        *reinterpret_cast<uint4*>(smem + XHALF + hp * 16) = lo;
      }
    }
  }
  __syncthreads();

  // ---- compute: wave q owns s0 row q >> 1, s1 rows 4 (q & 1) .. + 3, all 32 t.
  // Column j = lane & 15: s1 row j >> 2, base t = (j & 3) * 8; k-group kq reads
  // cell e = 4 s + kq after the base; row (delta, co) wants filter column
  // c = e - delta of the (a, b) filter row, zero outside 0..2.
  const int r0 = wave >> 1, r1 = 4 * (wave & 1) + (p >> 2);
  const unsigned base = (unsigned)(((r0 * XH1 + r1) * XH2 + (p & 3) * 8 + kq) * 16);
  const int delta = p >> 1, co = p & 1;
  unsigned aoff[3];
  bool aval[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int c = 4 * s + kq - delta;
    aval[s] = c >= 0 && c <= 2;
    aoff[s] = (unsigned)(XTAB + ((aval[s] ? c : 0) * 2 + co) * 32);
  }
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
  f32x4 acc0 = (f32x4){b0, b1, b0, b1}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 27; ++f) {
    const int ab = f / 3, s = f % 3;
    const unsigned off = (unsigned)((((ab / 3) * XH1 + ab % 3) * XH2 + 4 * s) * 16);
    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(smem + base + off);
    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(smem + XHALF + base + off);
    uint4 uh = *reinterpret_cast<const uint4*>(smem + aoff[s] + ab * 192);
    uint4 ul = *reinterpret_cast<const uint4*>(smem + aoff[s] + ab * 192 + 16);
    if (!aval[s]) { uh = make_uint4(0, 0, 0, 0); ul = uh; }
    const bf16x8 ah = __builtin_bit_cast(bf16x8, uh), al = __builtin_bit_cast(bf16x8, ul);
    if (f & 1) {
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc0, 0, 0, 0);
    }
  }
  // lane (j, kq) holds positions base t + 2 kq, + 1 x channels 0, 1: one float4
  const int o0 = org0 + r0, o1 = org1 + r1, o2 = org2 + (p & 3) * 8 + 2 * kq;
  if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2]) {
    const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
    float* yp = y + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * 2;
    const f32x4 t = acc0 + acc1;
    const float v0 = act_sel(t[0], slope), v1 = act_sel(t[1], slope),
                v2 = act_sel(t[2], slope), v3 = act_sel(t[3], slope);
    if (o2 + 1 < g.O[2]) {
      *reinterpret_cast<f32x4*>(yp) = (f32x4){v0, v1, v2, v3};
    } else {
      yp[0] = v0; yp[1] = v1;
    }
  }
}

}  // namespace

bool conv_tail_x3_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16X3 || s3_opt_has(S3O_NO_TAIL_X3)) return false;
  if (g.Cin != 8 || g.Cout != 2 || g.d2s != 1 || g.in_rep > 1) return false;
  if (g.pad_mode != S3_PAD_REFLECT && g.pad_mode != S3_PAD_ZERO) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1) return false;
  // tiles of 32 along t would be mostly masked on short series
  return g.O[2] >= 16;
}

int launch_conv_tail_x3(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* w,
                        const float* bias, float* y) {
  const int tiles0 = (g.O[0] + XT0 - 1) / XT0, tiles1 = (g.O[1] + XT1 - 1) / XT1,
            tiles2 = (g.O[2] + XT2 - 1) / XT2;
  const int n_tiles = g.N * tiles0 * tiles1 * tiles2;
  if (n_tiles <= 0) return S3_OK;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_x3_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, XLDS));
    attr_set.mark(ctx->device);
  }
  const int per_xcd = (n_tiles + 7) / 8;
  hipLaunchKernelGGL(conv_tail_x3_kernel, dim3(8 * per_xcd), dim3(XNT), XLDS, ctx->stream, x, w, bias, y, g,
                     tiles0, tiles1, tiles2, n_tiles, per_xcd);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
