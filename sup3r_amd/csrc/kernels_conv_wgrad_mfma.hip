// Weight gradient of the 3x3x3 stride-1 trunk convs (C_in = 64) on MFMA:
//
//   dW[tap][ci][co] = sum_{n, p} Xpad[n, p + tap - 1][ci] * dPre[n, p][co]
//
// i.e. 27 GEMMs  (64 x P)(P x 64)  with K = P = all output positions.
// PERSISTENT workgroups (one per CU, 8 waves), each owning a 32-wide cout
// tile: wave w holds the 16x16 block (ci-block w/2, co-block w%2) of ALL 27
// taps — 27 f32x4 accumulators = 108 VGPRs — so the 27x64x32 gradient tile
// (42 % of the CU's register file) lives in registers while the workgroup
// streams position tiles (2 x 4 x 16) through LDS: the x halo
// (4 x 6 x 18 cells x 64 ch fp32, boundary handled at the load) and the dPre
// tile.  v_mfma_f32_16x16x4_f32 (exact fp32): A[i = ci][k = pos] and
// B[k = pos][j = co] are single-dword LDS reads, conflict-free with the
// (row & 1) << 4 channel swizzle.  dPre fragments are shared by the 27 taps.
// Each workgroup writes one partial; a fixed-order reduction makes the result
// deterministic (identical on every rank).
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WT0 = 2, WT1 = 4, WT2 = 16;
constexpr int WH0 = WT0 + 2, WH1 = WT1 + 2, WH2 = WT2 + 2;
constexpr int WHP = WH0 * WH1 * WH2;          // 432 halo cells
constexpr int WNP = WT0 * WT1 * WT2;          // 128 positions per tile
constexpr int WNT = 512;
constexpr int WCT = 32;                         // cout tile per workgroup
constexpr size_t WG_LDS = (size_t)WHP * 64 * 4 + (size_t)WNP * WCT * 4;   // 126,976 B

__global__ __launch_bounds__(512) void conv3_wgrad_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);              // [WHP][64] swizzled
  float* ds = xs + (size_t)WHP * 64;                       // [WNP][32] swizzled
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, kq = lane >> 4;
  const int cib = (wave >> 1) * 16, cob = (wave & 1) * 16;
  const int ct = blockIdx.y;                                // cout tile of 32
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int tr = tile;
    const int t2i = tr % tiles2; tr /= tiles2;
    const int t1i = tr % tiles1; tr /= tiles1;
    const int t0i = tr % tiles0; tr /= tiles0;
    const int n = tr;
    const int org0 = t0i * WT0, org1 = t1i * WT1, org2 = t2i * WT2;
    __syncthreads();   // previous tile fully consumed
    // ---- stage x halo: 432 cells x 16 float4
    for (int item = tid; item < WHP * 16; item += WNT) {
      const int hp = item >> 4, ch = item & 15;
      int h = hp;
      const int c2 = h % WH2; h /= WH2;
      const int c1 = h % WH1; h /= WH1;
      const int c0 = h;
      int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
      bool valid = true;
      if (g.pad_mode == S3_PAD_REFLECT) {
        i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
      } else {
        valid = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2;
      }
      // cells feeding only out-of-range outputs are multiplied by zero dPre
      i0 = i0 < 0 ? 0 : (i0 > D0 - 1 ? D0 - 1 : i0);
      i1 = i1 < 0 ? 0 : (i1 > D1 - 1 ? D1 - 1 : i1);
      i2 = i2 < 0 ? 0 : (i2 > D2 - 1 ? D2 - 1 : i2);
      float4 v = make_float4(0, 0, 0, 0);
      if (valid)
        v = *reinterpret_cast<const float4*>(
            x + ((((size_t)n * D0 + i0) * D1 + i1) * D2 + i2) * 64 + ch * 4);
      const int col = (ch * 4) ^ ((hp & 1) << 4);
      *reinterpret_cast<float4*>(xs + (size_t)hp * 64 + col) = v;
    }
    // ---- stage dPre tile: 128 positions x 8 float4 (zero outside / beyond C_out)
    for (int item = tid; item < WNP * (WCT / 4); item += WNT) {
      const int pl = item >> 3, ch = item & 7;
      const int row = pl / WT2, tt = pl % WT2;
      const int o0 = org0 + row / WT1, o1 = org1 + row % WT1, o2 = org2 + tt;
      const int co = ct * WCT + ch * 4;
      float4 v = make_float4(0, 0, 0, 0);
      if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && co < g.Cout)
        v = *reinterpret_cast<const float4*>(
            dy + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * g.Cout + co);
      const int col = (ch * 4) ^ ((pl & 1) << 4);
      *reinterpret_cast<float4*>(ds + (size_t)pl * WCT + col) = v;
    }
    __syncthreads();
    // ---- accumulate: 8 (s1,s2) rows x 4 k-steps of 4 consecutive t
    for (int row = 0; row < WT0 * WT1; ++row) {
      const int r0 = row / WT1, r1 = row % WT1;
#pragma unroll 1
      for (int tq = 0; tq < WT2 / 4; ++tq) {
        const int pl = row * WT2 + tq * 4 + kq;              // this lane's k
        const float bv = ds[(size_t)pl * WCT + ((cob + fi) ^ ((pl & 1) << 4))];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const int hp = ((r0 + a) * WH1 + (r1 + b)) * WH2 + tq * 4 + kq + c;
              const float av = xs[(size_t)hp * 64 + ((cib + fi) ^ ((hp & 1) << 4))];
              acc[(a * 3 + b) * 3 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  av, bv, acc[(a * 3 + b) * 3 + c], 0, 0, 0);
            }
      }
    }
  }
  // ---- partial[bid][tap][ci][co]: C/D map col = lane&15 (co), row = kq*4 + r (ci)
  float* out = partial + (size_t)blockIdx.x * 27 * 64 * g.Cout;
  const int co = ct * WCT + cob + fi;
  if (co < g.Cout) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[((size_t)t * 64 + cib + kq * 4 + r) * g.Cout + co] = acc[t][r];
  }
}

__global__ void wgrad_partial_reduce(const float* __restrict__ partial,
                                     int n_part, int64_t wsize,
                                     float* __restrict__ dw, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < wsize;
       i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < n_part; ++s) t += partial[(int64_t)s * wsize + i];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// many partials of a small filter (the hi-res few-channel convs: up to 2 048
// partials of a few thousand elements): one thread per element walking all of
// them is a 130 us chain on a handful of workgroups.  Segments of the partials
// are summed side by side (blockIdx.y), then the segment sums in order — a
// fixed order either way
__global__ void wgrad_partial_reduce_seg(const float* __restrict__ partial, int n_part, int64_t wsize,
                                         float* __restrict__ seg_out, int n_seg) {
  const int seg = blockIdx.y;
  const int s0 = (int)((int64_t)n_part * seg / n_seg), s1 = (int)((int64_t)n_part * (seg + 1) / n_seg);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < wsize;
       i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = s0; s < s1; ++s) t += partial[(int64_t)s * wsize + i];
    seg_out[(int64_t)seg * wsize + i] = t;
  }
}

int launch_wgrad_partial_reduce(s3_ctx* ctx, const float* partial, int n_part, int64_t wsize, float* dw,
                                int accumulate) {
  int rg = (int)((wsize + 255) / 256);
  if (rg > 2048) rg = 2048;
  // (only where it pays: enough partials, too few workgroups to hide the walk)
  const int n_seg = (n_part >= 128 && rg * 4 <= ctx->num_cu && !s3_opt_has(S3O_NO_SEG_REDUCE)) ? 16 : 1;
  if (n_seg > 1 && ensure_scratch(ctx, (size_t)n_seg * wsize * sizeof(float)) == S3_OK) {
    hipLaunchKernelGGL(wgrad_partial_reduce_seg, dim3(rg, n_seg), dim3(256), 0, ctx->stream, partial, n_part, wsize,
                       ctx->scratch, n_seg);
    hipLaunchKernelGGL(wgrad_partial_reduce, dim3(rg), dim3(256), 0, ctx->stream, (const float*)ctx->scratch, n_seg,
                       wsize, dw, accumulate);
  } else {
    hipLaunchKernelGGL(wgrad_partial_reduce, dim3(rg), dim3(256), 0, ctx->stream, partial, n_part, wsize, dw,
                       accumulate);
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int wgrad_grid(const s3_ctx* ctx, const ConvGeom& g, int* n_tiles_out, int* t0,
               int* t1, int* t2) {
  const int tiles0 = (g.O[0] + WT0 - 1) / WT0, tiles1 = (g.O[1] + WT1 - 1) / WT1,
            tiles2 = (g.O[2] + WT2 - 1) / WT2;
  const int n_tiles = g.N * tiles0 * tiles1 * tiles2;
  *n_tiles_out = n_tiles; *t0 = tiles0; *t1 = tiles1; *t2 = tiles2;
  const int n_ct = (g.Cout + WCT - 1) / WCT;
  int grid = ctx->num_cu / n_ct;
  if (grid < 1) grid = 1;
  if (grid > n_tiles) grid = n_tiles;
  return grid;
}

}  // namespace

bool conv_wgrad_mfma_supported(const ConvGeom& g) {
  if (g.Cin != 64 || g.Cout % 4 != 0 || g.Cout < 16) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] != 1 || g.O[d] != g.D[d]) return false;
  return g.D[2] >= 8;
}

size_t conv_wgrad_mfma_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  int nt, a, b, c;
  const int grid = wgrad_grid(ctx, g, &nt, &a, &b, &c);
  return (size_t)grid * 27 * 64 * g.Cout * sizeof(float);
}

int launch_conv_wgrad_mfma(s3_ctx* ctx, const ConvGeom& g, const float* x,
                           const float* dy, float* dw, float* partial,
                           size_t partial_bytes, int accumulate) {
  int n_tiles, tiles0, tiles1, tiles2;
  const int grid = wgrad_grid(ctx, g, &n_tiles, &tiles0, &tiles1, &tiles2);
  if (partial_bytes < conv_wgrad_mfma_partial_bytes(ctx, g))
    S3_FAIL(ctx, S3_EINVAL, "wgrad_mfma: partial buffer too small");
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_mfma_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS));
    attr_set.mark(ctx->device);
  }
  const int n_ct = (g.Cout + WCT - 1) / WCT;
  hipLaunchKernelGGL(conv3_wgrad_mfma_kernel, dim3(grid, n_ct), dim3(WNT), WG_LDS,
                     ctx->stream, x, dy, partial, g, tiles0, tiles1, tiles2, n_tiles);
  const int64_t wsize = (int64_t)27 * 64 * g.Cout;
  return launch_wgrad_partial_reduce(ctx, partial, grid, wsize, dw, accumulate);
}

// ===========================================================================
// General variant: any C_in (tiles of <= 64, padded to CIB blocks of 16), any C_out,
// stride 1 or 2, any low padding / valid extents — the discriminator convs
// (32->32 s2, 32->64, 64->64 s2, 64->128, 4->32) and the generator's small
// head / tail convs.  Same scheme as above: persistent workgroups of 8 waves,
// wave = one 16 x 16 (ci, co) block of ALL 27 taps in registers, exact fp32
// v_mfma_f32_16x16x4_f32 with single-dword LDS operand reads (A[i = ci][k =
// position] needs no transpose and tolerates any tap shift / stride).
//   CIB blocks of 16 input channels x NB blocks of 16 output channels per
//   workgroup; when CIB * NB < 8 (small C_out: the hi-res 4->32, 8->2 convs)
//   the spare waves split the positions of a tile PS = 8 / (CIB NB) ways and
//   write their own partials (the fixed-order reduction sums them).
//   stride 1: 2 x 4 x 16 positions per tile (halo 4 x 6 x 18 cells)
//   stride 2: 1 x 2 x 16 positions per tile (halo 3 x 5 x 33 cells)
namespace {

template <int CIB, int STR, int NB>
struct GenW {
  static constexpr int CIP = CIB * 16;              // padded input channels
  static constexpr int COT = NB * 16;               // output-channel tile
  static constexpr int PS = 8 / (CIB * NB);         // position splits
  static_assert(PS >= 1 && PS * CIB * NB == 8, "8 waves = CIB x NB x PS");
  static constexpr int T0 = STR == 1 ? 2 : 1, T1 = STR == 1 ? 4 : 2, T2 = 16;
  static constexpr int G0 = (T0 - 1) * STR + 3, G1 = (T1 - 1) * STR + 3, G2 = (T2 - 1) * STR + 3;
  static constexpr int HP = G0 * G1 * G2;
  static constexpr int NP = T0 * T1 * T2;
  static constexpr size_t LDS = (size_t)HP * CIP * 4 + (size_t)NP * COT * 4;
};

template <int CIB, int STR, int NB>
__global__ __launch_bounds__(512) void conv_wgrad_gen_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles0, int tiles1, int tiles2,
    int n_tiles) {
  using W = GenW<CIB, STR, NB>;
  constexpr int CIP = W::CIP, COT = W::COT, T1 = W::T1, T2 = W::T2, PS = W::PS;
  constexpr int G1 = W::G1, G2 = W::G2, HP = W::HP, NP = W::NP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);              // [HP][CIP] swizzled
  float* ds = xs + (size_t)HP * CIP;                       // [NP][COT] swizzled
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, kq = lane >> 4;
  const int cib = (wave % CIB) * 16, cob = ((wave / CIB) % NB) * 16;
  const int ps_id = wave / (CIB * NB);
  const int ct = blockIdx.y;
  const int ci0 = blockIdx.z * CIP;                 // input-channel tile
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  const int Cin = g.Cin, Cout = g.Cout;
  // swizzles: the two positions a 32-lane ds_read_b32 group touches must not
  // share banks (x: key on the cell's t index / stride; dy: position parity)
  constexpr int XSW = CIP >= 32 ? 16 : 0;
  constexpr int DSW = COT >= 32 ? 16 : 0;

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int tr = tile;
    const int t2i = tr % tiles2; tr /= tiles2;
    const int t1i = tr % tiles1; tr /= tiles1;
    const int t0i = tr % tiles0; tr /= tiles0;
    const int n = tr;
    const int org0 = t0i * W::T0, org1 = t1i * T1, org2 = t2i * T2;
    __syncthreads();   // previous tile fully consumed
    // ---- stage the x halo: cells x (CIP / 4) float4 (zero beyond C_in)
    constexpr int CH = CIP / 4;
    for (int item = tid; item < HP * CH; item += 512) {
      const int hp = item / CH, ch = item % CH;
      int h = hp;
      const int c2 = h % G2; h /= G2;
      const int c1 = h % G1; h /= G1;
      const int c0 = h;
      int i0 = org0 * STR + c0 - g.lo[0], i1 = org1 * STR + c1 - g.lo[1],
          i2 = org2 * STR + c2 - g.lo[2];
      bool valid = true;
      if (g.pad_mode == S3_PAD_REFLECT) {
        i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
      }
      // outside the tensor: zero padding, or cells feeding only masked outputs
      valid = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        const float* src = x + ((((size_t)n * D0 + i0) * D1 + i1) * D2 + i2) * Cin + ci0 + ch * 4;
        if ((Cin & 3) == 0 && ci0 + ch * 4 + 3 < Cin) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ci0 + ch * 4 + e < Cin) v[e] = src[e];
        }
      }
      const int col = (ch * 4) ^ (((c2 / STR) & 1) * XSW);
      *reinterpret_cast<float4*>(xs + (size_t)hp * CIP + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // ---- stage the dPre tile: NP positions x (COT / 4) float4
    constexpr int DH = COT / 4;
    for (int item = tid; item < NP * DH; item += 512) {
      const int pl = item / DH, ch = item % DH;
      const int row = pl / T2, tt = pl % T2;
      const int o0 = org0 + row / T1, o1 = org1 + row % T1, o2 = org2 + tt;
      const int co = ct * COT + ch * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && co < Cout) {
        const float* src = dy + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * Cout + co;
        if ((Cout & 3) == 0) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < Cout) v[e] = src[e];
        }
      }
      const int col = (ch * 4) ^ ((pl & 1) * DSW);
      *reinterpret_cast<float4*>(ds + (size_t)pl * COT + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    // ---- accumulate: (s1, s2) rows x k-steps of 4 consecutive t
#pragma unroll 1
    for (int ks = ps_id; ks < W::T0 * T1 * (T2 / 4); ks += PS) {
      const int row = ks / (T2 / 4), tq = ks % (T2 / 4);
      const int r0 = row / T1, r1 = row % T1;
      {
        const int pl = row * T2 + tq * 4 + kq;               // this lane's k
        const float bv = ds[(size_t)pl * COT + ((cob + fi) ^ ((pl & 1) * DSW))];
        const int tcell = (tq * 4 + kq) * STR;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const int hp = ((r0 * STR + a) * G1 + (r1 * STR + b)) * G2 + tcell + c;
              const float av = xs[(size_t)hp * CIP + ((cib + fi) ^ ((((tcell + c) / STR) & 1) * XSW))];
              acc[(a * 3 + b) * 3 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  av, bv, acc[(a * 3 + b) * 3 + c], 0, 0, 0);
            }
      }
    }
  }
  // ---- partial[bid][tap][ci][co]: C/D map col = lane&15 (co), row = kq*4 + r (ci)
  float* out = partial + ((size_t)blockIdx.x * PS + ps_id) * 27 * Cin * Cout;
  const int co = ct * COT + cob + fi;
  if (co < Cout) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + cib + kq * 4 + r;
        if (ci < Cin) out[((size_t)t * Cin + ci) * Cout + co] = acc[t][r];
      }
  }
}

template <int CIB, int STR, int NB>
int wgrad_gen_grid(const s3_ctx* ctx, const ConvGeom& g, int* n_tiles, int* t0, int* t1, int* t2) {
  using W = GenW<CIB, STR, NB>;
  *t0 = (g.O[0] + W::T0 - 1) / W::T0; *t1 = (g.O[1] + W::T1 - 1) / W::T1;
  *t2 = (g.O[2] + W::T2 - 1) / W::T2;
  *n_tiles = g.N * *t0 * *t1 * *t2;
  const int n_ct = (g.Cout + W::COT - 1) / W::COT;
  const int n_cit = (g.Cin + W::CIP - 1) / W::CIP;
  int grid = ctx->num_cu / (n_ct * n_cit);
  if (grid < 1) grid = 1;
  if (grid > *n_tiles) grid = *n_tiles;
  return grid;
}

template <int CIB, int STR, int NB>
int wgrad_gen_launch(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                     float* dw, float* partial, size_t partial_bytes, int accumulate) {
  using W = GenW<CIB, STR, NB>;
  int n_tiles, t0, t1, t2;
  const int grid = wgrad_gen_grid<CIB, STR, NB>(ctx, g, &n_tiles, &t0, &t1, &t2);
  const size_t need = (size_t)grid * W::PS * 27 * g.Cin * g.Cout * sizeof(float);
  if (partial_bytes < need) S3_FAIL(ctx, S3_EINVAL, "wgrad_gen: partial buffer too small");
  auto kern = conv_wgrad_gen_kernel<CIB, STR, NB>;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS));
    attr_set.mark(ctx->device);
  }
  const int n_ct = (g.Cout + W::COT - 1) / W::COT;
  const int n_cit = (g.Cin + W::CIP - 1) / W::CIP;
  hipLaunchKernelGGL(kern, dim3(grid, n_ct, n_cit), dim3(512), W::LDS, ctx->stream, x, dy,
                     partial, g, t0, t1, t2, n_tiles);
  const int64_t wsize = (int64_t)27 * g.Cin * g.Cout;
  return launch_wgrad_partial_reduce(ctx, partial, grid * W::PS, wsize, dw, accumulate);
}

int wgrad_gen_cib(const ConvGeom& g) { return g.Cin <= 16 ? 1 : (g.Cin <= 32 ? 2 : 4); }   // C_in > 64: tiles of 64

}  // namespace

bool conv_wgrad_gen_supported(const ConvGeom& g) {
  if (s3_opt_has(S3O_NO_GCONV)) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != g.s[0] || (g.s[d] != 1 && g.s[d] != 2)) return false;
  return g.O[2] >= 2;
}

// co blocks per workgroup: as many as the conv has, rounded to a power of two
int wgrad_gen_nb(const ConvGeom& g) {
  const int cap = 8 / wgrad_gen_cib(g);
  const int want = (g.Cout + 15) / 16;
  int nb = 1;
  while (nb < want && nb < cap) nb *= 2;
  return nb;
}

size_t conv_wgrad_gen_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  // one partial per (workgroup, position split); the grid never exceeds the
  // CU count
  const int ps = 8 / (wgrad_gen_cib(g) * wgrad_gen_nb(g));
  return (size_t)ctx->num_cu * ps * 27 * g.Cin * g.Cout * sizeof(float);
}

int launch_conv_wgrad_gen(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                          float* dw, float* partial, size_t partial_bytes, int accumulate) {
  const int cib = wgrad_gen_cib(g), nb = wgrad_gen_nb(g);
  const bool s2 = g.s[0] == 2;
#define S3_WG(C, B)                                                                          \
  if (cib == C && nb == B) {                                                                 \
    if (s2) return wgrad_gen_launch<C, 2, B>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate); \
    return wgrad_gen_launch<C, 1, B>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);  \
  }
  S3_WG(1, 1) S3_WG(1, 2) S3_WG(1, 4) S3_WG(1, 8)
  S3_WG(2, 1) S3_WG(2, 2) S3_WG(2, 4)
  S3_WG(4, 1) S3_WG(4, 2)
#undef S3_WG
  S3_FAIL(ctx, S3_EINVAL, "wgrad_gen: no kernel for this channel split");
}
