// The few-feature OUTPUT conv of a 2-D generator: 64 -> C_out <= 7, 3 x 3, stride 1,
// reflect 'same', fp32 out, no skip operand (spatial/gen_*: 64 -> 1 / 2;
// sup3rcc/gen_*_5x_1x_*: 64 -> 1 / 6 — at hi-res, e.g. 48 x 150 x 150 or 96 x 750 x
// 750 cells), gfx950 only.  bf16 plans read bf16 cells, BF16X3 plans fp32 cells
// (hi*hi + hi*lo + lo*hi).
//
// This layer is HBM-class: 1152 C_out MACs per 128-B (256-B) cell.  On the
// weights-stationary tile kernel (conv2d_ws_kernel<1>) it is nine taps x two
// k-steps of a 16-wide MFMA of which C_out columns are used — 36 MFMAs and 5 LDS
// fragment reads per 4 of them for 16 positions: 57 us at 48 x 150 x 150 (2.6 TB/s);
// BF16X3 plans ran it on the logical-axes tile kernel: 272 us.  Here the TAPS are
// columns of the matrix product:
//
//   P[cell][tap, co] = sum_ci x[cell][ci] w[tap][ci][co]      one [16 cells x 64] x
//                                                            [64 x 9 C_out] product
//   y[r][c][co] = bias + sum_tap P[(r, c) + tap - 1][tap, co]  a 9-term gather-add
//
// so a halo cell is multiplied ONCE (2 k-steps x ceil(9 C_out / 16) MFMAs per 16
// cells: 4 for C_out = 2) straight from the registers its global load filled — the
// A fragment of v_mfma_f32_16x16x32_bf16 (row = cell, 8 consecutive channels per
// lane) IS a coalesced 16-B load of a channels-last cell — and the filter (64 x 9
// C_out values) lives in registers for the whole launch.  Only the partial products
// go through LDS (a ring of four halo rows of 16 G cells x 9 C_out floats).
//
// A workgroup of G waves owns a strip of 16 G halo columns (16 G - 2 outputs) of one
// image and walks a segment of rows; a wave = 16 cells of each halo row; loads run
// four rows ahead in registers; one barrier per row.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 obf16x2 __attribute__((ext_vector_type(2)));
typedef float of32x2 __attribute__((ext_vector_type(2)));

constexpr int OUT_MAX_COUT = 7;
constexpr int OUT_SLOTS = 4;
constexpr int OUT_AHEAD = 4;          // halo rows in flight per wave

__device__ inline unsigned out_pk(float a, float b) {
  of32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, obf16x2));
}
__device__ inline float out_lo16(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float out_hi16(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// canonical fp32 w[tap 9][ci 64][co C_out] -> B fragments, one 16 B per lane:
// image[sel][ks][nf][lane] = 8 bf16: k = ks 32 + (lane >> 4) 8 + e, column nf 16 +
// (lane & 15) = tap C_out + co (zero beyond 9 C_out); sel 0 = bf16 rounding of w
// (hi), sel 1 = bf16 rounding of the residue (lo, BF16X3 plans only)
__global__ void pack_out_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (sel, ks, nf, lane, e)
  if (idx >= 2 * 2 * 4 * 64 * 8) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, nf = (idx >> 9) & 3, ks = (idx >> 11) & 1, sel = idx >> 12;
  const int k = ks * 32 + (lane >> 4) * 8 + e, col = nf * 16 + (lane & 15);
  float v = 0.f;
  if (col < 9 * cout) {
    const int tap = col / cout, co = col % cout;
    v = w[((size_t)tap * 64 + k) * cout + co];
  }
  const unsigned hi = out_pk(v, 0.f) & 0xFFFFu;
  const unsigned lo = out_pk(v - __uint_as_float(hi << 16), 0.f) & 0xFFFFu;
  out[idx] = (unsigned short)(sel ? lo : hi);
}

struct OutGeom {
  int N, H, W, Cout;
  int strips, segs, seg_rows;     // work items: image x 14-column strip x row segment
  int ncolp;                      // floats per cell in the LDS ring (9 C_out, odd)
  float slope;
};

constexpr int OUT_WAVES = 4;      // waves per workgroup (independent of each other)
constexpr int OUT_OCOLS = 14;     // output columns of a wave's 16-cell strip

// X3: fp32 cells (BF16X3 plan); else bf16 cells.  NF = ceil(9 C_out / 16).
// Every WAVE is on its own: it owns a strip of 16 halo columns (14 outputs) of one
// image and walks a segment of rows — its loads run OUT_AHEAD halo rows ahead in
// registers, its partial products go through a wave-private ring of four rows in
// LDS (LDS executes a wave's instructions in order: the 9-term gather-add reads
// what the same wave wrote, no barrier anywhere), so waves of different progress
// interleave freely on a SIMD.
template <bool X3, int NF>
__global__ __launch_bounds__(64 * OUT_WAVES) void conv2d_out_kernel(
    const void* __restrict__ xv, const uint4* __restrict__ wimg, const float* __restrict__ bias,
    float* __restrict__ y, OutGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, kq = lane >> 4;
  float* P = reinterpret_cast<float*>(smem) + (size_t)wave * OUT_SLOTS * 16 * g.ncolp;   // this wave's ring

  // ---- the filter: B fragments in registers for the whole launch
  bf16x8 wh[2][NF], wl[2][X3 ? NF : 1];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      wh[ks][nf] = __builtin_bit_cast(bf16x8, wimg[((0 * 2 + ks) * 4 + nf) * 64 + lane]);
      if constexpr (X3) wl[ks][nf] = __builtin_bit_cast(bf16x8, wimg[((1 * 2 + ks) * 4 + nf) * 64 + lane]);
    }
  // this lane's outputs of a row: o = lane (+ 64): column o / C_out, feature o % C_out
  const int nout = OUT_OCOLS * g.Cout;                  // <= 98
  const int o_c0 = lane / g.Cout, o_co0 = lane - o_c0 * g.Cout;
  const int o_c1 = (lane + 64) / g.Cout, o_co1 = (lane + 64) - o_c1 * g.Cout;
  const float b0 = bias ? bias[o_co0] : 0.f, b1 = bias ? bias[o_co1] : 0.f;

  const int items = g.N * g.strips * g.segs;
  const int n_wave = gridDim.x * OUT_WAVES;
  // (neighbouring strips — which share two halo columns — go to the waves of one
  // workgroup: the shared cells are L1 / L2 hits)
  for (int item = blockIdx.x * OUT_WAVES + wave; item < items; item += n_wave) {
    int q = item;
    const int strip = q % g.strips; q /= g.strips;
    const int seg = q % g.segs;
    const int im = q / g.segs;
    const int r0 = seg * g.seg_rows;
    const int rows = r0 + g.seg_rows <= g.H ? g.seg_rows : g.H - r0;      // output rows r0 .. r0 + rows - 1
    const int c0 = strip * OUT_OCOLS;                                      // output columns c0 .. c0 + 13
    // this lane's halo cell column (reflected, clamped: a clamped column beyond the
    // image's last output is computed and never read)
    int cc = s3_reflect(c0 - 1 + frow, g.W);
    cc = cc < 0 ? 0 : (cc > g.W - 1 ? g.W - 1 : cc);
    const size_t lane_off = (size_t)cc * 64 + kq * 8;
#define OUT_ROW_OFF(hr_)                                                                       \
    (((size_t)im * g.H + ({ int r_ = s3_reflect(r0 - 1 + (hr_), g.H);                         \
                            r_ < 0 ? 0 : (r_ > g.H - 1 ? g.H - 1 : r_); })) * g.W * 64 + lane_off)
    // ---- register ring of raw loads, OUT_AHEAD halo rows deep (named registers)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rc0, rc1, rc2, rc3, rd0, rd1, rd2, rd3;
#define OUT_ISSUE(R0, R1, R2, R3, hr_)                                                         \
    {                                                                                          \
      const size_t o_ = OUT_ROW_OFF(hr_);                                                      \
      if constexpr (X3) {                                                                      \
        const uint4* s_ = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(xv) + o_); \
        R0 = s_[0]; R1 = s_[1]; R2 = s_[8]; R3 = s_[9];                                        \
      } else {                                                                                 \
        const uint4* s_ = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(xv) + o_); \
        R0 = s_[0]; R1 = s_[4];                                                                \
      }                                                                                        \
    }
    const int hrows = rows + 2;
    OUT_ISSUE(ra0, ra1, ra2, ra3, 0)
    OUT_ISSUE(rb0, rb1, rb2, rb3, 1)
    OUT_ISSUE(rc0, rc1, rc2, rc3, 2 < hrows ? 2 : hrows - 1)
    OUT_ISSUE(rd0, rd1, rd2, rd3, 3 < hrows ? 3 : hrows - 1)

    // one halo row: partial products of the 16 cells -> ring slot hr & 3
    auto produce = [&](const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, int hr)
        __attribute__((always_inline)) {
      f32x4 acc[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (X3) {
          const uint4 a = ks ? q2 : q0, b = ks ? q3 : q1;
          const float v[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                              __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
          unsigned h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h[e] = out_pk(v[2 * e], v[2 * e + 1]);
            l[e] = out_pk(v[2 * e] - out_lo16(h[e]), v[2 * e + 1] - out_hi16(h[e]));
          }
          const bf16x8 xh = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
          const bf16x8 xl = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh[ks][nf], acc[nf], 0, 0, 0);
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl[ks][nf], acc[nf], 0, 0, 0);
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh[ks][nf], acc[nf], 0, 0, 0);
          }
        } else {
          const bf16x8 xf = __builtin_bit_cast(bf16x8, ks ? q1 : q0);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wh[ks][nf], acc[nf], 0, 0, 0);
        }
      }
      // D[cell 4 kq + e][column frow] of fragment nf -> P[ring][cell][column]
      // (only the 9 C_out live columns have room in the ring)
      float* dst = P + ((size_t)(hr & (OUT_SLOTS - 1)) * 16 + 4 * kq) * g.ncolp + frow;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        if (nf * 16 + frow < 9 * g.Cout) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e * g.ncolp + nf * 16] = acc[nf][e];
        }
    };
    // output row orow (halo rows orow, orow + 1, orow + 2 of the ring)
    auto emit = [&](int orow) __attribute__((always_inline)) {
      float* yrow = y + (((size_t)im * g.H + r0 + orow) * g.W + c0) * g.Cout;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int o = lane + 64 * half;
        const int c = half ? o_c1 : o_c0, co = half ? o_co1 : o_co0;
        if (o >= nout) continue;
        float s = half ? b1 : b0;
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          const float* pr = P + ((size_t)((orow + tb) & (OUT_SLOTS - 1)) * 16 + c) * g.ncolp + tb * 3 * g.Cout + co;
#pragma unroll
          for (int tc = 0; tc < 3; ++tc) s += pr[tc * g.ncolp + tc * g.Cout];
        }
        if (c0 + c < g.W) yrow[o] = s > 0.f ? s : g.slope * s;
      }
    };

    // ---- halo rows 0 .. rows + 1; output row hr - 2 is complete once row hr is in the ring
    for (int hr0 = 0; hr0 < hrows; hr0 += OUT_AHEAD) {
#define OUT_STEP(R0, R1, R2, R3, a_)                                                           \
      {                                                                                        \
        const int hr = hr0 + (a_);                                                             \
        if (hr < hrows) {                                                                      \
          produce(R0, R1, R2, R3, hr);                                                         \
          { const int nx_ = hr + OUT_AHEAD < hrows ? hr + OUT_AHEAD : hrows - 1;               \
            OUT_ISSUE(R0, R1, R2, R3, nx_) }                                                   \
          if (hr >= 2) emit(hr - 2);                                                           \
        }                                                                                      \
      }
      OUT_STEP(ra0, ra1, ra2, ra3, 0)
      OUT_STEP(rb0, rb1, rb2, rb3, 1)
      OUT_STEP(rc0, rc1, rc2, rc3, 2)
      OUT_STEP(rd0, rd1, rd2, rd3, 3)
#undef OUT_STEP
    }
#undef OUT_ISSUE
#undef OUT_ROW_OFF
  }
}

}  // namespace

// geometry: the few-feature output conv of kernels_conv2d_ws.hip's tail form
bool conv2d_out_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res) {
  if (s3_opt_on(S3O_NO_CONV2D_WS) || s3_opt_on(S3O_NO_CONV2D_OUT)) return false;
  if (has_res || g.res2 || g.w_cin || io.out_bf16) return false;
  if (g.Cout < 1 || g.Cout > OUT_MAX_COUT || !conv2d_ws_tail_geom_ok(g)) return false;
  // per IMAGE (the choice must not depend on the batch size: chunk by chunk == batched,
  // bit for bit): the row walk pays for itself from ~64 x 64 cells on
  if ((int64_t)g.D[0] * g.D[1] < 4096) return false;
  if (precision == S3_PREC_BF16) return io.in_bf16 != 0;
  if (precision == S3_PREC_BF16X3) return io.in_bf16 == 0;
  return false;
}

size_t conv2d_out_image_bytes(const ConvGeom&) { return (size_t)2 * 2 * 4 * 64 * 16; }

int launch_conv2d_out_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image) {
  hipLaunchKernelGGL(pack_out_kernel, dim3(2 * 2 * 4 * 64 * 8 / 256), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)image, g.Cout);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv2d_out(s3_ctx* ctx, const ConvGeom& g, int precision, const void* x, const void* image,
                      const float* bias, void* y) {
  OutGeom o;
  o.N = g.N; o.H = g.D[0]; o.W = g.D[1]; o.Cout = g.Cout;
  o.slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  const int nf = (9 * g.Cout + 15) / 16;
  o.ncolp = (9 * g.Cout) | 1;
  o.strips = (o.W + OUT_OCOLS - 1) / OUT_OCOLS;
  const size_t lds = (size_t)OUT_WAVES * OUT_SLOTS * 16 * o.ncolp * sizeof(float);
  // wave slots of the chip (8 waves per SIMD would need <= 64 VGPRs; the BF16X3 forms
  // with their hi / lo filter fragments take up to ~180: 2 - 5 waves per SIMD)
  const int64_t slots = (int64_t)ctx->num_cu * 4 * (precision == S3_PREC_BF16X3 ? 4 : 3);
  // row segments: the launch lasts as long as its busiest wave — rounds of items x
  // (rows + 2 halo rows + ~4 rows of pipeline fill); >= 8 rows per segment
  int segs = 1;
  {
    int64_t best_cost = -1;
    for (int cand = 1; cand <= o.H / 8 || cand == 1; ++cand) {
      const int sr = (o.H + cand - 1) / cand, sg = (o.H + sr - 1) / sr;
      const int64_t it = (int64_t)o.N * o.strips * sg;
      const int64_t cost = ((it + slots - 1) / slots) * (sr + 6);
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; segs = sg; }
    }
  }
  o.seg_rows = (o.H + segs - 1) / segs;
  o.segs = (o.H + o.seg_rows - 1) / o.seg_rows;
  const int64_t items = (int64_t)o.N * o.strips * o.segs;
  int64_t grid = (items + OUT_WAVES - 1) / OUT_WAVES;
  if (grid > slots / OUT_WAVES) grid = slots / OUT_WAVES;
  const bool x3 = precision == S3_PREC_BF16X3;
#define OUT_LAUNCH(X3_, NF_)                                                                                   \
  {                                                                                                            \
    static S3DeviceOnce once;                                                                                  \
    if (!once.done(ctx->device)) {                                                                             \
      std::lock_guard<std::mutex> lk(once.m);                                                                  \
      S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_out_kernel<X3_, NF_>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
      once.mark(ctx->device);                                                                                  \
    }                                                                                                          \
    hipLaunchKernelGGL((conv2d_out_kernel<X3_, NF_>), dim3((unsigned)grid), dim3(64 * OUT_WAVES), lds, ctx->stream, x, \
                       (const uint4*)image, bias, (float*)y, o);                                               \
  }
  switch (nf + (x3 ? 4 : 0)) {
    case 1: OUT_LAUNCH(false, 1) break;
    case 2: OUT_LAUNCH(false, 2) break;
    case 3: OUT_LAUNCH(false, 3) break;
    case 4: OUT_LAUNCH(false, 4) break;
    case 5: OUT_LAUNCH(true, 1) break;
    case 6: OUT_LAUNCH(true, 2) break;
    case 7: OUT_LAUNCH(true, 3) break;
    default: OUT_LAUNCH(true, 4) break;
  }
#undef OUT_LAUNCH
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
