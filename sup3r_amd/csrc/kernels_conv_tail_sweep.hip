// Plane-sweep form of the hi-res tail conv for C_out = 2 (Conv3D 8 -> 2, k 3,
// stride 1, reflect padding, bf16 in, fp32 out): the same banded MFMAs per
// output position, in the same order, as conv_tail_mfma_kernel<true> and
// conv_tail_slide_kernel (kernels_conv_tail_mfma.hip) — identical output bits
// — with the data movement rearranged around what bounds the op, the ~10 B per
// clock one CU can move (loads + stores; profiles/r06/tail_sweep.md: 9.2 B / clk
// in the sliding-window kernel, 10.7 here):
//
//   * the slide kernel keeps the three input planes of an output row in LDS
//     (3 + 2 in flight = 5 slots of 19 KB), which limits its column to 16 x 64
//     positions: 18 x 66 halo cells per 1 024 positions, and 5 columns of 64
//     over a series of 288 steps — every input cell crosses the L2 -> CU path
//     1.35 times at the C2 shape;
//   * here a plane is read from LDS ONCE, when it arrives, and feeds the three
//     output rows it belongs to (taps a = 0 / 1 / 2 of rows q / q - 1 / q - 2),
//     whose accumulators live in registers.  LDS then holds the plane being
//     read and the two in flight, so a plane may be 53 KB: 40 x 72 positions
//     (42 x 74 cells: 1.08 x) at the C2 shape, chosen per launch from the
//     divisors of the output extents, and a third of the LDS reads;
//   * no staging waves: the 8 waves issue their share of the next plane's
//     LDS-DMA pieces (64 cells each, in linear plane order — any row length)
//     before the MFMAs of the current one, stores go out before that DMA, so
//     the counted vmcnt in front of the row barrier covers exactly the newest
//     plane; the plane stream runs across the units of a workgroup without a
//     pipeline refill.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int SW_WAVES = 8;
constexpr int SW_NTH = SW_WAVES * 64;
constexpr int SW_NSLOT = 3;
constexpr int SW_MAXP = 7;                      // DMA pieces per wave and plane
constexpr int SW_LDS = 160 * 1024;

struct SweepShape {
  int S1, S2, G2, P2;          // positions per plane (s1, t), t groups of 8, cells per plane row
  int plane_cells, npieces, plane_bytes;
  int seg, segs0, tiles1, tiles2, n_units;
};

__device__ inline unsigned sw_pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float sw_act(float v, float slope) { return v > 0.f ? v : slope * v; }
__device__ inline float sw_affine2(float v, float sc, float sh) {
  float t = v * sc;
  asm volatile("" : "+v"(t));
  return t + sh;
}

template <int NSET>
__global__ __launch_bounds__(SW_NTH) void conv_tail_sweep_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, ConvGeom g, SweepShape sh,
    const float* __restrict__ aff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, kq = lane >> 4;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  auto clampi = [](int i, int d) { return i < 0 ? 0 : (i > d - 1 ? d - 1 : i); };

  // XCD-contiguous unit ranges (conv_tail_mfma_kernel)
  int u_first, u_step, u_end;
  {
    const int G = gridDim.x, b = blockIdx.x, xcd = b % 8;
    int before = 0;
    for (int q = 0; q < xcd; ++q) before += (G - q + 7) / 8;
    const int mine = (G - xcd + 7) / 8;
    u_first = (int)((long long)sh.n_units * before / G) + b / 8;
    u_step = mine;
    u_end = (int)((long long)sh.n_units * (before + mine) / G);
  }
  auto unit_org = [&](int u, int& n, int& r0, int& o1, int& o2) __attribute__((always_inline)) {
    int tr = u;
    o2 = (tr % sh.tiles2) * sh.S2; tr /= sh.tiles2;
    o1 = (tr % sh.tiles1) * sh.S1; tr /= sh.tiles1;
    r0 = (tr % sh.segs0) * sh.seg; tr /= sh.segs0;
    n = tr;
  };

  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;

  // ---- compute side: column group cg = (j * 8 + wave) * 16 + p of the plane
  // (plane row cg / G2, t base 8 (cg % G2)); k-group kq reads cell 4 s + kq
  unsigned lane_base[NSET];
  int row1[NSET], t8[NSET];
#pragma unroll
  for (int j = 0; j < NSET; ++j) {
    const int cg = (j * SW_WAVES + wave) * 16 + p;
    const bool ok = cg < sh.S1 * sh.G2;
    row1[j] = ok ? cg / sh.G2 : -1;
    t8[j] = ok ? (cg % sh.G2) * 8 : 0;
    lane_base[j] = ok ? (unsigned)((row1[j] * sh.P2 + t8[j] + kq) * 16) : 0u;
  }

  // ---- issue side: pieces wave, wave + 8, ... of the plane stream
  const int my_np = (sh.npieces - wave + SW_WAVES - 1) / SW_WAVES;
  int iu = u_first, iq = 0, irows = 0, in_ = 0, ir0 = 0, islot = 0;
  unsigned poff[SW_MAXP];
  auto issue_setup = [&]() __attribute__((always_inline)) {
    int o1, o2;
    unit_org(iu, in_, ir0, o1, o2);
    irows = ir0 + sh.seg <= g.O[0] ? sh.seg : g.O[0] - ir0;
#pragma unroll
    for (int i = 0; i < SW_MAXP; ++i) {
      int q = ((wave + SW_WAVES * i) << 6) + lane;
      q = q < sh.plane_cells ? q : sh.plane_cells - 1;     // (pad cells: any finite value)
      const int row = q / sh.P2, col = q - row * sh.P2;
      const int i1 = clampi(s3_reflect(o1 + row - g.lo[1], D1), D1);
      const int i2 = clampi(s3_reflect(o2 + col - g.lo[2], D2), D2);
      poff[i] = (unsigned)((i1 * D2 + i2) * 8);
    }
  };
  // next plane of the stream -> slot islot; false at the end of the stream
  auto issue_next = [&]() __attribute__((always_inline)) -> bool {
    if (iu >= u_end) return false;
    const int i0 = clampi(s3_reflect(ir0 + iq - g.lo[0], D0), D0);
    const unsigned short* pl = x + ((size_t)in_ * D0 + i0) * D1 * D2 * 8;
    char* dst = smem + islot * sh.plane_bytes + wave * 1024;
#pragma unroll
    for (int i = 0; i < SW_MAXP; ++i)
      if (i < my_np)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(pl + poff[i]),
            (__attribute__((address_space(3))) void*)(dst + i * (SW_WAVES * 1024)), 16, 0, 0);
    islot = islot == SW_NSLOT - 1 ? 0 : islot + 1;
    if (++iq == irows + 2) {
      iq = 0;
      iu += u_step;
      if (iu < u_end) issue_setup();
    }
    return true;
  };
  // everything but the newest plane's pieces of this wave has landed
  auto wait_but_newest = [&](bool newest_in_flight) __attribute__((always_inline)) {
    switch (newest_in_flight ? my_np : 0) {
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
#define SWEEP_BARRIER() asm volatile("s_barrier" ::: "memory")

  if (iu >= u_end) return;
  issue_setup();
  issue_next();
  const bool second = issue_next();
  // banded filter fragments, as in conv_tail_mfma_kernel<true>, built from an LDS
  // copy of the 432 filter values (slot 2, which the first DMA reaches only after the
  // barrier below) while planes 0 and 1 are in flight: 27 dependent trips to L2 per
  // lane otherwise
  bf16x8 wf[27];
  {
    float* wl = reinterpret_cast<float*>(smem + 2 * sh.plane_bytes);
    if (tid < 27 * 8 * 2) wl[tid] = w[tid];
    __syncthreads();
    const int delta = p >> 1, co = p & 1;
#pragma unroll
    for (int f = 0; f < 27; ++f) {
      const int ab = f / 3, s = f % 3;
      const int c = 4 * s + kq - delta;
      unsigned u[4] = {0u, 0u, 0u, 0u};
      if (c >= 0 && c <= 2) {
        const float* wp = wl + (ab * 3 + c) * 8 * 2 + co;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = sw_pk2(wp[(2 * e) * 2], wp[(2 * e + 1) * 2]);
      }
      uint4 uv = make_uint4(u[0], u[1], u[2], u[3]);
      wf[f] = __builtin_bit_cast(bf16x8, uv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  wait_but_newest(second);
  SWEEP_BARRIER();

  // rows q - 2 / q - 1 / q of the plane being read: R0 / R1 / R2
  f32x4 R0[NSET][2], R1[NSET][2], R2[NSET][2];
#pragma unroll
  for (int j = 0; j < NSET; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) R0[j][h] = R1[j][h] = R2[j][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the row completed by the previous plane, stored at the top of the next step
  bool pend = false;
  int pn = 0, po0 = 0, po1 = 0, po2 = 0;
  auto store_pending = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) {
      const int oo1 = po1 + row1[j], oo2 = po2 + t8[j] + 2 * kq;
      if (row1[j] >= 0 && oo1 < g.O[1] && oo2 < g.O[2]) {
        float* yp = y + ((((size_t)pn * g.O[0] + po0) * g.O[1] + oo1) * g.O[2] + oo2) * 2;
        const f32x4 t = R0[j][0] + R0[j][1];
        float v0 = sw_act(t[0], slope), v1 = sw_act(t[1], slope),
              v2 = sw_act(t[2], slope), v3 = sw_act(t[3], slope);
        if (aff) {
          v0 = sw_affine2(v0, aff[0], aff[2]); v1 = sw_affine2(v1, aff[1], aff[3]);
          v2 = sw_affine2(v2, aff[0], aff[2]); v3 = sw_affine2(v3, aff[1], aff[3]);
        }
        if (oo2 + 1 < g.O[2]) {
          __builtin_nontemporal_store((f32x4){v0, v1, v2, v3}, reinterpret_cast<f32x4*>(yp));
        } else {
          yp[0] = v0; yp[1] = v1;
        }
      }
    }
  };

  int cslot = 0;
  for (int u = u_first; u < u_end; u += u_step) {
    int n, r0, o1, o2;
    unit_org(u, n, r0, o1, o2);
    const int rows = r0 + sh.seg <= g.O[0] ? sh.seg : g.O[0] - r0;
    for (int q = 0; q < rows + 2; ++q) {
      if (pend) { store_pending(); pend = false; }
#pragma unroll
      for (int j = 0; j < NSET; ++j) {
        R0[j][0] = R1[j][0]; R0[j][1] = R1[j][1];
        R1[j][0] = R2[j][0]; R1[j][1] = R2[j][1];
        R2[j][0] = (f32x4){b0, b1, b0, b1};
        R2[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const bool newest = issue_next();
      const char* plane = smem + cslot * sh.plane_bytes;
#pragma unroll
      for (int j = 0; j < NSET; ++j) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const char* rowp = plane + lane_base[j] + b * (sh.P2 * 16);
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(rowp + s * 64);
            const int f0 = b * 3 + s;
            // f = 9 a + f0; the accumulator of tap f is f & 1 (conv_tail_slide_kernel)
            R2[j][f0 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f0], xf, R2[j][f0 & 1], 0, 0, 0);
            R1[j][(f0 + 1) & 1] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[9 + f0], xf, R1[j][(f0 + 1) & 1], 0, 0, 0);
            R0[j][f0 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[18 + f0], xf, R0[j][f0 & 1], 0, 0, 0);
          }
        }
      }
      if (q >= 2) { pend = true; pn = n; po0 = r0 + q - 2; po1 = o1; po2 = o2; }
      cslot = cslot == SW_NSLOT - 1 ? 0 : cslot + 1;
      wait_but_newest(newest);
      SWEEP_BARRIER();
    }
  }
  if (pend) store_pending();
#undef SWEEP_BARRIER
}

// plane shape and rows per unit for a launch: least time of the busiest
// workgroup, a plane costing ~1000 clk (barrier, exposed DMA latency) + what it
// moves through the CU at 12.5 B / clk (its DMA pieces + its row of fp32
// stores), or its MFMAs, whichever is longer
bool sweep_shape(const s3_ctx* ctx, const ConvGeom& g, SweepShape& best, int& nset) {
  static const int s1c[] = {4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96};
  static const int s2c[] = {16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 144, 160, 192, 256, 288};
  static const int sgc[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128};
  double best_cost = -1.0;
  // TAIL_SWEEP_SHAPE = S1 * 1e6 + S2 * 1e3 + rows per unit: one shape only (tests, sweeps)
  const long long forced = s3_opt_int(S3O_TAIL_SWEEP_SHAPE, 0);
  const int f1 = (int)(forced / 1000000), f2 = (int)(forced / 1000 % 1000), fseg = (int)(forced % 1000);
  for (int S1 : s1c) {
    if (forced ? S1 != f1 : (S1 > g.O[1] && S1 != s1c[0])) continue;
    for (int S2 : s2c) {
      if (forced ? S2 != f2 : (S2 > g.O[2] + 7 && S2 != s2c[0])) continue;
      SweepShape s;
      s.S1 = S1; s.S2 = S2; s.G2 = S2 / 8; s.P2 = S2 + 2;
      const int groups = S1 * s.G2;
      if (groups > SW_WAVES * 16 * 3) continue;
      const int ns = (groups + SW_WAVES * 16 - 1) / (SW_WAVES * 16);
      s.plane_cells = (S1 + 2) * s.P2;
      s.npieces = (s.plane_cells + 2 + 63) / 64;       // (+ 2: the band reads two cells past a row's end)
      if (s.npieces > SW_MAXP * SW_WAVES) continue;
      s.plane_bytes = s.npieces * 1024;
      if (SW_NSLOT * s.plane_bytes > SW_LDS) continue;
      s.tiles1 = (g.O[1] + S1 - 1) / S1;
      s.tiles2 = (g.O[2] + S2 - 1) / S2;
      // (fitted to HIP-event times of forced shapes at the C2 output, batch 8 and 32:
      // profiles/r06/tail_sweep.md)
      const double data_clk = 1000.0 + (s.plane_bytes + S1 * S2 * 8.0) / 12.5;
      const double mfma_clk = ns * 27 * 16.0 * 2 + 250.0;
      const double plane_clk = data_clk > mfma_clk ? data_clk : mfma_clk;
      for (int seg : sgc) {
        if (forced && seg != fseg) continue;
        int sg = seg;
        if (sg > g.O[0]) sg = g.O[0];
        s.seg = sg;
        s.segs0 = (g.O[0] + sg - 1) / sg;
        const long long units = (long long)g.N * s.segs0 * s.tiles1 * s.tiles2;
        if (units > 0x7fffffffLL) continue;
        s.n_units = (int)units;
        const long long rounds = (units + ctx->num_cu - 1) / ctx->num_cu;
        const double cost = (double)rounds * (sg + 2) * plane_clk + 2.0 * plane_clk;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; nset = ns; }
        if (sg == g.O[0]) break;
      }
    }
  }
  return best_cost >= 0;
}

}  // namespace

bool conv_tail_sweep_supported(const ConvGeom& g) {
  if (g.Cin != 8 || g.Cout != 2 || g.d2s != 1 || g.pad_mode != S3_PAD_REFLECT) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1) return false;
  // (element offsets within a plane are 32-bit)
  if ((long long)g.D[1] * g.D[2] * 8 > 0x7fffffffLL) return false;
  return g.O[0] >= 4 && g.O[2] >= 16;
}

int launch_conv_tail_sweep(s3_ctx* ctx, const ConvGeom& g, const void* x, const float* w,
                           const float* bias, float* y, const float* aff) {
  SweepShape sh;
  int nset = 1;
  if (!sweep_shape(ctx, g, sh, nset)) return S3_EINVAL;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_sweep_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_sweep_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_sweep_kernel<3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
    attr_set.mark(ctx->device);
  }
  int grid = ctx->num_cu;
  if (grid > sh.n_units) grid = sh.n_units;
  auto kern = nset == 1 ? conv_tail_sweep_kernel<1>
                        : (nset == 2 ? conv_tail_sweep_kernel<2> : conv_tail_sweep_kernel<3>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(SW_NTH), SW_NSLOT * sh.plane_bytes, ctx->stream,
                     (const unsigned short*)x, w, bias, y, g, sh, aff);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
