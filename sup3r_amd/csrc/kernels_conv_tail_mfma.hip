// Hi-res tail conv of the generator (Conv3D 8 -> C_out <= 16, k 3, stride 1,
// bf16 input from the depth-to-space store, fp32 output = the model output) on
// the matrix cores.  K2 "8 -> 2 @ (80,80,288)" of SURVEY.md §8: 0.8 GMAC and
// 44 MB per sample, AI ~ 36 FLOP/B — an HBM-class op that the direct kernel ran
// at ~1 TB/s because 27 x 8 x 2 scalar FMAs per position with 13x re-read
// of every cell through L1 is VALU/L1-bound.
//
// With C_in = 8 a halo cell IS one 16-B MFMA operand chunk (8 bf16), so the
// im2col is free: K = 27 taps x 8 channels is walked in 7 k-steps of 32 = 4
// taps x 8 ci, and lane (position p, k-group kq) of a B fragment simply reads
// the cell of position p shifted by tap 4s + kq — one ds_read_b128 at
// (per-lane base) + (per-lane tap offset).  The A operand is the filter,
// rows = output channels (zero-padded to 16), 7 fragments held in 28 VGPRs
// for the whole workgroup.  D[co][pos]: lane (pos, kq) owns channels
// kq*4 .. kq*4+3 of one position, so C_out <= 4 stores one float2/float4 per
// position, contiguous across the 16 lanes of a fragment.
//
// Persistent workgroups (one per CU) walk 4 x 8 x 64-position tiles with a
// DOUBLE-BUFFERED halo (2 x 62 KB of LDS) and two kinds of waves: 8 compute
// waves (ds_read_b128 + MFMA + stores — they never execute a load, so they
// never wait on vmcnt, and their stores are fire-and-forget) and 4 staging
// waves that bring in the halo of tile i+1 by LDS-DMA (global_load_lds_dwordx4:
// lane l of a wave-instruction fills LDS cell hp0 + l from its own, reflect-
// resolved global address — no VGPR staging) while tile i is on the matrix
// cores.  One raw s_barrier per tile hands the buffers over.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int T0 = 4, T1 = 8, T2 = 64;
constexpr int H0 = T0 + 2, H1 = T1 + 2, H2 = T2 + 2;
constexpr int HP = H0 * H1 * H2;                 // 3960 cells
constexpr int NCW = 8;                           // compute (MFMA + store) waves
constexpr int NDW = 4;                           // staging (LDS-DMA) waves
constexpr int NTH = (NCW + NDW) * 64;            // 768
constexpr int NDMA = (HP + 63) / 64;             // 62 wave-instructions per halo
constexpr int BUF_BYTES = NDMA * 64 * 16;        // 63,488 (tail lanes land in the pad)
constexpr int GROUPS = T0 * T1 * (T2 / 16);      // 128 fragments of 16 positions
constexpr int KS = 7;                            // ceil(27 / 4) k-steps

__device__ inline unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float act_sel(float v, float slope) { return v > 0.f ? v : slope * v; }
// (x * scale) + shift with two roundings, like numpy's un-normalisation
// (s3_chunk_epilogue does the same): the windowed forward of the C3 executor
__device__ inline float affine2(float v, float sc, float sh) {
  float t = v * sc;
  asm volatile("" : "+v"(t));
  return t + sh;
}

// BAND (C_out == 2 only): the 16 MFMA rows are 8 consecutive output positions
// x 2 channels instead of 2 channels + 14 rows of padding — see the compute
// loop.
template <bool BAND>
__global__ __launch_bounds__(NTH) void conv_tail_mfma_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, ConvGeom g,
    int tiles0, int tiles1, int tiles2, int n_tiles, const float* __restrict__ aff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, kq = lane >> 4;
  const int Cout = g.Cout;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  auto tile_org = [&](int tile, int& n, int& o0, int& o1, int& o2) __attribute__((always_inline)) {
    int tr = tile;
    o2 = (tr % tiles2) * T2; tr /= tiles2;
    o1 = (tr % tiles1) * T1; tr /= tiles1;
    o0 = (tr % tiles0) * T0; tr /= tiles0;
    n = tr;
  };
  // halo of `tile` -> buffer `buf`, by the 4 staging waves.  One LDS-DMA
  // wave-instruction per halo row (c0, c1) brings its first 64 cells: the lane
  // = the cell's t index, so the per-lane part of the address (reflect of t)
  // is computed once per tile and the row part is scalar.  The two remaining
  // cells of each of the 60 rows go through registers (120 lanes, one 16-B
  // load + ds_write each).
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    int n, org0, org1, org2;
    tile_org(tile, n, org0, org1, org2);
    const unsigned short* xn = x + (size_t)n * D0 * D1 * D2 * 8;
    auto clampi = [](int i, int d) { return i < 0 ? 0 : (i > d - 1 ? d - 1 : i); };
    // (ragged tiles: clamped addresses stay legal; results are masked at the store)
    const int i2 = clampi(s3_reflect(org2 + lane - g.lo[2], D2), D2);
    char* bufp = smem + buf * BUF_BYTES;
    const int sw = wave - NCW;
    for (int row = sw; row < H0 * H1; row += NDW) {
      const int c0 = row / H1, c1 = row % H1;
      const int i0 = clampi(s3_reflect(org0 + c0 - g.lo[0], D0), D0);
      const int i1 = clampi(s3_reflect(org1 + c1 - g.lo[1], D1), D1);
      const unsigned short* src = xn + (((size_t)i0 * D1 + i1) * D2 + i2) * 8;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(bufp + row * (H2 * 16)), 16, 0, 0);
    }
    const int st = (tid - NCW * 64);               // 0 .. 255
    if (st < 2 * H0 * H1) {
      const int row = st >> 1, c2 = 64 + (st & 1);
      const int c0 = row / H1, c1 = row % H1;
      const int i0 = clampi(s3_reflect(org0 + c0 - g.lo[0], D0), D0);
      const int i1 = clampi(s3_reflect(org1 + c1 - g.lo[1], D1), D1);
      const int j2 = clampi(s3_reflect(org2 + c2 - g.lo[2], D2), D2);
      const uint4 v = *reinterpret_cast<const uint4*>(xn + (((size_t)i0 * D1 + i1) * D2 + j2) * 8);
      *reinterpret_cast<uint4*>(bufp + (row * H2 + c2) * 16) = v;
    }
  };

  // ---- this workgroup's tiles.  Block b runs on XCD b % 8 and every XCD has
  // its own L2: an XCD owns a CONTIGUOUS range of tiles and its workgroups walk
  // it interleaved, so the tiles in flight on one XCD are neighbours and the
  // halo cells they share (1.9x of the input is read per tile) are L2 hits
  // instead of a second trip to memory.
  int t_first, t_step, t_end;
  {
    const int G = gridDim.x, b = blockIdx.x, xcd = b % 8;
    int before = 0;                       // workgroups on lower XCDs
    for (int q = 0; q < xcd; ++q) before += (G - q + 7) / 8;
    const int mine = (G - xcd + 7) / 8;   // workgroups on this XCD
    const long long lo = (long long)n_tiles * before / G;
    const long long hi = (long long)n_tiles * (before + mine) / G;
    t_first = (int)lo + b / 8;
    t_step = mine;
    t_end = (int)hi;
  }

  // ---- filter fragments (plain): lane (row = co = lane & 15, k-group kq) of
  // k-step s holds w[tap 4s+kq][ci 0..7][co] as bf16 (zero beyond 27 taps / C_out)
  constexpr int NWF = BAND ? 27 : KS;
  bf16x8 wf[NWF];
  unsigned toff[KS];       // byte offset of tap 4s+kq in the halo (B operand)
  if constexpr (!BAND) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int tap = 4 * s + kq;
      unsigned u[4] = {0u, 0u, 0u, 0u};
      if (tap < 27 && p < Cout) {
        const float* wp = w + (size_t)tap * 8 * Cout + p;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pk2(wp[(2 * e) * Cout], wp[(2 * e + 1) * Cout]);
      }
      uint4 uv = make_uint4(u[0], u[1], u[2], u[3]);
      wf[s] = __builtin_bit_cast(bf16x8, uv);
      const int tt = tap < 27 ? tap : 0;
      const int a = tt / 9, b = tt / 3 % 3, c = tt % 3;
      toff[s] = (unsigned)(((a * H1 + b) * H2 + c) * 16);
    }
  } else {
    // BAND: MFMA row i = (delta = i >> 1, co = i & 1) is output position
    // base + delta, column j is base position 8 j; per (a, b) the contraction
    // runs over the 12 cells e = 0..11 after the base (10 are touched):
    // A[(delta, co)][(e, ci)] = w[a, b, c = e - delta][ci][co] for 0 <= c <= 2,
    // else 0.  27 fragments = 9 (a, b) x 3 k-steps of 4 cells, 108 VGPRs;
    // 27 MFMAs per 128 positions instead of 56.
    const int delta = p >> 1, co = p & 1;
#pragma unroll
    for (int f = 0; f < 27; ++f) {
      const int ab = f / 3, s = f % 3;
      const int c = 4 * s + kq - delta;
      unsigned u[4] = {0u, 0u, 0u, 0u};
      if (c >= 0 && c <= 2) {
        const float* wp = w + (size_t)(ab * 3 + c) * 8 * 2 + co;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pk2(wp[(2 * e) * 2], wp[(2 * e + 1) * 2]);
      }
      uint4 uv = make_uint4(u[0], u[1], u[2], u[3]);
      wf[f] = __builtin_bit_cast(bf16x8, uv);
    }
  }
  // bias of this lane's four channels
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (bias && kq * 4 + r < Cout) ? bias[kq * 4 + r] : 0.f;
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);

  // raw workgroup barrier: the hand-over is ordered by the staging waves'
  // own vmcnt(0) before it ("memory": no compiler motion across it)
#define TAIL_BARRIER() asm volatile("s_barrier" ::: "memory")
  int cur = 0;
  if (wave >= NCW) {
    // ---------------------------------------------------- staging waves
    // the pad cells behind each halo are read (against zero filter taps) by
    // the last fragments: keep them finite
    {
      const int st = tid - NCW * 64;
      if (st < 2 * (NDMA * 64 - HP))
        *reinterpret_cast<uint4*>(smem + (st & 1) * BUF_BYTES + (HP + (st >> 1)) * 16) =
            make_uint4(0, 0, 0, 0);
    }
    if (t_first < t_end) stage(t_first, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TAIL_BARRIER();
    for (int tile = t_first; tile < t_end; tile += t_step) {
      const int next = tile + t_step;
      if (next < t_end) stage(next, cur ^ 1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      TAIL_BARRIER();   // next halo landed, this one free
      cur ^= 1;
    }
    return;
  }
  // ------------------------------------------------------ compute waves
  TAIL_BARRIER();
  for (int tile = t_first; tile < t_end; tile += t_step) {
    int n, org0, org1, org2;
    tile_org(tile, n, org0, org1, org2);
    const char* halo = smem + cur * BUF_BYTES;
    if constexpr (BAND) {
      // 16 fragment sets of 128 positions (two s1 rows x 64 t); a compute wave
      // owns sets `wave` and `wave + 8`, each accumulated in two chains (four
      // independent MFMA chains per wave).  Column j = lane & 15: s1 row
      // j >> 3, base t = (j & 7) * 8; k-group kq reads cell e = 4 s + kq.
      const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
      unsigned base[2];
      f32x4 acc[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = wave + u * NCW;
        const int r0 = q >> 2, r1 = 2 * (q & 3) + (p >> 3);
        base[u] = (unsigned)(((r0 * H1 + r1) * H2 + (p & 7) * 8 + kq) * 16);
        acc[u][0] = (f32x4){b0, b1, b0, b1};
        acc[u][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int f = 0; f < 27; ++f) {
        const int ab = f / 3, s = f % 3;
        const unsigned off = (unsigned)((((ab / 3) * H1 + ab % 3) * H2 + 4 * s) * 16);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(halo + base[u] + off);
          acc[u][f & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf, acc[u][f & 1], 0, 0, 0);
        }
      }
      // lane (j, kq) holds positions base t + 2 kq, + 1 x channels 0, 1: one
      // float4; a wave stores 1 KB contiguous per s1 row pair
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = wave + u * NCW;
        const int o0 = org0 + (q >> 2), o1 = org1 + 2 * (q & 3) + (p >> 3);
        const int o2 = org2 + (p & 7) * 8 + 2 * kq;
        if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2]) {
          float* yp = y + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * 2;
          const f32x4 t = acc[u][0] + acc[u][1];
          float v0 = act_sel(t[0], slope), v1 = act_sel(t[1], slope),
                v2 = act_sel(t[2], slope), v3 = act_sel(t[3], slope);
          if (aff) {
            v0 = affine2(v0, aff[0], aff[2]); v1 = affine2(v1, aff[1], aff[3]);
            v2 = affine2(v2, aff[0], aff[2]); v3 = affine2(v3, aff[1], aff[3]);
          }
          if (o2 + 1 < g.O[2]) {
            __builtin_nontemporal_store((f32x4){v0, v1, v2, v3}, reinterpret_cast<f32x4*>(yp));
          } else {
            yp[0] = v0; yp[1] = v1;
          }
        }
      }
    } else {
    // ---- 128 fragments of 16 positions, 16 per wave, four at a time: four
    // independent accumulator chains keep the matrix pipe issuing (one chain
    // of 7 dependent MFMAs would wait out the full MFMA latency 7 times)
    constexpr int GU = 4;
    for (int g0 = wave * GU; g0 < GROUPS; g0 += NCW * GU) {
      unsigned base[GU];
      f32x4 acc[GU];
#pragma unroll
      for (int j = 0; j < GU; ++j) {
        const int gi = g0 + j;
        const int tq = gi % (T2 / 16);
        const int r1 = (gi / (T2 / 16)) % T1;
        const int r0 = gi / ((T2 / 16) * T1);
        base[j] = (unsigned)(((r0 * H1 + r1) * H2 + tq * 16 + p) * 16);
        acc[j] = (f32x4){bv[0], bv[1], bv[2], bv[3]};
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < GU; ++j) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(halo + base[j] + toff[s]);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xf, acc[j], 0, 0, 0);
        }
      if (Cout == 2) {
        // the four fragments are the 64 consecutive t positions of one
        // (s0, s1) row and lane (p, kq = 0) of fragment j holds position
        // 16 j + p: gather them so that lane L stores position L — one
        // full-wave 512-B store instead of four quarter-wave ones
        const int src = (lane & 15) << 2;             // byte index of lane p (kq = 0)
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int j = 0; j < GU; ++j) {
          const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
              src, __builtin_bit_cast(int, act_sel(acc[j][0], slope))));
          const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
              src, __builtin_bit_cast(int, act_sel(acc[j][1], slope))));
          if ((lane >> 4) == j) { v0 = a0; v1 = a1; }
        }
        if (aff) { v0 = affine2(v0, aff[0], aff[2]); v1 = affine2(v1, aff[1], aff[3]); }
        const int gi = g0;
        const int r1 = (gi / (T2 / 16)) % T1;
        const int r0 = gi / ((T2 / 16) * T1);
        const int o0 = org0 + r0, o1 = org1 + r1, o2 = org2 + lane;
        if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2]) {
          float* yp = y + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * 2;
          // (the model output is never re-read on the device)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          __builtin_nontemporal_store((f32x2){v0, v1}, reinterpret_cast<f32x2*>(yp));
        }
      } else {
#pragma unroll
        for (int j = 0; j < GU; ++j) {
          const int gi = g0 + j;
          const int tq = gi % (T2 / 16);
          const int r1 = (gi / (T2 / 16)) % T1;
          const int r0 = gi / ((T2 / 16) * T1);
          const int o0 = org0 + r0, o1 = org1 + r1, o2 = org2 + tq * 16 + p;
          if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && kq * 4 < Cout) {
            float* yp = y + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * Cout + kq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kq * 4 + r < Cout) {
                const float v = act_sel(acc[j][r], slope);
                yp[r] = aff ? affine2(v, aff[kq * 4 + r], aff[Cout + kq * 4 + r]) : v;
              }
          }
        }
      }
    }
    }   // !BAND
    TAIL_BARRIER();
    cur ^= 1;
  }
#undef TAIL_BARRIER
}


// ---------------------------------------------------------------------------
// Sliding-window variant for C_out = 2 (the banded form above): the kernel is
// bound by what one CU can pull in (~10 B / clk / CU, MI355X_MICROARCH.md),
// not by HBM — with 4 x 8 x 64 tiles every input cell is fetched 1.93 times
// (6 x 10 x 66 halo cells per 2048 positions; the XCD-contiguous tile order
// makes the repeats L2 hits, FETCH_SIZE = 1.00x algorithmic, but they still
// cross the L2 -> CU path).  Here a workgroup owns a COLUMN of 16 x 64
// positions and walks it along s0: per output row it brings in ONE new input
// plane (18 x 66 cells) into a 4-slot ring — planes s0 - 1, s0, s0 + 1 are the
// taps a = 0, 1, 2 of row s0 — so a cell is fetched 1.16 x (in-plane halo)
// x 22 / 20 (two extra planes per 20-row segment) = 1.28 times.  Same wave
// roles as above: 8 compute waves (one 2 x 64 position set per row each, 27
// banded MFMAs in the same order as the tile kernel: bit-identical results),
// 4 staging waves (LDS-DMA of the next plane under the current row's MFMAs),
// one raw barrier per row.
constexpr int S1 = 16, S2 = 64;   // (rows per segment: chosen per launch, see launch_conv_tail_mfma)
constexpr int P1 = S1 + 2, P2 = S2 + 2;
constexpr int PLANE_CELLS = P1 * P2;                    // 1188
constexpr int PLANE_BYTES = ((PLANE_CELLS + 63) / 64) * 64 * 16;   // 19,456 incl. pad
// PD planes are in flight ahead of the row being computed (the round trip of a
// plane's LDS-DMA under load is longer than one row's 27 MFMAs: with PD = 1
// every row barrier waited for it).  Measured at C2 x 32 chunks: PD 1 / 2 / 3 /
// 4 = 0.369 / 0.338 / 0.342 / 0.352 ms; a conflict-free swizzle of the plane
// rows (the banded B reads are 4-way bank conflicts in column order) changed
// nothing at PD 1 — the rows wait on the fetch path, not on LDS.
constexpr int PD = 2;
constexpr int NSLOT = 3 + PD;
constexpr int SLIDE_LDS = NSLOT * PLANE_BYTES;          // 97,280

__global__ __launch_bounds__(NTH) void conv_tail_slide_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, ConvGeom g,
    int segs0, int tiles1, int tiles2, int n_units, int SEG0, const float* __restrict__ aff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, kq = lane >> 4;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  auto clampi = [](int i, int d) { return i < 0 ? 0 : (i > d - 1 ? d - 1 : i); };

  // XCD-contiguous unit ranges (see conv_tail_mfma_kernel)
  int u_first, u_step, u_end;
  {
    const int G = gridDim.x, b = blockIdx.x, xcd = b % 8;
    int before = 0;
    for (int q = 0; q < xcd; ++q) before += (G - q + 7) / 8;
    const int mine = (G - xcd + 7) / 8;
    u_first = (int)((long long)n_units * before / G) + b / 8;
    u_step = mine;
    u_end = (int)((long long)n_units * (before + mine) / G);
  }
  // unit -> (n, s0 segment, s1 tile, t tile); t fastest so neighbours share L2
  auto unit_org = [&](int u, int& n, int& r0, int& o1, int& o2) __attribute__((always_inline)) {
    int tr = u;
    o2 = (tr % tiles2) * S2; tr /= tiles2;
    o1 = (tr % tiles1) * S1; tr /= tiles1;
    r0 = (tr % segs0) * SEG0; tr /= segs0;
    n = tr;
  };
  // input plane of padded row index hr (= output row hr - lo0 ... reflect) -> ring slot
  auto stage_plane = [&](int n, int hr, int o1, int o2, int slot) __attribute__((always_inline)) {
    const unsigned short* xn = x + (size_t)n * D0 * D1 * D2 * 8;
    const int i0 = clampi(s3_reflect(hr - g.lo[0], D0), D0);
    const int i2 = clampi(s3_reflect(o2 + lane - g.lo[2], D2), D2);
    char* bufp = smem + slot * PLANE_BYTES;
    const int sw = wave - NCW;
    // (columns 64 / 65: a second DMA piece per row with two active lanes — no
    // register-path load, so every wait below is a counted vmcnt on DMA pieces)
    const int j2 = clampi(s3_reflect(o2 + 64 + (lane & 1) - g.lo[2], D2), D2);
    for (int row = sw; row < P1; row += NDW) {
      const int i1 = clampi(s3_reflect(o1 + row - g.lo[1], D1), D1);
      const unsigned short* rowp = xn + ((size_t)i0 * D1 + i1) * D2 * 8;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(rowp + (size_t)i2 * 8),
          (__attribute__((address_space(3))) void*)(bufp + row * (P2 * 16)), 16, 0, 0);
      if (lane < 2)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(rowp + (size_t)j2 * 8),
            (__attribute__((address_space(3))) void*)(bufp + (row * P2 + 64) * 16), 16, 0, 0);
    }
  };
  // vector-memory instructions stage_plane issues in this wave
  const int plane_ops = 2 * ((P1 - (wave - NCW) + NDW - 1) / NDW);
  auto wait_planes = [&](int in_flight) __attribute__((always_inline)) {
    // all but the youngest `in_flight` planes of this wave have landed
    switch (in_flight * plane_ops) {
      case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
      case 20: asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); break;
      case 24: asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); break;
      case 30: asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
  };

  // banded filter fragments, as in conv_tail_mfma_kernel<true>
  bf16x8 wf[27];
  {
    const int delta = p >> 1, co = p & 1;
#pragma unroll
    for (int f = 0; f < 27; ++f) {
      const int ab = f / 3, s = f % 3;
      const int c = 4 * s + kq - delta;
      unsigned u[4] = {0u, 0u, 0u, 0u};
      if (c >= 0 && c <= 2) {
        const float* wp = w + (size_t)(ab * 3 + c) * 8 * 2 + co;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pk2(wp[(2 * e) * 2], wp[(2 * e + 1) * 2]);
      }
      uint4 uv = make_uint4(u[0], u[1], u[2], u[3]);
      wf[f] = __builtin_bit_cast(bf16x8, uv);
    }
  }
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
#define SLIDE_BARRIER() asm volatile("s_barrier" ::: "memory")

  if (wave >= NCW) {
    // ---------------------------------------------------- staging waves
    {
      // pad cells behind each plane are read against zero filter taps: finite
      const int st = tid - NCW * 64;
      const int npad = PLANE_BYTES / 16 - PLANE_CELLS;
      for (int q = st; q < NSLOT * npad; q += NDW * 64)
        *reinterpret_cast<uint4*>(smem + (q / npad) * PLANE_BYTES + (PLANE_CELLS + q % npad) * 16) =
            make_uint4(0, 0, 0, 0);
    }
    for (int u = u_first; u < u_end; u += u_step) {
      int n, r0, o1, o2;
      unit_org(u, n, r0, o1, o2);
      const int rows = (r0 + SEG0 <= g.O[0] ? SEG0 : g.O[0] - r0);
      // padded rows r0 .. r0 + rows + 1 feed output rows r0 .. r0 + rows - 1;
      // plane q (slot q % NSLOT) is needed from row q - 2 on
      int issued = 0;                        // planes staged so far
      for (; issued < 2 + PD && issued < rows + 2; ++issued) stage_plane(n, r0 + issued, o1, o2, issued % NSLOT);
      wait_planes(issued - 3);               // planes 0 .. 2 are in
      SLIDE_BARRIER();
      for (int r = 0; r < rows; ++r) {
        // slot (r + 2 + PD) % NSLOT held plane r - 1: free since the last barrier
        if (issued < rows + 2) { stage_plane(n, r0 + issued, o1, o2, issued % NSLOT); ++issued; }
        // row r + 1 reads planes up to r + 3
        const int need = r + 4 < rows + 2 ? r + 4 : rows + 2;
        wait_planes(issued - need);
        SLIDE_BARRIER();                     // row r computed, plane r + 3 landed
      }
    }
    return;
  }
  // ------------------------------------------------------ compute waves
  // set `wave`: s1 rows 2 wave, 2 wave + 1 of the column; column j = lane & 15:
  // s1 row j >> 3, base t = (j & 7) * 8; k-group kq reads cell e = 4 s + kq
  const unsigned lane_base = (unsigned)((((2 * wave + (p >> 3)) * P2) + (p & 7) * 8 + kq) * 16);
  for (int u = u_first; u < u_end; u += u_step) {
    int n, r0, o1, o2;
    unit_org(u, n, r0, o1, o2);
    const int rows = (r0 + SEG0 <= g.O[0] ? SEG0 : g.O[0] - r0);
    SLIDE_BARRIER();
    for (int r = 0; r < rows; ++r) {
      f32x4 acc0 = (f32x4){b0, b1, b0, b1}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < 27; ++f) {
        const int ab = f / 3, s = f % 3;
        const int a = ab / 3, b = ab % 3;
        const unsigned off = (unsigned)(((r + a) % NSLOT) * PLANE_BYTES) + (unsigned)((b * P2 + 4 * s) * 16);
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(smem + lane_base + off);
        if (f & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf, acc0, 0, 0, 0);
      }
      const int o0 = r0 + r, oo1 = o1 + 2 * wave + (p >> 3);
      const int oo2 = o2 + (p & 7) * 8 + 2 * kq;
      if (oo1 < g.O[1] && oo2 < g.O[2]) {
        float* yp = y + ((((size_t)n * g.O[0] + o0) * g.O[1] + oo1) * g.O[2] + oo2) * 2;
        const f32x4 t = acc0 + acc1;
        float v0 = act_sel(t[0], slope), v1 = act_sel(t[1], slope),
              v2 = act_sel(t[2], slope), v3 = act_sel(t[3], slope);
        if (aff) {
          v0 = affine2(v0, aff[0], aff[2]); v1 = affine2(v1, aff[1], aff[3]);
          v2 = affine2(v2, aff[0], aff[2]); v3 = affine2(v3, aff[1], aff[3]);
        }
        if (oo2 + 1 < g.O[2]) {
          __builtin_nontemporal_store((f32x4){v0, v1, v2, v3}, reinterpret_cast<f32x4*>(yp));
        } else {
          yp[0] = v0; yp[1] = v1;
        }
      }
      SLIDE_BARRIER();
    }
  }
#undef SLIDE_BARRIER
}

}  // namespace

bool conv_tail_mfma_supported(const ConvGeom& g) {
  if (g.Cin != 8 || g.Cout < 1 || g.Cout > 16 || g.d2s != 1) return false;
  if (g.pad_mode != S3_PAD_REFLECT) return false;   // LDS-DMA cannot write the zeros
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1) return false;
  // tiles of 64 along t would be mostly masked on short series
  return g.O[2] >= 16;
}

int launch_conv_tail_mfma(s3_ctx* ctx, const ConvGeom& g, const void* x,
                          const float* w, const float* bias, float* y, const float* aff) {
  const int tiles0 = (g.O[0] + T0 - 1) / T0, tiles1 = (g.O[1] + T1 - 1) / T1,
            tiles2 = (g.O[2] + T2 - 1) / T2;
  const int n_tiles = g.N * tiles0 * tiles1 * tiles2;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_mfma_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_mfma_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
    attr_set.mark(ctx->device);
  }
  const bool band = g.Cout == 2 && !s3_opt_has(S3O_NO_TAIL_BAND);
  // read per call: the parity tests flip it between two forwards
  const bool noslide = s3_opt_on(S3O_NO_TAIL_SLIDE);
  // (plane-sweep form: kernels_conv_tail_sweep.hip — same bits, 0.8 of the bytes through the CU)
  if (band && !noslide && !s3_opt_on(S3O_NO_TAIL_SWEEP) && conv_tail_sweep_supported(g))
    return launch_conv_tail_sweep(ctx, g, x, w, bias, y, aff);
  if (band && g.O[0] >= 4 && !noslide) {
    static S3DeviceOnce slide_attr;
    if (!slide_attr.done(ctx->device)) {
      std::lock_guard<std::mutex> lk_slide_attr(slide_attr.m);
      S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail_slide_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SLIDE_LDS));
      slide_attr.mark(ctx->device);
    }
    const int st1 = (g.O[1] + S1 - 1) / S1, st2 = (g.O[2] + S2 - 1) / S2;
    // rows per segment: a workgroup brings in seg + 2 planes per unit and the
    // launch lasts as long as its busiest workgroup — ceil(units / CUs) units.
    // 20 rows (1.10 planes per row) suit large batches; at C2 batch 8 (800
    // units of 20 rows on 256 CUs: 4 rounds of 22 planes) 16 rows make it 1000
    // units: 4 rounds of 18 planes.
    int seg = 20;
    {
      long long best = -1;
      for (int cand : {8, 10, 12, 16, 20, 24, 32, 40}) {
        if (cand > g.O[0] && cand != 8) continue;
        const int sg = (g.O[0] + cand - 1) / cand;
        const long long units = (long long)g.N * sg * st1 * st2;
        const long long rounds = (units + ctx->num_cu - 1) / ctx->num_cu;
        // planes of the busiest workgroup (+ 3 per unit: pipeline fill)
        const long long cost = rounds * (cand + 2 + 3);
        if (best < 0 || cost < best) { best = cost; seg = cand; }
      }
    }
    const int segs0 = (g.O[0] + seg - 1) / seg;
    const int n_units = g.N * segs0 * st1 * st2;
    int sgrid = ctx->num_cu;
    if (sgrid > n_units) sgrid = n_units;
    hipLaunchKernelGGL(conv_tail_slide_kernel, dim3(sgrid), dim3(NTH), SLIDE_LDS, ctx->stream,
                       (const unsigned short*)x, w, bias, y, g, segs0, st1, st2, n_units, seg, aff);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  int grid = ctx->num_cu;
  if (grid > n_tiles) grid = n_tiles;
  auto kern = band ? conv_tail_mfma_kernel<true> : conv_tail_mfma_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), 2 * BUF_BYTES, ctx->stream,
                     (const unsigned short*)x, w, bias, y, g, tiles0, tiles1, tiles2, n_tiles, aff);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
