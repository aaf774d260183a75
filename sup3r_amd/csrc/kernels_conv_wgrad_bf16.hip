// Trunk weight gradient on bf16 MFMA (gfx950) — S3_PREC_BF16 training plans.
//
//   dW[tap][ci][co] = sum_{n, p} Xpad[n, p + tap - 1][ci] * dPre[n, p][co]
//
// = 27 GEMMs (64 x P)(P x C_out) whose contraction index is the POSITION, while
// both operands live channels-last.  v_mfma_f32_16x16x32_bf16 wants 8
// consecutive k per lane, i.e. 8 positions of one channel: the operands are
// read with ds_read_b64_tr_b16 (LDS transpose read: within a 16-lane group
// lane q gets column q of the 4 x 16 bf16 block whose row r is the 32 B that
// lanes 4r..4r+3 point at; the row addresses are free), so the LDS images stay
// in the natural [cell][channel] order — the tap shift is a cell offset and
// the staging is a plain convert + 8-byte store.
//
// PERSISTENT workgroups of 12 waves (3 per SIMD), each owning a 32-wide cout
// tile of the whole 27 x 64 x 32 gradient in registers: wave w = (ci block
// w & 3, s1-tap a = w >> 2) holds 9 taps x 2 cout blocks = 18 f32x4
// accumulators.  Position tiles of 4 x 4 x 16 stream through LDS: x halo
// 6 x 6 x 18 cells x 128 B (bf16, 32-B segments XOR-swizzled by the cell's t
// index) + dPre 256 x 64 B.  Per 32-position k-step a wave issues 18 A + 4 B
// transpose reads (2 LDS cycles each) for 18 MFMAs (16 cycles each).
// Partials per workgroup, reduced in fixed order (deterministic on every rank).
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int BT0 = 4, BT1 = 4, BT2 = 16;
constexpr int BH0 = BT0 + 2, BH1 = BT1 + 2, BH2 = BT2 + 2;
constexpr int BHP = BH0 * BH1 * BH2;            // 648 halo cells
constexpr int BNP = BT0 * BT1 * BT2;            // 256 positions per tile
constexpr int BNT = 768;                        // 12 waves
constexpr int BCT = 32;                         // cout tile per workgroup
constexpr int XS_BYTES = BHP * 128;             // 82,944
constexpr int DS_BYTES = BNP * 64;              // 16,384
constexpr size_t BF_LDS = (size_t)XS_BYTES + DS_BYTES;

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

// 32-B segment swizzle of a halo cell: cells of equal parity among the 8 cells
// one half-wave transpose read touches ({t..t+3} U {t+8..t+11}) get distinct keys
__device__ __forceinline__ int xs_key(int th) { return ((th >> 1) & 1) | (((th >> 3) & 1) << 1); }

__device__ __forceinline__ s16x4 lds_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(p));
}

// DY16: dPre arrives as bf16 (the frame fold's bf16-only store): 8-B loads, no convert
template <bool IN16, bool DY16 = false>
__global__ __launch_bounds__(BNT) void conv3_wgrad_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int n_tiles, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                       // [BHP cells][4 segs of 32 B] bf16
  char* ds = smem + XS_BYTES;            // [BNP positions][2 segs of 32 B] bf16
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, kg = lane >> 4;
  const int cb = wave & 3, ta = wave >> 2;                 // ci block, s1 tap
  const int ct = blockIdx.y;                                // cout tile of 32
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // per-lane operand addresses.  k = 8 kg + e  <->  tile row 2 ks + (kg >> 1),
  // t = 8 (kg & 1) + e; transpose read h covers e = 4 h .. 4 h + 3 and this
  // lane points at row e = 4 h + (q >> 2), chunk q & 3 of the 16-channel block
  int a_off[3][2];                       // [c][h], without the k-step / b part
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int th = 8 * (kg & 1) + 4 * h + (q >> 2) + c;
      a_off[c][h] = ((ta * BH1 + (kg >> 1)) * BH2 + th) * 128 +
                    ((cb ^ xs_key(th)) << 5) + ((q & 3) << 3);
    }
  int b_off[2][2];                       // [cout block][h]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = 8 * kg + 4 * h + (q >> 2);
      b_off[nb][h] = pl * 64 + ((nb ^ ((pl >> 3) & 1)) << 5) + ((q & 3) << 3);
    }

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int tr = tile;
    const int t2i = tr % tiles2; tr /= tiles2;
    const int t1i = tr % tiles1; tr /= tiles1;
    const int t0i = tr % tiles0; tr /= tiles0;
    const int n = tr;
    const int org0 = t0i * BT0, org1 = t1i * BT1, org2 = t2i * BT2;
    __syncthreads();   // previous tile fully consumed
    // ---- stage x halo: 648 cells x 16 float4 -> bf16
    // (dbg: timing-only ablations, results invalid — bit 0 stages the first
    // tile only, bit 1 skips the MFMA loop)
    // (IN16: x is a bf16 tensor — 16-B chunks of 8 channels, no convert)
    constexpr int XCH = IN16 ? 8 : 16;          // chunks per cell
    for (int item = tid; item < (((dbg & 1) && tile != (int)blockIdx.x) ? 0 : BHP * XCH); item += BNT) {
      const int hp = item / XCH, ch = item % XCH;
      int h = hp;
      const int c2 = h % BH2; h /= BH2;
      const int c1 = h % BH1; h /= BH1;
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
      const size_t cell = (((size_t)n * D0 + i0) * D1 + i1) * D2 + i2;
      if constexpr (IN16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (valid)
          v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(x) + cell * 64 + ch * 8);
        *reinterpret_cast<uint4*>(xs + hp * 128 + (((ch >> 1) ^ xs_key(c2)) << 5) + ((ch & 1) << 4)) = v;
      } else {
        float4 v = make_float4(0, 0, 0, 0);
        if (valid) v = *reinterpret_cast<const float4*>(x + cell * 64 + ch * 4);
        uint2 pk = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
        *reinterpret_cast<uint2*>(xs + hp * 128 + (((ch >> 2) ^ xs_key(c2)) << 5) + ((ch & 3) << 3)) = pk;
      }
    }
    // ---- stage dPre tile: 256 positions x 8 float4 (zero outside / beyond C_out)
    for (int item = tid; item < (((dbg & 1) && tile != (int)blockIdx.x) ? 0 : BNP * (BCT / 4)); item += BNT) {
      const int pl = item >> 3, ch = item & 7;
      const int row = pl / BT2, tt = pl % BT2;
      const int o0 = org0 + row / BT1, o1 = org1 + row % BT1, o2 = org2 + tt;
      const int co = ct * BCT + ch * 4;
      uint2 pk = make_uint2(0u, 0u);
      const bool in = o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && co < g.Cout;
      const size_t de = ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * g.Cout + co;
      if constexpr (DY16) {
        if (in) pk = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dy) + de);
      } else {
        float4 v = make_float4(0, 0, 0, 0);
        if (in) v = *reinterpret_cast<const float4*>(dy + de);
        pk = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
      }
      *reinterpret_cast<uint2*>(ds + pl * 64 + (((ch >> 2) ^ ((pl >> 3) & 1)) << 5) + ((ch & 3) << 3)) = pk;
    }
    __syncthreads();
    if (dbg & 2) continue;
    // ---- 8 k-steps of 32 positions (2 rows x 16 t)
#pragma unroll
    for (int ks = 0; ks < BNP / 32; ++ks) {
      // rows 2 ks, 2 ks + 1: r0 = ks >> 1, r1 = 2 (ks & 1) + (kg >> 1)
      const int rowb = (((ks >> 1) * BH1) + 2 * (ks & 1)) * BH2 * 128;
      bf16x8 bfr[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const s16x4 lo = lds_tr(ds + b_off[nb][0] + ks * 32 * 64);
        const s16x4 hi = lds_tr(ds + b_off[nb][1] + ks * 32 * 64);
        bfr[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const s16x4 lo = lds_tr(xs + a_off[c][0] + rowb + b * BH2 * 128);
          const s16x4 hi = lds_tr(xs + a_off[c][1] + rowb + b * BH2 * 128);
          const bf16x8 afr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[b * 3 + c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[0], acc[b * 3 + c][0], 0, 0, 0);
          acc[b * 3 + c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[1], acc[b * 3 + c][1], 0, 0, 0);
        }
    }
  }
  // ---- partial[bid][tap][ci][co]: C/D map col = lane & 15 (co), row = 4 kg + r (ci)
  float* out = partial + (size_t)blockIdx.x * 27 * 64 * g.Cout;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int co = ct * BCT + nb * 16 + q;
    if (co < g.Cout) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((size_t)(ta * 9 + t) * 64 + cb * 16 + kg * 4 + r) * g.Cout + co] = acc[t][nb][r];
    }
  }
}

// ---------------------------------------------------------------------------
// Wave-specialised variant for the all-bf16, reflect-padded, exact-fit case (the
// trunk convs of a training step).  PMC of conv3_wgrad_bf16_kernel: MFMA busy
// 27 % of the SIMD cycles, waves waiting 58 % — all 12 waves stand in front of
// every tile's staging loads, and next to 72 accumulators there is no register
// room to prefetch (three attempts).  Here the 4 x 4 x 16 tile
// is walked as two 2 x 4 x 16 halves (same k-step order: bit-identical sums),
// a half's images are 62 KB (x halo 4 x 6 x 18 cells + 128 dPre rows), so TWO
// fit in LDS, and 4 extra producer waves fill the other buffer by LDS-DMA
// (global_load_lds_dwordx4: bf16 in HBM = bf16 in LDS, the swizzles are applied
// by choosing each lane's SOURCE chunk) while the 12 consumer waves run the
// 4 k-steps of the current half.  One workgroup barrier per half hands over.
constexpr int WH0 = 2;                                   // s0 rows of a half tile
constexpr int WHP = (WH0 + 2) * BH1 * BH2;               // 432 halo cells
constexpr int WNP = WH0 * BT1 * BT2;                     // 128 positions
constexpr int WXS = WHP * 128;                           // 55,296 B
constexpr int WBUF = WXS + WNP * 64;                     // 63,488 B per buffer
constexpr int WS_LDS = 2 * WBUF;                         // 126,976 B
constexpr int WS_NT = 1024;                              // 12 consumer + 4 producer waves
constexpr int WS_NXI = WXS / 1024;                       // 54 wave-DMAs per x halo
constexpr int WS_NDI = WNP * 64 / 1024;                  // 8 wave-DMAs per dPre block

__global__ __launch_bounds__(WS_NT) void conv3_wgrad_bf16_ws_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // first channel of this workgroup's 32-wide cout tile.  C_out need not be a
  // multiple of 32 (the 64 -> 200 conv): the last tile is moved back to end at
  // C_out, its overlap with the tile before is computed twice — the same sums
  // in the same order, so both workgroups store identical values there
  const int cbase = (int)(blockIdx.y * BCT + BCT) <= g.Cout ? (int)(blockIdx.y * BCT) : g.Cout - BCT;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  // this workgroup's half tiles: item it = (tile blockIdx.x + (it >> 1) gridDim.x, half it & 1)
  // (round-robin tiles: XCD-contiguous shares, which help the HBM-class
  // weight gradients, ran this MFMA-bound kernel 6 % slower — 121 vs 114 us)
  const int my_tiles = (int)blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int n_items = 2 * my_tiles;
  auto item_org = [&](int it, int& n, int& o0, int& o1, int& o2) __attribute__((always_inline)) {
    int tr = blockIdx.x + (it >> 1) * gridDim.x;
    o2 = (tr % tiles2) * BT2; tr /= tiles2;
    o1 = (tr % tiles1) * BT1; tr /= tiles1;
    o0 = (tr % tiles0) * BT0 + (it & 1) * WH0; tr /= tiles0;
    n = tr;
  };

  if (wave >= 12) {
    // ================================================= producer waves
    const int pw = wave - 12;
    // tile-invariant part of this lane's DMA sources: halo cell coordinates and
    // the swizzle-resolved source chunk of every wave-DMA it takes part in
    constexpr int NXJ = (WS_NXI + 3) / 4, NDJ = (WS_NDI + 3) / 4;
    int xt[NXJ];          // c0 | c1 << 4 | c2 << 8 | ch << 16
#pragma unroll
    for (int k = 0; k < NXJ; ++k) {
      const int j = pw + 4 * k;
      const int cell = 8 * j + (lane >> 3), slot = lane & 7;
      int h = cell;
      const int c2 = h % BH2; h /= BH2;
      const int c1 = h % BH1; h /= BH1;
      // slot = (((ch >> 1) ^ key) << 1) | (ch & 1)  <=>  ch = (((slot >> 1) ^ key) << 1) | (slot & 1)
      const int ch = ((((slot >> 1) ^ xs_key(c2)) << 1) | (slot & 1));
      xt[k] = h | (c1 << 4) | (c2 << 8) | (ch << 16);
    }
    int dt[NDJ];          // s0 row | s1 row << 4 | t << 8 | source channel << 16
#pragma unroll
    for (int k = 0; k < NDJ; ++k) {
      const int j = pw + 4 * k;
      const int pl = 16 * j + (lane >> 2), s16 = lane & 3;
      const int row = pl / BT2, tt = pl % BT2;
      // LDS: 32-B segment seg ^ ((pl >> 3) & 1) holds channels 16 seg ..; slot s16 = its 16-B half
      const int seg = (s16 >> 1) ^ ((pl >> 3) & 1);
      dt[k] = (row / BT1) | ((row % BT1) << 4) | (tt << 8) | ((seg * 16 + (s16 & 1) * 8) << 16);
    }
    auto load_item = [&](int it, char* buf) __attribute__((always_inline)) {
      int n, o0, o1, o2;
      item_org(it, n, o0, o1, o2);
      const unsigned short* xn = x + (size_t)n * D0 * D1 * D2 * 64;
      // x halo: wave-DMA j fills cells 8 j .. 8 j + 7; lane -> (cell, 16-B slot)
#pragma unroll
      for (int k = 0; k < NXJ; ++k) {
        const int j = pw + 4 * k;
        if (j >= WS_NXI) break;
        const int t = xt[k];
        const int i0 = s3_reflect(o0 + (t & 15) - 1, D0), i1 = s3_reflect(o1 + ((t >> 4) & 15) - 1, D1),
                  i2 = s3_reflect(o2 + ((t >> 8) & 255) - 1, D2);
        const unsigned short* src = xn + (unsigned)(((i0 * D1 + i1) * D2 + i2) * 64 + (t >> 16) * 8);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(buf + j * 1024), 16, 0, 0);
      }
      // dPre rows: wave-DMA j fills positions 16 j .. 16 j + 15; lane -> (position, 16-B slot)
      const unsigned short* dn = dy + (size_t)n * g.O[0] * g.O[1] * g.O[2] * g.Cout + cbase;
#pragma unroll
      for (int k = 0; k < NDJ; ++k) {
        const int j = pw + 4 * k;
        if (j >= WS_NDI) break;
        const int t = dt[k];
        const int p0 = o0 + (t & 15), p1 = o1 + ((t >> 4) & 15), p2 = o2 + ((t >> 8) & 255);
        const unsigned short* src = dn + (unsigned)(((p0 * g.O[1] + p1) * g.O[2] + p2) * g.Cout + (t >> 16));
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(buf + WXS + j * 1024), 16, 0, 0);
      }
    };
    if (n_items > 0) load_item(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA landed before the hand-over
    __syncthreads();
    for (int it = 0; it < n_items; ++it) {
      if (it + 1 < n_items) load_item(it + 1, smem + ((it + 1) & 1) * WBUF);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }

  // =================================================== consumer waves
  const int q = lane & 15, kg = lane >> 4;
  const int cb = wave & 3, ta = wave >> 2;
  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int a_off[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int th = 8 * (kg & 1) + 4 * h + (q >> 2) + c;
      a_off[c][h] = ((ta * BH1 + (kg >> 1)) * BH2 + th) * 128 +
                    ((cb ^ xs_key(th)) << 5) + ((q & 3) << 3);
    }
  int b_off[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = 8 * kg + 4 * h + (q >> 2);
      b_off[nb][h] = pl * 64 + ((nb ^ ((pl >> 3) & 1)) << 5) + ((q & 3) << 3);
    }
  __syncthreads();            // first half landed
  for (int it = 0; it < n_items; ++it) {
    const char* xs = smem + (it & 1) * WBUF;
    const char* ds = xs + WXS;
#pragma unroll
    for (int ks = 0; ks < WNP / 32; ++ks) {
      // rows 2 ks, 2 ks + 1 of the half: r0 = ks >> 1, r1 = 2 (ks & 1) + (kg >> 1)
      const int rowb = (((ks >> 1) * BH1) + 2 * (ks & 1)) * BH2 * 128;
      bf16x8 bfr[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const s16x4 lo = lds_tr(ds + b_off[nb][0] + ks * 32 * 64);
        const s16x4 hi = lds_tr(ds + b_off[nb][1] + ks * 32 * 64);
        bfr[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const s16x4 lo = lds_tr(xs + a_off[c][0] + rowb + b * BH2 * 128);
          const s16x4 hi = lds_tr(xs + a_off[c][1] + rowb + b * BH2 * 128);
          const bf16x8 afr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[b * 3 + c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[0], acc[b * 3 + c][0], 0, 0, 0);
          acc[b * 3 + c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[1], acc[b * 3 + c][1], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * 27 * 64 * g.Cout;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int co = cbase + nb * 16 + q;
    if (co < g.Cout) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((size_t)(ta * 9 + t) * 64 + cb * 16 + kg * 4 + r) * g.Cout + co] = acc[t][nb][r];
    }
  }
}

// ---------------------------------------------------------------------------
// S3_PREC_BF16X3 training plans: the same contraction with both operands split
// on the fly into bf16 pairs (hi = bf16(v), lo = bf16(v - hi)) and every
// product taken as hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate —
// fp32-class weight gradients (the dropped lo*lo term is ~2^-16 relative) at
// three MFMAs per product instead of the exact-fp32 MFMA's eight-times-slower
// rate.  x and dPre are fp32 in HBM.  The hi and lo images together are as
// large as the fp32 data, so a workgroup walks HALF tiles (2 x 4 x 16
// positions, x halo 4 x 6 x 18 cells) like the wave-specialised kernel:
// x hi | x lo | dPre hi | dPre lo = 2 x 55,296 + 2 x 8,192 = 126,976 B.
// Same lane maps, swizzles and k-step order as conv3_wgrad_bf16_kernel.
__global__ __launch_bounds__(BNT) void conv3_wgrad_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H0 = 2;                             // s0 rows of a half tile
  constexpr int HP = (H0 + 2) * BH1 * BH2;          // 432 halo cells
  constexpr int NP = H0 * BT1 * BT2;                // 128 positions
  char* xh = smem;                                  // [HP][128 B] hi
  char* xl = smem + HP * 128;                       // [HP][128 B] lo
  char* dh = smem + 2 * HP * 128;                   // [NP][64 B] hi
  char* dl = dh + NP * 64;                          // [NP][64 B] lo
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, kg = lane >> 4;
  const int cb = wave & 3, ta = wave >> 2;
  const int ct = blockIdx.y;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int a_off[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int th = 8 * (kg & 1) + 4 * h + (q >> 2) + c;
      a_off[c][h] = ((ta * BH1 + (kg >> 1)) * BH2 + th) * 128 +
                    ((cb ^ xs_key(th)) << 5) + ((q & 3) << 3);
    }
  int b_off[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = 8 * kg + 4 * h + (q >> 2);
      b_off[nb][h] = pl * 64 + ((nb ^ ((pl >> 3) & 1)) << 5) + ((q & 3) << 3);
    }
  // split two fp32 pairs into packed bf16 hi / lo words
  auto split = [](const float4& v, uint2& hi, uint2& lo) __attribute__((always_inline)) {
    hi = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
    lo = make_uint2(pk2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xFFFF0000u)),
                    pk2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xFFFF0000u)));
  };

  for (int item = 2 * blockIdx.x; item < 2 * n_tiles; item = (item & 1) ? item - 1 + 2 * (int)gridDim.x : item + 1) {
    // item = (tile, half): both halves of a tile, then the tile gridDim.x further on
    int tr = item >> 1;
    const int t2i = tr % tiles2; tr /= tiles2;
    const int t1i = tr % tiles1; tr /= tiles1;
    const int t0i = tr % tiles0; tr /= tiles0;
    const int n = tr;
    const int org0 = t0i * BT0 + (item & 1) * H0, org1 = t1i * BT1, org2 = t2i * BT2;
    __syncthreads();   // previous half fully consumed
    for (int it = tid; it < HP * 16; it += BNT) {
      const int hp = it >> 4, ch = it & 15;
      int h = hp;
      const int c2 = h % BH2; h /= BH2;
      const int c1 = h % BH1; h /= BH1;
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
      const size_t cell = (((size_t)n * D0 + i0) * D1 + i1) * D2 + i2;
      float4 v = make_float4(0, 0, 0, 0);
      if (valid) v = *reinterpret_cast<const float4*>(x + cell * 64 + ch * 4);
      uint2 hi, lo;
      split(v, hi, lo);
      const int o = hp * 128 + (((ch >> 2) ^ xs_key(c2)) << 5) + ((ch & 3) << 3);
      *reinterpret_cast<uint2*>(xh + o) = hi;
      *reinterpret_cast<uint2*>(xl + o) = lo;
    }
    for (int it = tid; it < NP * (BCT / 4); it += BNT) {
      const int pl = it >> 3, ch = it & 7;
      const int row = pl / BT2, tt = pl % BT2;
      const int o0 = org0 + row / BT1, o1 = org1 + row % BT1, o2 = org2 + tt;
      const int co = ct * BCT + ch * 4;
      const bool in = o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && co < g.Cout;
      const size_t de = ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * g.Cout + co;
      float4 v = make_float4(0, 0, 0, 0);
      if (in) v = *reinterpret_cast<const float4*>(dy + de);
      uint2 hi, lo;
      split(v, hi, lo);
      const int o = pl * 64 + (((ch >> 2) ^ ((pl >> 3) & 1)) << 5) + ((ch & 3) << 3);
      *reinterpret_cast<uint2*>(dh + o) = hi;
      *reinterpret_cast<uint2*>(dl + o) = lo;
    }
    __syncthreads();
    // ---- 4 k-steps of 32 positions (2 s1 rows x 16 t)
#pragma unroll
    for (int ks = 0; ks < NP / 32; ++ks) {
      const int rowb = (((ks >> 1) * BH1) + 2 * (ks & 1)) * BH2 * 128;
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const s16x4 h0 = lds_tr(dh + b_off[nb][0] + ks * 32 * 64);
        const s16x4 h1 = lds_tr(dh + b_off[nb][1] + ks * 32 * 64);
        bh[nb] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x4 l0 = lds_tr(dl + b_off[nb][0] + ks * 32 * 64);
        const s16x4 l1 = lds_tr(dl + b_off[nb][1] + ks * 32 * 64);
        bl[nb] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int ao0 = a_off[c][0] + rowb + b * BH2 * 128, ao1 = a_off[c][1] + rowb + b * BH2 * 128;
          const s16x4 h0 = lds_tr(xh + ao0);
          const s16x4 h1 = lds_tr(xh + ao1);
          const bf16x8 ah = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          const s16x4 l0 = lds_tr(xl + ao0);
          const s16x4 l1 = lds_tr(xl + ao1);
          const bf16x8 al = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 a = acc[b * 3 + c][nb];
            // small terms first
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nb], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nb], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nb], a, 0, 0, 0);
            acc[b * 3 + c][nb] = a;
          }
        }
    }
  }
  float* out = partial + (size_t)blockIdx.x * 27 * 64 * g.Cout;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int co = ct * BCT + nb * 16 + q;
    if (co < g.Cout) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((size_t)(ta * 9 + t) * 64 + cb * 16 + kg * 4 + r) * g.Cout + co] = acc[t][nb][r];
    }
  }
}
constexpr int X3_LDS = 2 * (4 * BH1 * BH2) * 128 + 2 * (2 * BT1 * BT2) * 64;   // 126,976 B

// workgroups [0, wblocks): the weight gradient's partials; workgroups past
// them (b_c of them, when a bias job rides along): bias_grad_stage2's channels
__global__ void wgrad_bf16_partial_reduce(const float* __restrict__ partial,
                                          int n_part, int64_t wsize,
                                          float* __restrict__ dw, int accumulate, int wblocks,
                                          const float* __restrict__ b_partial, int b_nblk, int b_c,
                                          float* __restrict__ b_db, int b_accumulate) {
  if ((int)blockIdx.x >= wblocks) {
    __shared__ float sm[256];
    s3_bias_stage2_body(b_partial, b_nblk, b_c, (int)blockIdx.x - wblocks, b_db, b_accumulate, sm);
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < wsize;
       i += (int64_t)wblocks * blockDim.x) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int s = 0;
    for (; s + 4 <= n_part; s += 4) {
      t0 += partial[(int64_t)s * wsize + i];
      t1 += partial[(int64_t)(s + 1) * wsize + i];
      t2 += partial[(int64_t)(s + 2) * wsize + i];
      t3 += partial[(int64_t)(s + 3) * wsize + i];
    }
    for (; s < n_part; ++s) t0 += partial[(int64_t)s * wsize + i];
    const float t = (t0 + t1) + (t2 + t3);
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

static int launch_wgrad_bf16_reduce(s3_ctx* ctx, const float* partial, int n_part, int64_t wsize, float* dw,
                                    int accumulate, int wblocks) {
  // a bias gradient's second stage waiting in the context rides along
  const s3_ctx::PendingBias j = ctx->pend_bias;
  ctx->pend_bias.partial = nullptr;
  const int extra = j.partial ? j.c : 0;
  hipLaunchKernelGGL(wgrad_bf16_partial_reduce, dim3(wblocks + extra), dim3(256), 0, ctx->stream, partial,
                     n_part, wsize, dw, accumulate, wblocks, j.partial, j.nblk, j.c, j.db, j.accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int bf_grid(const s3_ctx* ctx, const ConvGeom& g, int* n_tiles_out, int* t0,
            int* t1, int* t2) {
  const int tiles0 = (g.O[0] + BT0 - 1) / BT0, tiles1 = (g.O[1] + BT1 - 1) / BT1,
            tiles2 = (g.O[2] + BT2 - 1) / BT2;
  const int n_tiles = g.N * tiles0 * tiles1 * tiles2;
  *n_tiles_out = n_tiles; *t0 = tiles0; *t1 = tiles1; *t2 = tiles2;
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  int grid = ctx->num_cu / n_ct;
  if (grid < 1) grid = 1;
  if (grid > n_tiles) grid = n_tiles;
  return grid;
}


// ===========================================================================
// General variant for the discriminator convs: C_in = 32 (CIB = 2 blocks of 16)
// or tiles of 64 channels (CIB = 4; blockIdx.z walks C_in = 128 / 256), stride 1
// or 2, any low padding / valid extents.  Same scheme: 12 waves = (s1 tap a) x
// (ci block, and for CIB = 2 the cout block), 9 taps x NBW accumulators per
// wave, operands by LDS transpose reads from the natural [cell][channel] bf16
// images.
//   stride 1: 4 x 4 x 16 positions per tile (halo 6 x 6 x 18 cells)
//   stride 2: 2 x 2 x 16 positions per tile (halo 5 x 5 x 33 cells, stored with
//             the t index de-interleaved by parity so that the cells 2 t + c of
//             consecutive positions are consecutive in LDS)
template <int CIB, int STR>
struct BfGen {
  static constexpr int T0 = STR == 1 ? 4 : 2, T1 = STR == 1 ? 4 : 2, T2 = 16;
  static constexpr int G0 = (T0 - 1) * STR + 3, G1 = (T1 - 1) * STR + 3, G2 = (T2 - 1) * STR + 3;
  static constexpr int HP = G0 * G1 * G2;
  static constexpr int NP = T0 * T1 * T2;
  static constexpr int CB = CIB * 32;               // bytes per halo cell
  static constexpr int NBW = CIB / 2;               // cout blocks per wave
  static constexpr int XS = HP * CB;
  static constexpr size_t LDS = (size_t)XS + NP * 64;
  // LDS slot of halo t index th
  __device__ static __forceinline__ int tau(int th) {
    return STR == 1 ? th : (th & 1) * ((G2 + 1) / 2) + (th >> 1);
  }
  // 32-B segment swizzle key of LDS slot u (see conv3_wgrad_bf16_kernel)
  __device__ static __forceinline__ int key(int u) {
    return CIB == 4 ? (((u >> 1) & 1) | (((u >> 3) & 1) << 1)) : ((u >> 3) & 1);
  }
};

// PF (C_in = 32 blocks, bf16 x): with one cout block per wave (36 accumulator
// registers) there IS room to prefetch — the next tile's x chunks and dPre rows
// are fetched into registers before the k-steps of the current tile and dropped
// into LDS after them (zero padding / ragged tiles: per-chunk zero masks), so
// the staging traffic runs under the MFMAs.  Same operands, same order.
// X3 (S3_PREC_BF16X3 plans; fp32 x and dPre, C_in walked in tiles of 32 so
// that the hi AND lo images of both operands fit LDS): operands split in the
// staging, hi*hi + hi*lo + lo*hi per fragment pair, as conv3_wgrad_x3_kernel.
template <int CIB, int STR, bool IN16, bool PF = false, bool X3 = false>
__global__ __launch_bounds__(BNT) void conv_wgrad_bf16_gen_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int n_tiles) {
  using W = BfGen<CIB, STR>;
  constexpr int T1 = W::T1, T2 = W::T2, G1 = W::G1, G2 = W::G2, HP = W::HP, NP = W::NP;
  constexpr int CB = W::CB, NBW = W::NBW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!X3 || (CIB == 2 && !IN16 && !PF), "the split-bf16 variant: fp32 operands, 32-channel tiles");
  char* xs = smem;
  char* ds = smem + (X3 ? 2 : 1) * W::XS;
  char* xl = smem + W::XS;                  // X3: residue images
  char* dl = ds + NP * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, kg = lane >> 4;
  const int ta = wave >> 2, w4 = wave & 3;
  const int cb = CIB == 4 ? w4 : (w4 & 1);
  const int nb0 = CIB == 4 ? 0 : (w4 >> 1);
  const int ct = blockIdx.y;
  const int ci0 = blockIdx.z * (CIB * 16);
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  const int Cin = g.Cin, Cout = g.Cout;

  f32x4 acc[9][NBW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int a_off[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int th = (8 * (kg & 1) + 4 * h + (q >> 2)) * STR + c;
      const int u = W::tau(th);
      a_off[c][h] = ((ta * G1 + (kg >> 1) * STR) * G2 + u) * CB + ((cb ^ W::key(u)) << 5) + ((q & 3) << 3);
    }
  int b_off[NBW][2];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = 8 * kg + 4 * h + (q >> 2);
      b_off[nb][h] = pl * 64 + (((nb0 + nb) ^ ((pl >> 3) & 1)) << 5) + ((q & 3) << 3);
    }

  // (XCD-contiguous tile shares: s3_xcd_share, common.h)
  int64_t xt_lo, xt_hi;
  int xt_k, xt_nk;
  s3_xcd_share(n_tiles, xt_lo, xt_hi, xt_k, xt_nk);
  if constexpr (PF) {
    static_assert(IN16, "the prefetching variant stages bf16 cells");
    constexpr int CHP = CIB * 2;                               // 16-B chunks per cell
    constexpr int NXI = (HP * CHP + BNT - 1) / BNT, NDI = (NP * (BCT / 4) + BNT - 1) / BNT;
    uint4 xr[NXI];
    float4 dr[NDI];
    unsigned xz = 0, dz = 0;                                   // chunks stored as zero
    auto fetch = [&](int tile) __attribute__((always_inline)) {
      int tr = tile;
      const int t2i = tr % tiles2; tr /= tiles2;
      const int t1i = tr % tiles1; tr /= tiles1;
      const int t0i = tr % tiles0; tr /= tiles0;
      const int n = tr;
      const int org0 = t0i * W::T0, org1 = t1i * T1, org2 = t2i * T2;
      xz = 0; dz = 0;
#pragma unroll
      for (int k = 0; k < NXI; ++k) {
        const int item = tid + k * BNT;
        const int hp = item / CHP, ch = item % CHP;
        int h = hp;
        const int c2 = h % G2; h /= G2;
        const int c1 = h % G1; h /= G1;
        int i0 = org0 * STR + h - g.lo[0], i1 = org1 * STR + c1 - g.lo[1], i2 = org2 * STR + c2 - g.lo[2];
        if (g.pad_mode == S3_PAD_REFLECT) {
          i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
        }
        const bool valid = item < HP * CHP && i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2 &&
                           ci0 + ch * 8 < Cin;
        if (!valid) xz |= 1u << k;
        else
          xr[k] = *reinterpret_cast<const uint4*>(
              reinterpret_cast<const unsigned short*>(x) +
              ((((size_t)n * D0 + i0) * D1 + i1) * D2 + i2) * Cin + ci0 + ch * 8);
      }
#pragma unroll
      for (int k = 0; k < NDI; ++k) {
        const int item = tid + k * BNT;
        const int pl = item >> 3, ch = item & 7;
        const int row = pl / T2, tt = pl % T2;
        const int o0 = org0 + row / T1, o1 = org1 + row % T1, o2 = org2 + tt;
        const int co = ct * BCT + ch * 4;
        const bool in = item < NP * (BCT / 4) && o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && co < Cout;
        if (!in) dz |= 1u << k;
        else
          dr[k] = *reinterpret_cast<const float4*>(
              dy + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * Cout + co);
      }
    };
    auto put = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NXI; ++k) {
        const int item = tid + k * BNT;
        if (item >= HP * CHP) continue;
        const int hp = item / CHP, ch = item % CHP;
        int h = hp;
        const int c2 = h % G2; h /= G2;
        const int c1 = h % G1; h /= G1;
        const int u = W::tau(c2);
        char* cell = xs + ((h * G1 + c1) * G2 + u) * CB;
        *reinterpret_cast<uint4*>(cell + (((ch >> 1) ^ W::key(u)) << 5) + ((ch & 1) << 4)) =
            (xz >> k) & 1u ? make_uint4(0u, 0u, 0u, 0u) : xr[k];
      }
#pragma unroll
      for (int k = 0; k < NDI; ++k) {
        const int item = tid + k * BNT;
        if (item >= NP * (BCT / 4)) continue;
        const int pl = item >> 3, ch = item & 7;
        const float4 v = (dz >> k) & 1u ? make_float4(0.f, 0.f, 0.f, 0.f) : dr[k];
        *reinterpret_cast<uint2*>(ds + pl * 64 + (((ch >> 2) ^ ((pl >> 3) & 1)) << 5) + ((ch & 3) << 3)) =
            make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
      }
    };
    if ((int)xt_lo + xt_k < (int)xt_hi) {
      fetch((int)xt_lo + xt_k);
      put();
    }
    __syncthreads();
    for (int tile = (int)xt_lo + xt_k; tile < (int)xt_hi; tile += xt_nk) {
      const int next = tile + xt_nk;
      if (next < (int)xt_hi) fetch(next);
#pragma unroll
      for (int ks = 0; ks < NP / 32; ++ks) {
        const int rowb = ((((2 * ks) / T1) * STR * G1) + ((2 * ks) % T1) * STR) * G2 * CB;
        bf16x8 bfr[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          const s16x4 lo = lds_tr(ds + b_off[nb][0] + ks * 32 * 64);
          const s16x4 hi = lds_tr(ds + b_off[nb][1] + ks * 32 * 64);
          bfr[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const s16x4 lo = lds_tr(xs + a_off[c][0] + rowb + b * G2 * CB);
            const s16x4 hi = lds_tr(xs + a_off[c][1] + rowb + b * G2 * CB);
            const bf16x8 afr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
              acc[b * 3 + c][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nb], acc[b * 3 + c][nb], 0, 0, 0);
          }
      }
      __syncthreads();
      if (next < (int)xt_hi) put();
      __syncthreads();
    }
  } else
  for (int tile = (int)xt_lo + xt_k; tile < (int)xt_hi; tile += xt_nk) {
    int tr = tile;
    const int t2i = tr % tiles2; tr /= tiles2;
    const int t1i = tr % tiles1; tr /= tiles1;
    const int t0i = tr % tiles0; tr /= tiles0;
    const int n = tr;
    const int org0 = t0i * W::T0, org1 = t1i * T1, org2 = t2i * T2;
    __syncthreads();
    // ---- stage the x halo: cells x (CIB * 4) float4 -> bf16 (IN16: CIB * 2
    // 16-B chunks of a bf16 tensor, no convert)
    constexpr int CH = IN16 ? CIB * 2 : CIB * 4;
    constexpr int CW = IN16 ? 8 : 4;              // channels per item
    for (int item = tid; item < HP * CH; item += BNT) {
      const int hp = item / CH, ch = item % CH;
      int h = hp;
      const int c2 = h % G2; h /= G2;
      const int c1 = h % G1; h /= G1;
      const int c0 = h;
      int i0 = org0 * STR + c0 - g.lo[0], i1 = org1 * STR + c1 - g.lo[1],
          i2 = org2 * STR + c2 - g.lo[2];
      if (g.pad_mode == S3_PAD_REFLECT) {
        i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
      }
      // outside the tensor: zero padding, or cells feeding only masked outputs
      const bool valid = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2 &&
                         ci0 + ch * CW < Cin;
      const size_t e = ((((size_t)n * D0 + i0) * D1 + i1) * D2 + i2) * Cin + ci0 + ch * CW;
      const int u = W::tau(c2);
      char* cell = xs + ((c0 * G1 + c1) * G2 + u) * CB;
      if constexpr (IN16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (valid) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(x) + e);
        *reinterpret_cast<uint4*>(cell + (((ch >> 1) ^ W::key(u)) << 5) + ((ch & 1) << 4)) = v;
      } else {
        float4 v = make_float4(0, 0, 0, 0);
        if (valid) v = *reinterpret_cast<const float4*>(x + e);
        const uint2 hi = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
        char* at = cell + (((ch >> 2) ^ W::key(u)) << 5) + ((ch & 3) << 3);
        *reinterpret_cast<uint2*>(at) = hi;
        if constexpr (X3)
          *reinterpret_cast<uint2*>(at + W::XS) = make_uint2(
              pk2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xFFFF0000u)),
              pk2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xFFFF0000u)));
      }
    }
    // ---- stage the dPre tile: NP positions x 8 float4
    for (int item = tid; item < NP * (BCT / 4); item += BNT) {
      const int pl = item >> 3, ch = item & 7;
      const int row = pl / T2, tt = pl % T2;
      const int o0 = org0 + row / T1, o1 = org1 + row % T1, o2 = org2 + tt;
      const int co = ct * BCT + ch * 4;
      float4 v = make_float4(0, 0, 0, 0);
      if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2] && co < Cout)
        v = *reinterpret_cast<const float4*>(
            dy + ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * Cout + co);
      const uint2 hi = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
      char* at = ds + pl * 64 + (((ch >> 2) ^ ((pl >> 3) & 1)) << 5) + ((ch & 3) << 3);
      *reinterpret_cast<uint2*>(at) = hi;
      if constexpr (X3)
        *reinterpret_cast<uint2*>(at + NP * 64) = make_uint2(
            pk2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xFFFF0000u)),
            pk2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xFFFF0000u)));
    }
    __syncthreads();
    // ---- k-steps of 32 positions (2 rows x 16 t)
#pragma unroll
    for (int ks = 0; ks < NP / 32; ++ks) {
      // rows 2 ks, 2 ks + 1 of the tile: r0 = (2 ks) / T1, r1 = (2 ks) % T1 + (kg >> 1)
      const int rowb = ((((2 * ks) / T1) * STR * G1) + ((2 * ks) % T1) * STR) * G2 * CB;
      bf16x8 bfr[NBW], bfl[X3 ? NBW : 1];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const s16x4 lo = lds_tr(ds + b_off[nb][0] + ks * 32 * 64);
        const s16x4 hi = lds_tr(ds + b_off[nb][1] + ks * 32 * 64);
        bfr[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (X3) {
          const s16x4 l0 = lds_tr(dl + b_off[nb][0] + ks * 32 * 64);
          const s16x4 l1 = lds_tr(dl + b_off[nb][1] + ks * 32 * 64);
          bfl[nb] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const s16x4 lo = lds_tr(xs + a_off[c][0] + rowb + b * G2 * CB);
          const s16x4 hi = lds_tr(xs + a_off[c][1] + rowb + b * G2 * CB);
          const bf16x8 afr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          bf16x8 afl = afr;
          if constexpr (X3) {
            const s16x4 l0 = lds_tr(xl + a_off[c][0] + rowb + b * G2 * CB);
            const s16x4 l1 = lds_tr(xl + a_off[c][1] + rowb + b * G2 * CB);
            afl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) {
            if constexpr (X3) {   // small terms first
              acc[b * 3 + c][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl, bfr[nb], acc[b * 3 + c][nb], 0, 0, 0);
              acc[b * 3 + c][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfl[nb], acc[b * 3 + c][nb], 0, 0, 0);
            }
            acc[b * 3 + c][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nb], acc[b * 3 + c][nb], 0, 0, 0);
          }
        }
    }
  }
  float* out = partial + (size_t)blockIdx.x * 27 * Cin * Cout;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int co = ct * BCT + (nb0 + nb) * 16 + q;
    if (co < Cout) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = ci0 + cb * 16 + kg * 4 + r;
          if (ci < Cin) out[((size_t)(ta * 9 + t) * Cin + ci) * Cout + co] = acc[t][nb][r];
        }
    }
  }
}

template <int CIB, int STR>
int bf_gen_grid(const s3_ctx* ctx, const ConvGeom& g, int* n_tiles, int* t0, int* t1, int* t2) {
  using W = BfGen<CIB, STR>;
  *t0 = (g.O[0] + W::T0 - 1) / W::T0; *t1 = (g.O[1] + W::T1 - 1) / W::T1;
  *t2 = (g.O[2] + W::T2 - 1) / W::T2;
  *n_tiles = g.N * *t0 * *t1 * *t2;
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  const int n_cit = (g.Cin + CIB * 16 - 1) / (CIB * 16);
  int grid = ctx->num_cu / (n_ct * n_cit);
  if (grid < 1) grid = 1;
  if (grid > *n_tiles) grid = *n_tiles;
  return grid;
}

// BF16X3 plans: 32-channel C_in tiles, hi + lo images (2 x the LDS of the bf16 form)
template <int STR>
int bf_gen_launch_x3(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy, float* dw,
                     float* partial, size_t partial_bytes, int accumulate) {
  using W = BfGen<2, STR>;
  int n_tiles, t0, t1, t2;
  const int grid = bf_gen_grid<2, STR>(ctx, g, &n_tiles, &t0, &t1, &t2);
  const size_t need = (size_t)grid * 27 * g.Cin * g.Cout * sizeof(float);
  if (partial_bytes < need) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16_gen: partial buffer too small");
  auto kern = conv_wgrad_bf16_gen_kernel<2, STR, false, false, true>;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * W::LDS)));
    attr_set.mark(ctx->device);
  }
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  const int n_cit = (g.Cin + 31) / 32;
  hipLaunchKernelGGL(kern, dim3(grid, n_ct, n_cit), dim3(BNT), 2 * W::LDS, ctx->stream, x, dy, partial,
                     g, t0, t1, t2, n_tiles);
  S3_HIP(ctx, hipGetLastError());
  const int64_t wsize = (int64_t)27 * g.Cin * g.Cout;
  int rg = (int)((wsize + 255) / 256);
  if (rg > 4096) rg = 4096;
  return launch_wgrad_bf16_reduce(ctx, partial, grid, wsize, dw, accumulate, rg);
}

template <int CIB, int STR, bool IN16 = false>
int bf_gen_launch(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy, float* dw,
                  float* partial, size_t partial_bytes, int accumulate) {
  using W = BfGen<CIB, STR>;
  int n_tiles, t0, t1, t2;
  const int grid = bf_gen_grid<CIB, STR>(ctx, g, &n_tiles, &t0, &t1, &t2);
  const size_t need = (size_t)grid * 27 * g.Cin * g.Cout * sizeof(float);
  if (partial_bytes < need) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16_gen: partial buffer too small");
  constexpr bool CAN_PF = CIB == 2 && IN16;
  auto kern = conv_wgrad_bf16_gen_kernel<CIB, STR, IN16>;
  if constexpr (CAN_PF)
    if (!s3_opt_has(S3O_NO_WGRAD_GEN_PF)) kern = conv_wgrad_bf16_gen_kernel<CIB, STR, IN16, true>;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_gen_kernel<CIB, STR, IN16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS));
    if constexpr (CAN_PF)
      S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_gen_kernel<CIB, STR, IN16, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS));
    attr_set.mark(ctx->device);
  }
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  const int n_cit = (g.Cin + CIB * 16 - 1) / (CIB * 16);
  hipLaunchKernelGGL(kern, dim3(grid, n_ct, n_cit), dim3(BNT), W::LDS, ctx->stream, x, dy, partial,
                     g, t0, t1, t2, n_tiles);
  S3_HIP(ctx, hipGetLastError());
  const int64_t wsize = (int64_t)27 * g.Cin * g.Cout;
  int rg = (int)((wsize + 255) / 256);
  if (rg > 4096) rg = 4096;
  return launch_wgrad_bf16_reduce(ctx, partial, grid, wsize, dw, accumulate, rg);
}


// ===========================================================================
// 2-D convs (spatial models: k = 3 x 3 x 1 on (N, s1, s2, 1, C)): same scheme
// with s2 as the 16-long run axis.  Tiles of 8 x 16 positions (halo 10 x 18
// cells, stride 2: 17 x 33), 12 waves = (s1 tap) x (ci block[, cout block]),
// 3 taps x NBW accumulators per wave, 4 k-steps per tile.
template <int CIB, int STR>
struct Bf2D {
  static constexpr int T1 = 8, T2 = 16;
  static constexpr int G1 = (T1 - 1) * STR + 3, G2 = (T2 - 1) * STR + 3;
  static constexpr int HP = G1 * G2;
  static constexpr int NP = T1 * T2;
  static constexpr int CB = CIB * 32;
  static constexpr int NBW = CIB / 2;
  static constexpr int XS = HP * CB;
  static constexpr size_t LDS = (size_t)XS + NP * 64;
  __device__ static __forceinline__ int tau(int th) {
    return STR == 1 ? th : (th & 1) * ((G2 + 1) / 2) + (th >> 1);
  }
  __device__ static __forceinline__ int key(int u) {
    return CIB == 4 ? (((u >> 1) & 1) | (((u >> 3) & 1) << 1)) : ((u >> 3) & 1);
  }
};

// IN16: x is bf16 cells (C_in % 8 == 0; the activations a 2-D training plan
// saves behind the weights-stationary / logical-axes forward kernels)
// DY16: dPre is bf16 cells as well (C_out % 4 == 0; the masked fold of the
// consumer's data gradient stored it as bf16 only)
template <int CIB, int STR, bool IN16 = false, bool DY16 = false>
__global__ __launch_bounds__(BNT) void conv2_wgrad_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int tiles1, int tiles2, int n_tiles) {
  using W = Bf2D<CIB, STR>;
  constexpr int T1 = W::T1, T2 = W::T2, G2 = W::G2, HP = W::HP, NP = W::NP;
  constexpr int CB = W::CB, NBW = W::NBW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ds = smem + W::XS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, kg = lane >> 4;
  const int tb = wave >> 2, w4 = wave & 3;
  const int cb = CIB == 4 ? w4 : (w4 & 1);
  const int nb0 = CIB == 4 ? 0 : (w4 >> 1);
  const int ct = blockIdx.y;
  const int ci0 = blockIdx.z * (CIB * 16);
  const int D0 = g.D[0], D1 = g.D[1];
  const int Cin = g.Cin, Cout = g.Cout;

  f32x4 acc[3][NBW];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int a_off[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int th = (8 * (kg & 1) + 4 * h + (q >> 2)) * STR + c;
      const int u = W::tau(th);
      a_off[c][h] = ((tb + (kg >> 1) * STR) * G2 + u) * CB + ((cb ^ W::key(u)) << 5) + ((q & 3) << 3);
    }
  int b_off[NBW][2];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = 8 * kg + 4 * h + (q >> 2);
      b_off[nb][h] = pl * 64 + (((nb0 + nb) ^ ((pl >> 3) & 1)) << 5) + ((q & 3) << 3);
    }

  // (XCD-contiguous tile shares: s3_xcd_share, common.h)
  int64_t xt_lo, xt_hi;
  int xt_k, xt_nk;
  s3_xcd_share(n_tiles, xt_lo, xt_hi, xt_k, xt_nk);
  for (int tile = (int)xt_lo + xt_k; tile < (int)xt_hi; tile += xt_nk) {
    int tr = tile;
    const int t2i = tr % tiles2; tr /= tiles2;
    const int t1i = tr % tiles1; tr /= tiles1;
    const int n = tr;
    const int org1 = t1i * T1, org2 = t2i * T2;
    __syncthreads();
    constexpr int CH = CIB * 4;
    if constexpr (IN16) {
      constexpr int CH8 = CIB * 2;
      const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
      for (int item = tid; item < HP * CH8; item += BNT) {
        const int hp = item / CH8, ch = item % CH8;
        const int c2 = hp % G2, c1 = hp / G2;
        int i0 = org1 * STR + c1 - g.lo[0], i1 = org2 * STR + c2 - g.lo[1];
        if (g.pad_mode == S3_PAD_REFLECT) { i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); }
        const bool valid = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && ci0 + ch * 8 < Cin;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (valid)
          v = *reinterpret_cast<const uint4*>(x16 + (((size_t)n * D0 + i0) * D1 + i1) * Cin + ci0 + ch * 8);
        const int u = W::tau(c2);
        *reinterpret_cast<uint4*>(xs + (c1 * G2 + u) * CB + (((ch >> 1) ^ W::key(u)) << 5) + ((ch & 1) << 4)) = v;
      }
    } else
    for (int item = tid; item < HP * CH; item += BNT) {
      const int hp = item / CH, ch = item % CH;
      const int c2 = hp % G2, c1 = hp / G2;
      int i0 = org1 * STR + c1 - g.lo[0], i1 = org2 * STR + c2 - g.lo[1];
      if (g.pad_mode == S3_PAD_REFLECT) { i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); }
      const bool valid = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && ci0 + ch * 4 < Cin;
      float4 v = make_float4(0, 0, 0, 0);
      if (valid) {
        const float* xp = x + (((size_t)n * D0 + i0) * D1 + i1) * Cin + ci0 + ch * 4;
        if ((Cin & 3) == 0) {
          v = *reinterpret_cast<const float4*>(xp);
        } else {            // (C_in 1, 2, 3, 6, 7, 65: channel by channel, zeros behind the last)
          const int left = Cin - (ci0 + ch * 4);
          v.x = xp[0];
          if (left > 1) v.y = xp[1];
          if (left > 2) v.z = xp[2];
          if (left > 3) v.w = xp[3];
        }
      }
      const int u = W::tau(c2);
      *reinterpret_cast<uint2*>(xs + (c1 * G2 + u) * CB + (((ch >> 2) ^ W::key(u)) << 5) +
                                ((ch & 3) << 3)) = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
    }
    for (int item = tid; item < NP * (BCT / 4); item += BNT) {
      const int pl = item >> 3, ch = item & 7;
      const int o0 = org1 + pl / T2, o1 = org2 + pl % T2;
      const int co = ct * BCT + ch * 4;
      if constexpr (DY16) {
        uint2 v16 = make_uint2(0, 0);
        if (o0 < g.O[0] && o1 < g.O[1] && co < Cout)
          v16 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dy) +
                                                (((size_t)n * g.O[0] + o0) * g.O[1] + o1) * Cout + co);
        *reinterpret_cast<uint2*>(ds + pl * 64 + (((ch >> 2) ^ ((pl >> 3) & 1)) << 5) + ((ch & 3) << 3)) = v16;
        continue;
      }
      float4 v = make_float4(0, 0, 0, 0);
      if (o0 < g.O[0] && o1 < g.O[1] && co < Cout) {
        const float* dp = dy + (((size_t)n * g.O[0] + o0) * g.O[1] + o1) * Cout + co;
        if ((Cout & 3) == 0) {
          v = *reinterpret_cast<const float4*>(dp);
        } else {            // (C_out 1, 2, 6, 14)
          const int left = Cout - co;
          v.x = dp[0];
          if (left > 1) v.y = dp[1];
          if (left > 2) v.z = dp[2];
          if (left > 3) v.w = dp[3];
        }
      }
      *reinterpret_cast<uint2*>(ds + pl * 64 + (((ch >> 2) ^ ((pl >> 3) & 1)) << 5) + ((ch & 3) << 3)) =
          make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < NP / 32; ++ks) {
      const int rowb = (2 * ks) * STR * G2 * CB;
      bf16x8 bfr[NBW];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const s16x4 lo = lds_tr(ds + b_off[nb][0] + ks * 32 * 64);
        const s16x4 hi = lds_tr(ds + b_off[nb][1] + ks * 32 * 64);
        bfr[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const s16x4 lo = lds_tr(xs + a_off[c][0] + rowb);
        const s16x4 hi = lds_tr(xs + a_off[c][1] + rowb);
        const bf16x8 afr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
          acc[c][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nb], acc[c][nb], 0, 0, 0);
      }
    }
  }
  float* out = partial + (size_t)blockIdx.x * 9 * Cin * Cout;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int co = ct * BCT + (nb0 + nb) * 16 + q;
    if (co < Cout) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = ci0 + cb * 16 + kg * 4 + r;
          if (ci < Cin) out[((size_t)(tb * 3 + c) * Cin + ci) * Cout + co] = acc[c][nb][r];
        }
    }
  }
}

template <int CIB, int STR>
int bf_2d_grid(const s3_ctx* ctx, const ConvGeom& g, int* n_tiles, int* t1, int* t2) {
  using W = Bf2D<CIB, STR>;
  *t1 = (g.O[0] + W::T1 - 1) / W::T1; *t2 = (g.O[1] + W::T2 - 1) / W::T2;
  *n_tiles = g.N * *t1 * *t2;
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  const int n_cit = (g.Cin + CIB * 16 - 1) / (CIB * 16);
  // one partial per workgroup: more workgroups than CUs would make the
  // fixed-order reduction (grid x 9 x C_in x C_out floats) the larger kernel
  int grid = ctx->num_cu / (n_ct * n_cit);
  if (grid < 1) grid = 1;
  if (grid > *n_tiles) grid = *n_tiles;
  return grid;
}

template <int CIB, int STR, bool IN16 = false, bool DY16 = false>
int bf_2d_launch(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy, float* dw,
                 float* partial, size_t partial_bytes, int accumulate) {
  using W = Bf2D<CIB, STR>;
  int n_tiles, t1, t2;
  const int grid = bf_2d_grid<CIB, STR>(ctx, g, &n_tiles, &t1, &t2);
  const size_t need = (size_t)grid * 9 * g.Cin * g.Cout * sizeof(float);
  if (partial_bytes < need) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16_2d: partial buffer too small");
  auto kern = conv2_wgrad_bf16_kernel<CIB, STR, IN16, DY16>;
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS));
    attr_set.mark(ctx->device);
  }
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  const int n_cit = (g.Cin + CIB * 16 - 1) / (CIB * 16);
  hipLaunchKernelGGL(kern, dim3(grid, n_ct, n_cit), dim3(BNT), W::LDS, ctx->stream, x, dy, partial,
                     g, t1, t2, n_tiles);
  S3_HIP(ctx, hipGetLastError());
  const int64_t wsize = (int64_t)9 * g.Cin * g.Cout;
  int rg = (int)((wsize + 255) / 256);
  if (rg > 4096) rg = 4096;
  return launch_wgrad_bf16_reduce(ctx, partial, grid, wsize, dw, accumulate, rg);
}

}  // namespace

bool conv_wgrad_bf16_supported(const ConvGeom& g, int precision) {
  // (BF16X3 plans: conv3_wgrad_x3_kernel, fp32 operands split in the staging)
  if (precision == S3_PREC_BF16X3 ? s3_opt_has(S3O_NO_WGRAD_X3) : precision != S3_PREC_BF16) return false;
  if (s3_opt_has(S3O_NO_WGRAD_BF16)) return false;
  if (g.Cin != 64 || g.Cout % 4 != 0 || g.Cout < 16) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] != 1 || g.O[d] != g.D[d]) return false;
  // enough positions to amortise the 27 x 64 x C_out partial per workgroup
  return g.D[2] >= 8 && (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] >= 2048;
}

size_t conv_wgrad_bf16_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  int nt, a, b, c;
  const int grid = bf_grid(ctx, g, &nt, &a, &b, &c);
  return (size_t)grid * 27 * 64 * g.Cout * sizeof(float);
}

int launch_conv_wgrad_bf16(s3_ctx* ctx, const ConvGeom& g, const float* x,
                           const float* dy, float* dw, float* partial,
                           size_t partial_bytes, int accumulate, int x_bf16, int dy_bf16, int x3) {
  int n_tiles, tiles0, tiles1, tiles2;
  const int grid = bf_grid(ctx, g, &n_tiles, &tiles0, &tiles1, &tiles2);
  if (partial_bytes < conv_wgrad_bf16_partial_bytes(ctx, g))
    S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16: partial buffer too small");
  if (x3 && (x_bf16 || dy_bf16)) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16: the split-bf16 kernel takes fp32 operands");
  if (dy_bf16 && (!x_bf16 || (g.Cout & 3))) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16: bf16 dPre needs bf16 x and C_out % 4 == 0");
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_bf16_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_bf16_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_bf16_kernel<true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_bf16_ws_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_x3_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS));
    attr_set.mark(ctx->device);
  }
  const int n_ct = (g.Cout + BCT - 1) / BCT;
  const int dbg = (int)s3_opt_int(S3O_WGRAD_DBG, 0);
  // wave-specialised variant: reflect padding, tiles that fit exactly, whole 16-B channel chunks
  const bool ws = dy_bf16 && !dbg && !s3_opt_has(S3O_NO_WGRAD_WS) && g.pad_mode == S3_PAD_REFLECT &&
                  g.O[0] % BT0 == 0 && g.O[1] % BT1 == 0 && g.O[2] % BT2 == 0 && g.Cout >= BCT && g.Cout % 8 == 0 &&
                  (int64_t)g.D[0] * g.D[1] * g.D[2] * 64 < ((int64_t)1 << 31) &&
                  (int64_t)g.O[0] * g.O[1] * g.O[2] * g.Cout < ((int64_t)1 << 31);
  if (x3)
    hipLaunchKernelGGL(conv3_wgrad_x3_kernel, dim3(grid, n_ct), dim3(BNT), X3_LDS, ctx->stream,
                       x, dy, partial, g, tiles0, tiles1, tiles2, n_tiles);
  else if (ws)
    hipLaunchKernelGGL(conv3_wgrad_bf16_ws_kernel, dim3(grid, n_ct), dim3(WS_NT), WS_LDS, ctx->stream,
                       (const unsigned short*)x, (const unsigned short*)dy, partial, g, tiles0, tiles1, tiles2,
                       n_tiles);
  else if (dy_bf16)
    hipLaunchKernelGGL((conv3_wgrad_bf16_kernel<true, true>), dim3(grid, n_ct), dim3(BNT), BF_LDS,
                       ctx->stream, x, dy, partial, g, tiles0, tiles1, tiles2, n_tiles, dbg);
  else if (x_bf16)
    hipLaunchKernelGGL(conv3_wgrad_bf16_kernel<true>, dim3(grid, n_ct), dim3(BNT), BF_LDS,
                       ctx->stream, x, dy, partial, g, tiles0, tiles1, tiles2, n_tiles, dbg);
  else
    hipLaunchKernelGGL(conv3_wgrad_bf16_kernel<false>, dim3(grid, n_ct), dim3(BNT), BF_LDS,
                       ctx->stream, x, dy, partial, g, tiles0, tiles1, tiles2, n_tiles, dbg);
  S3_HIP(ctx, hipGetLastError());
  const int64_t wsize = (int64_t)27 * 64 * g.Cout;
  return launch_wgrad_bf16_reduce(ctx, partial, grid, wsize, dw, accumulate, (int)((wsize + 255) / 256));
}

// ---- general variant (discriminator convs)
bool conv_wgrad_bf16_gen_supported(const ConvGeom& g, int precision) {
  if (precision == S3_PREC_BF16X3 ? s3_opt_has(S3O_NO_WGRAD_X3) : precision != S3_PREC_BF16) return false;
  if (s3_opt_has(S3O_NO_WGRAD_BF16)) return false;
  // (dPre is in the conv's own [position][C_out] layout whatever the store
  // permutation of its forward pass: a depth-to-space conv is no special case)
  if (g.Cin % 32 != 0 || g.Cin < 32 || g.Cout % 4 != 0 || g.Cout < 16) return false;
  if (g.Cin > 64 && g.Cin % 64 != 0) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != g.s[0] || (g.s[d] != 1 && g.s[d] != 2)) return false;
  // (t extents below the 16-wide run of a tile are masked through dPre: still
  // several times the exact-fp32 MFMA kernel's rate — 256 -> 256 s2 at 3 x 3 x 6
  // outputs, batch 32: 388 us there)
  // (round 5: T = 3 — sup3rcc/gen_solar_1x_8x_1f — took the generic kernel at
  // 480 us per conv; 13 of 16 run positions masked is still 10x that)
  return g.O[2] >= 2 && (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] >= 512;
}

size_t conv_wgrad_bf16_gen_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  return (size_t)ctx->num_cu * 27 * g.Cin * g.Cout * sizeof(float);   // grid <= CU count
}

int launch_conv_wgrad_bf16_gen(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                               float* dw, float* partial, size_t partial_bytes, int accumulate,
                               int x_bf16, int x3) {
  const bool s2 = g.s[0] == 2;
  if (x3) {
    if (x_bf16) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16_gen: the split-bf16 kernel takes fp32 operands");
    return s2 ? bf_gen_launch_x3<2>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
              : bf_gen_launch_x3<1>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
  }
  if (x_bf16) {
    if (g.Cin == 32)
      return s2 ? bf_gen_launch<2, 2, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
                : bf_gen_launch<2, 1, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
    return s2 ? bf_gen_launch<4, 2, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
              : bf_gen_launch<4, 1, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
  }
  if (g.Cin == 32)
    return s2 ? bf_gen_launch<2, 2>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
              : bf_gen_launch<2, 1>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
  return s2 ? bf_gen_launch<4, 2>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
            : bf_gen_launch<4, 1>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
}

// ---- 2-D variant (spatial models)
bool conv_wgrad_bf16_2d_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_WGRAD_BF16)) return false;
  // (dPre is in the conv's own [position][C_out] layout whatever the store
  // permutation of its forward pass: a depth-to-space conv is no special case)
  // (round 5: any channel counts — the staging pads a ragged C_in / C_out with
  // zeros channel by channel.  The head and output convs of the 2-D specs,
  // 2 / 3 / 7 -> 64, 64 -> 1 / 2 / 6, and the 65 -> 64 conv behind a
  // Sup3rConcat were 0.85 of a 4.6 ms fwd + bwd on the generic kernel)
  if (g.Cin < 1 || g.Cout < 1) return false;
  if (g.k[0] != 3 || g.k[1] != 3 || g.k[2] != 1 || g.D[2] != 1 || g.O[2] != 1) return false;
  if (g.s[0] != g.s[1] || (g.s[0] != 1 && g.s[0] != 2)) return false;
  return g.O[1] >= 8 && (int64_t)g.N * g.O[0] * g.O[1] >= 1024;
}

size_t conv_wgrad_bf16_2d_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  return (size_t)ctx->num_cu * 9 * g.Cin * g.Cout * sizeof(float);   // grid <= CU count
}

int launch_conv_wgrad_bf16_2d(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                              float* dw, float* partial, size_t partial_bytes, int accumulate, int x_bf16,
                              int dy_bf16) {
  const bool s2 = g.s[0] == 2;
  if (dy_bf16) {
    if (!x_bf16 || g.Cin % 8 != 0 || g.Cout % 4 != 0 || s2)
      S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16_2d: bf16 dPre needs bf16 x cells, stride 1, C_in % 8 == 0, C_out % 4 == 0");
    if (g.Cin <= 32) return bf_2d_launch<2, 1, true, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
    return bf_2d_launch<4, 1, true, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
  }
  if (x_bf16) {
    if (g.Cin % 8 != 0) S3_FAIL(ctx, S3_EINVAL, "wgrad_bf16_2d: bf16 cells need C_in % 8 == 0");
    if (g.Cin <= 32)
      return s2 ? bf_2d_launch<2, 2, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
                : bf_2d_launch<2, 1, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
    return s2 ? bf_2d_launch<4, 2, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
              : bf_2d_launch<4, 1, true>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
  }
  if (g.Cin <= 32)
    return s2 ? bf_2d_launch<2, 2>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
              : bf_2d_launch<2, 1>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
  return s2 ? bf_2d_launch<4, 2>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate)
            : bf_2d_launch<4, 1>(ctx, g, x, dy, dw, partial, partial_bytes, accumulate);
}
