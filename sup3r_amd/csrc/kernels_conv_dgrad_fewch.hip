// Data gradient of the 2-channel hi-res conv that opens the discriminator
// (forward 2 -> 32 over N x 78 x 78 x 286 positions; the gradient reaches the
// generator through it) on bf16 MFMA with an LDS halo — S3_PREC_BF16 plans.
//
//   dx[i][ci] = sum_{tap, co} W[tap][ci][co] * dPre[i + lo - tap][co]
//
// A 32 -> 2 "full correlation": 864 MACs per output value, and each dPre cell
// (32 channels) feeds 27 taps.  The gather kernel re-reads a cell 27 times
// through L1 (6.2 ms at C2 batch 8); here a workgroup stages the
// (4+2) x (8+2) x (16+2) dPre halo of its 4 x 8 x 16 output tile ONCE into LDS
// (fp32 -> bf16, 64-B cells, 16-B chunks XOR-swizzled by (t >> 1) & 3:
// conflict-free for the four 16-lane groups of ds_read_b128 at all three tap
// shifts, brute-forced) and every tap reads its shifted window as the MFMA B
// operand (K = 32 output channels of the forward conv = one k-step).  The A
// operand is the flipped filter: rows = the C_in <= 4 input channels (the
// other rows are zero), all 27 fragments live in registers.  Lanes 0..15 own
// the C_in values of 16 consecutive t: one contiguous store per fragment.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int DT0 = 4, DT1 = 8, DT2 = 16;
constexpr int DH0 = DT0 + 2, DH1 = DT1 + 2, DH2 = DT2 + 2;
constexpr int DHP = DH0 * DH1 * DH2;         // 1080 halo cells
constexpr int DNW = 4;                       // waves; 8 (s1, s2) rows each
constexpr int DNT = DNW * 64;
constexpr int DLDS = DHP * 64;               // 69,120 B

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

// fp32 w[tap][cin][32] -> bf16 img[tap'][16 rows][32], tap' = 26 - tap, rows >= cin zero
__global__ void dgrad_c2_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                     int cin) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 27 * 16 * 32;
       idx += gridDim.x * blockDim.x) {
    const int co = idx & 31, row = (idx >> 5) & 15, tp = idx >> 9;
    const float v = row < cin ? w[((size_t)(26 - tp) * cin + row) * 32 + co] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

__global__ __launch_bounds__(DNT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dgrad_c2_kernel(
    const float* __restrict__ dy, const unsigned short* __restrict__ img,
    float* __restrict__ dx, ConvGeom g, int tiles0, int tiles1, int tiles2, int dy16) {
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

  // ---- stage the dPre halo: cell (c0, c1, c2) = dPre[org + c + lo - 2], zero outside
  // (dy16: dPre is a bf16 tensor — the stride-2 layer above stored it that way —
  // one 16-B chunk per item, no convert)
  if (dy16) {
    const unsigned short* d16 = reinterpret_cast<const unsigned short*>(dy);
    for (int base = tid; base < DHP * 4; base += DNT * 3) {
      uint4 v[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int item = base + u * DNT;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (item < DHP * 4) {
          const int hp = item >> 2, ch = item & 3;
          int h = hp;
          const int c2 = h % DH2; h /= DH2;
          const int c1 = h % DH1; h /= DH1;
          const int c0 = h;
          const int i0 = org0 + c0 + g.lo[0] - 2, i1 = org1 + c1 + g.lo[1] - 2,
                    i2 = org2 + c2 + g.lo[2] - 2;
          if (i0 >= 0 && i0 < O0 && i1 >= 0 && i1 < O1 && i2 >= 0 && i2 < O2)
            v[u] = *reinterpret_cast<const uint4*>(d16 + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int item = base + u * DNT;
        if (item < DHP * 4) {
          const int hp = item >> 2, ch = item & 3;
          const int key = ((hp % DH2) >> 1) & 3;
          *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) = v[u];
        }
      }
    }
  } else
  for (int base = tid; base < DHP * 4; base += DNT * 3) {
    float4 va[3], vb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
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
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * DNT;
      if (item < DHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        const int key = ((hp % DH2) >> 1) & 3;
        *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) =
            make_uint4(pk2(va[u].x, va[u].y), pk2(va[u].z, va[u].w), pk2(vb[u].x, vb[u].y),
                       pk2(vb[u].z, vb[u].w));
      }
    }
  }
  // ---- filter fragments: lane (row = j, kg) holds co 8 kg .. 8 kg + 7 of every tap
  // (loaded after the staging so that its registers are free during it)
  bf16x8 afr[27];
#pragma unroll
  for (int tp = 0; tp < 27; ++tp)
    afr[tp] = *reinterpret_cast<const bf16x8*>(img + (tp * 16 + j) * 32 + kg * 8);
  __syncthreads();

  // ---- 8 rows per wave x 27 taps
  int off_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) off_c[c] = (j + c) * 64 + ((kg ^ (((j + c) >> 1) & 3)) << 4);
  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int tp = (a * 3 + b) * 3 + c;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          // row r = wave * 8 + m: r0 = r / 8 = wave, r1 = m
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(
              halo + (((wave + a) * DH1 + (m + b)) * DH2) * 64 + off_c[c]);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[tp], bfr, acc[m], 0, 0, 0);
        }
      }

  // ---- C/D: col = lane & 15 (t), row = 4 kg + r (ci): lanes kg == 0 own ci 0..3
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


// ---------------------------------------------------------------------------
// Round 3: the same gradient as a SLIDING WINDOW with the three t-taps packed
// into the MFMA's M dimension.
//
// conv_dgrad_c2_kernel fetches a 6 x 10 x 18-cell halo (69 KB) per 512 outputs
// = 2.1 cells per output cell through the ~10 B/clk L2 -> CU path, then runs
// 216 MFMAs per wave of which 2 of 16 A rows carry data: 645 us = 0.19 of the
// HBM roofline at C2 batch 8 (profiles/r02/pmc_train_final.txt).  Here:
//   * a workgroup owns a column of 16 (s1) x 14 (t) outputs and walks it along
//     s0; output row r needs the dPre planes r .. r + 2 (18 x 16 cells of 64 B,
//     the 16th / 17th t cell are the window's own halo), one NEW plane per row
//     into a 5-slot LDS ring by LDS-DMA (bf16 dPre in HBM is bf16 in LDS, no
//     registers; two planes in flight): 1.47 cells fetched per output cell
//     instead of 2.1, the fetch spread over the compute;
//   * A = the flipped filter with rows (c, ci) = the three t-taps x C_in = 2:
//     ONE MFMA per (a, b) tap yields, for 16 consecutive cells and K = 32
//     forward output channels, the three t-tap partial sums of both input
//     channels; dx[t] = P_0[t] + P_1[t + 1] + P_2[t + 2] is two ds_bpermute
//     shifts per value.  9 MFMAs and 9 ds_read_b128 per 14 outputs instead of
//     27 and 27 per 16.
// Out-of-range cells / planes (the zero boundary of the full correlation) are
// DMA'd from a 64-byte run of zeros (row 15 of the packed filter image).
constexpr int GS1 = 16, GS2 = 14;            // outputs per row of a column
constexpr int GP1 = GS1 + 2, GP2 = 16;       // plane: 18 rows of 16 cells
constexpr int GPLANE = GP1 * GP2 * 64;       // 18,432 B
#ifndef GSLIDE_PD
#define GSLIDE_PD 2
#endif
#ifndef GSLIDE_SEG0
#define GSLIDE_SEG0 40
#endif
constexpr int GPD = GSLIDE_PD, GNSLOT = 3 + GPD;   // planes in flight, ring slots
constexpr int GNCW = 8, GNDW = 4;            // compute / staging waves
constexpr int GNT = (GNCW + GNDW) * 64;
constexpr int GSEG0 = GSLIDE_SEG0;           // s0 rows per work unit
constexpr int GLDS = GNSLOT * GPLANE;        // 92,160 B

// fp32 w[tap][cin = 2][32] -> bf16 img[ab][16 rows][32]: row 2 c + ci of image
// ab = a * 3 + b holds the flipped filter of halo offset (a, b, c), i.e.
// w[26 - ((a * 3 + b) * 3 + c)][ci][:]; rows 6 .. 15 are zero
__global__ void dgrad_c2_slide_pack_kernel(const float* __restrict__ w,
                                           unsigned short* __restrict__ img) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 9 * 16 * 32;
       idx += gridDim.x * blockDim.x) {
    const int co = idx & 31, row = (idx >> 5) & 15, ab = idx >> 9;
    float v = 0.f;
    if (row < 6) {
      const int c = row >> 1, ci = row & 1;
      v = w[((size_t)(26 - (ab * 3 + c)) * 2 + ci) * 32 + co];
    }
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

__global__ __launch_bounds__(GNT) void conv_dgrad_c2_slide_kernel(
    const unsigned short* __restrict__ dy, const unsigned short* __restrict__ img,
    float* __restrict__ dx, ConvGeom g, int segs0, int tiles1, int tiles2, int n_units) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2];

  // XCD-contiguous unit ranges (conv_tail_slide_kernel)
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
  auto unit_org = [&](int u, int& n, int& r0, int& o1, int& o2) __attribute__((always_inline)) {
    int tr = u;
    o2 = (tr % tiles2) * GS2; tr /= tiles2;
    o1 = (tr % tiles1) * GS1; tr /= tiles1;
    r0 = (tr % segs0) * GSEG0; tr /= segs0;
    n = tr;
  };
#define GSL_BARRIER() asm volatile("s_barrier" ::: "memory")

  if (wave >= GNCW) {
    // ---------------------------------------------------- staging waves
    // one 1-KB DMA piece per plane row: LDS slot s = lane -> cell s >> 2,
    // physical chunk s & 3 = logical chunk ^ ((cell >> 1) & 3)
    const int sw = wave - GNCW;
    const int cell = lane >> 2;
    const int lchunk = (lane & 3) ^ ((cell >> 1) & 3);
    const unsigned short* zeros = img + 15 * 32;          // row 15 of image 0: 64 B of zeros
    auto stage_plane = [&](int n, int pr, int o1, int o2, int slot) __attribute__((always_inline)) {
      // plane pr = dPre row pr + lo0 - 2 ... (x row o0 reads dPre rows o0 + a + lo0 - 2)
      const int i0 = pr + g.lo[0] - 2;
      const int i2 = o2 + cell + g.lo[2] - 2;
      const bool ok02 = i0 >= 0 && i0 < O0 && i2 >= 0 && i2 < O2;
      char* bufp = smem + slot * GPLANE;
      for (int row = sw; row < GP1; row += GNDW) {
        const int i1 = o1 + row + g.lo[1] - 2;
        const unsigned short* src = zeros + lchunk * 8;
        if (ok02 && i1 >= 0 && i1 < O1)
          src = dy + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + lchunk * 8;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(bufp + row * (GP2 * 64)), 16, 0, 0);
      }
    };
    const int plane_ops = (GP1 - sw + GNDW - 1) / GNDW;     // 5 or 4
    auto wait_planes = [&](int in_flight) __attribute__((always_inline)) {
      switch (in_flight * plane_ops) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
    for (int u = u_first; u < u_end; u += u_step) {
      int n, r0, o1, o2;
      unit_org(u, n, r0, o1, o2);
      const int rows = (r0 + GSEG0 <= g.D[0] ? GSEG0 : g.D[0] - r0);
      // planes r0 .. r0 + rows + 1 feed x rows r0 .. r0 + rows - 1; plane q
      // (slot q % GNSLOT) is needed from row q - 2 on
      int issued = 0;
      for (; issued < 2 + GPD && issued < rows + 2; ++issued)
        stage_plane(n, r0 + issued, o1, o2, issued % GNSLOT);
      wait_planes(issued - 3 > 0 ? issued - 3 : 0);          // planes 0 .. 2 are in
      GSL_BARRIER();
      for (int r = 0; r < rows; ++r) {
        // slot (r + 2 + GPD) % GNSLOT held plane r - 1: free since the last barrier
        if (issued < rows + 2) { stage_plane(n, r0 + issued, o1, o2, issued % GNSLOT); ++issued; }
        const int need = r + 4 < rows + 2 ? r + 4 : rows + 2;   // row r + 1 reads planes up to r + 3
        wait_planes(issued - need);
        GSL_BARRIER();
      }
    }
    return;
  }
  // ------------------------------------------------------ compute waves
  // wave w owns the s1 rows 2 w, 2 w + 1 of the column; lane (j = cell, kg)
  const int j = lane & 15, kg = lane >> 4;
  bf16x8 afr[9];
#pragma unroll
  for (int ab = 0; ab < 9; ++ab)
    afr[ab] = *reinterpret_cast<const bf16x8*>(img + (ab * 16 + j) * 32 + kg * 8);
  const unsigned lane_off = (unsigned)(j * 64 + ((kg ^ ((j >> 1) & 3)) << 4));
  for (int u = u_first; u < u_end; u += u_step) {
    int n, r0, o1, o2;
    unit_org(u, n, r0, o1, o2);
    const int rows = (r0 + GSEG0 <= g.D[0] ? GSEG0 : g.D[0] - r0);
    GSL_BARRIER();
    for (int r = 0; r < rows; ++r) {
      f32x4 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const unsigned pl = (unsigned)(((r + a) % GNSLOT) * GPLANE);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(
                smem + pl + (unsigned)((2 * wave + m + b) * (GP2 * 64)) + lane_off);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[a * 3 + b], bfr, acc[m], 0, 0, 0);
          }
      }
      // C/D: col = cell j, rows 4 kg + q = (c, ci): lanes kg 0 hold (c0: q 0, 1),
      // (c1: q 2, 3), lanes kg 1 hold c2 (q 0, 1).  dx[t = j] = c0[j] + c1[j + 1]
      // + c2[j + 2]
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int l1 = (j + 1) & 15, l2 = 16 + ((j + 2) & 15);
        const float s10 = __int_as_float(__builtin_amdgcn_ds_bpermute(l1 << 2, __float_as_int(acc[m][2])));
        const float s11 = __int_as_float(__builtin_amdgcn_ds_bpermute(l1 << 2, __float_as_int(acc[m][3])));
        const float s20 = __int_as_float(__builtin_amdgcn_ds_bpermute(l2 << 2, __float_as_int(acc[m][0])));
        const float s21 = __int_as_float(__builtin_amdgcn_ds_bpermute(l2 << 2, __float_as_int(acc[m][1])));
        const int oo0 = r0 + r, oo1 = o1 + 2 * wave + m, oo2 = o2 + j;
        if (kg == 0 && j < GS2 && oo1 < g.D[1] && oo2 < g.D[2]) {
          float* dst = dx + ((((size_t)n * g.D[0] + oo0) * g.D[1] + oo1) * g.D[2] + oo2) * 2;
          *reinterpret_cast<float2*>(dst) = make_float2(acc[m][0] + s10 + s20, acc[m][1] + s11 + s21);
        }
      }
      GSL_BARRIER();
    }
  }
#undef GSL_BARRIER
}

}  // namespace

bool conv_dgrad_c2_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_DGRAD_C2)) return false;
  if (g.Cin < 1 || g.Cin > 4 || g.Cout != 32 || g.d2s != 1) return false;
  if (g.pad_mode == S3_PAD_REFLECT) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] < 0 || g.lo[d] > 2) return false;
    if (g.O[d] != g.D[d] + 2 * g.lo[d] - 2) return false;
  }
  // enough output tiles to fill the chip
  return (int64_t)g.N * ((g.D[0] + DT0 - 1) / DT0) * ((g.D[1] + DT1 - 1) / DT1) *
             ((g.D[2] + DT2 - 1) / DT2) >= 64;
}

// the sliding-window kernel: bf16 dPre, C_in = 2, enough columns for the chip
static bool dgrad_c2_slide_ok(const s3_ctx* ctx, const ConvGeom& g, int dy_bf16) {
  if (!dy_bf16 || g.Cin != 2 || s3_opt_has(S3O_NO_DGRAD_C2_SLIDE)) return false;
  const int64_t units = (int64_t)g.N * ((g.D[0] + GSEG0 - 1) / GSEG0) * ((g.D[1] + GS1 - 1) / GS1) *
                        ((g.D[2] + GS2 - 1) / GS2);
  return units >= s3_opt_int(S3O_DGRAD_C2_SLIDE_MIN_UNITS, 2 * (long long)ctx->num_cu);
}

// (both images: the 27-tap one of conv_dgrad_c2_kernel, then the 9 packed ones)
size_t conv_dgrad_c2_packed_bytes() { return (size_t)(27 + 9) * 16 * 32 * 2; }

int launch_conv_dgrad_c2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  hipLaunchKernelGGL(dgrad_c2_pack_kernel, dim3(54), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)img, g.Cin);
  if (g.Cin == 2)
    hipLaunchKernelGGL(dgrad_c2_slide_pack_kernel, dim3(18), dim3(256), 0, ctx->stream, w,
                       (unsigned short*)img + 27 * 16 * 32);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_dgrad_c2(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img,
                         float* dx, int dy_bf16) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_c2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, DLDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_c2_slide_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GLDS));
    attr_set.mark(ctx->device);
  }
  if (dgrad_c2_slide_ok(ctx, g, dy_bf16)) {
    const int segs0 = (g.D[0] + GSEG0 - 1) / GSEG0, t1 = (g.D[1] + GS1 - 1) / GS1,
              t2 = (g.D[2] + GS2 - 1) / GS2;
    const int n_units = g.N * segs0 * t1 * t2;
    const int grid = n_units < ctx->num_cu ? n_units : ctx->num_cu;
    hipLaunchKernelGGL(conv_dgrad_c2_slide_kernel, dim3((unsigned)grid), dim3(GNT), GLDS, ctx->stream,
                       (const unsigned short*)dy, (const unsigned short*)img + 27 * 16 * 32, dx, g,
                       segs0, t1, t2, n_units);
    S3_HIP(ctx, hipGetLastError());
    ctx->stat[S3_STAT_DGRAD_C2_SLIDE]++;
    return S3_OK;
  }
  const int tiles0 = (g.D[0] + DT0 - 1) / DT0, tiles1 = (g.D[1] + DT1 - 1) / DT1,
            tiles2 = (g.D[2] + DT2 - 1) / DT2;
  hipLaunchKernelGGL(conv_dgrad_c2_kernel, dim3((unsigned)(g.N * tiles0 * tiles1 * tiles2)),
                     dim3(DNT), DLDS, ctx->stream, dy, (const unsigned short*)img, dx, g, tiles0,
                     tiles1, tiles2, dy_bf16);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
