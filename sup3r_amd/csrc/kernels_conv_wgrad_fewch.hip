// Weight gradient of the 2-channel hi-res conv that opens the discriminator
// (C_in = 2 -> C_out = 32 over N x 78 x 78 x 286 positions at C2) on bf16 MFMA,
// without LDS staging — S3_PREC_BF16 training plans.
//
//   dW[m = tap * 2 + ci][co] = sum_p X[p * s + tap][ci] * dPre[p][co]
//
// M = 54 (4 blocks of 16), N = C_out, contraction over positions.  The layer is
// bound by reading dPre once (128 B per position) — the kernel's job is to do
// exactly that: every WAVE owns the whole M x N gradient (4 x NB accumulators)
// and walks k-steps of 32 consecutive t of one (n, s1, s2) row, so each operand
// element is loaded once: lane (i, kg) of the A fragment loads the float2 cells
// of its own tap at t = 8 kg .. 8 kg + 7 (lanes 2 tap, 2 tap + 1 share them),
// lane (j, kg) of the B fragment 8 dPre values of its channel; immediates carry
// the t offsets.  Rows m >= 54 and t >= O2 are masked through the B operand.
// Waves of a workgroup are summed in LDS; one partial per workgroup, reduced in
// fixed order.
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

constexpr int C2_WAVES = 4;

// DY16: dPre is a bf16 tensor with C_out = 32.  The 32 t-rows of a k-step are
// 2 KB: every lane fetches 32 B of them (two 16-B loads instead of sixteen 4-B
// ones), drops them into a wave-private LDS block in the natural [t][co]
// order, and the K-major B fragments come back through ds_read_b64_tr_b16
// exactly as in conv3_wgrad_bf16_kernel (same 32-B segment swizzle).  No
// workgroup barrier: a wave's LDS operations complete in issue order.  (Per-
// lane 4-B pair loads + unpacking were measured 1.6x SLOWER than the fp32 path.)
// X3 (S3_PREC_BF16X3 plans, fp32 operands): both fragments are split into
// bf16 pairs in registers (hi = bf16(v), lo = bf16(v - hi)), three MFMAs per
// fragment pair — fp32-class gradients of the hi-res few-channel convs.
template <int CIN, int NB, bool DY16 = false, bool X3 = false>
__global__ __launch_bounds__(C2_WAVES * 64) void conv_wgrad_c2_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    float* __restrict__ partial, ConvGeom g, int chunks, int64_t n_steps) {
  constexpr int MT = 27 * CIN;                  // rows of the gradient
  constexpr int MB = (MT + 15) / 16;
  // DY16, stride 1, no padding: the x window of a k-step — 9 (ta, tb) rows of 34
  // cells — is fetched once per wave (5 coalesced 8-B loads per lane instead of
  // 32 gathered 4-B ones) and the A fragments are read back from here
  constexpr int XW_CELLS = 9 * 34;
  // ONE LDS block: the staging areas of the k-step loop (dst: [wave][32 t][64 B],
  // xw) and the end-of-kernel reduction buffer `red` are never live together.
  // Separate arrays were 50 KB per workgroup = 3 workgroups per CU for a grid
  // of 4 per CU: the fourth ran as a tail (round 3: 32 KB, all four resident).
  constexpr int RED_BYTES = C2_WAVES * MB * NB * 64 * 4 * 4;
  constexpr int DST_BYTES = DY16 ? C2_WAVES * 2048 : 16;
  constexpr int XW_BYTES = (DY16 && CIN == 2) ? C2_WAVES * XW_CELLS * 8 : 8;
  constexpr int STAGE_BYTES = DST_BYTES + XW_BYTES;
  __shared__ __attribute__((aligned(16))) char lds_all[RED_BYTES > STAGE_BYTES ? RED_BYTES : STAGE_BYTES];
  float (*red)[MB * NB][64][4] = reinterpret_cast<float (*)[MB * NB][64][4]>(lds_all);
  char* dst = lds_all;
  float2* xw = reinterpret_cast<float2*>(lds_all + DST_BYTES);
  // (zero padding in front — TF 'same' — is window cells outside the tensor: zeros)
  const bool xwin = DY16 && CIN == 2 && !(g.pad_mode == S3_PAD_REFLECT) && g.s[0] == 1 && g.s[1] == 1 &&
                    g.s[2] == 1 && g.lo[0] >= 0 && g.lo[0] <= 1 && g.lo[1] >= 0 && g.lo[1] <= 1 &&
                    g.lo[2] >= 0 && g.lo[2] <= 1;
  // (wave index on the scalar unit: the k-step -> (sample, row, chunk) chains of
  // 64-bit divisions then run there, not on a VALU that PMC showed 71 % busy)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int S1 = g.D[1], S2 = g.D[2];
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2], Cout = g.Cout;
  int tinfo[MB];                                // (ta, tb, tc, ci) of row m, -1 = padding row
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + i;
    const int tap = m / CIN;
    const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
    tinfo[mb] = m < MT ? (ta | (tb << 4) | (tc << 8) | ((m % CIN) << 12)) : -1;
  }
  const int D0 = g.D[0];
  const bool reflect = g.pad_mode == S3_PAD_REFLECT;
  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // k-steps of this workgroup: its XCD's contiguous share of the step list
  // (steps run along t, then s2, then s1: the 3 x 3 window rows of a step are
  // the neighbours' rows — with block b on XCD b % 8 and steps dealt round
  // robin, all eight L2s fetched every x row: FETCH 1.86 GB for 1.0 GB of
  // operands, profiles/r03/pmc_train.txt)
  int64_t st_lo, st_hi;
  int xk, xnk;
  s3_xcd_share(n_steps, st_lo, st_hi, xk, xnk);
  const int64_t n_waves = (int64_t)xnk * C2_WAVES;
  n_steps = st_hi;
  if constexpr (DY16 && CIN == 2) {
    if (xwin) {
      // Round 3: the same k-steps, software-pipelined.  The generic loop
      // below loads a step's dPre rows and x window and waits for them on the
      // spot — waves sat in s_waitcnt 65 % of the time (profiles/r03/
      // pmc_train_start.txt) — here the NEXT step's 18 registers of loads are
      // in flight while this step goes through LDS and the MFMAs.
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      const int r = lane >> 1, hf = lane & 1;
      uint4 q0, q1;
      float2 c5[5];
      // (sample, s1 row, s2 row, t chunk) of the step being fetched, advanced by
      // the grid stride with carries: the 64-bit divisions of step -> (n, o0,
      // o1, chunk) were ~450 scalar instructions per k-step and wave next to
      // its 8 MFMAs
      int f_n, f_o0, f_o1, f_ch;
      {
        const int64_t st0 = st_lo + (int64_t)xk * C2_WAVES + wave;
        int64_t row = st0 / chunks;
        f_ch = (int)(st0 % chunks);
        f_o1 = (int)(row % O1); row /= O1;
        f_o0 = (int)(row % O0); row /= O0;
        f_n = (int)row;
      }
      int d_n, d_o0, d_o1, d_ch;
      {
        int64_t row = n_waves / chunks;
        d_ch = (int)(n_waves % chunks);
        d_o1 = (int)(row % O1); row /= O1;
        d_o0 = (int)(row % O0); row /= O0;
        d_n = (int)row;
      }
      auto advance = [&]() __attribute__((always_inline)) {
        f_ch += d_ch;
        int c = f_ch >= chunks ? 1 : 0;
        f_ch -= c * chunks;
        f_o1 += d_o1 + c;
        c = f_o1 >= O1 ? 1 : 0;
        f_o1 -= c * O1;
        f_o0 += d_o0 + c;
        c = f_o0 >= O0 ? 1 : 0;
        f_o0 -= c * O0;
        f_n += d_n + c;
      };
      auto fetch = [&]() __attribute__((always_inline)) {
        const int tch = f_ch * 32, o1 = f_o1, o0 = f_o0, n = f_n;
        const int tr = tch + r;
        q0 = make_uint4(0u, 0u, 0u, 0u); q1 = q0;
        if (tr < O2) {
          const uint4* src = reinterpret_cast<const uint4*>(
              reinterpret_cast<const unsigned short*>(dy) +
              ((((int64_t)n * O0 + o0) * O1 + o1) * O2 + tr) * 32 + hf * 16);
          q0 = src[0]; q1 = src[1];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int item = lane + 64 * k;
          const int rw = item / 34, cl = item - rw * 34;
          // (cells outside the tensor: the zero padding, or past the row end
          // where they feed t >= O2 only)
          const int i0 = o0 + rw / 3 - g.lo[0], i1 = o1 + rw % 3 - g.lo[1], tt = tch + cl - g.lo[2];
          c5[k] = make_float2(0.f, 0.f);
          if (item < XW_CELLS && i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < S1 && tt >= 0 && tt < S2)
            c5[k] = *reinterpret_cast<const float2*>(x + ((((int64_t)n * D0 + i0) * S1 + i1) * S2 + tt) * 2);
        }
      };
      int64_t step = st_lo + (int64_t)xk * C2_WAVES + wave;
      if (step < n_steps) fetch();
      for (; step < n_steps; step += n_waves) {
        char* d = dst + wave * 2048 + r * 64 + ((hf ^ ((r >> 3) & 1)) << 5);
        float2* xb = xw + wave * XW_CELLS;
        // (other lanes read what this lane writes: keep the compiler from
        // moving LDS accesses across these points; the hardware runs a wave's
        // LDS operations in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<uint4*>(d) = q0;
        *reinterpret_cast<uint4*>(d + 16) = q1;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (lane + 64 * k < XW_CELLS) xb[lane + 64 * k] = c5[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the next step's loads fly under this step's LDS reads and MFMAs
        if (step + n_waves < n_steps) { advance(); fetch(); }
        bf16x8 bfr[NB];
        const char* wb = dst + wave * 2048;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          s16x4 lh[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int pl = 8 * kg + 4 * h + (i >> 2);
            lh[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(wb + pl * 64 + ((nb ^ ((pl >> 3) & 1)) << 5) + ((i & 3) << 3)));
          }
          bfr[nb] = __builtin_shufflevector(lh[0], lh[1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int ti = tinfo[mb];
          const int ta = ti & 15, tb = (ti >> 4) & 15, tc = (ti >> 8) & 15, ci = (ti >> 12) & 15;
          const float* xr = reinterpret_cast<const float*>(xw + wave * XW_CELLS) +
                            ((ta * 3 + tb) * 34 + kg * 8 + tc) * 2 + ci;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ti >= 0 ? xr[e * 2] : 0.f;
          const uint4 u = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
          const bf16x8 afr = __builtin_bit_cast(bf16x8, u);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nb], acc[mb][nb], 0, 0, 0);
        }
      }
      goto reduce;
    }
  }
  // (the step's (sample, rows, chunk) advanced with carries, as in the
  // pipelined loop above: no 64-bit division per k-step)
  int g_n, g_o0, g_o1, g_ch, e_n, e_o0, e_o1, e_ch;
  {
    const int64_t st0 = st_lo + (int64_t)xk * C2_WAVES + wave;
    int64_t row = st0 / chunks;
    g_ch = (int)(st0 % chunks);
    g_o1 = (int)(row % O1); row /= O1;
    g_o0 = (int)(row % O0); row /= O0;
    g_n = (int)row;
    row = n_waves / chunks;
    e_ch = (int)(n_waves % chunks);
    e_o1 = (int)(row % O1); row /= O1;
    e_o0 = (int)(row % O0); row /= O0;
    e_n = (int)row;
  }
  for (int64_t step = st_lo + (int64_t)xk * C2_WAVES + wave; step < n_steps; step += n_waves) {
    const int s_ch = g_ch, o1 = g_o1, o0 = g_o0, n = g_n;
    {
      g_ch += e_ch;
      int c = g_ch >= chunks ? 1 : 0;
      g_ch -= c * chunks;
      g_o1 += e_o1 + c;
      c = g_o1 >= O1 ? 1 : 0;
      g_o1 -= c * O1;
      g_o0 += e_o0 + c;
      c = g_o0 >= O0 ? 1 : 0;
      g_o0 -= c * O0;
      g_n += e_n + c;
    }
    const int t0 = s_ch * 32 + kg * 8;
    // clamp the lane's first t so that the 8 loads stay inside the row; the
    // shifted-out positions are masked through dPre
    // (DY16: no shift — the staged dPre rows carry the t >= O2 mask, the x
    // loads past the row end take the per-element path below)
    const int tl = DY16 ? t0 : (t0 + 8 <= O2 ? t0 : (O2 - 8 > 0 ? O2 - 8 : 0));
    const int shift = t0 - tl;                 // 0, or how many of the 8 slots repeat earlier t
    const float* dr = dy + ((((int64_t)n * O0 + o0) * O1 + o1) * O2 + tl) * Cout;
    const int xstep = CIN * g.s[2];
    static_assert(!(X3 && DY16), "the split-bf16 variant takes fp32 dPre");
    bf16x8 bfr[NB], bfl[X3 ? NB : 1];
    // hi | lo split of 8 fp32 values (lo = the rounding residue)
    auto split = [](const float* v, bf16x8& hi, bf16x8& lo) __attribute__((always_inline)) {
      const uint4 h = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
      hi = __builtin_bit_cast(bf16x8, h);
      if constexpr (X3) {
        auto lf = [](unsigned u) { return __uint_as_float(u << 16); };
        auto hf = [](unsigned u) { return __uint_as_float(u & 0xFFFF0000u); };
        const uint4 l = make_uint4(pk2(v[0] - lf(h.x), v[1] - hf(h.x)), pk2(v[2] - lf(h.y), v[3] - hf(h.y)),
                                   pk2(v[4] - lf(h.z), v[5] - hf(h.z)), pk2(v[6] - lf(h.w), v[7] - hf(h.w)));
        lo = __builtin_bit_cast(bf16x8, l);
      }
    };
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if constexpr (DY16) {
        if (nb == 0) {
          // stage rows t = 32 chunk + r, r = lane >> 1; this lane's 32-B half
          const int r = lane >> 1, hf = lane & 1;
          const int tr = s_ch * 32 + r;
          uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0;
          if (tr < O2) {
            const uint4* src = reinterpret_cast<const uint4*>(
                reinterpret_cast<const unsigned short*>(dy) +
                ((((int64_t)n * O0 + o0) * O1 + o1) * O2 + tr) * 32 + hf * 16);
            q0 = src[0]; q1 = src[1];
          }
          char* d = dst + wave * 2048 + r * 64 + ((hf ^ ((r >> 3) & 1)) << 5);
          // (other lanes read what this lane writes: keep the compiler from
          // moving LDS accesses across these points; the hardware runs a
          // wave's LDS operations in order)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          *reinterpret_cast<uint4*>(d) = q0;
          *reinterpret_cast<uint4*>(d + 16) = q1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const char* wb = dst + wave * 2048;
        s16x4 lh[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pl = 8 * kg + 4 * h + (i >> 2);
          lh[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(wb + pl * 64 + ((nb ^ ((pl >> 3) & 1)) << 5) + ((i & 3) << 3)));
        }
        bfr[nb] = __builtin_shufflevector(lh[0], lh[1], 0, 1, 2, 3, 4, 5, 6, 7);
        continue;
      }
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int co = nb * 16 + i;
        const float t = co < Cout ? dr[(int64_t)e * Cout + co] : 0.f;
        // slot e holds t = tl + e; it belongs to this k-step iff e >= shift
        v[e] = (e >= shift && tl + e < O2) ? t : 0.f;
      }
      split(v, bfr[nb], bfl[X3 ? nb : 0]);
    }
    if constexpr (DY16 && CIN == 2) {
      if (xwin) {
        const int tch = s_ch * 32;
        float2* xb = xw + wave * XW_CELLS;
        float2 c5[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int item = lane + 64 * k;
          const int rw = item / 34, cl = item - rw * 34;
          int tt = tch + cl;
          tt = tt > S2 - 1 ? S2 - 1 : tt;          // (past the row: feeds t >= O2 only, zero dPre)
          c5[k] = make_float2(0.f, 0.f);
          if (item < XW_CELLS)
            c5[k] = *reinterpret_cast<const float2*>(
                x + ((((int64_t)n * D0 + o0 + rw / 3) * S1 + o1 + rw % 3) * S2 + tt) * 2);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (lane + 64 * k < XW_CELLS) xb[lane + 64 * k] = c5[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float v[8];
      if (DY16 && CIN == 2 && xwin) {
        const int ti = tinfo[mb];
        const int ta = ti & 15, tb = (ti >> 4) & 15, tc = (ti >> 8) & 15, ci = (ti >> 12) & 15;
        const float* xr = reinterpret_cast<const float*>(xw + wave * XW_CELLS) +
                          ((ta * 3 + tb) * 34 + kg * 8 + tc) * 2 + ci;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ti >= 0 ? xr[e * 2] : 0.f;
      } else
      {
        // this lane's tap: source row (i0, i1) under the (virtual) padding,
        // then 8 cells along t — linear when they are all inside the row
        const int ti = tinfo[mb];
        const int ta = ti & 15, tb = (ti >> 4) & 15, tc = (ti >> 8) & 15, ci = (ti >> 12) & 15;
        int i0 = o0 * g.s[0] + ta - g.lo[0], i1 = o1 * g.s[1] + tb - g.lo[1];
        if (reflect) { i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, S1); }
        const bool rok = ti >= 0 && i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < S1;
        i0 = i0 < 0 ? 0 : (i0 > D0 - 1 ? D0 - 1 : i0);
        i1 = i1 < 0 ? 0 : (i1 > S1 - 1 ? S1 - 1 : i1);
        const float* xrow = x + (((int64_t)n * D0 + i0) * S1 + i1) * S2 * CIN + ci;
        const int t_first = tl * g.s[2] + tc - g.lo[2];
        if (t_first >= 0 && t_first + 7 * g.s[2] < S2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rok ? xrow[(int64_t)t_first * CIN + e * xstep] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            int i2 = t_first + e * g.s[2];
            if (reflect) i2 = s3_reflect(i2, S2);
            const bool ok = rok && i2 >= 0 && i2 < S2;
            i2 = i2 < 0 ? 0 : (i2 > S2 - 1 ? S2 - 1 : i2);
            v[e] = ok ? xrow[(int64_t)i2 * CIN] : 0.f;
          }
        }
      }
      bf16x8 afr, afl;
      split(v, afr, afl);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if constexpr (X3) {   // small terms first
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl, bfr[nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfl[nb], acc[mb][nb], 0, 0, 0);
        }
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nb], acc[mb][nb], 0, 0, 0);
      }
    }
  }
reduce:
  // ---- sum the workgroup's waves, write partial[bid][m][co]
  __syncthreads();          // (red aliases the staging areas of the slower waves)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][mb * NB + nb][lane][r] = acc[mb][nb][r];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * MT * Cout;
  for (int item = threadIdx.x; item < MB * NB * 64 * 4; item += C2_WAVES * 64) {
    const int r = item & 3, ln = (item >> 2) & 63, f = item >> 8;
    const int mb = f / NB, nb = f % NB;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < C2_WAVES; ++w) t += red[w][f][ln][r];
    const int m = mb * 16 + (ln >> 4) * 4 + r, co = nb * 16 + (ln & 15);
    if (m < MT && co < Cout) out[(size_t)m * Cout + co] = t;
  }
}

__global__ void wgrad_c2_partial_reduce(const float* __restrict__ partial, int n_part,
                                        int wsize, float* __restrict__ dw, int accumulate) {
  // one workgroup per 16 elements; 16 lane groups split the partials and each
  // walks its share with four independent sums (up to 1 024 partials of 1 728
  // elements: with 4 groups and one dependent sum per lane the 2 -> 32 layer's
  // reduction took 66 us).  Fixed order: deterministic, identical on every rank.
  __shared__ float s[16][16];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (e < wsize) {
    int p = q;
    for (; p + 48 < n_part; p += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] += partial[(size_t)(p + 16 * u) * wsize + e];
    }
    for (int u = 0; p < n_part; p += 16, ++u) t[u & 3] += partial[(size_t)p * wsize + e];
  }
  s[q][el] = (t[0] + t[1]) + (t[2] + t[3]);
  __syncthreads();
  if (q == 0 && e < wsize) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += s[g][el];
    dw[e] = accumulate ? dw[e] + v : v;
  }
}

int c2_grid(const s3_ctx* ctx, int64_t n_steps) {
  int64_t grid = 4 * (int64_t)ctx->num_cu;
  const int64_t need = (n_steps + C2_WAVES - 1) / C2_WAVES;
  if (grid > need) grid = need;
  return (int)(grid < 1 ? 1 : grid);
}

int64_t c2_steps(const ConvGeom& g, int* chunks) {
  *chunks = (g.O[2] + 31) / 32;
  return (int64_t)g.N * g.O[0] * g.O[1] * *chunks;
}

}  // namespace

bool conv_wgrad_c2_supported(const ConvGeom& g, int precision) {
  if (precision == S3_PREC_BF16X3 ? s3_opt_has(S3O_NO_WGRAD_X3) : precision != S3_PREC_BF16) return false;
  if (s3_opt_has(S3O_NO_WGRAD_C2)) return false;
  // 2 -> 16 / 32 / 64 (discriminator input layer) or 8 -> C_out <= 16 (hi-res tail)
  const bool a = g.Cin == 2 && (g.Cout == 16 || g.Cout == 32 || g.Cout == 64);
  const bool b = g.Cin == 8 && g.Cout <= 16;
  if (!a && !b) return false;
  if (g.d2s != 1) return false;
  for (int d = 0; d < 3; ++d) if (g.k[d] != 3) return false;
  for (int d = 0; d < 3; ++d)
    if (g.lo[d] < 0 || g.lo[d] > 2) return false;
  return g.O[2] >= 8 && (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] >= 4096;
}

size_t conv_wgrad_c2_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  int chunks;
  const int64_t n_steps = c2_steps(g, &chunks);
  return (size_t)c2_grid(ctx, n_steps) * 27 * g.Cin * g.Cout * sizeof(float);
}

int launch_conv_wgrad_c2(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                         float* dw, float* partial, size_t partial_bytes, int accumulate, int dy_bf16, int x3) {
  int chunks;
  const int64_t n_steps = c2_steps(g, &chunks);
  const int grid = c2_grid(ctx, n_steps);
  if (partial_bytes < conv_wgrad_c2_partial_bytes(ctx, g))
    S3_FAIL(ctx, S3_EINVAL, "wgrad_c2: partial buffer too small");
  const int nb = (g.Cout + 15) / 16;
#define S3_C2(C, B)                                                                          \
  hipLaunchKernelGGL((conv_wgrad_c2_kernel<C, B>), dim3(grid), dim3(C2_WAVES * 64), 0, ctx->stream, \
                     x, dy, partial, g, chunks, n_steps)
#define S3_C2X(C, B)                                                                                        \
  hipLaunchKernelGGL((conv_wgrad_c2_kernel<C, B, false, true>), dim3(grid), dim3(C2_WAVES * 64), 0, ctx->stream, \
                     x, dy, partial, g, chunks, n_steps)
  if (x3) {
    if (dy_bf16) S3_FAIL(ctx, S3_EINVAL, "wgrad_c2: the split-bf16 kernel takes fp32 dPre");
    if (g.Cin == 8) S3_C2X(8, 1);
    else if (nb == 1) S3_C2X(2, 1);
    else if (nb == 2) S3_C2X(2, 2);
    else S3_C2X(2, 4);
  } else if (dy_bf16) {
    if (g.Cin != 2 || g.Cout != 32) S3_FAIL(ctx, S3_EINVAL, "wgrad_c2: bf16 dPre needs C_in = 2, C_out = 32");
    hipLaunchKernelGGL((conv_wgrad_c2_kernel<2, 2, true>), dim3(grid), dim3(C2_WAVES * 64), 0, ctx->stream,
                       x, dy, partial, g, chunks, n_steps);
  } else if (g.Cin == 8) S3_C2(8, 1);
  else if (nb == 1) S3_C2(2, 1);
  else if (nb == 2) S3_C2(2, 2);
  else S3_C2(2, 4);
#undef S3_C2
#undef S3_C2X
  S3_HIP(ctx, hipGetLastError());
  const int wsize = 27 * g.Cin * g.Cout;
  hipLaunchKernelGGL(wgrad_c2_partial_reduce, dim3((wsize + 15) / 16), dim3(256), 0, ctx->stream,
                     partial, grid, wsize, dw, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// ===========================================================================
// LDS-halo weight gradient of the hi-res tail conv (C_in = 8 -> C_out <= 16,
// stride 1, any padding): dW[(tap, ci)][co] = sum_p X[p + tap][ci] dPre[p][co].
// The 8-channel bf16 cell is 16 B, so the 32-B row of an LDS transpose read
// (ds_read_b64_tr_b16) is two cells adjacent in t = the taps (c, c + 1) of one
// position: one M block of 16 rows = 2 taps x 8 channels, 18 blocks for the 27
// taps (the second block of every (a, b) carries c = 2 and a discarded c = 3).
// dPre sits transposed in LDS ([co][position], one ds_read_b128 per fragment).
// A workgroup stages the halo of a 4 x 8 x 32 tile once (padding rule applied
// there); its 4 waves split the 32 k-steps and keep all 18 accumulators, summed
// through LDS at the end; one partial per workgroup.
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TW0 = 4, TW1 = 8, TW2 = 32;
constexpr int TG0 = TW0 + 2, TG1 = TW1 + 2, TG2 = TW2 + 2;
constexpr int TWP = TG0 * TG1 * TG2;            // 2040 halo cells of 16 B
constexpr int TWN = TW0 * TW1 * TW2;            // 1024 positions
constexpr int TW_XS = (TWP + 2) * 16;           // + over-read of the junk tap
constexpr int TW_CO = 4;                        // C_out rows kept in LDS (the tail convs have 2 - 4)
constexpr int TW_DPT = TW_CO * TWN / 256;       // dPre values per thread (8)
constexpr int TW_DS = TW_CO * TWN * 2;          // dPre^T: C_out rows x 1024 positions bf16
// (the epilogue's reduction image needs 4 waves x 9 blocks x 256 floats)
constexpr int TW_LDS = (TW_XS + TW_DS) > 36864 ? (TW_XS + TW_DS) : 36864;

template <bool IN16>
__global__ __launch_bounds__(256) void conv_wgrad_tail_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
    ConvGeom g, int tiles0, int tiles1, int tiles2, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  unsigned short* dsT = reinterpret_cast<unsigned short*>(smem + TW_XS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, kg = lane >> 4;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2], Cout = g.Cout;

  // Round 6: the TAPS (a, b) are columns of the matrix product.  Until then a k-step
  // was one output row of 32 t with dPre as the B fragment (2 of its 16 columns
  // alive) and 18 A fragments — the 9 (a, b) halo rows x 2 t-shifts — for 18 MFMAs:
  // 37 LDS reads per k-step, and the kernel ran at the speed of those reads (0.21 of
  // the HBM roofline; three or four workgroups per CU changed nothing).  Now a k-step
  // is one HALO row H: its two A fragments (t-shift 0 / 2: rows (s, ci) = taps c = s,
  // 2 + s) are read once and multiplied with B[q][(a, b, co)] = dPre[H - (a, b)][q][co]
  // — 9 C_out live columns (18 in two fragments for the 8 -> 2 conv), zero where H -
  // (a, b) leaves the tile — so a tile is 60 halo rows x (4 transpose reads + 2 reads +
  // 4 MFMAs) instead of 32 x (36 + 1 + 18), and the accumulators are 4 - 6 fragments
  // instead of 18.
  constexpr int NBK = (9 * TW_CO + 15) / 16;      // column fragments (3 for C_out <= 4)
  const int ncol = 9 * Cout;
  f32x4 acc[2][NBK];
#pragma unroll
  for (int cp = 0; cp < 2; ++cp)
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) acc[cp][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this lane's B columns: jc = nb 16 + q -> (tap (a, b), co), alive while jc < 9 C_out
  int col_a[NBK], col_b[NBK], col_co[NBK];
  bool col_ok[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    const int jc = nb * 16 + q, ab = jc / Cout;
    col_ok[nb] = jc < ncol;
    col_a[nb] = ab / 3; col_b[nb] = ab % 3; col_co[nb] = jc - ab * Cout;
  }

  // Round 4: the tile loop is software-pipelined.  It ran load -> LDS ->
  // barrier -> 144 MFMAs per wave back to back: ~10 us of exposed load latency
  // per 1.1 us of matrix work (0.14 of the HBM roofline).  Now the NEXT tile's
  // halo cells (8 x 16 B per thread) and dPre values (TW_DPT per thread) are
  // fetched into registers before the current tile's k-steps and dropped into
  // LDS after them, and dPre^T keeps only its C_out <= TW_CO rows (the B
  // fragment's other columns are zero registers), which takes the LDS image
  // from 65 to 37 KB: four workgroups per CU instead of two.
  constexpr int NH = (TWP + 255) / 256;           // halo cells per thread (8)
  uint4 hreg[NH];
  float dreg[TW_DPT];
  auto tile_org = [&](int tile, int& n, int& o0, int& o1, int& o2) {
    int tr = tile;
    o2 = (tr % tiles2) * TW2; tr /= tiles2;
    o1 = (tr % tiles1) * TW1; tr /= tiles1;
    o0 = (tr % tiles0) * TW0; tr /= tiles0;
    n = tr;
  };
  auto fetch = [&](int tile) {
    int n, org0, org1, org2;
    tile_org(tile, n, org0, org1, org2);
    // ---- x halo: cell = 8 channels -> 16 B of bf16
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int hp = tid + 256 * k;
      int h = hp < TWP ? hp : TWP - 1;
      const int c2 = h % TG2; h /= TG2;
      const int c1 = h % TG1; h /= TG1;
      const int c0 = h;
      int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
      if (g.pad_mode == S3_PAD_REFLECT) {
        i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
      }
      const bool ok = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2;
      const int64_t cell = (((int64_t)n * D0 + i0) * D1 + i1) * D2 + i2;
      uint4 v = make_uint4(0, 0, 0, 0);
      if constexpr (IN16) {                      // bf16 cells (bf16 saved activations)
        if (ok) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(x) + cell * 8);
      } else {
        if (ok) {
          const float4 a = *reinterpret_cast<const float4*>(x + cell * 8);
          const float4 b = *reinterpret_cast<const float4*>(x + cell * 8 + 4);
          v = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
        }
      }
      hreg[k] = v;
    }
    // ---- dPre of the tile: item = co * TWN + position (coalesced along t)
#pragma unroll
    for (int k = 0; k < TW_DPT; ++k) {
      const int item = tid + 256 * k;
      const int pl = item % TWN, co = item / TWN;
      const int row = pl / TW2, tt = pl % TW2;
      const int o0 = org0 + row / TW1, o1 = org1 + row % TW1, o2 = org2 + tt;
      float v = 0.f;
      if (co < Cout && o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2])
        v = dy[((((int64_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * Cout + co];
      dreg[k] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int hp = tid + 256 * k;
      if (hp < TWP) *reinterpret_cast<uint4*>(xs + hp * 16) = hreg[k];
    }
#pragma unroll
    for (int k = 0; k < TW_DPT; ++k) {
      const int item = tid + 256 * k;
      if (item < Cout * TWN) dsT[item] = (unsigned short)(pk2(dreg[k], 0.f) & 0xFFFFu);
    }
  };
  if (tid < 2) *reinterpret_cast<uint4*>(xs + (TWP + tid) * 16) = make_uint4(0, 0, 0, 0);
  // (XCD-contiguous tile shares: s3_xcd_share, common.h)
  int64_t xt_lo, xt_hi;
  int xt_k, xt_nk;
  s3_xcd_share(n_tiles, xt_lo, xt_hi, xt_k, xt_nk);
  int tile = (int)xt_lo + xt_k;
  if (tile < (int)xt_hi) fetch(tile);
  for (; tile < (int)xt_hi; tile += xt_nk) {
    __syncthreads();                 // every wave is done with the previous image
    commit();
    __syncthreads();
    if (tile + xt_nk < (int)xt_hi) fetch(tile + xt_nk);   // in flight under the k-steps
    // ---- 60 k-steps (one halo row of 34 t each), 15 per wave
    for (int ks = wave; ks < TG0 * TG1; ks += 4) {
      const int h0 = ks / TG1, h1 = ks % TG1;
      bf16x8 afr[2];
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        // rows of the transpose read: cells t = 8 kg + 4 h + (q >> 2) (+ 2 cp)
        const char* base = xs + (((h0 * TG1 + h1) * TG2) + 8 * kg + (q >> 2) + 2 * cp) * 16 + ((q & 3) << 3);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(base + 4 * 16));
        afr[cp] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        if (nb * 16 >= ncol) continue;          // (uniform: no live column in this fragment)
        // output row of this column's tap: H - (a, b), inside the tile or nothing
        const int r0 = h0 - col_a[nb], r1 = h1 - col_b[nb];
        bf16x8 bfr = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (col_ok[nb] && r0 >= 0 && r0 < TW0 && r1 >= 0 && r1 < TW1)
          bfr = *reinterpret_cast<const bf16x8*>(dsT + col_co[nb] * TWN + (r0 * TW1 + r1) * TW2 + kg * 8);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp)
          acc[cp][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[cp], bfr, acc[cp][nb], 0, 0, 0);
      }
    }
  }
  // ---- sum the 4 waves: red[wave][cp][nb][row m 16][column 16]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);        // 4 x 2 NBK x 256 floats of the (dead) LDS image
  float* out = partial + (size_t)blockIdx.x * 27 * 8 * Cout;
#pragma unroll
  for (int cp = 0; cp < 2; ++cp)
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(((wave * 2 + cp) * NBK + nb) * 16 + kg * 4 + r) * 16 + q] = acc[cp][nb][r];
  __syncthreads();
  for (int item = tid; item < 2 * NBK * 256; item += 256) {
    const int j = item & 15, m16 = (item >> 4) & 15, blk = item >> 8;
    const int nb = blk % NBK, cp = blk / NBK;
    const int jc = nb * 16 + j, ab = jc / Cout, co = jc - ab * Cout;
    const int c = 2 * cp + (m16 >> 3), ci = m16 & 7;
    if (c < 3 && jc < ncol) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(((w * 2 + cp) * NBK + nb) * 16 + m16) * 16 + j];
      out[((size_t)(ab * 3 + c) * 8 + ci) * Cout + co] = t;
    }
  }
}


// ===========================================================================
// Plane-sweep form of the tail conv's weight gradient (round 6; C_in = 8 bf16 cells,
// C_out = 2, reflect padding): the same matrix product per halo row as
// conv_wgrad_tail_kernel above — A = two transpose-read fragments of an x row (taps
// c = 0 .. 3 x 8 channels), B[t][(a, b, co)] = dPre[row - (a, b)][t][co] — with the
// data movement of conv_tail_sweep_kernel (kernels_conv_tail_sweep.hip): a workgroup
// owns a column of S1 x S2 positions and walks it along s0; every step ONE x plane
// ((S1 + 2) x (S2 + 2) cells, LDS-DMA pieces in linear plane order into a 3-slot ring,
// two planes in flight) meets the three dPre rows it belongs to (taps a = 0 / 1 / 2 of
// rows q / q - 1 / q - 2), which sit transposed and bf16-rounded in a 4-slot ring
// (the raw fp32 row travels with the x plane of its step — more DMA pieces behind the
// plane's — and the lane that requested a piece converts it when it has landed).  The 4 x 8 x 32 tiles of the kernel above fetch every x cell twice (2 040 halo
// cells per 1 024 positions) one tile deep; here 1.1 - 1.2 times, two steps deep.
constexpr int WS_WAVES = 8, WS_NTH = WS_WAVES * 64, WS_NSLOT = 3, WS_NDS = 4;
constexpr int WS_MAXP = 7;                     // DMA pieces per wave and plane
constexpr int WS_DPT = 3;                      // dPre position pairs per thread and row
constexpr int WS_LDS = 160 * 1024;
constexpr int WS_RED = WS_WAVES * 4 * 256 * 4; // the final cross-wave sum: 32 KB

struct WSweepShape {
  int S1, S2, NCH, P2;         // positions per plane row set, 32-step chunks per row, cells per plane row
  int plane_cells, npieces, plane_bytes, ds_bytes;
  int ndpieces, slot_bytes;    // raw dPre pieces behind a plane's, bytes of a stream slot
  int seg, segs0, tiles1, tiles2, n_units;
};

// LDS reads the compiler does not see: in a wave that also issues LDS-DMA it puts an
// s_waitcnt vmcnt(0) in front of every LDS read it knows of (the DMA writes LDS) — which
// would wait for the plane requested a moment ago.  Ordered by hand (lgkmcnt below).
template <int OFF>
__device__ inline s16x4 ws_lds_tr(unsigned addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ inline bf16x8 ws_lds_b128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

__global__ __launch_bounds__(WS_NTH) void conv_wgrad_tail_sweep_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
    ConvGeom g, WSweepShape sh) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, kg = lane >> 4;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2];
  auto clampi = [](int i, int d) { return i < 0 ? 0 : (i > d - 1 ? d - 1 : i); };
  char* dsT0 = smem + WS_NSLOT * sh.slot_bytes;
  const int pos_plane = sh.S1 * sh.S2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // 16 B of zeros behind the dPre^T ring: the B fragment of a tap that leaves the unit
  const unsigned zero_addr = lds0 + (unsigned)(WS_NSLOT * sh.slot_bytes + WS_NDS * sh.ds_bytes);
  if (tid < 4) reinterpret_cast<unsigned*>(dsT0 + WS_NDS * sh.ds_bytes)[tid] = 0u;

  // XCD-contiguous unit ranges (conv_tail_sweep_kernel)
  int u_first, u_step, u_end;
  {
    const int G = gridDim.x, b = blockIdx.x, xcd = b % 8;
    int before = 0;
    for (int k = 0; k < xcd; ++k) before += (G - k + 7) / 8;
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
  auto unit_rows = [&](int r0) __attribute__((always_inline)) { return r0 + sh.seg <= O0 ? sh.seg : O0 - r0; };

  f32x4 acc[2][2];
#pragma unroll
  for (int cp = 0; cp < 2; ++cp)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[cp][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* out = partial + (size_t)blockIdx.x * 27 * 8 * 2;

  if (u_first < u_end) {
    // this lane's B columns: jc = nb 16 + q -> tap (a, b), co; 18 live columns
    int col_a[2], col_b[2], col_off[2];
    bool col_ok[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int jc = nb * 16 + q, ab = jc >> 1;
      col_ok[nb] = jc < 18;
      col_a[nb] = ab / 3; col_b[nb] = ab % 3;
      col_off[nb] = (jc & 1) * pos_plane;
    }
    // lane parts of the fragment addresses: A = transpose reads of cells 8 kg + (q >> 2) (+ 2 cp,
    // + 4) of an x row, B = eight t of dPre^T row (h1 - b) of channel co
    const unsigned a_lane = (unsigned)((8 * kg + (q >> 2)) * 16 + ((q & 3) << 3));
    unsigned b_lane[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) b_lane[nb] = (unsigned)((col_off[nb] - col_b[nb] * sh.S2 + kg * 8) * 2);
    const int nks = (sh.S1 + 2) * sh.NCH;
    const int h1_first = wave / sh.NCH, tc_first = wave % sh.NCH;
    const int h1_step = WS_WAVES / sh.NCH, tc_step = WS_WAVES % sh.NCH;
    // this lane's dPre position pairs (two consecutive t of one row: 16 B of fp32): lane l of
    // raw piece wave + 8 k holds pair (wave + 8 k) 64 + l — it converts what it requested
    int drow[WS_DPT], dt[WS_DPT];
    bool dact[WS_DPT];
#pragma unroll
    for (int k = 0; k < WS_DPT; ++k) {
      const int pos = 2 * (((wave + WS_WAVES * k) << 6) + lane);
      dact[k] = pos < pos_plane;
      drow[k] = dact[k] ? pos / sh.S2 : 0;
      dt[k] = dact[k] ? pos - drow[k] * sh.S2 : 0;
    }
    const int my_nd = (sh.ndpieces - wave + WS_WAVES - 1) / WS_WAVES;
    // raw fp32 row in stream slot `rslot` (landed: this wave's own pieces) -> dPre^T slot
    // `slot`: [co][row][t] bf16, zero outside the output (o1, o2: the unit's column)
    auto dpre_convert = [&](int rslot, int slot, int o1, int o2) __attribute__((always_inline)) {
      unsigned short* base = reinterpret_cast<unsigned short*>(dsT0 + slot * sh.ds_bytes);
      const unsigned raw = lds0 + (unsigned)(rslot * sh.slot_bytes + sh.plane_bytes + wave * 1024 + lane * 16);
#pragma unroll
      for (int k = 0; k < WS_DPT; ++k) {
        if (k >= my_nd || !dact[k]) continue;
        bf16x8 rv = ws_lds_b128(raw + k * (WS_WAVES * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv) :: "memory");
        float4 v = __builtin_bit_cast(float4, rv);
        if (!(o1 + drow[k] < O1 && o2 + dt[k] < O2)) v = make_float4(0.f, 0.f, 0.f, 0.f);   // (O2 is even)
        const int idx = drow[k] * sh.S2 + dt[k];
        *reinterpret_cast<unsigned*>(base + idx) = pk2(v.x, v.z);
        *reinterpret_cast<unsigned*>(base + pos_plane + idx) = pk2(v.y, v.w);
      }
    };

    // ---- the stream (issue side), as in conv_tail_sweep_kernel: item q of a unit = x plane
    // q + the raw dPre row q (rows past the unit's last: that row again — the piece count
    // per item stays fixed for the counted wait)
    const int my_np = (sh.npieces - wave + WS_WAVES - 1) / WS_WAVES;
    int iu = u_first, iq = 0, irows = 0, in_ = 0, ir0 = 0, islot = 0;
    unsigned poff[WS_MAXP], doff[WS_DPT];
    auto issue_setup = [&]() __attribute__((always_inline)) {
      int o1, o2;
      unit_org(iu, in_, ir0, o1, o2);
      irows = unit_rows(ir0);
#pragma unroll
      for (int i = 0; i < WS_MAXP; ++i) {
        int c = ((wave + WS_WAVES * i) << 6) + lane;
        c = c < sh.plane_cells ? c : sh.plane_cells - 1;
        const int row = c / sh.P2, col = c - row * sh.P2;
        const int i1 = clampi(s3_reflect(o1 + row - g.lo[1], D1), D1);
        const int i2 = clampi(s3_reflect(o2 + col - g.lo[2], D2), D2);
        poff[i] = (unsigned)((i1 * D2 + i2) * 8);
      }
#pragma unroll
      for (int k = 0; k < WS_DPT; ++k) {
        const int oo1 = o1 + drow[k] < O1 ? o1 + drow[k] : O1 - 1;
        const int oo2 = o2 + dt[k] < O2 ? o2 + dt[k] : O2 - 2;
        doff[k] = (unsigned)((oo1 * O2 + oo2) * 2);
      }
    };
    auto issue_next = [&]() __attribute__((always_inline)) -> bool {
      if (iu >= u_end) return false;
      const int i0 = clampi(s3_reflect(ir0 + iq - g.lo[0], D0), D0);
      const unsigned short* pl = x + ((size_t)in_ * D0 + i0) * D1 * D2 * 8;
      char* dst = smem + islot * sh.slot_bytes + wave * 1024;
#pragma unroll
      for (int i = 0; i < WS_MAXP; ++i)
        if (i < my_np)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(pl + poff[i]),
              (__attribute__((address_space(3))) void*)(dst + i * (WS_WAVES * 1024)), 16, 0, 0);
      const int orow = ir0 + (iq < irows ? iq : irows - 1);
      const float* dr = dy + ((size_t)in_ * O0 + orow) * O1 * O2 * 2;
#pragma unroll
      for (int k = 0; k < WS_DPT; ++k)
        if (k < my_nd)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(dr + doff[k]),
              (__attribute__((address_space(3))) void*)(dst + sh.plane_bytes + k * (WS_WAVES * 1024)), 16, 0, 0);
      islot = islot == WS_NSLOT - 1 ? 0 : islot + 1;
      if (++iq == irows + 2) {
        iq = 0;
        iu += u_step;
        if (iu < u_end) issue_setup();
      }
      return true;
    };
    // everything but the newest item's pieces of this wave has landed
    auto wait_but_newest = [&](bool newest_in_flight) __attribute__((always_inline)) {
      switch (newest_in_flight ? my_np + my_nd : 0) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
#define WSW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- prologue: items 0 and 1 in flight, dPre row 0 of the first unit -> slot 0
    issue_setup();
    issue_next();
    {
      const bool second = issue_next();
      wait_but_newest(second);
      int n, r0, o1, o2;
      unit_org(u_first, n, r0, o1, o2);
      dpre_convert(0, 0, o1, o2);
    }
    WSW_BARRIER();

    int cslot = 0, dbase = 0;
    for (int u = u_first; u < u_end; u += u_step) {
      int n, r0, o1, o2;
      unit_org(u, n, r0, o1, o2);
      const int rows = unit_rows(r0);
      // the unit after this one (its dPre row 0 is converted at the end of this unit's last step)
      int n2 = 0, r02 = 0, o12 = 0, o22 = 0;
      const bool more = u + u_step < u_end;
      if (more) unit_org(u + u_step, n2, r02, o12, o22);
      for (int qq = 0; qq < rows + 2; ++qq) {
        const bool newest = issue_next();
        // ---- this plane's halo rows x 32-step chunks, one in eight per wave
        const unsigned plane_a = lds0 + (unsigned)(cslot * sh.slot_bytes) + a_lane;
        // per plane: the ring slot of each column's dPre row qq - a, and whether it is in the unit
        unsigned b_base[2];
        bool b_in[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int r = qq - col_a[nb];
          b_in[nb] = col_ok[nb] && r >= 0 && r < rows;
          b_base[nb] = lds0 + (unsigned)(WS_NSLOT * sh.slot_bytes + ((dbase + r) & 3) * sh.ds_bytes) + b_lane[nb];
        }
        int h1 = h1_first, tc = tc_first;
        for (int ks = wave; ks < nks; ks += WS_WAVES) {
          const unsigned aa = plane_a + (unsigned)((h1 * sh.P2 + tc * 32) * 16);
          const s16x4 lo0 = ws_lds_tr<0>(aa), hi0 = ws_lds_tr<64>(aa);
          const s16x4 lo1 = ws_lds_tr<32>(aa), hi1 = ws_lds_tr<96>(aa);
          bf16x8 bfr[2];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            // row h1 - b of that dPre row set — inside the column or the zero fragment
            const int r1 = h1 - col_b[nb];
            const bool ok = b_in[nb] && r1 >= 0 && r1 < sh.S1;
            const unsigned ba = ok ? b_base[nb] + (unsigned)((h1 * sh.S2 + tc * 32) * 2) : zero_addr;
            bfr[nb] = ws_lds_b128(ba);
          }
          s16x4 l0 = lo0, g0 = hi0, l1 = lo1, g1 = hi1;
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(l0), "+v"(g0), "+v"(l1), "+v"(g1), "+v"(bfr[0]), "+v"(bfr[1]) :: "memory");
          const bf16x8 afr0 = __builtin_shufflevector(l0, g0, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 afr1 = __builtin_shufflevector(l1, g1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr0, bfr[nb], acc[0][nb], 0, 0, 0);
            acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr1, bfr[nb], acc[1][nb], 0, 0, 0);
          }
          // ks += 8 in (row, chunk) form
          h1 += h1_step; tc += tc_step;
          if (tc >= sh.NCH) { tc -= sh.NCH; ++h1; }
        }
        const int nslot = cslot == WS_NSLOT - 1 ? 0 : cslot + 1;
        wait_but_newest(newest);
        // the NEXT step's dPre row (its raw copy came with the item that has just landed)
        if (qq + 1 < rows) dpre_convert(nslot, (dbase + qq + 1) & 3, o1, o2);
        else if (qq == rows + 1 && more) dpre_convert(nslot, (dbase + rows) & 3, o12, o22);
        cslot = nslot;
        WSW_BARRIER();
      }
      dbase = (dbase + rows) & 3;
    }
#undef WSW_BARRIER
  }

  // ---- sum the 8 waves: red[wave][cp][nb][row m 16][column 16]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int cp = 0; cp < 2; ++cp)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(((wave * 2 + cp) * 2 + nb) * 16 + kg * 4 + r) * 16 + q] = acc[cp][nb][r];
  __syncthreads();
  for (int item = tid; item < 4 * 256; item += WS_NTH) {
    const int j = item & 15, m16 = (item >> 4) & 15, blk = item >> 8;
    const int nb = blk & 1, cp = blk >> 1;
    const int jc = nb * 16 + j, ab = jc >> 1, co = jc & 1;
    const int c = 2 * cp + (m16 >> 3), ci = m16 & 7;
    if (c < 3 && jc < 18) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WS_WAVES; ++w) t += red[(((w * 2 + cp) * 2 + nb) * 16 + m16) * 16 + j];
      out[((size_t)(ab * 3 + c) * 8 + ci) * 2 + co] = t;
    }
  }
}

// column shape and rows per unit of a launch (the model of conv_tail_sweep's sweep_shape():
// a plane costs ~1 000 clk + its bytes through the CU at 12.5 B / clk)
bool wsweep_shape(const s3_ctx* ctx, const ConvGeom& g, WSweepShape& best) {
  static const int s1c[] = {4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48};
  static const int s2c[] = {32, 64, 96, 128, 160, 192};
  static const int sgc[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128};
  const long long forced = s3_opt_int(S3O_TAIL_SWEEP_SHAPE, 0);
  const int f1 = (int)(forced / 1000000), f2 = (int)(forced / 1000 % 1000), fseg = (int)(forced % 1000);
  double best_cost = -1.0;
  for (int S1 : s1c) {
    if (forced ? S1 != f1 : (S1 > g.O[1] && S1 != s1c[0])) continue;
    for (int S2 : s2c) {
      if (forced ? S2 != f2 : (S2 > g.O[2] + 31 && S2 != s2c[0])) continue;
      if (S1 * S2 > 2 * WS_NTH * WS_DPT) continue;
      WSweepShape s;
      s.S1 = S1; s.S2 = S2; s.NCH = S2 / 32; s.P2 = S2 + 2;
      s.plane_cells = (S1 + 2) * s.P2;
      s.npieces = (s.plane_cells + 2 + 63) / 64;      // (+ 2: the transpose read of the junk tap c = 3)
      if (s.npieces > WS_MAXP * WS_WAVES) continue;
      s.plane_bytes = s.npieces * 1024;
      s.ds_bytes = S1 * S2 * 2 * 2;
      s.ndpieces = (S1 * S2 / 2 + 63) / 64;
      s.slot_bytes = s.plane_bytes + s.ndpieces * 1024;
      if (WS_NSLOT * s.slot_bytes + WS_NDS * s.ds_bytes + 64 > WS_LDS) continue;
      s.tiles1 = (g.O[1] + S1 - 1) / S1;
      s.tiles2 = (g.O[2] + S2 - 1) / S2;
      const double plane_clk = 1000.0 + (s.plane_bytes + S1 * S2 * 8.0) / 12.5;
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
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
        if (sg == g.O[0]) break;
      }
    }
  }
  return best_cost >= 0;
}

bool wgrad_tail_sweep_ok(const ConvGeom& g, int x_bf16) {
  if (!x_bf16 || g.Cout != 2 || g.pad_mode != S3_PAD_REFLECT || s3_opt_on(S3O_NO_WGRAD_TAIL_SWEEP)) return false;
  if ((g.O[2] & 1) || g.O[0] < 4 || g.O[2] < 32) return false;
  return (long long)g.D[1] * g.D[2] * 8 <= 0x7fffffffLL;
}

}  // namespace

bool conv_wgrad_tail_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_WGRAD_TAIL)) return false;
  if (g.Cin != 8 || g.Cout > TW_CO || g.Cout < 1 || g.d2s != 1) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] < 0 || g.lo[d] > 2) return false;
  return (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] >= 65536;
}

static int tail_grid(const s3_ctx* ctx, const ConvGeom& g, int* t0, int* t1, int* t2, int* n_tiles) {
  *t0 = (g.O[0] + TW0 - 1) / TW0; *t1 = (g.O[1] + TW1 - 1) / TW1; *t2 = (g.O[2] + TW2 - 1) / TW2;
  *n_tiles = g.N * *t0 * *t1 * *t2;
  // (round 6: 3 or 4 workgroups per CU are no faster — 247 / 216 us against 211 —
  // the kernel is bound by its LDS fragment reads, not by occupancy)
  int grid = 2 * ctx->num_cu;
  if (grid > *n_tiles) grid = *n_tiles;
  return grid;
}

size_t conv_wgrad_tail_partial_bytes(const s3_ctx* ctx, const ConvGeom& g) {
  int a, b, c, nt;
  return (size_t)tail_grid(ctx, g, &a, &b, &c, &nt) * 27 * 8 * g.Cout * sizeof(float);
}

int launch_conv_wgrad_tail(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                           float* dw, float* partial, size_t partial_bytes, int accumulate,
                           int x_bf16) {
  int t0, t1, t2, n_tiles;
  const int grid = tail_grid(ctx, g, &t0, &t1, &t2, &n_tiles);
  if (partial_bytes < conv_wgrad_tail_partial_bytes(ctx, g))
    S3_FAIL(ctx, S3_EINVAL, "wgrad_tail: partial buffer too small");
  WSweepShape wsh;
  if (wgrad_tail_sweep_ok(g, x_bf16) && wsweep_shape(ctx, g, wsh)) {
    static S3DeviceOnce sweep_attr;
    if (!sweep_attr.done(ctx->device)) {
      std::lock_guard<std::mutex> lk(sweep_attr.m);
      S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tail_sweep_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
      sweep_attr.mark(ctx->device);
    }
    // (at most num_cu partials: the buffer is sized for 2 num_cu)
    int sgrid = ctx->num_cu;
    if (sgrid > wsh.n_units) sgrid = wsh.n_units;
    if (sgrid > grid) sgrid = grid;
    int lds = WS_NSLOT * wsh.slot_bytes + WS_NDS * wsh.ds_bytes + 64;
    if (lds < WS_RED) lds = WS_RED;
    hipLaunchKernelGGL(conv_wgrad_tail_sweep_kernel, dim3(sgrid), dim3(WS_NTH), lds, ctx->stream,
                       (const unsigned short*)x, dy, partial, g, wsh);
    S3_HIP(ctx, hipGetLastError());
    const int wsz = 27 * 8 * g.Cout;
    hipLaunchKernelGGL(wgrad_c2_partial_reduce, dim3((wsz + 15) / 16), dim3(256), 0, ctx->stream,
                       partial, sgrid, wsz, dw, accumulate);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tail_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tail_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS));
    attr_set.mark(ctx->device);
  }
  if (x_bf16)
    hipLaunchKernelGGL(conv_wgrad_tail_kernel<true>, dim3(grid), dim3(256), TW_LDS, ctx->stream, x, dy,
                       partial, g, t0, t1, t2, n_tiles);
  else
    hipLaunchKernelGGL(conv_wgrad_tail_kernel<false>, dim3(grid), dim3(256), TW_LDS, ctx->stream, x, dy,
                       partial, g, t0, t1, t2, n_tiles);
  S3_HIP(ctx, hipGetLastError());
  const int wsize = 27 * 8 * g.Cout;
  hipLaunchKernelGGL(wgrad_c2_partial_reduce, dim3((wsize + 15) / 16), dim3(256), 0, ctx->stream,
                     partial, grid, wsize, dw, accumulate);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
