// Halo-tile implicit-GEMM Conv3D for the generator body of Sup3rGan
// (K2: 33 x [REFLECT pad 3 -> Conv3D 64->64 k3 -> crop 2] + the 64->200
// expansion conv = 99.7 % of the generator FLOPs), written for gfx950.
//
//   y[n, p, co] = act(b[co] + sum_{tap, ci} x[n, reflect(p + tap - 1), ci] *
//                 w[tap][ci][co]) (+ residual) (store optionally permuted by
//                 depth-to-space)
//
// im2col-free: a workgroup owns TS0 x TS1 x 16 output positions (16 = run along
// t, the innermost spatial axis) and all 64 output channels of one cout tile.
// The (TS0+2)(TS1+2)(18) x 64-channel input halo is staged ONCE in LDS with the
// reflect/zero boundary evaluated as index math in the load; every one of the
// 27 taps then reads its shifted window straight out of LDS as MFMA A
// fragments.  The 64x64 filter slab of tap+1 is prefetched into registers while
// tap is on the matrix cores and lands in the other half of a 2-slab LDS ring
// (one barrier per tap).  K = 27 * 64 = 1728 is contracted on MFMA:
//
//   S3_PREC_BF16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Halo and filter
//                  rows are 128 B (64 x bf16); 16-B chunks are XOR-swizzled
//                  (halo: chunk ^ (cell_t & 7) — conflict-free for all three
//                  tap t-shifts, checked by brute force over the four 16-lane
//                  groups; filters: chunk ^ ((row >> 1) & 7)) so
//                  that the 16-lane groups of ds_read_b128 hit 16 distinct
//                  16-B slots of the 256-B bank row.  Every LDS read address
//                  in the 27-tap loop is (per-lane register) + (immediate):
//                  the loop is ds_read_b128 + MFMA only.  Activations may be
//                  fp32 or bf16 in HBM (IN16 / OUT16): inference plans keep
//                  the 64-channel trunk in bf16, which halves the halo and
//                  epilogue bytes and removes the convert from the staging.
//   S3_PREC_BF16X3 : fp32 activations and weights split on the fly into bf16
//                  pairs (hi = bf16(v), lo = bf16(v - hi)); every product is
//                  hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate —
//                  the dropped lo*lo term and the split residue are ~2^-16
//                  relative, i.e. fp32-class results at a third of the bf16
//                  rate (5x the fp32-MFMA rate).  A 128-B LDS cell holds 32
//                  channels as [hi x 32 | lo x 32], so the tile, the swizzles
//                  and the read addresses are those of the bf16 mode; the
//                  K = 64 channels are contracted in two passes of 32 with the
//                  halo re-staged in between (accumulators stay in registers).
//   S3_PREC_F32  : v_mfma_f32_16x16x4_f32 (exact fp32, == fmaf chain): parity
//                  mode.  Halo rows padded to 66 dwords, filter rows to 80, so
//                  the per-lane ds_read_b32 of the (row, k) fragments are
//                  conflict-free.
//
// Fragment maps (guide §3): A lane l: row l&15, k-group l>>4; B lane l: col
// l&15, k-group l>>4; C/D lane l reg r: col l&15, row (l>>4)*4 + r.
#include "conv_mfma_tile.h"

namespace {

// pack canonical fp32 w[tap][ci][co] -> bf16 slabs [ct][tap][co 64][ci 64],
// 16-B chunks pre-swizzled (the slab is the LDS image)
__global__ void pack_bf16_kernel(const float* __restrict__ w,
                                 unsigned short* __restrict__ out, int taps,
                                 int cout, int n_ct) {
  const int64_t total = (int64_t)n_ct * taps * CT * CIN;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int ci = (int)(r % CIN); r /= CIN;
    const int row = (int)(r % CT); r /= CT;
    const int tap = (int)(r % taps); r /= taps;
    const int ct = (int)r;
    const int co = ct * CT + row;
    const float v = co < cout ? w[((int64_t)tap * CIN + ci) * cout + co] : 0.f;
    const int chunk = ci >> 3, e = ci & 7;
    const int slot = chunk ^ ((row >> 1) & 7);
    out[(((int64_t)ct * taps + tap) * CT + row) * CIN + slot * 8 + e] = f2bf(v);
  }
}

// BF16X3: canonical fp32 w[tap][ci][co] -> slabs [ct][pass 2][tap][co 64][128 B],
// a row = channels 32 pass .. 32 pass + 31 as [hi x 32 | lo x 32], 16-B chunks
// pre-swizzled like the bf16 slabs
__global__ void pack_bf16x3_kernel(const float* __restrict__ w,
                                   unsigned short* __restrict__ out, int taps,
                                   int cout, int n_ct) {
  const int64_t total = (int64_t)n_ct * 2 * taps * CT * 32;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int cl = (int)(r % 32); r /= 32;
    const int row = (int)(r % CT); r /= CT;
    const int tap = (int)(r % taps); r /= taps;
    const int pass = (int)(r % 2); r /= 2;
    const int ct = (int)r;
    const int co = ct * CT + row, ci = pass * 32 + cl;
    const float v = co < cout ? w[((int64_t)tap * CIN + ci) * cout + co] : 0.f;
    const unsigned short hi = f2bf(v);
    const unsigned short lo = f2bf(v - __uint_as_float((unsigned)hi << 16));
    const int sw = (row >> 1) & 7;
    unsigned short* o = out + ((((int64_t)ct * 2 + pass) * taps + tap) * CT + row) * CIN;
    o[(((cl >> 3)) ^ sw) * 8 + (cl & 7)] = hi;
    o[((4 + (cl >> 3)) ^ sw) * 8 + (cl & 7)] = lo;
  }
}

template <int TS0, int TS1, int NW>
int launch_bf16(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* wpk,
                const float* bias, const void* res, void* y, ConvIO io) {
  if (io.in_bf16 && io.out_bf16)
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, true, true>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  // (fp32-out data gradients with <= 32 output channels: two N fragments)
  if (g.Cout <= 32 && !io.out_bf16 && TS0 == 4 && TS1 == 8 && NW == 16 && !s3_opt_has(S3O_NO_TILE_NF2)) {
    if (io.in_bf16)
      return launch_io<S3_PREC_BF16, TS0, TS1, NW, true, false, 2>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, false, false, 2>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  }
  if (io.in_bf16)
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, true, false>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  if (io.out_bf16)
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, false, true>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  return launch_io<S3_PREC_BF16, TS0, TS1, NW, false, false>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
}

}  // namespace

// the 64 -> C_out trunk instantiations of this file (every data-gradient
// predicate below builds on these); conv_mfma_supported adds the logical-axes
// instantiations of kernels_conv_mfma_gen.hip for the forward pass
static bool conv_mfma_trunk_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_F32 && precision != S3_PREC_BF16 && precision != S3_PREC_BF16X3) return false;
  if (g.Cin != CIN) return false;
  if (g.Cout % 4 != 0 || g.Cout < 16) return false;
  if (g.k[0] != 3 || g.k[1] != 3 || (g.k[2] != 3 && g.k[2] != 1)) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.s[d] != 1) return false;
    const int lo = g.k[d] == 3 ? 1 : 0;
    const bool same = g.lo[d] == lo && g.O[d] == g.D[d];
    // dgrad frame: full correlation over the padded extent, zero boundary
    const bool full = g.k[d] == 3 && g.lo[d] == 2 && g.O[d] == g.D[d] + 2 &&
                      g.pad_mode == S3_PAD_ZERO;
    // valid padding (the discriminator's 64 -> 128 conv): the halo never
    // leaves the tensor except under the masked overhang of ragged tiles
    const bool valid = g.k[d] == 3 && g.lo[d] == 0 && g.O[d] == g.D[d] - 2 &&
                       g.pad_mode != S3_PAD_REFLECT;
    if (!same && !full && !valid) return false;
  }
  if (g.d2s > 1 && (g.Cout / (g.d2s * g.d2s)) % 4 != 0) return false;
  if (g.k[2] == 1) return false;   // 2-D nets stay on the direct kernel for now
  if (g.D[2] < 8) return false;    // 16-long t runs would be mostly masked
  return true;
}

bool conv_mfma_is_gen(const ConvGeom& g, int precision) {
  return !conv_mfma_trunk_supported(g, precision) && conv_mfma_gen_supported(g, precision);
}

bool conv_mfma_supported(const ConvGeom& g, int precision) {
  return conv_mfma_trunk_supported(g, precision) || conv_mfma_gen_supported(g, precision);
}

// transposed + flipped filter of the data gradient:
// wt[tap'][co][ci] = w[26 - tap'][ci][co]
__global__ void pack_dgrad_kernel(const float* __restrict__ w,
                                  float* __restrict__ wt, int cin, int cout, int taps) {
  const int64_t total = (int64_t)taps * cin * cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int ci = (int)(r % cin); r /= cin;
    const int co = (int)(r % cout); r /= cout;
    const int tp = (int)r;
    wt[idx] = w[((int64_t)(taps - 1 - tp) * cin + ci) * cout + co];
  }
}

// chunk k of the data-gradient filter of a 64 -> C_out conv (C_out > 64):
// wt[tap'][ci' = co - 64 k][co' = ci] = w[26 - tap'][ci][co], zero rows past C_out
__global__ void pack_dgrad_chunk_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                        int cout, int k) {
  const int total = 27 * 64 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ci = idx & 63, cr = (idx >> 6) & 63, tp = idx >> 12;
    const int co = 64 * k + cr;
    wt[idx] = co < cout ? w[((size_t)(26 - tp) * 64 + ci) * cout + co] : 0.f;
  }
}

ConvGeom conv_dgrad_geom(const ConvGeom& g) {
  ConvGeom d = g;
  for (int q = 0; q < 3; ++q) {
    d.D[q] = g.O[q];            // input of the dgrad conv = dPre
    d.O[q] = g.D[q] + 2;        // padded frame of x
    d.lo[q] = 2;
  }
  d.Cin = g.Cout; d.Cout = g.Cin;
  d.pad_mode = S3_PAD_ZERO; d.act = S3_ACT_NONE; d.alpha = 0.f; d.d2s = 1;
  return d;
}

bool conv_dgrad_mfma_supported(const ConvGeom& g, int precision) {
  for (int q = 0; q < 3; ++q)
    if (g.k[q] != 3 || g.s[q] != 1 || g.lo[q] != 1 || g.O[q] != g.D[q]) return false;
  return conv_mfma_trunk_supported(conv_dgrad_geom(g), precision);
}

// 64 -> C_out 'same' conv with C_out > 64 (the 64 -> 200 expansion conv): its
// data gradient contracts over C_out; run it as ceil(C_out / 64) passes of the
// 64 -> 64 halo-tile kernel over 64-channel slices of dPre, accumulating in place
ConvGeom conv_dgrad_valid_geom(const ConvGeom& g);

// (a valid-padded conv's gradient lands on x's own grid: no frame, no fold)
ConvGeom conv_dgrad_chunk_geom(const ConvGeom& g, int k) {
  ConvGeom d = g.lo[0] == 0 ? conv_dgrad_valid_geom(g) : conv_dgrad_geom(g);
  d.Cin = 64; d.Cout = 64;
  d.in_cstride = g.Cout;
  d.in_cvalid = g.Cout - 64 * k < 64 ? g.Cout - 64 * k : 64;
  return d;
}

bool conv_dgrad_chunked_supported(const ConvGeom& g, int precision) {
  // (BF16X3 plans since round 4: the split-bf16 tile kernel over fp32 slices)
  if ((precision != S3_PREC_BF16 && (precision != S3_PREC_BF16X3 || s3_opt_has(S3O_NO_DGRAD_X3))) ||
      s3_opt_has(S3O_NO_DGRAD_CHUNKED))
    return false;
  if (g.Cin != 64 || g.Cout <= 64 || g.Cout % 8 != 0 || g.Cout > 512) return false;
  bool same = true, valid = g.pad_mode != S3_PAD_REFLECT && g.d2s == 1;
  for (int q = 0; q < 3; ++q) {
    if (g.k[q] != 3 || g.s[q] != 1) return false;
    same = same && g.lo[q] == 1 && g.O[q] == g.D[q];
    valid = valid && g.lo[q] == 0 && g.O[q] == g.D[q] - 2;
  }
  if (!same && !valid) return false;
  ConvGeom base = g;
  base.Cout = 64; base.d2s = 1;
  return conv_mfma_trunk_supported(valid ? conv_dgrad_valid_geom(base) : conv_dgrad_geom(base), precision);
}

int launch_conv_dgrad_chunk_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, float* wt, int k) {
  hipLaunchKernelGGL(pack_dgrad_chunk_kernel, dim3(432), dim3(256), 0, ctx->stream, w, wt, g.Cout, k);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// valid-padded forward conv (lo = 0, O = D - 2): its data gradient is the full
// correlation of dPre with the flipped filter and lands directly on x's grid
// (no padded frame, no fold)
ConvGeom conv_dgrad_valid_geom(const ConvGeom& g) {
  ConvGeom d = conv_dgrad_geom(g);
  for (int q = 0; q < 3; ++q) d.O[q] = g.D[q];
  return d;
}

bool conv_dgrad_mfma_valid_supported(const ConvGeom& g, int precision) {
  if (g.pad_mode == S3_PAD_REFLECT || g.d2s != 1) return false;
  for (int q = 0; q < 3; ++q)
    if (g.k[q] != 3 || g.s[q] != 1 || g.lo[q] != 0 || g.O[q] != g.D[q] - 2) return false;
  return conv_mfma_trunk_supported(conv_dgrad_valid_geom(g), precision);
}

int launch_conv_dgrad_pack(s3_ctx* ctx, const ConvGeom& g, const float* w,
                           float* wt) {
  // (k = 3 x 3 x 1 too: reversing the linear tap index flips every axis)
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int64_t total = (int64_t)taps * g.Cin * g.Cout;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, wt, g.Cin, g.Cout, taps);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// the bf16 store handles 8 consecutive channels per thread
bool conv_mfma_bf16_out_ok(const ConvGeom& g) {
  return g.Cout % 8 == 0 && (g.Cout / (g.d2s * g.d2s)) % 8 == 0;
}

size_t conv_mfma_packed_bytes(const ConvGeom& g, int precision) {
  if (conv_mfma_is_gen(g, precision)) return conv_mfma_gen_packed_bytes(g, precision);
  if (precision == S3_PREC_BF16) {
    const int n_ct = (g.Cout + CT - 1) / CT;
    size_t b = (size_t)n_ct * g.k[0] * g.k[1] * g.k[2] * CT * CIN * 2;
    if (conv_mfma_persist_geom_ok(g) || conv_mfma_persist_dgrad_geom_ok(g)) b += conv_mfma_persist_image_bytes(g);
    return b;
  }
  if (precision == S3_PREC_BF16X3)   // hi | lo images of both channel halves
    return (size_t)((g.Cout + CT - 1) / CT) * 2 * g.k[0] * g.k[1] * g.k[2] * CT * CIN * 2;
  return 16;  // f32 mode reads the canonical weights directly
}

int launch_conv_mfma_pack(s3_ctx* ctx, const ConvGeom& g, int precision,
                          const float* w, void* packed) {
  if (conv_mfma_is_gen(g, precision)) return launch_conv_mfma_gen_pack(ctx, g, precision, w, packed);
  if (precision == S3_PREC_BF16X3) {
    const int taps3 = g.k[0] * g.k[1] * g.k[2];
    const int n_ct3 = (g.Cout + CT - 1) / CT;
    const int64_t total3 = (int64_t)n_ct3 * 2 * taps3 * CT * 32;
    int grid3 = (int)((total3 + 255) / 256);
    if (grid3 > 2048) grid3 = 2048;
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(grid3), dim3(256), 0, ctx->stream, w, (unsigned short*)packed, taps3, g.Cout, n_ct3);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (precision != S3_PREC_BF16) return S3_OK;
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int n_ct = (g.Cout + CT - 1) / CT;
  const int64_t total = (int64_t)n_ct * taps * CT * CIN;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, (unsigned short*)packed, taps, g.Cout, n_ct);
  S3_HIP(ctx, hipGetLastError());
  if (conv_mfma_persist_geom_ok(g) || conv_mfma_persist_dgrad_geom_ok(g))
    return launch_conv_mfma_persist_pack(ctx, g, w, (char*)packed + total * 2);
  return S3_OK;
}

int launch_conv_mfma_fwd(s3_ctx* ctx, const ConvGeom& g, int precision,
                         const void* x, const void* packed, const float* bias,
                         const void* res, void* y, ConvIO io) {
  if (conv_mfma_is_gen(g, precision))
    return launch_conv_mfma_gen_fwd(ctx, g, precision, x, packed, bias, res, y, io);
  if (precision == S3_PREC_BF16) {
    if (!g.in_cstride && conv_mfma_persist2_supported(ctx, g, io, res != nullptr))
      return launch_conv_mfma_persist2(ctx, g, x, (const char*)packed + (size_t)((g.Cout + CT - 1) / CT) * 27 * CT * CIN * 2, bias, res, y);
    if (!g.in_cstride && conv_mfma_persist_supported(ctx, g, io, res != nullptr))
      return launch_conv_mfma_persist(ctx, g, x, (const char*)packed + (size_t)((g.Cout + CT - 1) / CT) * 27 * CT * CIN * 2, bias, res, y);
    if (g.in_rep > 1 || g.res_rep > 1) S3_FAIL(ctx, S3_ESTATE, "conv with a fused temporal repeat off the persistent kernel");
    // tile / wave configuration (SUP3R_AMD_MFMA_TILE overrides for A/B probes)
    const int tile_env = (int)s3_opt_int(S3O_MFMA_TILE, -1);
    int tile = tile_env;
    if (tile < 0) {
      // 512-position workgroups (16 waves) amortise the filter slabs best;
      // fall back to 128-position ones when the grid would not fill the chip
      const int64_t big = (int64_t)g.N * ((g.O[0] + 3) / 4) * ((g.O[1] + 7) / 8) * ((g.O[2] + 15) / 16);
      tile = big >= ctx->num_cu ? 4 : 3;
      // extents that are multiples of 6 rather than of 4 / 8 (the 18 x 18 x 290
      // padded frame of the trunk's data gradient): 6 x 6 x 16 tiles waste
      // 5 % of the positions instead of 55 %; compare rounds x tile size
      const int64_t six = (int64_t)g.N * ((g.O[0] + 5) / 6) * ((g.O[1] + 5) / 6) * ((g.O[2] + 15) / 16);
      const int64_t ncu = ctx->num_cu;
      if (tile == 4 && six >= ncu && ((six + ncu - 1) / ncu) * 576 < ((big + ncu - 1) / ncu) * 512 &&
          !s3_opt_has(S3O_NO_TILE66))
        tile = 7;
    }
    if (tile == 1) return launch_bf16<2, 4, 4>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 2) return launch_bf16<4, 4, 8>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 3) return launch_bf16<2, 4, 8>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 4) return launch_bf16<4, 8, 16>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 6) return launch_bf16<4, 8, 8>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 7) return launch_bf16<6, 6, 12>(ctx, g, x, packed, bias, res, y, io);
    return launch_bf16<4, 4, 4>(ctx, g, x, packed, bias, res, y, io);
  }
  if (precision == S3_PREC_BF16X3) {
    // fp32 activations, split on the fly; the 512-position tile when it fills
    // the chip, the 128-position one otherwise
    const int64_t big = (int64_t)g.N * ((g.O[0] + 3) / 4) * ((g.O[1] + 7) / 8) * ((g.O[2] + 15) / 16);
    // (option MFMA_TILE = 3: the 128-position tile on a full-size problem —
    // the rows-per-filter-slab ablation of profiles/r05/winograd_ablation.md)
    if (big >= ctx->num_cu && s3_opt_int(S3O_MFMA_TILE, -1) != 3)
      return launch_io<S3_PREC_BF16X3, 4, 8, 8, false, false>(ctx, g, x, packed, bias, res, y, 0);
    return launch_io<S3_PREC_BF16X3, 2, 4, 8, false, false>(ctx, g, x, packed, bias, res, y, 0);
  }
  // f32: the filters are read in canonical layout; `packed` is unused
  return launch_io<S3_PREC_F32, 2, 4, 4, false, false>(ctx, g, x, packed, bias, res, y, 0);
}
