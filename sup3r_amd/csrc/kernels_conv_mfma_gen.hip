// Logical-axes instantiations of the halo-tile implicit-GEMM conv
// (conv_mfma_tile.h) — what the 64 -> C_out 3x3x3 trunk kernel of
// kernels_conv_mfma.hip leaves to the gather / direct kernels, found by the
// round-5 census of the reference's shipped generator specs
// (profiles/r05/config_census_before.md):
//
//   * 2-D nets (Conv2D / Conv2DTranspose stacks of sup3r/configs/spatial and
//     sup3rcc/gen_*_5x_1x_*): tensors are (N, s1, s2, 1, C); the logical conv is
//     3-D over (a0 = batch, a1 = s1, a2 = s2) with ONE tap along a0 — a tile is
//     TS0 images x TS1 rows x 16 columns, the halo has no rows along a0;
//   * 3-D convs over few time steps (sup3rcc/gen_solar_1x_8x_1f: T = 3): the
//     16-position run goes along s2 instead of t (a0 = t, a1 = s1, a2 = s2);
//   * any C_in <= 256 (K passes of 64 channels; zero cells past C_in: the
//     14 / 18 / 32 / 65 -channel layers around expansions and concats) and any
//     C_out (the 1 / 2 / 3 / 6 / 14-feature output convs on ONE N fragment; the
//     64 -> 72 expansion whose 18-channel depth-to-space cells are not a
//     multiple of 4).
//
// Only the addressing differs from the trunk kernel: cell strides per logical
// axis (ConvGeom::xs / ys / yb), a tap permutation in the filter pack.  The
// per-position arithmetic (pass -> tap -> k-step order on the bf16 MFMA, fp32
// accumulate) does not depend on the tile shape, so chunk-by-chunk and
// batched inference stay bit-identical.
#include "conv_mfma_tile.h"

namespace {

// canonical fp32 w[ptap][ci][co] -> bf16 slabs [ct][pass][ltap][co 64][ci 64]
// (16-B chunks pre-swizzled like pack_bf16_kernel); ptap = ta tp0 + tb tp1 + tc tp2
__global__ void pack_gen_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                     int ltaps, int cin, int cout, int n_ct, int npass, int tp0, int tp1,
                                     int tp2) {
  const int64_t total = (int64_t)n_ct * npass * ltaps * CT * CIN;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int cl = (int)(r % CIN); r /= CIN;
    const int row = (int)(r % CT); r /= CT;
    const int lt = (int)(r % ltaps); r /= ltaps;
    const int pass = (int)(r % npass); r /= npass;
    const int ct = (int)r;
    const int co = ct * CT + row, ci = pass * CIN + cl;
    const int ta = lt / 9, tb = (lt / 3) % 3, tc = lt % 3;
    const int pt = ta * tp0 + tb * tp1 + tc * tp2;
    const float v = (co < cout && ci < cin) ? w[((int64_t)pt * cin + ci) * cout + co] : 0.f;
    const int slot = (cl >> 3) ^ ((row >> 1) & 7);
    out[((((int64_t)ct * npass + pass) * ltaps + lt) * CT + row) * CIN + slot * 8 + (cl & 7)] = f2bf(v);
  }
}

// BF16X3: slabs [ct][pass (32 ch)][ltap][co 64][hi x 32 | lo x 32]
__global__ void pack_gen_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int ltaps,
                                   int cin, int cout, int n_ct, int npass, int tp0, int tp1, int tp2) {
  const int64_t total = (int64_t)n_ct * npass * ltaps * CT * 32;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int cl = (int)(r % 32); r /= 32;
    const int row = (int)(r % CT); r /= CT;
    const int lt = (int)(r % ltaps); r /= ltaps;
    const int pass = (int)(r % npass); r /= npass;
    const int ct = (int)r;
    const int co = ct * CT + row, ci = pass * 32 + cl;
    const int ta = lt / 9, tb = (lt / 3) % 3, tc = lt % 3;
    const int pt = ta * tp0 + tb * tp1 + tc * tp2;
    const float v = (co < cout && ci < cin) ? w[((int64_t)pt * cin + ci) * cout + co] : 0.f;
    const unsigned short hi = f2bf(v);
    const unsigned short lo = f2bf(v - __uint_as_float((unsigned)hi << 16));
    const int sw = (row >> 1) & 7;
    unsigned short* o = out + ((((int64_t)ct * npass + pass) * ltaps + lt) * CT + row) * CIN;
    o[((cl >> 3) ^ sw) * 8 + (cl & 7)] = hi;
    o[((4 + (cl >> 3)) ^ sw) * 8 + (cl & 7)] = lo;
  }
}

struct GenMap {
  ConvGeom l;        // logical geometry (gen = 1)
  int tp[3];         // physical tap-index stride of each logical tap coordinate
  int ka;            // taps along a0 (1 or 3)
};

// physical -> logical geometry, or false when the layer is not one this kernel
// family takes.  (Stride 1; every axis 'same' with k = 3, or k = 1 on the t
// axis.)
bool gen_map(const ConvGeom& p, int precision, GenMap* out) {
  if (precision != S3_PREC_BF16 && precision != S3_PREC_BF16X3) return false;
  if (s3_opt_has(S3O_NO_MFMA_GEN)) return false;
  if (p.in_cstride || p.in_rep > 1 || p.res_rep > 1 || p.gen) return false;
  if (p.Cin < 1 || p.Cout < 1 || p.Cin > (precision == S3_PREC_BF16X3 ? 128 : 256)) return false;
  if (p.k[0] != 3 || p.k[1] != 3 || (p.k[2] != 3 && p.k[2] != 1)) return false;
  for (int d = 0; d < 3; ++d) {
    if (p.s[d] != 1) return false;
    if (p.k[d] == 1) {
      if (p.lo[d] != 0 || p.O[d] != p.D[d]) return false;
      continue;
    }
  }
  // 'same' extents with the REFLECT boundary (the generators' pad / conv / crop
  // groups), or the full correlation over that conv's padded frame with a zero
  // boundary (its data gradient, conv_dgrad_gen_geom): the zero-padded 'same'
  // and the valid discriminator layers keep their own LDS-halo kernels
  bool same = p.pad_mode == S3_PAD_REFLECT, full = p.pad_mode == S3_PAD_ZERO;
  for (int d = 0; d < 3; ++d) {
    if (p.k[d] != 3) continue;
    same = same && p.lo[d] == 1 && p.O[d] == p.D[d];
    full = full && p.lo[d] == 2 && p.O[d] == p.D[d] + 2;
  }
  if (!same && !full) return false;
  if (full && (p.d2s > 1 || p.act != S3_ACT_NONE)) return false;
  // few-channel heads and hi-res tails with kernels of their own (gather-MFMA
  // with the taps in K, LDS-DMA tail kernels and their backward forms)
  // (3-D: tuned for C2 / the discriminators.  The 2 / 4-feature heads of the 2-D
  // nets took the per-tap gather walk at 114 us for 270 000 positions)
  if ((p.Cin == 2 || p.Cin == 4 || p.Cin == 8) && !(p.k[2] == 1 && p.D[2] == 1 && p.Cin != 8)) return false;
  // (the reference's filters: 1 placeholder nets: nothing to put on a matrix core)
  if (p.Cin < 5 && p.Cout < 64) return false;
  // (few channels on BOTH sides — the 5 -> 2 output conv of that net at 1.4 M
  // positions: 138 us on the direct kernel, 323 us with K padded to 64 and N to 16)
  if (p.Cin < 16 && p.Cout < 16) return false;
  const int b = p.d2s < 1 ? 1 : p.d2s;
  if (p.Cout % (b * b) != 0) return false;
  // launch-bound sizes (the reference's own 5 x 5 / 10 x 10 test shapes) stay
  // on the few-position / whole-network kernels; per SAMPLE, so that an
  // inference plan's kernels do not change with the batch size
  if ((int64_t)p.O[0] * p.O[1] * p.O[2] < 256) return false;
  if ((int64_t)p.N * p.D[0] * p.D[1] * p.D[2] >= ((int64_t)1 << 31) ||
      (int64_t)p.N * p.O[0] * p.O[1] * p.O[2] * b * b >= ((int64_t)1 << 31))
    return false;
  if (!out) return true;

  GenMap& m = *out;
  ConvGeom& l = m.l;
  l = p;
  l.gen = 1;
  l.d2s = b;
  // physical cell strides: input (n, s1, s2, t), final output (n, s1 b, s2 b, t)
  const int64_t xi[3] = {(int64_t)p.D[1] * p.D[2], p.D[2], 1};
  const int64_t xin = (int64_t)p.D[0] * p.D[1] * p.D[2];
  const int64_t yo[3] = {(int64_t)b * (p.O[1] * b) * p.O[2], (int64_t)b * p.O[2], 1};
  const int64_t yon = (int64_t)(p.O[0] * b) * (p.O[1] * b) * p.O[2];
  l.yb[0] = (int64_t)(p.O[1] * b) * p.O[2];
  l.yb[1] = p.O[2];
  // physical tap layout [k0][k1][k2]
  const int ts[3] = {p.k[1] * p.k[2], p.k[2], 1};
  auto put = [&](int q, int ax) {   // logical axis q <- physical axis ax
    l.D[q] = p.D[ax]; l.O[q] = p.O[ax]; l.k[q] = p.k[ax]; l.lo[q] = p.lo[ax]; l.s[q] = 1;
    l.xs[q] = xi[ax]; l.ys[q] = yo[ax];
    m.tp[q] = p.k[ax] > 1 ? ts[ax] : 0;
  };
  if (p.k[2] == 1 && p.D[2] == 1) {
    // 2-D net: a0 = batch, a1 = s1, a2 = s2
    l.N = 1; l.xn = 0; l.yn = 0;
    l.D[0] = p.N; l.O[0] = p.N; l.k[0] = 1; l.lo[0] = 0; l.s[0] = 1;
    l.xs[0] = xin; l.ys[0] = yon; m.tp[0] = 0;
    put(1, 0); put(2, 1);
  } else if (p.k[2] == 1 || p.D[2] < 8) {
    // few time steps: a0 = t, a1 = s1, a2 = s2
    l.N = p.N; l.xn = xin; l.yn = yon;
    put(0, 2); put(1, 0); put(2, 1);
  } else {
    l.N = p.N; l.xn = xin; l.yn = yon;
    put(0, 0); put(1, 1); put(2, 2);
  }
  m.ka = l.k[0];
  return true;
}

template <int PREC, int TS0, int TS1, int NW, int KA>
int launch_gen_tile(s3_ctx* ctx, const ConvGeom& l, const void* x, const void* wpk, const float* bias,
                    const void* res, void* y, ConvIO io) {
  const bool nf1 = l.Cout <= 16 && !io.out_bf16;
  if constexpr (PREC == S3_PREC_BF16X3) {
    if (nf1) return launch_io<PREC, TS0, TS1, NW, false, false, 1, KA, true>(ctx, l, x, wpk, bias, res, y, 0);
    return launch_io<PREC, TS0, TS1, NW, false, false, 4, KA, true>(ctx, l, x, wpk, bias, res, y, 0);
  } else {
    if (nf1) {
      // the few-feature output convs (fp32 out): one N fragment
      if (io.in_bf16) return launch_io<PREC, TS0, TS1, NW, true, false, 1, KA, true>(ctx, l, x, wpk, bias, res, y, io.res_bf16);
      return launch_io<PREC, TS0, TS1, NW, false, false, 1, KA, true>(ctx, l, x, wpk, bias, res, y, io.res_bf16);
    }
    if (io.in_bf16 && io.out_bf16)
      return launch_io<PREC, TS0, TS1, NW, true, true, 4, KA, true>(ctx, l, x, wpk, bias, res, y, io.res_bf16);
    if (io.in_bf16)
      return launch_io<PREC, TS0, TS1, NW, true, false, 4, KA, true>(ctx, l, x, wpk, bias, res, y, io.res_bf16);
    if (io.out_bf16)
      return launch_io<PREC, TS0, TS1, NW, false, true, 4, KA, true>(ctx, l, x, wpk, bias, res, y, io.res_bf16);
    return launch_io<PREC, TS0, TS1, NW, false, false, 4, KA, true>(ctx, l, x, wpk, bias, res, y, io.res_bf16);
  }
}

template <int PREC, int KA>
int launch_gen_prec(s3_ctx* ctx, const ConvGeom& l, const void* x, const void* wpk, const float* bias,
                    const void* res, void* y, ConvIO io) {
  // 512-position workgroups when they fill the chip, 256 / 128-position ones
  // otherwise (option MFMA_TILE: 4 / 3 / 5 forces one for A/B runs)
  const int64_t n512 = (int64_t)l.N * ((l.O[0] + 3) / 4) * ((l.O[1] + 7) / 8) * ((l.O[2] + 15) / 16);
  int tile = (int)s3_opt_int(S3O_MFMA_TILE, -1);
  if (tile != 3 && tile != 4 && tile != 5) tile = n512 >= 2 * ctx->num_cu ? 4 : (KA == 1 ? 5 : 3);
  if constexpr (PREC == S3_PREC_BF16X3) {
    if (tile == 4) return launch_gen_tile<PREC, 4, 8, 8, KA>(ctx, l, x, wpk, bias, res, y, io);
    return launch_gen_tile<PREC, 2, 4, 8, KA>(ctx, l, x, wpk, bias, res, y, io);
  } else {
    if (tile == 4) return launch_gen_tile<PREC, 4, 8, 16, KA>(ctx, l, x, wpk, bias, res, y, io);
    if (KA == 1 && tile == 5) return launch_gen_tile<PREC, 2, 8, 8, KA>(ctx, l, x, wpk, bias, res, y, io);
    return launch_gen_tile<PREC, 2, 4, 8, KA>(ctx, l, x, wpk, bias, res, y, io);
  }
}

}  // namespace

bool conv_mfma_gen_supported(const ConvGeom& g, int precision) { return gen_map(g, precision, nullptr); }

// data gradient of a reflect-'same' conv with k = 3 on some axes and k = 1 on the
// others: the full correlation of dPre with the flipped / transposed filter over
// the frame padded by one cell on the k = 3 axes, zero boundary; the fold
// (adjoint of the virtual padding) follows
ConvGeom conv_dgrad_gen_geom(const ConvGeom& g) {
  ConvGeom d = g;
  for (int q = 0; q < 3; ++q) {
    d.D[q] = g.O[q];
    if (g.k[q] == 3) { d.O[q] = g.D[q] + 2; d.lo[q] = 2; }
    else { d.O[q] = g.D[q]; d.lo[q] = 0; }
  }
  d.Cin = g.Cout; d.Cout = g.Cin;
  d.pad_mode = S3_PAD_ZERO; d.act = S3_ACT_NONE; d.alpha = 0.f; d.d2s = 1;
  return d;
}

bool conv_dgrad_gen_supported(const ConvGeom& g, int precision) {
  if (s3_opt_has(S3O_NO_DGRAD_GEN)) return false;
  if (g.pad_mode != S3_PAD_REFLECT) return false;
  for (int q = 0; q < 3; ++q) {
    if (g.s[q] != 1 || (g.k[q] != 3 && g.k[q] != 1)) return false;
    if (g.k[q] == 3 ? (g.lo[q] != 1 || g.O[q] != g.D[q]) : (g.lo[q] != 0 || g.O[q] != g.D[q])) return false;
  }
  return gen_map(conv_dgrad_gen_geom(g), precision, nullptr);
}

bool conv_mfma_gen_in16_ok(const ConvGeom& g) { return g.Cin % 8 == 0; }

// the few-feature output conv's fragment image rides last (kernels_conv2d_out.hip)
static bool out_geom(const ConvGeom& g, int precision) {
  return (precision == S3_PREC_BF16 || precision == S3_PREC_BF16X3) && g.Cout <= 7 && !g.w_cin &&
         conv2d_ws_tail_geom_ok(g);
}
static size_t out_image_offset(const ConvGeom& g, int precision, size_t tile_bytes) {
  return tile_bytes + (precision == S3_PREC_BF16 ? conv2d_ws_image_bytes(g) : 0);
}

size_t conv_mfma_gen_packed_bytes(const ConvGeom& g, int precision) {
  GenMap m;
  if (!gen_map(g, precision, &m)) return 16;
  const int kch = precision == S3_PREC_BF16X3 ? 32 : 64;
  const int npass = (g.Cin + kch - 1) / kch;
  size_t b = (size_t)((g.Cout + CT - 1) / CT) * npass * m.ka * 9 * CT * CIN * 2;
  // the weights-stationary 2-D kernel's image rides behind the tile image
  if (precision == S3_PREC_BF16 && (conv2d_ws_geom_ok(g) || conv2d_ws_tail_geom_ok(g) || conv2d_ws_frame_geom_ok(g))) b += conv2d_ws_image_bytes(g);
  if (precision == S3_PREC_BF16X3 && conv2d_ws_geom_ok(g) && !conv2d_ws_tail_geom_ok(g)) b += conv2d_ws_x3_image_bytes(g);
  if (out_geom(g, precision)) b += conv2d_out_image_bytes(g);
  // ... or the few-feature head kernel's (exclusive: C_in 1 / 2 there, 64 above)
  if (conv2d_head_geom_ok(g)) b += conv2d_head_image_bytes(g);
  return b;
}

static size_t gen_tile_image_bytes(const ConvGeom& g, int precision, int ka) {
  const int kch = precision == S3_PREC_BF16X3 ? 32 : 64;
  return (size_t)((g.Cout + CT - 1) / CT) * ((g.Cin + kch - 1) / kch) * ka * 9 * CT * CIN * 2;
}

int launch_conv_mfma_gen_pack(s3_ctx* ctx, const ConvGeom& g, int precision, const float* w, void* packed) {
  GenMap m;
  if (!gen_map(g, precision, &m)) S3_FAIL(ctx, S3_ESTATE, "gen MFMA conv: pack of an unsupported geometry");
  const bool x3 = precision == S3_PREC_BF16X3;
  const int kch = x3 ? 32 : 64;
  const int npass = (g.Cin + kch - 1) / kch, ltaps = m.ka * 9, n_ct = (g.Cout + CT - 1) / CT;
  if (out_geom(g, precision)) {
    const int rc = launch_conv2d_out_pack(
        ctx, g, w, (char*)packed + out_image_offset(g, precision, gen_tile_image_bytes(g, precision, m.ka)));
    if (rc) return rc;
  }
  // (a conv with an exogenous channel split off runs on the weights-stationary
  // kernel only: its tile image is never read)
  if (g.w_cin || (g.ws_only && precision == S3_PREC_BF16 && (conv2d_ws_geom_ok(g) || conv2d_ws_tail_geom_ok(g) || conv2d_ws_frame_geom_ok(g))))
    return launch_conv2d_ws_pack(ctx, g, w, (char*)packed + gen_tile_image_bytes(g, precision, m.ka));
  const int64_t total = (int64_t)n_ct * npass * ltaps * CT * (x3 ? 32 : CIN);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (x3)
    hipLaunchKernelGGL(pack_gen_x3_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, (unsigned short*)packed, ltaps,
                       g.Cin, g.Cout, n_ct, npass, m.tp[0], m.tp[1], m.tp[2]);
  else
    hipLaunchKernelGGL(pack_gen_bf16_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, (unsigned short*)packed, ltaps,
                       g.Cin, g.Cout, n_ct, npass, m.tp[0], m.tp[1], m.tp[2]);
  S3_HIP(ctx, hipGetLastError());
  if (precision == S3_PREC_BF16X3 && conv2d_ws_geom_ok(g) && !conv2d_ws_tail_geom_ok(g) && !g.w_cin)
    return launch_conv2d_ws_x3_pack(ctx, g, w, (char*)packed + gen_tile_image_bytes(g, precision, m.ka));
  if (precision == S3_PREC_BF16 && (conv2d_ws_geom_ok(g) || conv2d_ws_tail_geom_ok(g) || conv2d_ws_frame_geom_ok(g)))
    return launch_conv2d_ws_pack(ctx, g, w, (char*)packed + gen_tile_image_bytes(g, precision, m.ka));
  if (conv2d_head_geom_ok(g))
    return launch_conv2d_head_pack(ctx, g, w, (char*)packed + gen_tile_image_bytes(g, precision, m.ka),
                                   precision == S3_PREC_BF16X3);
  return S3_OK;
}

int launch_conv_mfma_gen_fwd(s3_ctx* ctx, const ConvGeom& g, int precision, const void* x, const void* packed,
                             const float* bias, const void* res, void* y, ConvIO io) {
  GenMap m;
  if (!gen_map(g, precision, &m)) S3_FAIL(ctx, S3_ESTATE, "gen MFMA conv: launch of an unsupported geometry");
  if (io.in_bf16 && g.Cin % 8 != 0) S3_FAIL(ctx, S3_ESTATE, "gen MFMA conv: bf16 input needs C_in % 8 == 0");
  if (out_geom(g, precision) && conv2d_out_supported(g, precision, io, res != nullptr))
    return launch_conv2d_out(ctx, g, precision, x,
                             (const char*)packed + out_image_offset(g, precision, gen_tile_image_bytes(g, precision, m.ka)),
                             bias, y);
  if (conv2d_ws_supported(g, precision, io, res != nullptr))
    return launch_conv2d_ws(ctx, g, x, (const char*)packed + gen_tile_image_bytes(g, precision, m.ka), bias, res, y);
  if (conv2d_ws_x3_supported(g, precision, io, res != nullptr))
    return launch_conv2d_ws_x3(ctx, g, x, (const char*)packed + gen_tile_image_bytes(g, precision, m.ka), bias, res, y);
  if (conv2d_head_supported(g, precision, io, res != nullptr))
    return launch_conv2d_head(ctx, g, x, (const char*)packed + gen_tile_image_bytes(g, precision, m.ka), bias, y,
                              precision == S3_PREC_BF16X3);
  if (g.w_cin || g.ws_only || g.res2) S3_FAIL(ctx, S3_ESTATE, "conv planned for the weights-stationary kernel launched off it");
  if (precision == S3_PREC_BF16X3) {
    if (m.ka == 1) return launch_gen_prec<S3_PREC_BF16X3, 1>(ctx, m.l, x, packed, bias, res, y, io);
    return launch_gen_prec<S3_PREC_BF16X3, 3>(ctx, m.l, x, packed, bias, res, y, io);
  }
  if (m.ka == 1) return launch_gen_prec<S3_PREC_BF16, 1>(ctx, m.l, x, packed, bias, res, y, io);
  return launch_gen_prec<S3_PREC_BF16, 3>(ctx, m.l, x, packed, bias, res, y, io);
}
