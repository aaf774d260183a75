// Stand-alone timing of the elementwise backward passes (frame fold, mask
// pass, bias gradient) at the C2 trunk geometry on rotating cold buffers.
#include "../../sup3r_amd/csrc/kernels_misc.hip"
#include <vector>
int main() {
  s3_ctx ctx; hipStreamCreate(&ctx.stream);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); ctx.num_cu = pr.multiProcessorCount;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  GatherGeom fg{}; fg.kind = S3_OP_PAD; fg.N = 8;
  const int D[3] = {22, 22, 30};
  for (int q = 0; q < 3; ++q) { fg.Di[q] = D[q]; fg.Do[q] = D[q] + 2; fg.lo[q] = 1; }
  fg.Ci = fg.Co = 64; fg.pad_mode = S3_PAD_REFLECT; fg.rep = 1; fg.d2s = 1;
  const size_t n = (size_t)8 * 22 * 22 * 30 * 64, nf = (size_t)8 * 24 * 24 * 32 * 64;
  const int NB = 24;
  std::vector<float*> fr(NB), in(NB), out(NB);
  for (int i = 0; i < NB; ++i) {
    hipMalloc(&fr[i], nf * 4); hipMalloc(&in[i], n * 4); hipMalloc(&out[i], n * 4);
    hipMemsetAsync(fr[i], 0, nf * 4, ctx.stream); hipMemsetAsync(in[i], 0, n * 4, ctx.stream);
  }
  ConvGeom cg{}; cg.N = 8; cg.O[0] = 22; cg.O[1] = 22; cg.O[2] = 30; cg.Cout = 64; cg.d2s = 1; cg.act = S3_ACT_LEAKY; cg.alpha = 0.2f;
  float* db; hipMalloc(&db, 1024);
  for (int kind = 0; kind < 6; ++kind)
    for (int pass = 0; pass < 2; ++pass) {
      hipEventRecord(e0, ctx.stream);
      for (int i = 0; i < NB; ++i) {
        int rc = 0;
        if (kind == 0) rc = launch_gather_bwd(&ctx, fg, fr[i], out[i]);
        if (kind == 1) rc = launch_gather_bwd_masked(&ctx, fg, fr[i], out[i], in[i], 1, 0.2f);
        if (kind == 2) rc = launch_gather_bwd_add(&ctx, fg, fr[i], out[i], in[i]);
        if (kind == 3) rc = launch_conv_epilogue_bwd(&ctx, cg, in[i], in[(i + 1) % NB], out[i], 0);
        if (kind == 4) rc = launch_bias_grad(&ctx, in[i], (int64_t)n / 64, 64, db, 0);
        if (kind == 5) rc = launch_axpy(&ctx, in[i], out[i], n);
        if (rc) { printf("rc %d %s\n", rc, ctx.err.c_str()); return 1; }
      }
      hipEventRecord(e1, ctx.stream); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const char* nm[] = {"fold", "fold+mask16", "fold+add", "mask pass", "bias grad", "axpy"};
      if (pass) printf("%-12s %8.1f us/launch\n", nm[kind], ms * 1e3 / NB);
    }
  return 0;
}
