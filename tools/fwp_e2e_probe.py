"""End-to-end ForwardPass throughput (C3-style): lo-res domain in host memory ->
hi-res domain in host memory, chunked (20,20,48) with halo padding.
Usage: python tools/fwp_e2e_probe.py [--batched]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--domain', default='80,80,192')
    ap.add_argument('--batched', action='store_true')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--direct', action='store_true')
    ap.add_argument('--repeat', type=int, default=1,
                    help='timed runs; the first one also pays the page faults '
                         'of the fresh output array and the batched plan build')
    args = ap.parse_args()
    import torch
    from sup3r_amd import ChunkSlicer, ForwardPass, Sup3rGan
    d = tuple(int(v) for v in args.domain.split(','))
    feats = ['u_10m', 'v_10m', 'temp', 'pres']
    Sup3rGan.seed(0)
    means = {f: np.float32(0.1 * i) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.0 + 0.1 * i) for i, f in enumerate(feats)}
    model = Sup3rGan(os.path.join(CFG, 'gen_5x_12x_2f.json'),
                     os.path.join(CFG, 'disc_st.json'), means=means,
                     stdevs=stds, precision='bf16')
    model.set_model_params(lr_features=feats, hr_out_features=feats[:2],
                           s_enhance=5, t_enhance=12)
    slicer = ChunkSlicer(d[:2], d[2], 5, 12, (20, 20, 48), spatial_pad=2,
                         temporal_pad=4)
    rng = np.random.default_rng(0)
    domain = rng.standard_normal(d + (4,)).astype(np.float32)
    out = np.zeros(slicer.hr_shape + (2,), np.float32)
    fwp = ForwardPass(model, slicer)
    # warm-up (plans, weights)
    fwp.run_domain_chunk(domain, 0)
    for rep in range(args.repeat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.batched:
            n = fwp.run_batched(domain, out=out, batch=args.batch,
                                n_host_threads=args.threads,
                                direct_placement=args.direct)
        else:
            n = fwp.run_domain(domain, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'run {rep}: domain {d}: {n} chunks in {dt:.3f} s = {n / dt:.1f} '
              f'chunks/s ({"batched" if args.batched else "sequential"}), '
              f'hi-res {out.nbytes / 2**30:.2f} GiB, checksum '
              f'{float(out.mean()):.6f}')


if __name__ == '__main__':
    main()
