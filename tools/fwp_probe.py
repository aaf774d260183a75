#!/usr/bin/env python
"""C3-style inference probe: generator forward on forward-pass chunks
(B, 20, 20, 48, 4) -> (B, 100, 100, 576, 2) incl. optional halo padding."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='4,20,20,48,4')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--precision', default='bf16')
    args = ap.parse_args()
    import torch
    from sup3r_amd.engine import Network
    spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs',
                                       'gen_5x_12x_2f.json')))
    shape = tuple(int(v) for v in args.shape.split(','))
    net = Network(spec, precision=args.precision)
    net.build(shape, seed=0)
    ph = net.plan(shape)
    x = net.dev.to_device(np.random.default_rng(0).standard_normal(shape)
                          .astype(np.float32))
    out = net.dev.empty(ph.out_shape)
    for _ in range(2):
        ph.forward(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        ph.forward(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    macs = 299.47e9 * (shape[1] * shape[2] * shape[3]) / (16 * 16 * 24)
    print(f'chunks {shape} -> {tuple(ph.out_shape)} {args.precision}: '
          f'{dt * 1e3:.2f} ms/step, {shape[0] / dt:.1f} chunks/s, '
          f'{2 * macs * shape[0] / dt / 1e12:.1f} TFLOP/s, workspace '
          f'{ph.workspace_bytes / 2**30:.2f} GiB')


if __name__ == '__main__':
    main()
