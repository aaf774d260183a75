#!/usr/bin/env python
"""Micro-probe: ONE generator-body layer (REFLECT pad 3 -> Conv3D 64->64 k3 ->
crop 2 [-> LeakyReLU]) on the C2 grid, looped; used under rocprofv3 (--pmc
passes, kernel trace) to study conv3_mfma_kernel in isolation.

    python tools/conv_probe.py --batch 8 --iters 20 --precision bf16
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--cout', type=int, default=64)
    ap.add_argument('--dims', default='16,16,288')
    args = ap.parse_args()
    import torch
    from sup3r_amd.configs.author_configs import pcc
    from sup3r_amd.engine import Network
    d = [int(v) for v in args.dims.split(',')]
    spec = pcc(3, args.cout)
    net = Network(spec, precision=args.precision)
    shape = (args.batch, d[0], d[1], d[2], 64)
    net.build(shape, seed=0)
    ph = net.plan(shape)
    x = net.dev.to_device(
        np.random.default_rng(0).standard_normal(shape).astype(np.float32))
    out = net.dev.empty(ph.out_shape)
    for _ in range(3):
        ph.forward(x, out=out)
    torch.cuda.synchronize()
    ph.profile_begin(args.iters)
    for _ in range(args.iters):
        ph.forward(x, out=out)
    n, ms = ph.profile_end()
    dt = ms[0] * 1e-3          # HIP-event time of the conv op alone
    flop = 2.0 * args.batch * d[0] * d[1] * d[2] * 64 * 27 * args.cout
    print(f'{args.precision} cout={args.cout} batch={args.batch} '
          f'tile={os.environ.get("SUP3R_AMD_MFMA_TILE", "0")}: '
          f'{dt * 1e3:.4f} ms/launch (HIP events, {n} launches), '
          f'{flop / dt / 1e12:.1f} TFLOP/s')


if __name__ == '__main__':
    main()
