#!/usr/bin/env python
"""What does a half-tile item of the persistent trunk kernel cost?  Three
64 -> 64 convs (bf16 in / out in the middle one) on (batch, rows, cols, t)
grids that differ only in the number of s0 rows.

    python tools/dbg/trunk_rows_probe.py --dims 20,16,624 22,16,624 24,16,624
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--dims', nargs='+', default=['20,16,624', '22,16,624',
                                                  '24,16,624'])
    args = ap.parse_args()
    import torch
    from sup3r_amd.configs.author_configs import pcc
    from sup3r_amd.engine import Network
    for dims in args.dims:
        d = [int(v) for v in dims.split(',')]
        spec = pcc(3, 64) + pcc(3, 64) + pcc(3, 64) + pcc(3, 64, act=False)
        net = Network(spec, precision='bf16')
        shape = (args.batch, d[0], d[1], d[2], 64)
        net.build(shape, seed=0)
        ph = net.plan(shape, training=False)
        x = net.dev.to_device(np.random.default_rng(0).standard_normal(
            shape).astype(np.float32))
        out = net.dev.empty(ph.out_shape)
        for _ in range(3):
            ph.forward(x, out=out)
        torch.cuda.synchronize()
        ph.profile_begin(args.iters)
        for _ in range(args.iters):
            ph.forward(x, out=out)
        n, ms = ph.profile_end()
        flop = 2.0 * args.batch * d[0] * d[1] * d[2] * 64 * 27 * 64
        cls = [ph.op_kernel_class(i) for i in range(len(ph.plan.ops))]
        print(dims, 'classes', cls, 'ms', ['%.4f' % m for m in ms],
              'middle TF/s %.1f' % (flop / (ms[1] * 1e-3) / 1e12), flush=True)
        del ph, net, x, out
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
