#!/usr/bin/env python
"""Uninitialised-read screen: fill the torch caching allocator's free blocks
with NaN, then run forward + backward of a small bf16 training plan and look
for NaN in the results (python tools/poison_check.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def poison(n_mb=2048):
    import torch
    blocks = [torch.full((s * 262144,), float('nan'), device='cuda')
              for s in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512) for _ in range(2)]
    big = torch.full((n_mb * 262144,), float('nan'), device='cuda')
    del blocks, big


def main():
    import torch
    from sup3r_amd.configs.author_configs import pcc
    from sup3r_amd.engine import Network
    rng = np.random.default_rng(6)
    spec = pcc(3, 64) + [{'class': 'SkipConnection', 'name': 'a'}] + \
        pcc(3, 64) + pcc(3, 64, act=False) + \
        [{'class': 'SkipConnection', 'name': 'a'}] + \
        pcc(3, 200, act=False) + \
        [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
         {'alpha': 0.2, 'class': 'LeakyReLU'}]
    shape = (2, 5, 7, 19, 4)
    x = rng.standard_normal(shape).astype(np.float32)
    for prec in ('f32', 'bf16'):
        poison()
        net = Network(spec, precision=prec)
        net.build(shape, seed=0)
        poison()
        ph = net.plan(shape, training=True)
        y = ph.forward(net.dev.to_device(x))
        dy = torch.randn_like(y)
        dx = ph.backward(dy, need_dx=True)
        bad = [i for i, g in enumerate(net.grads) if not np.isfinite(g).all()]
        print(prec, 'y nan', bool(torch.isnan(y).any()), 'dx nan',
              bool(torch.isnan(dx).any()), 'grads with nan', bad)


if __name__ == '__main__':
    main()
