#!/usr/bin/env python
"""Times the structured content losses (value + gradient) on a hi-res batch:
python tools/loss_probe.py [--shape 8,80,80,288,2]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='8,80,80,288,2')
    args = ap.parse_args()
    import torch
    from sup3r_amd import _lib
    from sup3r_amd.compute import (SLOTS_PER_TERM, HipGanCompute,
                                   parse_loss_spec)
    from sup3r_amd.engine import Device
    shape = tuple(int(v) for v in args.shape.split(','))
    cp = HipGanCompute.__new__(HipGanCompute)
    cp.dev = Device.get()
    cp._scal = None
    dev = cp.dev
    g = torch.randn(shape, device='cuda')
    t = torch.randn(shape, device='cuda')
    d = torch.zeros_like(g)
    scal = cp._scalars()
    L = _lib.lib()
    specs = ['MeanAbsoluteError', 'ExpLoss', 'MaterialDerivativeLoss',
             'SpatialDerivativeLoss', 'TemporalDerivativeLoss', 'CoarseMseLoss',
             'SpatialExtremesLoss', 'TemporalExtremesLoss',
             {'LowResLoss': {'s_enhance': 5, 't_enhance': 12}}, 'MmdLoss',
             'SpatiotemporalFftLoss', 'SlicedWassersteinLoss']
    for spec in specs:
        (name, kind, w, kw), = parse_loss_spec(spec)

        def run():
            if isinstance(kind, str):
                cp._structured_term(name, kind, kw, g, t, shape[-1], w, scal, 4, d)
            else:
                rc = L.s3_loss_content(dev.ctx, kind, cp._ptr(g), shape[-1],
                                       cp._ptr(t), shape[-1], shape[-1],
                                       g.numel() // shape[-1], w,
                                       cp._ptr(scal, 4), cp._ptr(d), 1)
                _lib.check(rc, dev.ctx, 's3_loss_content')
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        print(f'{name:26s} {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms '
              '(value + gradient)')


if __name__ == '__main__':
    main()
