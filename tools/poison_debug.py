#!/usr/bin/env python
"""Locate reads of plan buffers before their first write: run with
SUP3R_AMD_POISON_ALLOC=1 (all-ones bytes in every new plan buffer) and report
where the results differ from a clean run."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sup3r_amd.engine import Device  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run(spec, shape, training, seed=21, prec='bf16'):
    from sup3r_amd.engine import Network
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    net = Network(spec, precision=prec)
    net.build(shape, seed=0)
    ph = net.plan(shape, training=training)
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    out = {'y': y}
    if training:
        dy = rng.standard_normal(y.shape).astype(np.float32)
        out['dx'] = ph.backward(net.dev.to_device(dy), need_dx=True).cpu().numpy()
        for i, g in enumerate(net.grads):
            out['g%d' % i] = np.array(g)
    return out


def main():
    from sup3r_amd.configs.author_configs import pcc
    cfgd = os.path.join(ROOT, 'sup3r_amd', 'configs')
    cases = {
        'gen_2x_2f': (json.load(open(os.path.join(cfgd, 'gen_2x_2f.json'))), (3, 9, 8, 2)),
        'chunked': (pcc(3, 64) + pcc(3, 200, act=False) +
                    [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
                     {'alpha': 0.2, 'class': 'LeakyReLU'}], (2, 5, 7, 19, 4)),
    }
    full = json.load(open(os.path.join(cfgd, 'gen_2x_2f.json')))['hidden_layers']
    cases = {
        'pad+convT(2->64,relu)+crop': (full[0:3], (3, 9, 8, 2)),
        'pad+convT(64->64)+crop': (full[5:8], (3, 9, 8, 64)),
        'pad+convT(64->256)+crop+d2s+relu': (full[9:14], (3, 9, 8, 64)),
        'pad+convT(64->2)+crop': (full[14:17], (3, 18, 16, 64)),
        'chunked': cases['chunked'],
    }
    for name, (spec, shape) in cases.items():
        if isinstance(spec, dict):
            spec = spec['hidden_layers']
        for training in (True, False):
            Device.get().set_option('POISON_ALLOC', None)
            clean = run(spec, shape, training)
            Device.get().set_option('POISON_ALLOC', 1)
            if name == 'chunked' and training:
                Device.get().set_option('TRACE', 1)
            dirty = run(spec, shape, training)
            Device.get().set_option('TRACE', None)
            if name == 'chunked' and training:
                Device.get().set_option('NO_DGRAD_CHUNKED', 1)
                d2 = run(spec, shape, training)
                Device.get().set_option('NO_DGRAD_CHUNKED', None)
                for k in d2:
                    if np.isnan(d2[k]).any():
                        print('chunked (gather dgrad)', k, 'nan at', np.argwhere(np.isnan(d2[k]))[:4].tolist())
            for k in clean:
                a, b = clean[k], dirty[k]
                bad = ~np.isclose(a, b, rtol=0, atol=0, equal_nan=False)
                if bad.any():
                    idx = np.argwhere(bad)
                    print(f'{name} training={training} {k} shape {a.shape}: {bad.sum()} differ, '
                          f'nan {np.isnan(b).sum()}, first {idx[0]}, last {idx[-1]}, '
                          f'clean {a[tuple(idx[0])]:.4g} dirty {b[tuple(idx[0])]:.4g}')
            print(f'{name} training={training} done')


if __name__ == '__main__':
    main()
