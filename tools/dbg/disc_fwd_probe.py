#!/usr/bin/env python
"""Per-op times of the discriminator forward (training plan, bf16) at the C2
hi-res batch.  python tools/dbg/disc_fwd_probe.py [--shape 8,80,80,288,2]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='8,80,80,288,2')
    ap.add_argument('--disc', default='disc_st.json')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--ops', type=int, default=4)
    args = ap.parse_args()
    import torch
    from sup3r_amd.engine import Network
    shape = tuple(int(v) for v in args.shape.split(','))
    spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs', args.disc)))
    spec = spec.get('hidden_layers', spec) if isinstance(spec, dict) else spec
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=True)
    x = net.dev.to_device(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
    for _ in range(3):
        ph.forward(x)
    torch.cuda.synchronize()
    ph.profile_begin(args.iters)
    for _ in range(args.iters):
        ph.forward(x)
    n, ms = ph.profile_end()
    kinds = [ph.op_info(i)['fwd'] for i in range(len(ph.plan.ops))]
    print(' '.join(f'{k}:{m * 1e3:.0f}us' for k, m in list(zip(kinds, ms))[:args.ops]),
          f'| all ops {sum(ms) * 1e3:.0f} us', flush=True)


if __name__ == '__main__':
    main()
