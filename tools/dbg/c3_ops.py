#!/usr/bin/env python
"""per-op HIP-event profile of the C3 chunk-batch plan: gen_5x_12x_2f at (16, 22, 22, 52, 4), bf16"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S
from sup3r_amd.engine import Network
b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spec = json.load(open(os.path.join(ROOT, 'sup3r_amd/configs/gen_5x_12x_2f.json')))
shape = (b, 22, 22, 52, 4)
net = Network(spec, precision='bf16')
net.build(shape, seed=0)
ph = net.plan(shape, training=False)
dev = net.dev
x = dev.to_device(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
out = dev.empty(tuple(ph.out_shape))
for _ in range(3):
    ph.forward(x, out=out)
ph.profile_begin(10)
for _ in range(10):
    ph.forward(x, out=out)
dev.sync()
_, ms = ph.profile_end()
tot = sum(ms)
print(f'total {tot:.3f} ms per batch of {b} = {b / tot * 1e3:.0f} chunks/s (kernels only)')
for i, op in enumerate(ph.plan.ops):
    info = ph.op_info(i)
    print(i, op['kind'], op.get('cin'), op.get('cout'), ph.plan.tensors[op['out']], info.get('fwd'), f'{ms[i]*1e3:.1f} us', 'in_rep', info.get('in_rep'))
