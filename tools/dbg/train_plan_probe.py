#!/usr/bin/env python
"""forward + backward of ONE training plan of a shipped generator spec (the census's procedure) at a given
shape, with the kernel selection printed: python tools/dbg/train_plan_probe.py <config under sup3r/> <n,h,w[,t],c>"""
import collections
import json
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402
rel, shape = sys.argv[1], tuple(int(v) for v in sys.argv[2].split(','))
with open(os.path.join(ROOT, 'sup3r_amd', 'configs', 'sup3r', rel)) as f:
    spec = json.load(f)
rng = np.random.default_rng(0)
net = Network(spec, precision='bf16')
net.build(shape, seed=1)
dev = net.dev
pht = net.plan(shape, training=True)
xt = dev.to_device(rng.standard_normal(shape).astype(np.float32))
exot = {k: dev.to_device(rng.standard_normal(tuple(sh)).astype(np.float32)) for k, sh in pht.in_shapes.items() if k != 'x'}
dy = dev.to_device(rng.standard_normal(tuple(pht.out_shape)).astype(np.float32))
sel = collections.Counter()
for i, op in enumerate(pht.plan.ops):
    if op['kind'] == S.OP_CONV:
        info = pht.op_info(i)
        sel[(f"{op['cin']}->{op['cout']}", tuple(pht.plan.tensors[op['out']]), info['fwd'], info['wgrad'], info['dgrad'])] += 1
for k, v in sel.items():
    print(v, k, flush=True)
for _ in range(3):
    pht.forward(xt, exot)
    pht.backward(dy, need_dx=False)
dev.sync()
g = net.grads
print('ok: grads finite', all(np.isfinite(np.asarray(a)).all() for a in g), flush=True)
