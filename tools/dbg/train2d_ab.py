"""A/B of a 2-D training plan: bf16 cells kept through the trunk (weights-
stationary forward, bf16-staging weight gradient) vs option NO_TRAIN2D_BF16.
usage: python tools/dbg/train2d_ab.py [config] [n,h,w,c]"""
import collections
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402

rel = sys.argv[1] if len(sys.argv) > 1 else 'sup3r/spatial/gen_2x_2f.json'
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '12,75,75,2').split(','))
spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs', rel)))
rng = np.random.default_rng(0)
net = Network(spec, precision='bf16')
net.build(shape, seed=1)
dev = net.dev
grads = {}
for name, opts in (('bf16 cells', {}), ('NO_TRAIN2D_BF16', {'NO_TRAIN2D_BF16': 1})):
    ph = net.plan(shape, training=True, options=opts)
    x = dev.to_device(rng.standard_normal(shape).astype(np.float32))
    exo = {k: dev.to_device(rng.standard_normal(tuple(sh)).astype(np.float32))
           for k, sh in ph.in_shapes.items() if k != 'x'}
    dy = dev.to_device(np.random.default_rng(1).standard_normal(
        tuple(ph.out_shape)).astype(np.float32))
    for _ in range(3):
        ph.forward(x, exo)
        ph.backward(dy, need_dx=False)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ph.forward(x, exo)
        ph.backward(dy, need_dx=False)
    dev.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    sel = collections.Counter()
    for i, op in enumerate(ph.plan.ops):
        if op['kind'] == S.OP_CONV:
            info = ph.op_info(i)
            sel[(f"{op['cin']}->{op['cout']}", info['fwd'], info['wgrad'], info['dgrad'],
                 'in16' if ph.tensor_is_bf16(op['in0']) else 'in32',
                 'out16' if ph.tensor_is_bf16(op['out']) else 'out32')] += 1
    print(name, '%.3f ms fwd+bwd' % ms)
    for k, v in sel.items():
        print('   ', k, 'x', v)
    g = ph.grads() if hasattr(ph, 'grads') else None
    del ph
    net.clear_plans()
