import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from tests.test_parity_r02 import _oracle, _hip, _load
from tests.helpers import switch

spec = _load('disc_st_same.json')
shape = (2, 12, 12, 16, 2)
rng = np.random.default_rng(31)
x = rng.standard_normal(shape).astype(np.float32)
ref = _oracle(spec, x, None, seed=31)
outs = {}
for mode in ('fused', 'nofuse'):
    switch('NO_MASK_FUSE', 1 if mode == 'nofuse' else None)
    net = _hip(spec, ref.weights, 'f32')
    ph = net.plan(shape, training=True)
    for i in range(len(ph.plan.ops)):
        inf = ph.op_info(i)
        if inf['kind'] == 1 and mode == 'fused':
            print(i, {k: inf[k] for k in ('fwd', 'wgrad', 'dgrad', 'fewpos_mfma', 'mask_fused_from')}, ph.plan.tensors[ph.plan.ops[i]['out']])
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    dy = np.random.default_rng(4).standard_normal(y.shape).astype(np.float32)
    dx = ph.backward(net.dev.to_device(dy), need_dx=True).cpu().numpy()
    outs[mode] = [dx] + [np.array(g) for g in net.grads]
    net.clear_plans()
for i, (a, b) in enumerate(zip(outs['fused'], outs['nofuse'])):
    print(i, a.shape, float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)))
