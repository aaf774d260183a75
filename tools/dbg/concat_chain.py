"""debug: backward through two consecutive Sup3rConcatObs layers"""
import numpy as np
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_with_obs import gen_config
from tests.helpers import emulate_plan
from oracle.network import Network as ONet
from sup3r_amd.engine import Network

spec = gen_config()
rng = np.random.default_rng(8)
lr = rng.standard_normal((3, 10, 10, 2)).astype(np.float32)
exo = {k: rng.standard_normal((3, 20, 20, 1)).astype(np.float32)
       for k in ('u_10m_obs', 'v_10m_obs')}
og = ONet(spec)
og.init_weights(lr, exo, seed=2, bias_scale=0.1)
net = Network(spec, precision='f32')
net.set_weights(og.weights)
ph = net.plan(lr.shape, training=True)
dev = net.dev
print('inputs', ph.input_names, ph.in_shapes)
y = ph.forward(dev.to_device(lr), {k: dev.to_device(v.reshape(3, 20, 20, 1, 1)) for k, v in exo.items()}).cpu().numpy()
y_ref = og.forward(lr, exo)
print('fwd err', np.abs(y - y_ref).max())
emulate_plan(og, ph, masks=True, rounding=False)
dy = rng.standard_normal(y_ref.shape).astype(np.float32)
og.backward(dy)
ph.backward(dev.to_device(dy), need_dx=False)
for i, (a, b) in enumerate(zip(net.grads, og.grads)):
    print(i, a.shape, float(np.abs(a - b).max() / max(1e-12, np.abs(b).max())))
for i, op in enumerate(ph.plan.ops):
    print(i, {k: v for k, v in op.items() if k in ('kind', 'in0', 'in1', 'out', 'res', 'act', 'd2s')})
