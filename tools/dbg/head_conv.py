"""conv2d_head_kernel vs the logical-axes MFMA kernel on the head layers alone: where do they differ?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.engine import Network
from sup3r_amd import spec as S
from tests.test_ref_surface import load_surface

spec = load_surface('spatial/gen_2x_2f.json')
hl = spec['hidden_layers']
head = hl[:3] + [{'class': 'FlexiblePadding', 'paddings': [[0, 0], [3, 3], [3, 3], [0, 0]], 'mode': 'REFLECT'},
                   {'class': 'Conv2DTranspose', 'filters': 64, 'kernel_size': 3, 'strides': 1},
                   {'class': 'Cropping2D', 'cropping': 4}]
shape = (3, 33, 37, 2)
x = (3.0 * np.random.default_rng(7).standard_normal(shape)).astype(np.float32)
net = Network({'hidden_layers': head}, precision='bf16')
net.build(shape, seed=3)
a = net.plan(shape, training=False)
b = net.plan(shape, training=False, options={'NO_CONV2D_HEAD': 1})
print([a.op_info(i)['fwd'] for i, op in enumerate(a.plan.ops) if op['kind'] == S.OP_CONV],
      [b.op_info(i)['fwd'] for i, op in enumerate(b.plan.ops) if op['kind'] == S.OP_CONV])
ya = a.forward(net.dev.to_device(x)).cpu().numpy()
yb = b.forward(net.dev.to_device(x)).cpu().numpy()
d = np.abs(ya - yb)
print('max', d.max(), 'scale', np.abs(yb).max(), 'n differing', (d > 0).sum(), 'of', d.size)
idx = np.argwhere(d.max(axis=-1) > 0)
print('rows', np.unique(idx[:, 1])[:40], 'cols', np.unique(idx[:, 2])[:40])

# the whole network: run-to-run determinism, head kernel vs matrix path, ping-pong vs lockstep trunk
net = Network(spec, precision='bf16')
net.build(shape, seed=3)
xd = net.dev.to_device(x)
plans = {'a': net.plan(shape, training=False), 'nohead': net.plan(shape, training=False, options={'NO_CONV2D_HEAD': 1}),
         'nopp': net.plan(shape, training=False, options={'NO_WS_PP': 1}),
         'nopp_nohead': net.plan(shape, training=False, options={'NO_WS_PP': 1, 'NO_CONV2D_HEAD': 1})}
out = {}
for k, ph in plans.items():
    y1 = ph.forward(xd).cpu().numpy()
    y2 = ph.forward(xd).cpu().numpy()
    out[k] = y1
    print(k, 'run-to-run max diff', np.abs(y1 - y2).max())
for k in ('nohead', 'nopp', 'nopp_nohead'):
    d = np.abs(out['a'] - out[k])
    print('a vs', k, d.max(), 'differing', (d > 0).sum(), 'of', d.size, 'scale', np.abs(out[k]).max())
