import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S
from sup3r_amd.engine import Network
rel = sys.argv[1] if len(sys.argv) > 1 else 'spatial/gen_10x_2f.json'
spec = json.load(open(os.path.join(ROOT, 'sup3r_amd/configs/sup3r', rel)))
shape = (3, 16, 17, 2)
x = np.random.default_rng(53).standard_normal(shape).astype(np.float32)
net = Network(spec, precision='bf16x3')
net.build(shape, seed=1)
dev = net.dev
K = {'KEEP_ACTIVATIONS': 1, 'NO_DIRECT_OUTPUT': 1}
ph = net.plan(shape, training=False, options=K)
ph1 = net.plan((1,) + shape[1:], training=False, options=K)
for rep in range(2):
    y = ph.forward(dev.to_device(x)).cpu().numpy()
    for k in range(3):
        yk = ph1.forward(dev.to_device(x[k:k + 1])).cpu().numpy()
        bad = []
        for i, op in enumerate(ph.plan.ops):
            a = ph.tensor(op['out'])[k]
            b = ph1.tensor(op['out'])[0]
            if not np.array_equal(a, b):
                d = np.argwhere(a != b)
                bad.append((i, ph.op_info(i)['fwd'], op.get('cout'), op.get('d2s'), len(d), d[:3].tolist(), float(np.abs(a - b).max())))
        print('rep', rep, 'image', k, 'equal' if not bad else bad[:3])
