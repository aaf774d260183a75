"""Race screen for the persistent trunk conv: N forwards on ragged and even
shapes, each compared with the one-tile-per-workgroup kernel (same bf16
operands).  Usage: python tools/persist_check.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from sup3r_amd.engine import Device  # noqa: E402
from sup3r_amd.configs.author_configs import pcc  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spec = pcc(3, 64) + [{'class': 'SkipConnection', 'name': 'a'}] + \
    pcc(3, 64) + pcc(3, 64, act=False) + \
    [{'class': 'SkipConnection', 'name': 'a'}] + pcc(3, 64) + \
    pcc(3, 64) + pcc(3, 2, act=False)
# (22 x 22: 11 half rows, the C3 chunk; 21 x 19: an odd number of rows)
for shape in [(5, 18, 20, 72, 4), (8, 16, 16, 64, 4), (3, 16, 24, 112, 4),
              (6, 22, 22, 48, 4), (6, 21, 19, 40, 4), (8, 22, 22, 208, 4),
              (4, 20, 13, 400, 4), (16, 16, 16, 288, 4)]:
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    Device.get().set_option('NO_PERSIST', 1)
    y0 = net(x).cpu().numpy()
    Device.get().set_option('NO_PERSIST', None)
    ph = net.plan(shape, training=False)
    k = [ph.op_kernel_class(i) for i in range(len(ph.plan.ops))]
    worst = 0.0
    for r in range(reps):
        y = net(x).cpu().numpy()
        worst = max(worst, float(np.abs(y - y0).max()))
    print(shape, 'persistent ops', k.count(2), 'max|persist - tile|', worst,
          'scale', float(np.abs(y0).max()), flush=True)
