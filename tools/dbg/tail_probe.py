#!/usr/bin/env python
"""the hi-res tail conv (Conv3D 8 -> 2) of gen_5x_12x_2f at the bench shape: plane-sweep
kernel vs the slide kernel (option NO_TAIL_SWEEP) vs the tile kernel (NO_TAIL_SLIDE):
per-op HIP-event time, bytes in + out over it, output bits compared.
python tools/dbg/tail_probe.py [batch]"""
import json
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs', 'gen_5x_12x_2f.json')))
shape = (batch, 16, 16, 24, 4)
x = np.random.default_rng(0).standard_normal(shape).astype(np.float32)
net = Network(spec, precision='bf16')
net.build(shape, seed=3)
dev = net.dev
xd = dev.to_device(x)
ys = {}
for name, opts in (('sweep', {}), ('slide (NO_TAIL_SWEEP)', {'NO_TAIL_SWEEP': 1}),
                   ('tile (NO_TAIL_SLIDE)', {'NO_TAIL_SLIDE': 1})):
    ph = net.plan(shape, training=False, options=opts)
    out = dev.empty(tuple(ph.out_shape))
    for _ in range(3):
        ph.forward(xd, out=out)
    ph.profile_begin(10)
    for _ in range(10):
        ph.forward(xd, out=out)
    dev.sync()
    _, ms = ph.profile_end()
    ys[name] = out.cpu().numpy()
    cells = int(np.prod(ph.out_shape[:-1]))
    nb = cells * (16 + 8)
    print(f'{name}: tail conv {ms[-1] * 1e3:.1f} us = {nb / ms[-1] / 1e9:.2f} TB/s of in + out '
          f'= {nb / ms[-1] / 1e9 / 8:.3f} of 8 TB/s, kernel {ph.op_info(len(ph.plan.ops) - 1)["fwd"]}, '
          f'identical to sweep: {np.array_equal(ys[name], ys["sweep"])}', flush=True)
# forced plane shapes: SHAPES="40,72,80 20,72,40 ..."
for item in os.environ.get('SHAPES', '').split():
    s1, s2, seg = (int(v) for v in item.split(','))
    ph = net.plan(shape, training=False, options={'TAIL_SWEEP_SHAPE': s1 * 1000000 + s2 * 1000 + seg})
    out = dev.empty(tuple(ph.out_shape))
    for _ in range(3):
        ph.forward(xd, out=out)
    ph.profile_begin(10)
    for _ in range(10):
        ph.forward(xd, out=out)
    dev.sync()
    _, ms = ph.profile_end()
    print(f'sweep {item}: {ms[-1] * 1e3:.1f} us = {nb / ms[-1] / 1e9 / 8:.3f} of 8 TB/s, '
          f'identical: {np.array_equal(out.cpu().numpy(), ys["sweep"])}', flush=True)
