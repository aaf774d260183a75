"""hipGraph replay vs eager launches on the launch-bound configs: C1
(gen_2x_2f, 36 Conv2DTranspose layers on (15,5,5,2)) and a C3 chunk.
Usage: python tools/graph_probe.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sup3r_amd.engine import Device  # noqa: E402
import torch  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402

for cfg, shape, prec in [('gen_2x_2f.json', (15, 5, 5, 2), 'f32'),
                         ('gen_2x_2f.json', (15, 5, 5, 2), 'bf16'),
                         ('gen_5x_12x_2f.json', (1, 20, 20, 48, 4), 'bf16')]:
    spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs', cfg)))
    net = Network(spec, precision=prec)
    net.build(shape, seed=0)
    ph = net.plan(shape)
    rng = np.random.default_rng(0)
    x = net.dev.to_device(rng.standard_normal(shape).astype(np.float32))
    out = net.dev.empty(ph.out_shape)
    Device.get().set_option('GRAPH', None)
    ph.forward(x, out=out)
    ref = out.clone()
    res = {}
    for mode in ('eager', 'graph'):
        if mode == 'graph':
            Device.get().set_option('GRAPH', 1)
        for _ in range(4):
            ph.forward(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 200 if shape[0] == 15 else 20
        for _ in range(iters):
            ph.forward(x, out=out)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / iters * 1e3
        assert torch.equal(out, ref), mode
    print(f'{cfg} {shape} {prec}: eager {res["eager"]:.3f} ms, graph '
          f'{res["graph"]:.3f} ms per forward (bit-identical outputs)')
