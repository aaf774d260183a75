import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.engine import Network
spec = json.load(open(os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs', 'sup3r', 'spatial', 'gen_2x_2f.json')))
shape = (48, 75, 75, 2)
x = None
for opts in (None, {'GRAPH': 1}):
    net = Network(spec, precision='bf16'); net.build(shape, seed=0)
    ph = net.plan(shape, training=False, options=opts)
    if x is None:
        x = net.dev.to_device(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
    out = net.dev.empty(tuple(ph.out_shape))
    for _ in range(5):
        ph.forward(x, out=out)
    net.dev.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        ph.forward(x, out=out)
    net.dev.sync()
    print(opts, f'{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per forward', float(out.abs().mean()), flush=True)
