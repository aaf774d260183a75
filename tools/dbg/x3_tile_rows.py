"""BF16X3 trunk conv of C2: the 512-position tile (4 x 8 x 16, 8 waves) against
the 128-position one (2 x 4 x 16) — the MFMA work per 8 KB filter slab a
Winograd-domain kernel could afford (profiles/r05/winograd_ablation.md)"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.engine import Network
from sup3r_amd import spec as S

spec = json.load(open(os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs', 'gen_5x_12x_2f.json')))
B = 16
shape = (B, 16, 16, 24, 4)
x = None
for opts in (None, {'MFMA_TILE': 3}):
    net = Network(spec, precision='bf16x3')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=False, options=opts)
    if x is None:
        x = net.dev.to_device(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
    for _ in range(2):
        ph.forward(x)
    ph.profile_begin(4)
    for _ in range(4):
        ph.forward(x)
    net.dev.sync()
    _, ms = ph.profile_end()
    body = [i for i, op in enumerate(ph.plan.ops) if op['kind'] == S.OP_CONV and op['cin'] == 64 and op['cout'] == 64
            and ph.plan.tensors[op['out']][3] == 288]
    t = float(np.mean([ms[i] for i in body]))
    fl = 2.0 * B * 16 * 16 * 288 * 27 * 64 * 64
    print(opts, f'{len(body)} trunk convs, {t:.3f} ms each, {fl / t / 1e9:.0f} TFLOP/s fp32-equivalent, '
          f'{3 * fl / t / 1e9 / 2500:.3f} of the bf16 MFMA peak', flush=True)
