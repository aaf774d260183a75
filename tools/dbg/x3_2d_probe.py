#!/usr/bin/env python
"""spatial/gen_2x_2f forward in the BF16X3 mode at the config_fwp_spatial.json chunk
(48, 75, 75, 2): weights-stationary X3 kernel vs the logical-axes tile kernel (option
NO_WS_X3), per-op HIP-event times, and both against the fp32 oracle on one image.
python tools/dbg/x3_2d_probe.py [batch] [h] [w]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 48
h = int(sys.argv[2]) if len(sys.argv) > 2 else 75
w = int(sys.argv[3]) if len(sys.argv) > 3 else 75
rel = os.environ.get('SPEC', 'spatial/gen_2x_2f.json')
with open(os.path.join(ROOT, 'sup3r_amd', 'configs', 'sup3r', rel)) as f:
    spec = json.load(f)
cin = S.build_plan(S.parse_layers(spec), (1, 16, 16, 2)).ops[0]['cin'] if 'CIN' not in os.environ \
    else int(os.environ['CIN'])
shape = (b, h, w, cin)
x = np.random.default_rng(1).standard_normal(shape).astype(np.float32)
net = Network(spec, precision='bf16x3')
net.build(shape, seed=0)
dev = net.dev
xd = dev.to_device(x)
outs = {}
DBG = int(os.environ.get('DBG', '0'))
for name, opts in (('ws_x3', {'MFMA_DBG': DBG} if DBG else {}), ('tile (NO_WS_X3)', {'NO_WS_X3': 1})):
    ph = net.plan(shape, training=False, options=opts)
    sel = [ph.op_info(i)['fwd'] for i, op in enumerate(ph.plan.ops) if op['kind'] == S.OP_CONV]
    out = dev.empty(tuple(ph.out_shape))
    for _ in range(3):
        ph.forward(xd, out=out)
    dev.sync()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        ph.forward(xd, out=out)
    dev.sync()
    dt = (time.perf_counter() - t0) / n
    ph.profile_begin(n)
    for _ in range(n):
        ph.forward(xd, out=out)
    dev.sync()
    _, ms = ph.profile_end()
    trunk = [i for i, op in enumerate(ph.plan.ops) if op['kind'] == S.OP_CONV and op['cin'] == 64 and op['cout'] == 64]
    t_ms = float(np.mean([ms[i] for i in trunk]))
    flop = 2.0 * b * h * w * 9 * 64 * 64
    print(f'{name}: {dt * 1e3:.3f} ms per forward = {b / dt:.0f} samples/s; 64->64 conv {t_ms * 1e3:.1f} us = '
          f'{flop / t_ms / 1e9:.0f} TFLOP/s fp32-equivalent; kernels '
          f'{ {k: sel.count(k) for k in sorted(set(sel))} }', flush=True)
    print('   per-op ms:', [round(v, 3) for v in ms][:40])
    outs[name] = out.cpu().numpy()
a, c = outs['ws_x3'], outs['tile (NO_WS_X3)']
print('ws_x3 vs tile kernel: max |diff|', float(np.abs(a - c).max()), 'scale', float(np.abs(c).max()))
if os.environ.get('ORACLE', '1') == '1':
    from oracle.network import Network as ONet
    ref = ONet(spec)
    ref.init_weights(x[:1, :8, :8], seed=0)
    ref.set_weights(net.weights)
    y = ref.forward(x[:1])
    print('vs the fp32 oracle (image 0): ws_x3 L-inf', float(np.abs(a[:1] - y).max()),
          'tile', float(np.abs(c[:1] - y).max()), 'scale', float(np.abs(y).max()))
