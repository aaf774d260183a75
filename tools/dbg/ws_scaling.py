"""asymptotic rate of the weights-stationary 2-D conv: 6 x (64 -> 64) behind a 4 -> 64 head"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.configs.author_configs import pcc
from sup3r_amd.engine import Network
from sup3r_amd import spec as S

spec = pcc(2, 64)
for _ in range(8):
    spec = spec + pcc(2, 64)
spec = spec + pcc(2, 2, act=False)
opts = [None]
if len(sys.argv) > 1:
    for a in sys.argv[1:]:
        k, _, v = a.partition('=')
        opts.append({k: int(v or 1)})
for shape in ([(48, 75, 75, 6), (480, 75, 75, 6)] if len(sys.argv) > 1 else [(48, 75, 75, 6), (96, 75, 75, 6), (480, 75, 75, 6), (48, 150, 150, 6), (60, 16, 16, 6), (60, 80, 80, 6), (64, 64, 64, 6), (512, 64, 64, 6)]):
    for o in opts:
        net = Network(spec, precision='bf16')
        net.build(shape, seed=1)
        ph = net.plan(shape, training=False, options=o)
        x = net.dev.to_device(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
        for _ in range(3):
            ph.forward(x)
        ph.profile_begin(10)
        for _ in range(10):
            ph.forward(x)
        net.dev.sync()
        _, ms = ph.profile_end()
        sel = [ph.op_info(i)['fwd'] for i, op in enumerate(ph.plan.ops) if op['kind'] == S.OP_CONV]
        t = [m for m, s in zip(ms, sel) if s in ('conv2d_ws',)] or [m for m, s in zip(ms, sel)][1:-1]
        per = float(np.mean(t))
        fl = 2 * shape[0] * shape[1] * shape[2] * 9 * 64 * 64
        print(shape, o, sel[1], f'{per*1e3:.1f} us/conv, {fl/per/1e9:.0f} TF/s, {shape[0]*shape[1]*shape[2]*256/per/1e9:.2f} TB/s', flush=True)
