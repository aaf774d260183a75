#!/usr/bin/env python
"""the few-feature 2-D output conv on its own: 64 -> C_out (3 x 3 reflect) behind one
64-channel conv, bf16 and BF16X3 plans, conv2d_out kernel vs option NO_CONV2D_OUT, per-op
HIP-event time and both against the fp32 oracle.  python tools/dbg/out_conv_probe.py [n h w cout]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.configs.author_configs import pcc  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402
from oracle.network import Network as ONet  # noqa: E402
n, h, w, cout = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (48, 150, 150, 2)))
spec = pcc(2, 64) + pcc(2, cout, act=False)
shape = (n, h, w, 2)
x = np.random.default_rng(0).standard_normal(shape).astype(np.float32)
ref = ONet(spec)
ref.init_weights(x[:1, :8, :8], seed=3, bias_scale=0.1)
y_ref = ref.forward(x[:1])
for prec in ('bf16', 'bf16x3'):
    net = Network(spec, precision=prec)
    net.set_weights(ref.weights)
    dev = net.dev
    xd = dev.to_device(x)
    for name, opts in (('conv2d_out', {}), ('before (NO_CONV2D_OUT)', {'NO_CONV2D_OUT': 1})):
        ph = net.plan(shape, training=False, options=opts)
        out = dev.empty(tuple(ph.out_shape))
        for _ in range(3):
            ph.forward(xd, out=out)
        ph.profile_begin(10)
        for _ in range(10):
            ph.forward(xd, out=out)
        dev.sync()
        _, ms = ph.profile_end()
        yv = out.cpu().numpy()
        cells = n * h * w
        nb = cells * 64 * (2 if prec == 'bf16' else 4) + cells * cout * 4
        err = float(np.abs(yv[:1] - y_ref).max())
        print(f'{prec} {name}: output conv {ms[-1] * 1e3:.1f} us = {nb / ms[-1] / 1e9:.2f} TB/s of in + out, '
              f'kernel {ph.op_info(len(ph.plan.ops) - 1)["fwd"]}, L-inf vs oracle {err:.2e} '
              f'(scale {np.abs(y_ref).max():.2f})', flush=True)
