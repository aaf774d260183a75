"""single-conv probes of the logical-axes MFMA kernel vs the oracle"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.configs.author_configs import pcc
from sup3r_amd.engine import Network
from oracle.network import Network as ON
from sup3r_amd import spec as S

def run(nd, cin, cout, shape_sp, n=2, prec='bf16x3', d2s=0):
    spec = pcc(nd, cout, act=False)
    if d2s:
        spec = spec + [{'class': 'SpatialExpansion' if nd == 2 else 'SpatioTemporalExpansion', 'spatial_mult': d2s}]
    shape = (n,) + tuple(shape_sp) + (cin,)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = ON(spec); ref.init_weights(x, seed=1, bias_scale=0.1)
    y_ref = ref.forward(x)
    net = Network(spec, precision=prec); net.set_weights(ref.weights)
    ph = net.plan(shape, training=False)
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    sel = [ph.op_info(i)['fwd'] for i, op in enumerate(ph.plan.ops) if op['kind'] == S.OP_CONV]
    err = np.abs(y - y_ref).max() / max(1, np.abs(y_ref).max())
    print(f'{prec} nd{nd} {cin}->{cout} d2s{d2s} {shape}: err {err:.2e} {sel}', flush=True)


for prec in ('bf16', 'bf16x3'):
    for cin in (64, 129, 192, 200, 256) if prec == 'bf16' else (64, 65, 96):
        run(2, cin, 64, (20, 18), prec=prec)
        run(2, cin, 64, (20, 18), n=1, prec=prec)
    run(3, 65, 64, (8, 9, 8), prec=prec)
    run(3, 40, 64, (8, 9, 4), prec=prec)
    run(2, 64, 100, (20, 18), prec=prec, d2s=5)
    run(2, 64, 6, (20, 18), prec=prec)
