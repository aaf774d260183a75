"""debug: per-op times of the C2 generator forward at the C3 chunk shape"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.engine import Device, Network
import torch
spec = json.load(open('sup3r_amd/configs/gen_5x_12x_2f.json'))
dev = Device.get()
for shape, opts in (((8, 22, 22, 52, 4), {}), ((8, 22, 22, 52, 4), {'NO_FEWCH_HALO': 1}),
                    ((32, 16, 16, 24, 4), {}), ((8, 24, 24, 64, 4), {})):
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=False, options=opts)
    x = dev.to_device(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
    for _ in range(2):
        ph.forward(x)
    torch.cuda.synchronize()
    ph.profile_begin(4)
    for _ in range(4):
        ph.forward(x)
    torch.cuda.synchronize()
    n, ms = ph.profile_end()
    print(shape, opts, 'op0', ph.op_info(0)['fwd'], '%.3f ms' % ms[0], 'total %.2f' % sum(ms))
    del ph
    net.clear_plans()
