import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd.engine import Network
from sup3r_amd import spec as S
spec = json.load(open(os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs', 'sup3r', 'spatial', 'gen_2x_2f.json')))
rng = np.random.default_rng(0)
for hw in ((24, 22), (12, 13), (40, 40)):
    x = rng.standard_normal((42,) + hw + (2,)).astype(np.float32)
    net = Network(spec, precision='bf16'); net.build(x.shape, seed=1)
    outs = {}
    for n in (14, 42, 1):
        ph = net.plan((n,) + hw + (2,), training=False)
        sel = sorted(set(ph.op_info(i)['fwd'] for i, op in enumerate(ph.plan.ops) if op['kind'] == S.OP_CONV))
        y = ph.forward(net.dev.to_device(x[:n])).cpu().numpy()
        outs[n] = y
        print(hw, n, sel, flush=True)
    print(' 14 vs 42:', np.abs(outs[14] - outs[42][:14]).max(), ' 1 vs 42:', np.abs(outs[1] - outs[42][:1]).max())
