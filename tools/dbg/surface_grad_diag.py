"""per-gradient bf16 errors of one surface spec (tests/test_ref_surface.py)"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_ref_surface import CASES, load_surface
from tests.test_parity_r02 import _oracle, _hip
from tests.helpers import emulate_plan, teacher_forced_check, rel_max

rel = sys.argv[1] if len(sys.argv) > 1 else 'sup3rcc/gen_wind_1x_24x_6f.json'
shape, _ = CASES[rel]
if len(sys.argv) > 2:
    shape = tuple(int(v) for v in sys.argv[2].split(','))
spec = load_surface(rel)
for seed in (43, 44, 45):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = _oracle(spec, x, None, seed=seed)
    net = _hip(spec, ref.weights, 'bf16')
    dev = net.dev
    ph = net.plan(shape, training=True)
    y = ph.forward(dev.to_device(x)).cpu().numpy()
    y_ref = ref.forward(x)
    emulate_plan(ref, ph, masks=False)
    teacher_forced_check(ref, ph, x, None)
    emulate_plan(ref, ph, masks=True, rounding=False)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = ref.backward(dy)
    dx = ph.backward(dev.to_device(dy), need_dx=True).cpu().numpy().reshape(dx_ref.shape)
    gmax = max(float(np.abs(g).max()) for g in ref.grads)
    errs = [float(np.abs(g - gr).max() / max(np.abs(gr).max(), 1e-3 * gmax)) for g, gr in zip(net.grads, ref.grads)]
    print('seed', seed, 'dx', rel_max(dx, dx_ref))
    print(' '.join(f'{e:.1e}' for e in errs))
    infos = [ph.op_info(i) for i, op in enumerate(ph.plan.ops) if op['kind'] == 1]
    print([(i['fwd'], i['wgrad'], i['dgrad'], i['dgrad_frame16']) for i in infos[:6]], '...')
