"""No kernel may read a plan buffer before something wrote it (``-m gpu``).

Option ``POISON_ALLOC`` fills every plan buffer (activation arena, packed
filter images, partial-sum and frame scratch) with all-ones bytes — NaN as
fp32 and as bf16 — instead of zeros when it is allocated.  A forward +
backward pass over such a plan must give the same bits as over a zero-filled
one: round 5 closed the one known exception (the gather-MFMA kernels' K-tail
over-read of their filter image), so correctness no longer
depends on allocation hygiene."""
import json
import os

import numpy as np
import pytest

from tests.helpers import switch

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')

CASES = [
    ('test_gen_st_2x_4x_2f.json', (2, 5, 6, 4, 3)),      # 3 -> 16 head, small channels
    ('test_gen_st_64ch.json', (1, 6, 5, 12, 3)),         # 64-channel MFMA trunk
    ('test_gen_st_3x_4x_2f_topo.json', (1, 4, 5, 4, 2)),  # C_in = 2 head, concat
    ('test_disc_st_valid.json', (2, 14, 13, 15, 2)),     # valid / strided, dense
    ('test_disc_st_same.json', (2, 12, 12, 16, 2)),
    ('test_gen_s_2x_2f.json', (3, 7, 6, 2)),             # 2-D, Conv2DTranspose
    ('sup3r/sup3rcc/gen_solar_5x_1x_1f.json', (2, 17, 16, 3)),   # logical-axes kernels
    ('sup3r/spatiotemporal/gen_2x_2x_2f.json', (1, 8, 9, 8, 2)),
]


def _run(spec, shape, prec, poison, exo_shape=None):
    from sup3r_amd import spec as S
    from sup3r_amd.engine import Network
    switch('POISON_ALLOC', 1 if poison else None)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape).astype(np.float32)
    net = Network(spec, precision=prec)
    net.build(shape, seed=3)
    dev = net.dev
    out = {}
    for training in (False, True):
        ph = net.plan(shape, training=training,
                      options={'POISON_ALLOC': 1} if poison else None)
        if training:
            # the switch has teeth: an activation buffer nobody wrote yet
            probe = ph.tensor(ph.plan.ops[0]['out'])
            assert np.isnan(probe).all() == bool(poison), (poison, probe.ravel()[:4])
        exo = {k: dev.to_device(rng.standard_normal(tuple(sh)).astype(
            np.float32)) for k, sh in ph.in_shapes.items() if k != 'x'}
        y = ph.forward(dev.to_device(x), exo).cpu().numpy()
        out[('y', training)] = y
        if training:
            dy = rng.standard_normal(y.shape).astype(np.float32)
            dx = ph.backward(dev.to_device(dy), need_dx=True).cpu().numpy()
            out['dx'] = dx
            out['grads'] = [np.array(g) for g in net.grads]
    return out


@pytest.mark.parametrize('cfg,shape', CASES)
@pytest.mark.parametrize('prec', ['f32', 'bf16', 'bf16x3'])
def test_poisoned_plan_buffers_change_nothing(cfg, shape, prec):
    with open(os.path.join(CFG, cfg)) as f:
        spec = json.load(f)
    a = _run(spec, shape, prec, poison=False)
    b = _run(spec, shape, prec, poison=True)
    for key in (('y', False), ('y', True), 'dx'):
        assert np.isfinite(b[key]).all(), key
        np.testing.assert_array_equal(a[key], b[key])
    for ga, gb in zip(a['grads'], b['grads']):
        assert np.isfinite(gb).all()
        np.testing.assert_array_equal(ga, gb)
