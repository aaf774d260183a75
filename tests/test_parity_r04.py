"""GPU parity tests (``-m gpu``) added in round 4.

* ``conv3_mfma_persist2_kernel`` (option ``PERSIST2``: 128-position consumer
  waves, filter fragments from L1 / L2, four barriers per tile) against
  ``conv3_mfma_persist_kernel``: the same MFMA sequence per output element,
  so the results must be bit-identical — ragged tiles, odd row counts,
  SkipConnection residuals, C_out = 200 with the depth-to-space store and its
  32-wide last channel tile (sup3r/models/abstract.py:1131-1173 runs these
  layers through keras Conv3D + SpatioTemporalExpansion).
"""
import numpy as np
import pytest

from tests.helpers import switch

pytestmark = pytest.mark.gpu


def _trunk_spec(tail_d2s=False):
    from sup3r_amd.configs.author_configs import pcc
    spec = pcc(3, 64) + [{'class': 'SkipConnection', 'name': 'a'}] + \
        pcc(3, 64) + pcc(3, 64, act=False) + \
        [{'class': 'SkipConnection', 'name': 'a'}] + pcc(3, 64)
    if tail_d2s:
        spec += pcc(3, 200, act=False) + \
            [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
             {'alpha': 0.2, 'class': 'LeakyReLU'}] + pcc(3, 2, act=False)
    else:
        spec += pcc(3, 2, act=False)
    return spec


@pytest.mark.parametrize('shape,d2s', [
    ((16, 16, 16, 48, 4), False),      # whole tiles
    ((6, 22, 22, 48, 4), False),       # the C3 chunk: 11 half rows, ragged s1
    ((6, 21, 19, 40, 4), False),       # odd rows, ragged everywhere
    ((8, 16, 16, 32, 4), True),        # 64 -> 200 + depth-to-space, 4 channel tiles
])
def test_persist2_is_bit_identical_to_the_persistent_kernel(shape, d2s):
    from sup3r_amd.engine import Network
    spec = _trunk_spec(d2s)
    x = np.random.default_rng(5).standard_normal(shape).astype(np.float32)

    def run(on):
        switch('PERSIST2', 1 if on else None)
        net = Network(spec, precision='bf16')
        net.build(shape, seed=3)
        ph = net.plan(shape, training=False)
        kinds = [ph.op_kernel_class(i) for i in range(len(ph.plan.ops))]
        assert kinds.count(2) >= 2, kinds
        ys = [ph.forward(net.dev.to_device(x)).cpu().numpy() for _ in range(3)]
        net.clear_plans()
        return ys
    ref = run(False)
    got = run(True)
    switch('PERSIST2', None)
    assert np.isfinite(ref[0]).all() and np.abs(ref[0]).max() > 0
    for y in got + ref[1:]:
        np.testing.assert_array_equal(y, ref[0])
