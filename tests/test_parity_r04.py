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
    ((8, 16, 16, 64, 4), True),        # 64 -> 200 + depth-to-space, 4 channel tiles
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


def test_bf16x3_lds_halo_data_gradients_of_the_hires_discriminator_layers():
    """``conv_dgrad_c2_x3_kernel`` (32 -> 2 full correlation behind the first
    discriminator layer) and ``conv_dgrad_s2_x3_kernel`` (32 -> 32 stride 2),
    the split-bf16 forms BF16X3 plans use since round 4, on ragged tiles:
    forward 1e-4 / every gradient 1e-3 against the fp32 oracle under the
    device's masks (the BF16X3 bounds of tests/test_parity_r02.py), the two
    kernels selected, and agreement with the gather-MFMA adjoint they replace
    (option ``NO_DGRAD_X3``) to fp32 summation order.  Reference:
    sup3r/models/base.py:283-313 (``_tf_discriminate``) under
    ``abstract.py:1190-1238``."""
    from sup3r_amd import spec as S
    from sup3r_amd.engine import Network
    from tests.helpers import rel_max
    from tests.test_parity_r02 import _fwd_bwd_vs_oracle

    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(32, 2) + conv(16, 1)
    shape = (2, 21, 23, 37, 2)
    switch('DGRAD_S2_MIN_TILES', 1)
    ph = _fwd_bwd_vs_oracle(spec, shape, 'bf16x3', 53, 1e-4, 1e-3)
    dg = [ph.op_info(i)['dgrad'] for i, op in enumerate(ph.plan.ops)
          if op['kind'] == S.OP_CONV]
    assert dg[0] == 'c2' and dg[1] == 's2', dg
    rng = np.random.default_rng(54)
    x = rng.standard_normal(shape).astype(np.float32)

    def run():
        net = Network(spec, precision='bf16x3')
        net.build(shape, seed=7)
        p = net.plan(shape, training=True)
        y = p.forward(net.dev.to_device(x))
        dy = net.dev.to_device(np.random.default_rng(55).standard_normal(
            tuple(y.shape)).astype(np.float32))
        dx = p.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        net.clear_plans()
        return dx, g
    dx1, g1 = run()
    switch('NO_DGRAD_X3', 1)
    dx0, g0 = run()
    switch('NO_DGRAD_X3', None)
    assert np.abs(dx1 - dx0).max() > 0            # another kernel ran
    assert rel_max(dx1, dx0) < 1e-4, rel_max(dx1, dx0)
    for a, b in zip(g1, g0):
        assert rel_max(a, b) < 1e-4


@pytest.mark.parametrize('shape,d2s', [((8, 22, 22, 208, 4), False),
                                       ((4, 20, 13, 400, 4), False),
                                       ((8, 19, 9, 256, 4), False),
                                       ((8, 22, 22, 104, 4), True)])
def test_last_column_strip_on_six_column_tiles_is_bit_identical(shape, d2s):
    """An s1 extent of 8 k + r, 1 <= r <= 6 (the C3 chunk's 22 = 8 + 8 + 6)
    runs its last columns as a strip of 6-column tiles
    (``conv3_mfma_persist_kernel<.., TW = 6>``, second launch of the conv):
    same MFMA sequence per output as the 8-column tiles, so bit-identical to
    the single launch (``NO_PERSIST_STRIP``) — with residuals, r = 6 / 5 / 1,
    and the 64 -> 200 conv with its depth-to-space store (four channel tiles,
    the last one 32 wide)."""
    from sup3r_amd.engine import Network
    spec = _trunk_spec(d2s)
    x = np.random.default_rng(6).standard_normal(shape).astype(np.float32)

    def run(strip):
        switch('NO_PERSIST_STRIP', None if strip else 1)
        net = Network(spec, precision='bf16')
        net.build(shape, seed=4)
        ph = net.plan(shape, training=False)
        kinds = [ph.op_kernel_class(i) for i in range(len(ph.plan.ops))]
        assert kinds.count(2) >= 2, kinds
        ys = [ph.forward(net.dev.to_device(x)).cpu().numpy() for _ in range(3)]
        net.clear_plans()
        return ys
    got = run(True)
    ref = run(False)
    switch('NO_PERSIST_STRIP', None)
    assert np.isfinite(ref[0]).all() and np.abs(ref[0]).max() > 0
    for y in got + ref[1:]:
        np.testing.assert_array_equal(y, ref[0])


@pytest.mark.parametrize('padding,shape', [
    ('reflect', (2, 11, 13, 45, 8)),
    ('reflect', (1, 4, 8, 32, 8)),
    ('same', (3, 9, 10, 70, 8)),
])
def test_bf16x3_tail_conv_on_the_banded_split_mfma_kernel(padding, shape):
    """``conv_tail_x3_kernel``: the 8 -> 2 tail conv of a BF16X3 plan (fp32 in
    / out) as banded split-bf16 MFMAs instead of the direct fp32 kernel (4.4 ms
    -> of the 52.9 ms C2 forward at 32 chunks).  Forward 1e-4 against the fp32
    oracle (the BF16X3 forward bound of tests/test_parity_r02.py) on ragged
    tiles, reflect (FlexiblePadding) and zero ('same') borders, with the
    kernel selected; and against the direct kernel it replaces
    (``NO_TAIL_X3``).  Reference: the last Conv3D of
    sup3r/models/abstract.py:926-987 (``generate`` layer walk)."""
    from sup3r_amd.engine import Network
    from tests.helpers import rel_max
    from tests.test_parity_r02 import _fwd_bwd_vs_oracle
    if padding == 'reflect':
        spec = [{'class': 'FlexiblePadding',
                 'paddings': [[0, 0], [1, 1], [1, 1], [1, 1], [0, 0]],
                 'mode': 'REFLECT'},
                {'class': 'Conv3D', 'filters': 2, 'kernel_size': 3,
                 'strides': 1, 'padding': 'valid'}]
    else:
        spec = [{'class': 'Conv3D', 'filters': 2, 'kernel_size': 3,
                 'strides': 1, 'padding': 'same'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    ph = _fwd_bwd_vs_oracle(spec, shape, 'bf16x3', 61, 1e-4, 1e-3)
    fwd = [ph.op_info(i)['fwd'] for i in range(len(ph.plan.ops))]
    assert 'tail_mfma' in fwd, fwd
    x = np.random.default_rng(62).standard_normal(shape).astype(np.float32)

    def run():
        net = Network(spec, precision='bf16x3')
        net.build(shape, seed=9)
        p = net.plan(shape, training=False)
        y = p.forward(net.dev.to_device(x)).cpu().numpy()
        net.clear_plans()
        return y
    y1 = run()
    switch('NO_TAIL_X3', 1)
    y0 = run()
    switch('NO_TAIL_X3', None)
    assert np.abs(y1 - y0).max() > 0              # another kernel ran
    assert rel_max(y1, y0) < 2e-5, rel_max(y1, y0)


def test_inference_forward_writes_the_callers_buffer_directly():
    """An inference plan's last op writes ``s3_plan_forward``'s output buffer
    itself (no device-to-device copy of the whole model output; option
    ``NO_DIRECT_OUTPUT`` restores the copy): same bits, also into a buffer
    that is a window of a larger allocation, and ``out=None`` still returns
    the plan's own result."""
    import torch
    from sup3r_amd.engine import Network
    spec = _trunk_spec(True) + [
        {'class': 'FlexiblePadding',
         'paddings': [[0, 0], [1, 1], [1, 1], [1, 1], [0, 0]],
         'mode': 'REFLECT'},
        {'class': 'Conv3D', 'filters': 2, 'kernel_size': 3, 'strides': 1,
         'padding': 'valid'}]
    shape = (3, 9, 10, 40, 4)
    x = np.random.default_rng(8).standard_normal(shape).astype(np.float32)
    net = Network(spec, precision='bf16')
    net.build(shape, seed=3)
    ph = net.plan(shape, training=False)
    xd = net.dev.to_device(x)
    y_own = ph.forward(xd).cpu().numpy()
    n = int(np.prod(ph.out_shape))
    big = torch.full((n + 64,), 7.0, dtype=torch.float32, device='cuda')
    out = big[32:32 + n].view(*ph.out_shape)
    y_direct = ph.forward(xd, out=out).cpu().numpy()
    guard = big.cpu().numpy()
    assert (guard[:32] == 7.0).all() and (guard[32 + n:] == 7.0).all()
    switch('NO_DIRECT_OUTPUT', 1)          # (a plan keeps its options: new plan)
    ph2 = net.plan(shape, training=False)
    assert ph2 is not ph
    y_copy = ph2.forward(xd, out=net.dev.empty(ph.out_shape)).cpu().numpy()
    switch('NO_DIRECT_OUTPUT', None)
    y_again = ph.forward(xd).cpu().numpy()
    assert np.abs(y_own).max() > 0
    for y in (y_direct, y_copy, y_again):
        np.testing.assert_array_equal(y, y_own)


@pytest.mark.parametrize('shape,units', [
    ((11, 5, 5, 15, 4), (260, 512)),    # C_in 1500 (a ragged last 16-row step), ragged column tile, two batch passes
    ((3, 4, 4, 16, 8), (1024, 256)),    # C_in 2048
])
def test_dense_layers_on_the_16_byte_walks_vs_oracle(shape, units):
    """dense_fwd4_stage1 / dense_dgrad4_kernel / dense_wgrad4_kernel (a lane
    owns four consecutive outputs, a wave four rows of W per step; taken when
    C_in and C_out are multiples of 4 and C_out >= 256) against the oracle:
    forward, input gradient and weight gradients of Flatten + Dense stacks
    (sup3r/models/base.py:283-313 runs the discriminator's keras Dense
    layers), batch > 8 (two register passes), ragged column tiles and slabs."""
    from tests.test_parity_r02 import _fwd_bwd_vs_oracle
    spec = [{'class': 'Flatten'}]
    for u in units:
        spec += [{'class': 'Dense', 'units': u}, {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec += [{'class': 'Dense', 'units': 1}]
    _fwd_bwd_vs_oracle(spec, shape, 'f32', 31, 1e-4, 1e-3)
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 32, 1e-3, 1e-3)


def _fewpos_specs():
    from sup3r_amd.configs.author_configs import pcc

    def conv(nd, f, s=1, pad='valid', act=True):
        out = [{'class': f'Conv{nd}D', 'filters': f, 'kernel_size': 3,
                'strides': s, 'padding': pad}]
        if act:
            out.append({'alpha': 0.2, 'class': 'LeakyReLU'})
        return out
    # the C1 generator's shapes: reflect-padded 64 -> 64 convs with skip
    # connections, a 64 -> 256 conv + depth-to-space, 7 x 5 positions per sample
    gen2d = pcc(2, 64) + [{'class': 'SkipConnection', 'name': 'a'}] + \
        pcc(2, 64) + pcc(2, 64, act=False) + \
        [{'class': 'SkipConnection', 'name': 'a'}] + pcc(2, 256) + \
        [{'class': 'SpatialExpansion', 'spatial_mult': 2}] + pcc(2, 16)
    # the C1 discriminator's shapes: strides 1 / 2, valid and 'same' padding,
    # C_out 48 (one ragged 64-wide channel tile), Flatten + Dense behind
    disc2d = conv(2, 32) + conv(2, 64, 2, 'same') + conv(2, 48, 1, 'same') + \
        conv(2, 128, 2) + [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    # 27 taps, zero 'same' padding, a stride-2 layer
    st3d = conv(3, 16) + conv(3, 48, 1, 'same') + conv(3, 32, 2, 'same') + \
        conv(3, 16, 1, 'same', act=False)
    return [('gen2d', gen2d, (5, 7, 5, 16)),
            ('disc2d', disc2d, (3, 14, 13, 2)),
            ('st3d', st3d, (2, 6, 5, 7, 3))]


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
@pytest.mark.parametrize('case', [0, 1, 2])
def test_fewpos_one_launch_mfma_kernels_vs_oracle_and_split_k_family(case, precision):
    """``fewpos_mfma_kernel<0 / 1>`` and ``fewpos_wgrad_mfma_kernel`` (round 4:
    forward conv + epilogue, data gradient from the untransposed filter, weight
    + bias gradient — ONE launch each on ``v_mfma_f32_16x16x4_f32``; the
    activation adjoint applied as dy is read) against
    the oracle (forward 1e-4, every gradient 1e-4 under the device's masks:
    exact fp32 products) and against the split-K weight-streaming family they
    replace (option ``NO_FEWPOS_MFMA``) to fp32 summation order.  Ragged
    16-position tiles, reflect / zero / no padding, strides 1 and 2, skip
    connections, depth-to-space, C_out 48 / 256, 9 and 27 taps.  Reference:
    sup3r/models/abstract.py:1131-1238 runs these layers through keras
    Conv2D / Conv3D under ``tf.GradientTape``."""
    from tests.test_parity_r02 import _fwd_bwd_vs_oracle, _oracle, _hip
    name, spec, shape = _fewpos_specs()[case]
    ph = _fwd_bwd_vs_oracle(spec, shape, precision, 11 + case, 1e-4 if precision == 'f32' else 3e-2,
                            1e-4 if precision == 'f32' else 3e-2)
    infos = [ph.op_info(i) for i in range(len(ph.plan.ops))]
    n_new = sum(1 for i in infos if i['kind'] == 1 and i['fewpos_mfma'])
    assert n_new >= 2, (name, infos)

    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = _oracle(spec, x, None, seed=5)

    def run(old, side=False, fuse=True, small=True):
        switch('NO_FEWPOS_MFMA', 1 if old else None)
        switch('NO_FEWPOS_SMALL', None if small else 1)
        switch('WGRAD_SIDE_STREAM', 1 if side else None)
        switch('NO_FEWPOS_BWD_FUSE', None if fuse else 1)
        net = _hip(spec, ref.weights, precision)
        p = net.plan(shape, training=True)
        flags = [p.op_info(i)['fewpos_mfma'] for i in range(len(p.plan.ops))]
        assert (sum(flags) == 0) == bool(old)
        y = p.forward(net.dev.to_device(x)).cpu().numpy()
        dy = np.random.default_rng(4).standard_normal(y.shape).astype(np.float32)
        dx = p.backward(net.dev.to_device(dy), need_dx=True).cpu().numpy()
        out = [y, dx] + [np.array(g) for g in net.grads]
        if not old and fuse and not side:
            # a second shard: the kernels add into dW / db (abstract.py:785-805)
            p.backward(net.dev.to_device(dy), need_dx=True, accumulate_wgrad=True)
            for g1, g2 in zip(out[2:], net.grads):
                np.testing.assert_array_equal(np.array(g2), 2 * g1)
        net.clear_plans()
        return out
    new, new2, ser, old = run(False), run(False), run(False, side=True), run(True)
    switch('NO_FEWPOS_MFMA', None)
    # fixed summation order; with option WGRAD_SIDE_STREAM the weight gradients
    # run on a branch beside the data-gradient chain, joined before the pass
    # returns, and read finished buffers only: same bits
    # ... and so does the data + weight gradient pair as one launch
    # (``fewpos_bwd_kernel``) against the two separate launches
    two = run(False, fuse=False)
    for a, b, c, d in zip(new, new2, ser, two):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)
        np.testing.assert_array_equal(a, d)
    # (launch-bound plans also route their few-channel head / tail convs to
    # these kernels — exact fp32 where the gather-MFMA kernel of a bf16 plan
    # rounds its operands — so the family comparison runs without that routing;
    # the oracle check above covers it)
    cmp_new = run(False, small=False)
    switch('NO_FEWPOS_SMALL', None)
    tol = 2e-5 if precision == 'f32' else 2e-2       # (bf16 plans: other layers round)
    for i, (a, b) in enumerate(zip(cmp_new, old)):
        scale = max(float(np.abs(b).max()), 1e-6)
        assert float(np.abs(a - b).max()) / scale < tol, (name, i)
