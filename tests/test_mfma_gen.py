"""GPU parity tests (``-m gpu``) of the logical-axes halo-tile MFMA conv
(``sup3r_amd/csrc/kernels_conv_mfma_gen.hip``): the layers of the reference's
shipped generator specs that the 64 -> C_out 3x3x3 trunk kernel does not take —
2-D stacks (Conv2D / Conv2DTranspose, time axis folded into the batch:
``/root/reference/sup3r/pipeline/forward_pass.py:274-337``), 3-D convs over few
time steps, C_in in {1, 3, 6, 7, 14, 18, 32, 65}, C_out in {1, 2, 6, 14}, the
64 -> 72 / 1600 expansion convs with their depth-to-space stores — at shapes
where a SAMPLE has >= 256 positions (the kernel's selection threshold).

Bounds as everywhere (tests/test_parity_r02.py::_fwd_bwd_vs_oracle): bf16
plans per op teacher-forced (<= one bf16 spacing on <= 1 % of an op's
elements) + 3e-2 end to end, gradients 2e-2 on the device's masks; BF16X3
plans < 1e-3 end to end; and an inference plan is bit-identical sample by
sample whatever the batch size (the tile shape changes, the arithmetic per
position does not).
"""
import numpy as np
import pytest

from sup3r_amd import spec as S
from tests.test_ref_surface import BF16_GRAD_TOL, _exo_for, _fwd_bwd, load_surface

pytestmark = pytest.mark.gpu

# spec -> (low-res shape, exo, (layer description, min count of mfma_gen convs
# in the bf16 inference plan))
CASES = {
    # 2-D Conv2DTranspose archetype: 35 convs at 20 x 18 and 40 x 36
    'spatial/gen_2x_2f.json': ((3, 20, 18, 2), None, 34),
    # + the 64 -> 1600 expansion (depth-to-space 5) and the 64 -> 2 output conv
    'spatial/gen_10x_2f.json': ((2, 16, 17, 2), None, 36),
    # Conv2D + LeakyReLU, 3 -> 64 head, 64 -> 1 output
    'sup3rcc/gen_solar_5x_1x_1f.json': ((3, 17, 16, 3), None, 36),
    # ... Sup3rConcat mid-network: a 65 -> 64 conv (two K passes), 7 -> 64 head
    'sup3rcc/gen_wind_5x_1x_6f.json': ((2, 16, 17, 7), 'topography', 36),
    # T = 3: the 16-position run along s2; 64 -> 512 depth_to_time; 64 -> 1
    'sup3rcc/gen_solar_1x_8x_1f.json': ((2, 17, 16, 3, 3), None, 35),
    # 64 -> 72 with 18-channel depth-to-space cells, 18 -> 2 output conv
    'spatiotemporal/gen_2x_2x_2f.json': ((1, 8, 9, 8, 2), None, 2),
    # 64 -> 14 output conv at hi-res (one N fragment), 14 -> 64 head
    'spatiotemporal/gen_3x_4x_14f.json': ((1, 6, 7, 8, 14), None, 2),
    # 32 -> 2 behind depth_to_time x24
    'sup3rcc/gen_trh_1x_24x_2f.json': ((1, 8, 9, 4, 4), None, 1),
    # 64 -> 6 output conv, 6 -> 64 head at 24 x 7 x 6 x 3
    'sup3rcc/gen_wind_1x_24x_6f.json': ((1, 12, 8, 3, 6), None, 2),
}


def _selection(ph):
    return [ph.op_info(i)['fwd'] for i, op in enumerate(ph.plan.ops)
            if op['kind'] == S.OP_CONV]


def _n_gen(sel):
    """convs on the logical-axes tile kernel or (bf16 inference plans, 2-D
    64 -> 64 k trunks) on the weights-stationary Conv2D kernel"""
    return sel.count('mfma_gen') + sel.count('conv2d_ws') + sel.count('conv2d_head')


@pytest.mark.parametrize('rel', sorted(CASES))
def test_gen_kernel_bf16_forward_backward(rel):
    shape, exo_name, n_min = CASES[rel]
    ph = _fwd_bwd(load_surface(rel), shape, 'bf16', 51, 3e-2,
                  BF16_GRAD_TOL.get(rel, 2e-2), exo_name)
    sel = _selection(ph)
    print(rel, {k: sel.count(k) for k in sorted(set(sel))})
    assert _n_gen(sel) >= n_min, sel


@pytest.mark.parametrize('rel', sorted(CASES))
def test_gen_kernel_inference_bf16x3_and_batch_invariance(rel):
    from sup3r_amd.engine import Network
    from tests.helpers import rel_linf
    from tests.test_parity_r02 import _oracle
    shape, exo_name, n_min = CASES[rel]
    spec = load_surface(rel)
    rng = np.random.default_rng(53)
    n = max(3, shape[0])
    shape = (n,) + tuple(shape[1:])
    x = rng.standard_normal(shape).astype(np.float32)
    plan = S.build_plan(S.parse_layers(spec), shape)
    exo = _exo_for(plan, exo_name, rng, np.float32) if exo_name else None
    ref = _oracle(spec, x[:1], None if exo is None else
                  {k: v[:1] for k, v in exo.items()}, seed=53)
    y_ref = ref.forward(x[:1], None if exo is None else
                        {k: v[:1] for k, v in exo.items()})
    for prec, tol in (('bf16x3', 1e-3), ('bf16', 3e-2)):
        net = Network(spec, precision=prec)
        net.set_weights(ref.weights)
        dev = net.dev
        ph = net.plan(shape, training=False)
        sel = _selection(ph)
        assert _n_gen(sel) >= n_min, (prec, sel)
        y = ph.forward(dev.to_device(x), {k: dev.to_device(v) for k, v in
                                          (exo or {}).items()}).cpu().numpy()
        err = rel_linf(y[:1], y_ref)
        assert err < tol, (prec, err)
        # sample by sample == the batch (different tile shapes / grids)
        ph1 = net.plan((1,) + tuple(shape[1:]), training=False)
        for k in range(n):
            yk = ph1.forward(
                dev.to_device(x[k:k + 1]),
                {kk: dev.to_device(v[k:k + 1]) for kk, v in
                 (exo or {}).items()}).cpu().numpy()
            np.testing.assert_array_equal(yk[0], y[k])


def test_gen_kernel_can_be_switched_off():
    """option NO_MFMA_GEN: the same plan on the gather / direct kernels — the
    A/B switch of the census — agrees with the logical-axes kernel to bf16
    accumulation-order noise"""
    from sup3r_amd.engine import Network
    rel = 'sup3rcc/gen_solar_5x_1x_1f.json'
    spec = load_surface(rel)
    shape = (4, 17, 16, 3)
    x = np.random.default_rng(3).standard_normal(shape).astype(np.float32)
    net = Network(spec, precision='bf16')
    net.build(shape, seed=2)
    dev = net.dev
    a = net.plan(shape, training=False)
    b = net.plan(shape, training=False, options={'NO_MFMA_GEN': 1})
    assert _n_gen(_selection(a)) >= 36 and _n_gen(_selection(b)) == 0
    assert _selection(a).count('conv2d_ws') >= 33
    c = net.plan(shape, training=False, options={'NO_CONV2D_WS': 1})
    assert _selection(c).count('conv2d_ws') == 0
    assert _selection(c).count('mfma_gen') >= 36
    ya = a.forward(dev.to_device(x)).cpu().numpy()
    yb = b.forward(dev.to_device(x)).cpu().numpy()
    yc = c.forward(dev.to_device(x)).cpu().numpy()
    assert np.abs(ya - yb).max() < 3e-2 * max(1.0, np.abs(yb).max())
    assert np.abs(ya - yc).max() < 3e-2 * max(1.0, np.abs(yc).max())


def test_concat_of_one_exo_channel_runs_inside_the_ws_conv():
    """sup3rcc/gen_wind_5x_1x_6f: the Sup3rConcat of the 64-channel hi-res
    tensor and the topography in front of a 65 -> 64 Conv2D.  bf16 inference
    plans never write the 65-channel tensor: the conv runs as 64 -> 64 on the
    weights-stationary kernel, which adds the topography's nine taps per
    output from the fp32 field (option NO_WS_EXO: the concat + two-pass conv).
    Same result as the unsplit plan to bf16 accumulation-order noise, oracle
    parity per op in the tests above (they run on this plan), batch
    invariance, and the skip add behind a residual conv stays in bf16
    (add16, option NO_ADD16)."""
    from sup3r_amd.engine import Network
    rel = 'sup3rcc/gen_wind_5x_1x_6f.json'
    spec = load_surface(rel)
    shape = (3, 16, 17, 7)
    rng = np.random.default_rng(71)
    x = rng.standard_normal(shape).astype(np.float32)
    plan = S.build_plan(S.parse_layers(spec), shape)
    exo = _exo_for(plan, 'topography', rng, np.float32)
    net = Network(spec, precision='bf16')
    net.build(shape, seed=8)
    dev = net.dev
    xd = dev.to_device(x)
    ed = {k: dev.to_device(v) for k, v in exo.items()}
    a = net.plan(shape, training=False)
    kinds = [op['kind'] for op in a.plan.ops]
    ic = kinds.index(S.OP_CONCAT)
    assert a.op_info(ic)['in_rep'] == 1                    # fused away
    conv65 = next(i for i, op in enumerate(a.plan.ops)
                  if op['kind'] == S.OP_CONV and op['cin'] == 65)
    assert a.op_info(conv65)['fwd'] == 'conv2d_ws'
    # the d2s conv in front of it stores bf16 cells now (-> ws as well)
    d2s = next(i for i, op in enumerate(a.plan.ops)
               if op['kind'] == S.OP_CONV and op.get('d2s', 1) == 5)
    assert a.tensor_is_bf16(a.plan.ops[d2s]['out'])
    assert a.op_info(d2s)['fwd'] == 'conv2d_ws'
    iadd = [i for i, k in enumerate(kinds) if k == S.OP_ADD]
    assert iadd and all(a.tensor_is_bf16(a.plan.ops[i]['out']) for i in iadd)
    ya = a.forward(xd, ed).cpu().numpy()
    # ... and the LAST of them (the big skip right behind a conv that already
    # adds the block's skip) is absorbed into that conv's store: the same
    # bits as the plan that runs it as a pass of its own
    assert [a.op_info(i)['in_rep'] for i in iadd] == [0] * (len(iadd) - 1) + [1]
    c = net.plan(shape, training=False, options={'NO_WS_RES2': 1})
    assert [c.op_info(i)['in_rep'] for i in iadd] == [0] * len(iadd)
    np.testing.assert_array_equal(c.forward(xd, ed).cpu().numpy(), ya)
    b = net.plan(shape, training=False, options={'NO_WS_EXO': 1,
                                                  'NO_ADD16': 1})
    assert b.op_info(ic)['in_rep'] == 0
    assert b.op_info(conv65)['fwd'] == 'mfma_gen'
    assert not any(b.tensor_is_bf16(b.plan.ops[i]['out']) for i in iadd)
    yb = b.forward(xd, ed).cpu().numpy()
    assert np.isfinite(ya).all()
    assert np.abs(ya - yb).max() < 3e-2 * max(1.0, np.abs(yb).max())
    # the topography matters, and sample by sample == the batch
    e2 = {k: dev.to_device(v[::-1].copy()) for k, v in exo.items()}
    assert np.abs(a.forward(xd, e2).cpu().numpy() - ya).max() > 1e-3
    a1 = net.plan((1,) + shape[1:], training=False)
    for k in range(shape[0]):
        yk = a1.forward(dev.to_device(x[k:k + 1]),
                        {n_: dev.to_device(v[k:k + 1]) for n_, v in
                         exo.items()}).cpu().numpy()
        np.testing.assert_array_equal(yk[0], ya[k])


def test_2d_training_plan_keeps_bf16_cells():
    """a 2-D training plan: the 64 -> 64 k layers keep bf16 cells (forward on
    the weights-stationary kernel, weight gradient staged from bf16), EVERY
    conv's weight gradient — the 2 -> 64 head and the 64 -> 2 output conv
    included — on the transpose-read MFMA kernel; option NO_TRAIN2D_BF16 is
    the fp32-cell plan of the start of round 5 (same outputs to bf16 noise,
    gradients in the same direction).  (Parity with the oracle: the bf16 forward / backward tests
    above run on this plan.)"""
    from sup3r_amd.engine import Network
    rel = 'spatial/gen_2x_2f.json'
    spec = load_surface(rel)
    shape = (3, 20, 18, 2)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape).astype(np.float32)
    net = Network(spec, precision='bf16')
    net.build(shape, seed=4)
    dev = net.dev
    out = {}
    for name, opts in (('cells16', {}), ('cells32', {'NO_TRAIN2D_BF16': 1})):
        ph = net.plan(shape, training=True, options=opts)
        convs = [(i, op) for i, op in enumerate(ph.plan.ops)
                 if op['kind'] == S.OP_CONV]
        info = [ph.op_info(i) for i, _ in convs]
        assert all(f['wgrad'] == 'bf16_2d' for f in info), [f['wgrad'] for f in info]
        trunk = [f for f, (_, op) in zip(info, convs)
                 if op['cin'] == 64 and op['cout'] % 64 == 0]
        in16 = [ph.tensor_is_bf16(op['in0']) for _, op in convs if op['cin'] == 64]
        if name == 'cells16':
            assert all(f['fwd'] == 'conv2d_ws' for f in trunk) and len(trunk) >= 34
            assert all(in16)
        else:
            assert all(f['fwd'] == 'mfma_gen' for f in trunk)
            assert not any(in16)
        y = ph.forward(dev.to_device(x))
        dy = np.random.default_rng(6).standard_normal(
            tuple(ph.out_shape)).astype(np.float32)
        ph.backward(dev.to_device(dy), need_dx=False)
        out[name] = (y.cpu().numpy(), [np.array(g) for g in net.grads])
        del ph
        net.clear_plans()
    ya, ga = out['cells16']
    yb, gb = out['cells32']
    assert np.abs(ya - yb).max() < 3e-2 * max(1.0, np.abs(yb).max())
    # (two roundings of 35 ReLU layers' activations: the gradients of a random
    # net agree in direction, not element by element — the element-wise bound
    # is the oracle's, under the device's masks, in the tests above)
    for a, b in zip(ga, gb):
        cos = float((a * b).sum() / max(np.linalg.norm(a) * np.linalg.norm(b),
                                        1e-30))
        assert np.isfinite(a).all() and cos > 0.97, (a.shape, cos)


# ---------------------------------------------------------------------------
# exact tests: one-hot filters.  Every output channel of every conv copies ONE
# (tap, input channel) of its input (+ an integer bias); inputs are small
# integers.  The whole network is then a composition of shifts, reflections,
# channel permutations, skip adds and depth-to-space moves of integers below
# 2^8 — exact in bf16 and in every accumulation order — so the device must
# reproduce the oracle BIT FOR BIT, and a wrong tap order, row permutation,
# swizzle, K pass or store permutation of a kernel cannot hide behind a
# tolerance.
# ---------------------------------------------------------------------------
def _one_hot_weights(ref, rng, nd):
    for layer in ref.weight_layers:
        k = layer.kernel
        w = np.zeros(k.shape, np.float32)
        taps = int(np.prod(k.shape[:-2]))
        cin, cout = k.shape[-2], k.shape[-1]
        flat = w.reshape(taps, cin, cout)
        for co in range(cout):
            flat[rng.integers(taps), rng.integers(cin), co] = 1.0
        layer.kernel = w
        if layer.use_bias:
            layer.bias = rng.integers(-2, 3, size=layer.bias.shape).astype(
                np.float32)


ONE_HOT = [
    # 2-D: 3 -> 64 head (logical-axes kernel), 64 -> 64 x 3 with a skip
    # (weights-stationary kernel in bf16 inference plans), 64 -> 256 d2s 2,
    # 64 -> 2 output conv; ragged in rows and columns (21 x 19, 5 images)
    ('2d', 2, (5, 21, 19, 3)),
    # 2-D, one image, 16 x 16 exactly one tile
    ('2d', 2, (1, 16, 16, 3)),
    # 2-D at >= 64 x 64 hi-res cells per image: the 64 -> 2 output conv on the
    # taps-as-columns kernel (kernels_conv2d_out.hip), ragged 14-column strips
    # (72 = 5 x 14 + 2) and row segments
    ('2d', 2, (3, 40, 36, 3)),
    # 3-D over T = 3 (run along s2), 64 -> 128 depth_to_time 2
    ('3d_small_t', 3, (2, 17, 18, 3, 5)),
    # 3-D, long T, C_in = 70 (two K passes), C_out = 72 d2s 2 (18-channel
    # cells), 18 -> 6 output conv
    ('3d_odd', 3, (1, 6, 7, 16, 5)),
]


def _one_hot_spec(kind, nd):
    from sup3r_amd.configs.author_configs import pcc
    skip = {'class': 'SkipConnection', 'name': 'a'}
    if kind == '2d':
        return pcc(2, 64, act=False) + [skip] + pcc(2, 64, act=False) + \
            pcc(2, 64, act=False) + [dict(skip)] + pcc(2, 256, act=False) + \
            [{'class': 'SpatialExpansion', 'spatial_mult': 2}] + \
            pcc(2, 2, act=False)
    if kind == '3d_small_t':
        return pcc(3, 64, act=False, pad=[3, 3, 2], crop=[2, 2, 1]) + \
            [skip] + pcc(3, 64, act=False, pad=[3, 3, 2], crop=[2, 2, 1]) + \
            [dict(skip)] + \
            pcc(3, 128, act=False, pad=[3, 3, 2], crop=[2, 2, 1]) + \
            [{'class': 'SpatioTemporalExpansion', 'temporal_mult': 2,
              'temporal_method': 'depth_to_time', 't_roll': 1}] + \
            pcc(3, 1, act=False)
    return pcc(3, 70, act=False) + pcc(3, 64, act=False) + \
        pcc(3, 72, act=False) + \
        [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 2}] + \
        pcc(3, 6, act=False)


@pytest.mark.parametrize('kind,nd,shape', ONE_HOT)
@pytest.mark.parametrize('prec', ['bf16', 'bf16x3'])
def test_one_hot_filters_are_exact(kind, nd, shape, prec):
    from sup3r_amd.engine import Network
    from tests.test_parity_r02 import _oracle
    spec = _one_hot_spec(kind, nd)
    rng = np.random.default_rng(61)
    x = rng.integers(-8, 9, size=shape).astype(np.float32)
    ref = _oracle(spec, x[:1], None, seed=1)
    _one_hot_weights(ref, rng, nd)
    y_ref = ref.forward(x)
    assert np.abs(y_ref).max() < 256 and np.abs(y_ref).max() > 4
    net = Network(spec, precision=prec)
    net.set_weights(ref.weights)
    ph = net.plan(shape, training=False)
    sel = _selection(ph)
    print(kind, prec, sel)
    assert _n_gen(sel) >= len(sel) - 1, sel
    if prec == 'bf16' and kind == '2d':
        assert sel.count('conv2d_ws') == 4, sel     # 64 -> 64 x 2, 64 -> 256 d2s, 64 -> 2 output
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    np.testing.assert_array_equal(y, y_ref)


@pytest.mark.parametrize('rel,shape', [('spatial/gen_2x_2f.json', (3, 33, 37, 2)),
                                        ('spatial/gen_2x_1f.json', (2, 75, 75, 1))])
def test_head_conv_kernel_vs_the_matrix_path_and_the_oracle(rel, shape):
    """conv2d_head_kernel (C_in 1 / 2 -> 64, fp32 field in, bf16 cells out, the
    filter taps of a lane's eight channels in registers): same bf16 operand
    rounding and fp32 accumulation as the logical-axes MFMA kernel it replaces
    (option NO_CONV2D_HEAD) — the two differ by the order of 9 C_in terms, i.e. by
    bf16 output roundings that flip — and both sit inside the bf16 bound of the
    oracle."""
    from sup3r_amd.engine import Network
    from tests.helpers import rel_linf
    from tests.test_parity_r02 import _oracle
    spec = load_surface(rel)
    rng = np.random.default_rng(7)
    x = (3.0 * rng.standard_normal(shape)).astype(np.float32)
    ref = _oracle(spec, x[:1], None, seed=11)
    net = Network(spec, precision='bf16')
    net.set_weights(ref.weights)
    dev = net.dev
    a = net.plan(shape, training=False)
    b = net.plan(shape, training=False, options={'NO_CONV2D_HEAD': 1})
    sa, sb = _selection(a), _selection(b)
    assert sa[0] == 'conv2d_head' and sb[0] == 'mfma_gen', (sa[0], sb[0])
    assert sa[1:] == sb[1:]
    ya = a.forward(dev.to_device(x)).cpu().numpy()
    yb = b.forward(dev.to_device(x)).cpu().numpy()
    y_ref = ref.forward(x[:1])
    assert rel_linf(ya[:1], y_ref) < 3e-2 and rel_linf(yb[:1], y_ref) < 3e-2
    # (a flipped bf16 rounding of the first layer reaches the output amplified by
    # 35 more: the two valid bf16 evaluations agree like each does with the oracle)
    assert rel_linf(ya, yb) < 3e-2
    # the head layers alone (+ one conv that reads their bf16 cells): at most one
    # bf16 spacing apart, on a small fraction of the elements
    hl = spec['hidden_layers']
    head = {'hidden_layers': hl[:3] + [
        {'class': 'FlexiblePadding', 'paddings': [[0, 0], [3, 3], [3, 3], [0, 0]], 'mode': 'REFLECT'},
        {'class': 'Conv2DTranspose', 'filters': 64, 'kernel_size': 3, 'strides': 1},
        {'class': 'Cropping2D', 'cropping': 4}]}
    hnet = Network(head, precision='bf16')
    hnet.build(shape, seed=5)
    ha = hnet.plan(shape, training=False)
    hb = hnet.plan(shape, training=False, options={'NO_CONV2D_HEAD': 1})
    assert _selection(ha)[0] == 'conv2d_head' and _selection(hb)[0] == 'mfma_gen'
    za = ha.forward(hnet.dev.to_device(x)).cpu().numpy()
    zb = hb.forward(hnet.dev.to_device(x)).cpu().numpy()
    d = np.abs(za - zb)
    assert d.max() <= 2.0 ** -6 * np.abs(zb).max() and (d > 0).mean() < 1e-2, (d.max(), (d > 0).mean())
    # one sample == the batch
    a1 = net.plan((1,) + tuple(shape[1:]), training=False)
    assert _selection(a1)[0] == 'conv2d_head'
    for k in range(shape[0]):
        yk = a1.forward(dev.to_device(x[k:k + 1])).cpu().numpy()
        np.testing.assert_array_equal(yk[0], ya[k])


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_pingpong_conv2d_is_bit_identical_to_the_lockstep_kernel(seed):
    """conv2d_ws_pp_kernel (two half-workgroups half a period apart, single-image
    tiles, hand-ordered loads) against conv2d_ws_kernel (option NO_WS_PP): the
    same MFMAs in the same order per position and the same epilogue arithmetic,
    so every output bit agrees — on ragged extents, odd batch sizes, one-tile
    runs, with and without skip operands, through a depth-to-space store and
    with 25 output-channel tiles."""
    from sup3r_amd.engine import Network
    rng = np.random.default_rng(100 + seed)
    rel = ['spatial/gen_2x_2f.json', 'spatial/gen_10x_2f.json', 'spatial/gen_2x_1f.json'][seed]
    spec = load_surface(rel)
    cin = 1 if rel.endswith('1f.json') else 2
    for _ in range(3):
        n = int(rng.integers(1, 6))
        h, w = int(rng.integers(16, 41)), int(rng.integers(16, 41))
        shape = (n, h, w, cin)
        x = rng.standard_normal(shape).astype(np.float32)
        net = Network(spec, precision='bf16')
        net.build(shape, seed=int(rng.integers(1 << 20)))
        a = net.plan(shape, training=False)
        b = net.plan(shape, training=False, options={'NO_WS_PP': 1})
        assert _selection(a) == _selection(b) and _selection(a).count('conv2d_ws') >= 34
        xd = net.dev.to_device(x)
        ya = a.forward(xd).cpu().numpy()
        yb = b.forward(xd).cpu().numpy()
        np.testing.assert_array_equal(ya, yb, err_msg=str(shape))
