"""GPU parity tests (``-m gpu``) added in round 2: what ``bench.py`` measures is
what is checked here, at its size.

* the full-size C2 generator in the bf16 throughput mode (batch 8: the
  persistent trunk kernel is selected — asserted through ``s3_plan_op_info``)
  (i) PER OP against the oracle doing the same roundings, every op fed the
  device's own input (``helpers.teacher_forced_check``): at most one bf16
  spacing on at most 1 % of an op's elements, everything else identical —
  this isolates kernel errors from the precision of the mode, which an
  end-to-end bound cannot do for a deep bf16 stack (see that helper) — and
  (ii) end to end against the exact fp32 oracle — the stated accuracy of the
  bf16 mode, 3e-2;
* the same generator in the BF16X3 mode (hi*hi + hi*lo + lo*hi on the bf16
  MFMA): L-inf < 1e-3 against the exact fp32 oracle, north_star's tolerance;
* the production discriminators as whole networks, forward and backward;
* gradients against the oracle under the device's own LeakyReLU masks: the
  backward pass is then linear in its inputs and the bounds are those of the
  arithmetic (1e-3 fp32, 1e-2 bf16-emulated), not of mask flips;
* one full C2 ``_train_batch`` against the oracle: moved to
  ``tests/test_parity_r03.py`` (device masks, fp32 batch 1 + bf16 batch 8);
* the sharded gradient (two shards accumulated on one GPU) against the
  oracle's per-shard SUM;
* the configs the round-1 verdict found unexercised: C3 (a 20x20x48 chunk
  through ``run_batched`` with the C2 generator), C4 (the reference's
  ``filters: 1`` toy and the 3x/4x body), C5 (CondMom on the 3x/4x body).

Every tolerance is stated where it is asserted.
"""
import json
import os

import numpy as np
import pytest

from tests.helpers import (emulate_plan, rel_linf, rel_max, rel_rms, switch,
                           teacher_forced_check)

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _load(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


def _oracle(spec, x, exo=None, seed=3, bias_scale=0.1):
    from oracle.network import Network
    net = Network(spec)
    net.init_weights(x, exo, seed=seed, bias_scale=bias_scale)
    return net


def _hip(spec, weights, precision):
    from sup3r_amd.engine import Network
    net = Network(spec, precision=precision)
    net.set_weights(weights)
    return net


def _kernels(ph, field='fwd'):
    from sup3r_amd import spec as S
    return [ph.op_info(i)[field] for i, op in enumerate(ph.plan.ops)
            if op['kind'] == S.OP_CONV]


# ---------------------------------------------------------------- C2 forward
@pytest.fixture(scope='module')
def c2():
    """one C2 sample, oracle weights, exact fp32 oracle output (~40 s)"""
    rng = np.random.default_rng(42)
    spec = _load('gen_5x_12x_2f.json')
    x = rng.standard_normal((8, 16, 16, 24, 4)).astype(np.float32)
    # (fully convolutional: the lazy build runs on a tiny input)
    ref = _oracle(spec, x[:1, :6, :6, :6], seed=0, bias_scale=0.0)
    y_ref = ref.forward(x[:1])
    return spec, x, ref, y_ref


def _assert_per_op(stats, what, frac16=1e-2, frac32=1e-3):
    """every op agrees with the oracle on the device's own input: bf16-stored
    outputs differ on < 1 % of their elements and there by one bf16 spacing
    (+ fp32 accumulation noise, 2e-5 of the tensor's scale); fp32-stored ones
    agree to that noise"""
    worst = max(stats, key=lambda d: d['frac'])
    print(f'{what}: {len(stats)} ops, worst mismatch fraction '
          f'{worst["frac"]:.2e} (op {worst["op"]}), largest difference beyond '
          f'one bf16 spacing {max(d["excess"] for d in stats):.2e} of the '
          'tensor scale')
    for d in stats:
        assert d['excess'] <= (d['noise'] if d['bf16'] else 5 * d['noise']), d
        assert d['frac'] < (frac16 if d['bf16'] else frac32), d


def test_c2_bf16_full_size_persistent_kernel_vs_oracle(c2):
    spec, x, ref, y_ref = c2
    net = _hip(spec, ref.weights, 'bf16')
    ph = net.plan(x.shape, training=False)
    fwd = _kernels(ph)
    # the benchmarked kernel is the one under test: 33 body convs, the head
    # conv at T = 96 and the 64 -> 200 conv all run on the persistent kernel
    assert fwd.count('mfma_persist') >= 34, fwd
    assert fwd[0] == 'gconv_fewch' and fwd[-1] == 'tail_mfma', fwd
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    assert y.shape == (8, 80, 80, 288, 2) and np.isfinite(y).all()
    # (ii) the accuracy of the bf16 mode itself, end to end
    err_mode = rel_linf(y[0], y_ref[0])
    print(f'C2 bf16 batch 8, end to end vs the fp32 oracle: {err_mode:.2e}')
    assert err_mode < 3e-2, err_mode
    # samples are independent: every copy of sample 0 in the batch is
    # bit-identical to it, whatever tile / workgroup computed it
    xx = np.repeat(x[:1], 8, axis=0)
    yy = ph.forward(net.dev.to_device(xx)).cpu().numpy()
    for k in range(8):
        np.testing.assert_array_equal(yy[k], y[0])
    # (i) per op, on the plan that keeps its activations; it runs the trunk
    # on the same kernels as the inference plan
    pht = net.plan(x.shape, training=True)
    fwd_t = _kernels(pht)
    assert fwd_t.count('mfma_persist') >= 34, fwd_t
    yt = pht.forward(net.dev.to_device(x)).cpu().numpy()
    assert rel_linf(yt[0], y_ref[0]) < 3e-2
    emu = _oracle(spec, x[:1, :6, :6, :6], seed=0, bias_scale=0.0)
    n_ops, n_store, _ = emulate_plan(emu, pht)
    assert n_ops == 38 and n_store >= 30, (n_ops, n_store)
    stats = teacher_forced_check(emu, pht, x, sample=slice(0, 1))
    assert len(stats) == len(pht.plan.ops)
    _assert_per_op(stats, 'C2 bf16 batch 8, per op')


def test_c2_bf16x3_meets_the_fp32_tolerance(c2):
    """S3_PREC_BF16X3: north_star's L-inf < 1e-3 at full C2 size"""
    spec, x, ref, y_ref = c2
    net = _hip(spec, ref.weights, 'bf16x3')
    ph = net.plan((2,) + x.shape[1:], training=False)
    assert _kernels(ph).count('mfma_tile') >= 34
    y = ph.forward(net.dev.to_device(x[:2])).cpu().numpy()
    err = float(np.abs(y[0] - y_ref[0]).max())
    print(f'C2 bf16x3: L-inf {err:.2e} vs the fp32 oracle '
          f'(output scale {np.abs(y_ref).max():.2f})')
    assert err < 1e-3, err
    # and against the exact-fp32 MFMA mode of the device
    net32 = _hip(spec, ref.weights, 'f32')
    y32 = net32(x[:1]).cpu().numpy()
    assert float(np.abs(y[0] - y32[0]).max()) < 1e-3
    assert float(np.abs(y32[0] - y_ref[0]).max()) < 1e-3


def test_bf16x3_edge_shapes_and_backward():
    """ragged tiles, both tile shapes, d2s / residual epilogues, and the data
    gradient (the same kernel over the padded frame) in BF16X3"""
    from sup3r_amd.configs.author_configs import pcc
    rng = np.random.default_rng(5)
    spec = pcc(3, 64) + [{'class': 'SkipConnection', 'name': 'a'}] + \
        pcc(3, 64) + pcc(3, 64, act=False) + \
        [{'class': 'SkipConnection', 'name': 'a'}] + \
        pcc(3, 200, act=False) + \
        [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
         {'alpha': 0.2, 'class': 'LeakyReLU'}] + pcc(3, 2, act=False)
    for shape in ((2, 5, 7, 19, 4), (12, 12, 17, 40, 4)):
        x = rng.standard_normal(shape).astype(np.float32)
        ref = _oracle(spec, x)
        y_ref = ref.forward(x)
        net = _hip(spec, ref.weights, 'bf16x3')
        ph = net.plan(shape, training=True)
        if shape[0] == 12:
            # round 3: the 64 -> C_out weight gradients run split-bf16 too
            # (conv3_wgrad_x3_kernel), not on the exact-fp32 MFMA
            wg = [ph.op_info(i)['wgrad'] for i, op in enumerate(ph.plan.ops)
                  if op['kind'] == 1 and op.get('cin') == 64]
            assert wg.count('bf16_trunk') == 3, wg
        y = ph.forward(net.dev.to_device(x)).cpu().numpy()
        assert rel_linf(y, y_ref) < 2e-5, (shape, rel_linf(y, y_ref))
        dy = rng.standard_normal(y_ref.shape).astype(np.float32)
        emulate_plan(ref, ph, masks=True, rounding=False)
        dx_ref = ref.backward(dy)
        dx = ph.backward(net.dev.to_device(dy), need_dx=True).cpu().numpy()
        assert rel_max(dx.reshape(dx_ref.shape), dx_ref) < 1e-4
        for g, g_ref in zip(net.grads, ref.grads):
            assert rel_max(g, g_ref) < 1e-4


# ------------------------------------------------- production discriminators
def test_disc_st_production_forward_full_size():
    """disc_st.json at the C2 hi-res shape (37 M parameters, 15360 x 2048
    Dense): exact-fp32 mode vs the oracle, and the bf16 mode vs the oracle
    doing the same roundings"""
    rng = np.random.default_rng(8)
    spec = _load('disc_st.json')
    x = rng.standard_normal((1, 80, 80, 288, 2)).astype(np.float32)
    ref = _oracle(spec, x, seed=1)
    y_ref = ref.forward(x)
    assert y_ref.shape == (1, 1)
    n_par = sum(w.size for w in ref.weights)
    assert n_par == 37072513
    net = _hip(spec, ref.weights, 'f32')
    y = net(x).cpu().numpy()
    assert abs(float(y[0, 0] - y_ref[0, 0])) < 1e-4 * max(1, abs(y_ref[0, 0]))
    net16 = _hip(spec, ref.weights, 'bf16')
    ph = net16.plan(x.shape, training=True)
    y16 = ph.forward(net16.dev.to_device(x)).cpu().numpy()
    emulate_plan(ref, ph)                 # (ref now rounds like the device)
    y_emu = ref.forward(x)
    print('disc_st full size: f32', float(y[0, 0]), 'oracle',
          float(y_ref[0, 0]), 'bf16', float(y16[0, 0]), 'emulated',
          float(y_emu[0, 0]))
    scale = max(1.0, abs(float(y_ref[0, 0])))
    assert abs(float(y16[0, 0] - y_emu[0, 0])) < 2e-3 * scale
    assert abs(float(y16[0, 0] - y_ref[0, 0])) < 3e-2 * scale


def _fwd_bwd_vs_oracle(spec, shape, precision, seed, tol_y, tol_g,
                       exo_name=None, exo_shape=None, replicate=False):
    """forward + backward of one network on the device vs the oracle that
    does the device's roundings and uses the device's masks.  bf16: the
    forward is checked per op on the device's own inputs (teacher forcing,
    see ``helpers.teacher_forced_check``) plus ``tol_y`` end to end against
    the exact oracle; the backward pass then runs over the device's
    activations, so its bound is that of the arithmetic.

    ``replicate``: the device batch is ``shape[0]`` copies of ONE sample (and
    of one output gradient), the oracle runs that sample once — the kernels
    see the production batch size, the numpy side a 1 / N of the work; weight
    gradients are then N times the oracle's."""
    rng = np.random.default_rng(seed)
    n_rep = shape[0] if replicate else 1
    oshape = ((1,) + tuple(shape[1:])) if replicate else tuple(shape)
    x = rng.standard_normal(oshape).astype(np.float32)
    exo = None
    if exo_name:
        exo = {exo_name: rng.standard_normal(exo_shape).astype(np.float32)}
    ref = _oracle(spec, x, exo, seed=seed)
    net = _hip(spec, ref.weights, precision)
    dev = net.dev
    ph = net.plan(shape, training=True)
    exod = {k: dev.to_device(v) for k, v in (exo or {}).items()}
    xd = np.repeat(x, n_rep, axis=0) if replicate else x
    y = ph.forward(dev.to_device(xd), exod).cpu().numpy()
    y_ref = ref.forward(x, exo)
    err_y = rel_linf(y[:oshape[0]], y_ref)
    assert err_y < tol_y, (precision, err_y)
    if replicate:
        for k in range(1, n_rep):
            np.testing.assert_array_equal(y[k], y[0])
    if precision == 'bf16':
        emulate_plan(ref, ph, masks=False)
        stats = teacher_forced_check(ref, ph, x, exo,
                                     sample=slice(0, oshape[0]))
        _assert_per_op(stats, f'bf16 {shape} per op')
    emulate_plan(ref, ph, masks=True, rounding=False, sample=slice(
        0, oshape[0]))
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = ref.backward(dy)
    dyd = np.repeat(dy, n_rep, axis=0) if replicate else dy
    dx = ph.backward(dev.to_device(dyd), need_dx=True).cpu().numpy()
    dx = dx.reshape((-1,) + dx_ref.shape[1:])[:oshape[0]]
    errs = {'dx': rel_max(dx, dx_ref)}
    gmax = max(float(np.abs(g).max()) for g in ref.grads) * n_rep
    for i, (g, g_ref) in enumerate(zip(net.grads, ref.grads)):
        # tensors whose gradient is numerically nothing compare at the
        # round-off of the large ones
        g_ref = g_ref * n_rep
        errs[i] = float(np.abs(g - g_ref).max()
                        / max(np.abs(g_ref).max(), 1e-3 * gmax))
    worst = max(errs.values())
    print(f'{precision} {shape}: y {err_y:.2e} end to end, worst gradient '
          f'error {worst:.2e}')
    assert worst < tol_g, (precision, errs)
    return ph


@pytest.mark.parametrize('cfg,shape', [
    ('disc_st_same.json', (2, 12, 12, 16, 2)),   # the reference's test disc
    ('disc_s_same.json', (4, 20, 20, 2)),        # tests/data/config_disc_s_test
    ('disc_s.json', (2, 64, 64, 2)),
])
def test_production_discriminators_fwd_bwd(cfg, shape):
    """6.1 M / 2.2 M-parameter test discriminators of the reference and the
    2-D production one: fp32 1e-4 end to end / 1e-3 gradients; bf16 per op
    (teacher forced) + 3e-2 end to end / 2e-2 gradients on the device's
    activations and masks"""
    spec = _load(cfg)
    _fwd_bwd_vs_oracle(spec, shape, 'f32', 31, 1e-4, 1e-3)
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 31, 3e-2, 2e-2)
    # round 3: BF16X3 plans run these convs split-bf16 on the gather-MFMA
    # kernel (forward and data gradient) at the fp32 bounds
    ph = _fwd_bwd_vs_oracle(spec, shape, 'bf16x3', 31, 1e-4, 1e-3)
    fwd, dg = _kernels(ph, 'fwd'), _kernels(ph, 'dgrad')
    print('bf16x3 fwd', fwd, 'dgrad', dg)
    assert sum(k.startswith('gconv') for k in fwd) >= 2, fwd
    # (round 4: the hi-res layers' data gradients run on the split-bf16
    # LDS-halo kernels where their tile counts allow)
    assert sum(dg.count(k) for k in ('gconv', 'c2', 's2')) >= 2, dg


def test_disc_st_production_kernels_at_reduced_shape():
    """disc_st.json forward / backward at the smallest shape that still runs
    on the kernels of the C2 training step (selection asserted against the
    plan of the full-size batch-8 discriminator)"""
    spec = _load('disc_st.json')
    from sup3r_amd.engine import Network
    full = Network(spec, precision='bf16')
    full.build((8, 80, 80, 288, 2), seed=0)
    ph_full = full.plan((8, 80, 80, 288, 2), training=True)
    want = {f: set(_kernels(ph_full, f)) for f in ('fwd', 'wgrad', 'dgrad')}
    print('production kernels:', want)
    del ph_full
    full.clear_plans()
    chosen = None
    # (8 valid convs, strides 1 2 1 2 ...: 62 is the smallest extent)
    for shape in ((2, 64, 64, 112, 2), (4, 64, 64, 112, 2),
                  (4, 64, 64, 160, 2), (8, 64, 64, 160, 2)):
        probe = Network(spec, precision='bf16')
        probe.build(shape, seed=0)
        ph = probe.plan(shape, training=True)
        got = {f: set(_kernels(ph, f)) for f in want}
        del ph
        probe.clear_plans()
        if all(want[f] <= got[f] for f in want):
            chosen = shape
            break
        print('shape', shape, 'misses',
              {f: want[f] - got[f] for f in want if want[f] - got[f]})
    assert chosen is not None, 'no reduced shape selects every kernel'
    _fwd_bwd_vs_oracle(spec, chosen, 'bf16', 17, 3e-2, 2e-2, replicate=True)
    _fwd_bwd_vs_oracle(spec, chosen, 'f32', 17, 1e-4, 1e-3, replicate=True)
    ph = _fwd_bwd_vs_oracle(spec, chosen, 'bf16x3', 17, 1e-4, 1e-3,
                            replicate=True)
    assert sum(k.startswith('gconv') for k in _kernels(ph, 'fwd')) >= 5


# --------------------------------------------------- generators, masks fixed
@pytest.mark.parametrize('cfg,shape', [
    ('gen_2x_2f.json', (3, 9, 8, 2)),             # C1: 36 Conv2DTranspose
    ('gen_3x_4x_2f.json', (1, 5, 6, 4, 2)),       # C4 / C5 body
    ('gen_3x_4x_2f.json', (8, 16, 16, 24, 2)),    # ... on the bf16 kernels
])
def test_production_generators_gradients_under_device_masks(cfg, shape):
    """round 1 asserted 5e-2 (f32) / 3e-1 (bf16) relative rms on these and
    blamed LeakyReLU mask flips; with the device's masks given to the oracle
    and — in bf16 — the device's activations (teacher forcing) the bounds are
    1e-3 (f32) and 2e-2 (bf16)"""
    spec = _load(cfg)
    big = shape[0] >= 8
    if not big:
        _fwd_bwd_vs_oracle(spec, shape, 'f32', 21, 1e-4, 1e-3)
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 21, 3e-2, 2e-2, replicate=big)


def test_c4_toy_generator_filters_1():
    """sup3rcc/gen_wind_3x_4x_2f.json of the reference: every hidden conv has
    ONE filter (C_in = C_out = 1 geometries) + Sup3rConcat topography"""
    spec = _load('gen_wind_3x_4x_2f_toy.json')
    n_par = 0
    ph = _fwd_bwd_vs_oracle(spec, (4, 4, 4, 4, 2), 'f32', 3, 1e-5, 1e-3,
                            exo_name='topography',
                            exo_shape=(4, 12, 12, 16, 1))
    n_par = sum(int(np.prod(p['shape'])) for p in ph.plan.params)
    assert n_par == 1447
    _fwd_bwd_vs_oracle(spec, (4, 4, 4, 4, 2), 'bf16', 3, 3e-2, 2e-2,
                       exo_name='topography', exo_shape=(4, 12, 12, 16, 1))


# ----------------------------------------------------------- training steps
# (one full C2 ``_train_batch`` vs the oracle: tests/test_parity_r03.py, under
# the device's masks, fp32 batch 1 and bf16 batch 8)
def test_sharded_gradient_is_the_per_shard_sum():
    """abstract.py:785-805 on one GPU: shard 0 then shard 1 with
    ``accumulate_wgrad`` — the buffer holds the SUM of the two shards'
    gradients (not the gradient of the whole batch's mean loss)"""
    from oracle.gan import GanOracle
    from oracle.network import Network as ONet
    from sup3r_amd import Sup3rGan
    rng = np.random.default_rng(3)
    gspec, dspec = _load('test_gen_st_2x_4x_2f.json'), \
        _load('test_disc_st_same.json')
    lr = rng.standard_normal((4, 4, 4, 4, 2)).astype(np.float32)
    hr = rng.standard_normal((4, 8, 8, 16, 2)).astype(np.float32)
    og, od = ONet(gspec), ONet(dspec)
    og.init_weights(lr, seed=1, bias_scale=0.1)
    od.init_weights(hr, seed=2, bias_scale=0.1)
    m = Sup3rGan(os.path.join(CFG, 'test_gen_st_2x_4x_2f.json'),
                 os.path.join(CFG, 'test_disc_st_same.json'),
                 loss='MeanAbsoluteError', precision='f32')
    m.init_weights(lr.shape, hr.shape)
    m.generator.set_weights(og.weights)
    m.discriminator.set_weights(od.weights)
    orc = GanOracle(og, od, loss='MeanAbsoluteError')
    for which, kw in (('gen', dict(train_gen=True, train_disc=False)),
                      ('disc', dict(train_gen=False, train_disc=True))):
        total, losses = None, []
        for r in range(2):
            sl = slice(2 * r, 2 * r + 2)
            loss, _, g = orc.loss_and_grads(lr[sl], hr[sl], 1e-2, **kw)
            g = [np.array(x, np.float64) for x in g]
            total = g if total is None else [a + b for a, b in zip(total, g)]
            losses.append(float(loss))
        m.virtual_gpus = 2
        w_before = [w.copy() for w in m.weights]
        det = m.run_gradient_descent(lr, hr, weight_gen_advers=1e-2,
                                     multi_gpu=True, **kw)
        net = m.generator if which == 'gen' else m.discriminator
        gmax = max(float(np.abs(g).max()) for g in total)
        for g, g_ref in zip(net.grads, total):
            # (tensors whose gradient is numerically nothing compare at the
            # round-off of the large ones)
            assert np.abs(g - g_ref).max() < 2e-3 * max(
                float(np.abs(g_ref).max()), 1e-3 * gmax), which
        key = 'loss_gen' if which == 'gen' else 'loss_disc'
        assert abs(float(det[key]) - np.mean(losses)) < 1e-5
        # put the weights back so the next case starts from the oracle's
        m.generator.set_weights(w_before[:len(m.generator_weights)])
        m.discriminator.set_weights(w_before[len(m.generator_weights):])


def test_condmom_on_the_3x_4x_body():
    """C5: Sup3rCondMom over gen_3x_4x_2f (the production body, 3.9 M
    parameters): masked-MSE value and gradients vs the oracle"""
    from oracle.network import Network as ONet
    from sup3r_amd import Sup3rCondMom
    rng = np.random.default_rng(12)
    spec = _load('gen_3x_4x_2f.json')
    lr = rng.standard_normal((2, 4, 4, 4, 2)).astype(np.float32)
    out = rng.standard_normal((2, 12, 12, 16, 2)).astype(np.float32)
    mask = (rng.uniform(size=out.shape) > 0.3).astype(np.float32)
    og = ONet(spec)
    og.init_weights(lr, seed=4, bias_scale=0.1)
    m = Sup3rCondMom(os.path.join(CFG, 'gen_3x_4x_2f.json'), precision='f32')
    m.init_weights(lr.shape, out.shape)
    m.generator.set_weights(og.weights)
    y = og.forward(lr)
    d = y * mask - out * mask
    loss_ref = float((d * d).mean())
    _, det = m.get_single_grad(lr, out, mask=mask)
    assert abs(float(det['loss_gen']) - loss_ref) < 1e-5 * max(1, loss_ref)
    ph = m.generator.plan(lr.shape, training=True)
    emulate_plan(og, ph, masks=True, rounding=False)
    og.backward((2.0 * d * mask / d.size).astype(np.float32))
    worst = max(rel_max(a, b) if np.abs(b).max() > 0 else 0.0
                for a, b in zip(m.generator.grads, og.grads))
    print(f'CondMom on gen_3x_4x_2f: worst gradient error {worst:.2e}')
    assert worst < 2e-3, worst


# ------------------------------------------------------------- C3 executor
def test_c3_chunk_through_run_batched_with_the_c2_generator():
    """a 20x20x48 lo-res chunk (+ halo) through ``ForwardPass.run_batched``
    with the C2 generator in bf16 (the bench configuration): equals the
    chunk-by-chunk ``run`` bit for bit and the oracle within the bf16 bound"""
    from sup3r_amd import ChunkSlicer, ForwardPass, Sup3rGan
    feats = ['u_100m', 'v_100m', 'temperature_100m', 'pressure_0m']
    outs = ['u_100m', 'v_100m']
    Sup3rGan.seed(11)
    means = {f: np.float32(0.1 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.0 + 0.25 * i) for i, f in enumerate(feats)}
    m = Sup3rGan(os.path.join(CFG, 'gen_5x_12x_2f.json'),
                 os.path.join(CFG, 'test_disc_st_same.json'), means=means,
                 stdevs=stds, precision='bf16')
    m.set_model_params(lr_features=feats, hr_out_features=outs, s_enhance=5,
                       t_enhance=12)
    rng = np.random.default_rng(0)
    domain = rng.standard_normal((24, 22, 52, 4)).astype(np.float32)
    slicer = ChunkSlicer((24, 22), 52, 5, 12, (20, 20, 48), spatial_pad=1,
                         temporal_pad=2)
    assert slicer.n_chunks == 8
    m.init_weights((1, 22, 22, 52, 4), (1, 110, 110, 624, 2))
    fwp = ForwardPass(m, slicer)
    got = np.full(slicer.hr_shape + (2,), np.nan, np.float32)
    n = fwp.run_batched(domain, out=got, batch=4)
    assert n == 8 and np.isfinite(got).all()
    # chunk 0 (the full 20x20x48 one) again through the reference-shaped path
    c0 = fwp.run_domain_chunk(domain, 0)
    hs = slicer.chunks[0]['hr_slice']
    assert c0.shape == (100, 100, 576, 2)
    np.testing.assert_array_equal(got[hs], c0)
    # ... and against the fp32 oracle on that chunk's padded input
    from oracle.gan import norm_input, un_norm_output
    from oracle.network import Network as ONet
    og = ONet(_load('gen_5x_12x_2f.json'))
    xin = fwp.chunk_input(domain, 0)[None]
    xin = norm_input(xin, [means[f] for f in feats],
                     [stds[f] for f in feats]).astype(np.float32)
    og.forward(xin[:, :6, :6, :6])
    og.set_weights(m.generator_weights)
    y = og.forward(xin)
    y = un_norm_output(y, [means[f] for f in outs], [stds[f] for f in outs])
    y = y[0][tuple(slicer.chunks[0]['hr_crop'])]
    err = rel_linf(c0, y)
    print(f'C3 chunk through run_batched (bf16) vs oracle: {err:.2e}')
    assert err < 3e-2, err


# ----------------------------------------------- strided ConvNDTranspose (X1)
@pytest.mark.parametrize('spec,shape', [
    ([{'class': 'Conv3DTranspose', 'filters': 16, 'kernel_size': 3,
       'strides': 2},
      {'class': 'Cropping3D', 'cropping': 1},
      {'alpha': 0.2, 'class': 'LeakyReLU'},
      {'class': 'Conv3D', 'filters': 2, 'kernel_size': 3}], (2, 5, 6, 7, 3)),
    ([{'class': 'FlexiblePadding', 'mode': 'REFLECT',
       'paddings': [[0, 0], [1, 1], [1, 1], [0, 0]]},
      {'class': 'Conv2DTranspose', 'filters': 8, 'kernel_size': 3,
       'strides': [2, 3], 'activation': 'relu'},
      {'class': 'Cropping2D', 'cropping': 2},
      {'class': 'Conv2D', 'filters': 2, 'kernel_size': 3}], (3, 6, 5, 2)),
])
def test_strided_transpose_conv_vs_oracle(spec, shape):
    """Conv3DTranspose / Conv2DTranspose with strides > 1 (zero insertion +
    flipped-kernel conv on the device), forward and backward, fp32"""
    _fwd_bwd_vs_oracle(spec, shape, 'f32', 9, 1e-5, 1e-3)


def test_shared_disc_pass_over_the_true_field_changes_nothing(monkeypatch):
    """``_train_batch`` evaluates D(hi_res_true) once for the generator step
    and the discriminator step (same tensor, same discriminator weights in
    between): weights after the batch are bit-identical to recomputing it, and
    a changed batch or changed weights are never served from the cache"""
    from sup3r_amd import Sup3rGan
    rng = np.random.default_rng(2)
    lr = rng.standard_normal((4, 4, 4, 4, 2)).astype(np.float32)
    hr = rng.standard_normal((4, 8, 8, 16, 2)).astype(np.float32)

    def run(reuse):
        Sup3rGan.seed(3)
        m = Sup3rGan(os.path.join(CFG, 'test_gen_st_2x_4x_2f.json'),
                     os.path.join(CFG, 'test_disc_st_same.json'),
                     loss='MeanAbsoluteError', learning_rate=1e-3)
        m.init_weights(lr.shape, hr.shape)
        m._compute.share_dtrue_allowed = reuse

        class B:
            low_res, high_res = lr, hr
        out = [m._train_batch(B, True, False, False, True, False, False, 1e-2)
               for _ in range(3)]
        # a different batch through the same model
        B.high_res = hr[::-1].copy()
        out.append(m._train_batch(B, True, False, False, True, False, False,
                                  1e-2))
        return out, m.weights
    d1, w1 = run(True)
    d0, w0 = run(False)
    for a, b in zip(d1, d0):
        assert a == b
    for a, b in zip(w1, w0):
        np.testing.assert_array_equal(a, b)


def test_sliding_window_tail_conv_is_bit_identical(monkeypatch):
    """conv_tail_slide_kernel (a workgroup walks a 16 x 64 column along s0 with
    a 4-plane ring) issues the same banded MFMAs per position as the tile
    kernel: identical output bits on ragged columns / segments, and the
    oracle's values within the bf16-mode bound"""
    from sup3r_amd.configs.author_configs import pcc
    rng = np.random.default_rng(7)
    spec = pcc(3, 64) + pcc(3, 200, act=False) + \
        [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
         {'alpha': 0.2, 'class': 'LeakyReLU'}] + pcc(3, 2, act=False)
    for shape in ((2, 5, 7, 40, 4), (3, 9, 4, 70, 4), (9, 6, 4, 20, 4)):
        x = rng.standard_normal(shape).astype(np.float32)
        ref = _oracle(spec, x)
        y_ref = ref.forward(x)
        net = _hip(spec, ref.weights, 'bf16')
        ph = net.plan(shape, training=False)
        assert _kernels(ph)[-1] == 'tail_mfma'
        xd = net.dev.to_device(x)
        y_sweep = ph.forward(xd).cpu().numpy()
        switch('NO_TAIL_SWEEP', 1)
        y_slide = ph.forward(xd).cpu().numpy()
        switch('NO_TAIL_SWEEP', None)
        switch('NO_TAIL_SLIDE', 1)
        y_tile = ph.forward(xd).cpu().numpy()
        switch('NO_TAIL_SLIDE', None)
        np.testing.assert_array_equal(y_slide, y_tile)
        np.testing.assert_array_equal(y_sweep, y_tile)
        assert rel_linf(y_sweep, y_ref) < 3e-2
        # conv_tail_sweep_kernel with its plane shape forced (S1, S2, rows per
        # unit): one / two / three column sets per wave, planes of 3 .. 49 DMA
        # pieces, several units per workgroup (the plane stream crosses units),
        # ragged planes and segments
        for s1, s2, seg in ((4, 16, 4), (8, 64, 6), (16, 128, 10), (40, 72, 8),
                            (8, 288, 48), (5, 24, 4), (48, 64, 128)):
            switch('TAIL_SWEEP_SHAPE', s1 * 1000000 + s2 * 1000 + seg)
            y_f = ph.forward(xd).cpu().numpy()
            switch('TAIL_SWEEP_SHAPE', None)
            np.testing.assert_array_equal(y_f, y_tile, err_msg=str((s1, s2, seg)))


# ------------------------------------------------ whole-network 2-D kernel (C1)
def _t2(filters, act='relu'):
    return [{'class': 'FlexiblePadding', 'mode': 'REFLECT',
             'paddings': [[0, 0], [3, 3], [3, 3], [0, 0]]},
            {'class': 'Conv2DTranspose', 'filters': filters, 'kernel_size': 3,
             'strides': 1, 'activation': act},
            {'class': 'Cropping2D', 'cropping': 4}]


FUSED_CHAINS = {
    'one conv to the output': _t2(2, None),
    'two convs': _t2(64) + _t2(2, None),
    'residual': _t2(64) + [{'class': 'SkipConnection', 'name': 'a'}] + _t2(64)
    + _t2(64, None) + [{'class': 'SkipConnection', 'name': 'a'}] + _t2(2, None),
    'depth to space': _t2(64) + _t2(256, None) + [
        {'class': 'SpatialExpansion', 'spatial_mult': 2},
        {'class': 'Activation', 'activation': 'relu'}] + _t2(2, None),
    'zero padding': [
        {'class': 'Conv2D', 'filters': 32, 'kernel_size': 3, 'padding': 'same'},
        {'alpha': 0.2, 'class': 'LeakyReLU'},
        {'class': 'Conv2D', 'filters': 3, 'kernel_size': 3, 'padding': 'same'}],
}


@pytest.mark.parametrize('name', sorted(FUSED_CHAINS))
@pytest.mark.parametrize('shape', [(3, 10, 10, 2), (2, 5, 7, 2)])
def test_fused_2d_kernel_short_chains(name, shape, monkeypatch):
    """fused2d_kernel on chains short enough for the bf16 roundings not to
    cascade: against the oracle doing the same roundings (bf16 operands, bf16
    intermediate tensors) 2e-3 of the output scale — conv, multi-fragment
    C_out, residual, depth-to-space, reflect and zero borders, ragged
    fragments — and against the op-by-op plan of the same precision"""
    spec = FUSED_CHAINS[name]
    rng = np.random.default_rng(4)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = _oracle(spec, x)
    y_exact = ref.forward(x)
    net = _hip(spec, ref.weights, 'bf16')
    ph = net.plan(shape, training=False)
    assert set(_kernels(ph)) == {'fused2d'}, _kernels(ph)
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    assert y.shape == y_exact.shape
    emulate_plan(ref, ph)
    y_emu = ref.forward(x)
    err = rel_linf(y, y_emu)
    print(f'fused2d {name} {shape}: vs emulating oracle {err:.2e}, vs exact '
          f'{rel_linf(y, y_exact):.2e}')
    assert err < 2e-3, err
    switch('NO_FUSED2D', 1)
    y_ops = ph.forward(net.dev.to_device(x)).cpu().numpy()
    switch('NO_FUSED2D', None)
    assert rel_linf(y, y_ops) < 2e-2


def test_fused_2d_kernel_c1_generator():
    """BASELINE config C1: gen_2x_2f (36 Conv2DTranspose layers, 1 368 706
    parameters) at the reference's test shapes through ONE launch; the bf16
    mode's end-to-end bound against the exact oracle, samples independent"""
    spec = _load('gen_2x_2f.json')
    rng = np.random.default_rng(42)
    for shape in ((15, 5, 5, 2), (3, 10, 10, 2)):
        x = rng.standard_normal(shape).astype(np.float32)
        ref = _oracle(spec, x, seed=0)
        y_ref = ref.forward(x)
        net = _hip(spec, ref.weights, 'bf16')
        ph = net.plan(shape, training=False)
        assert set(_kernels(ph)) == {'fused2d'}
        y = ph.forward(net.dev.to_device(x)).cpu().numpy()
        assert y.shape == (shape[0], 2 * shape[1], 2 * shape[2], 2)
        err = rel_linf(y, y_ref)
        print(f'C1 fused forward {shape}: {err:.2e} vs the exact oracle')
        assert err < 3e-2, err
        y1 = net(x[1:2]).cpu().numpy()
        np.testing.assert_array_equal(y1[0], y[1])


def test_gather_mfma_four_fragments_per_wave_is_bit_identical(monkeypatch):
    """gconv_mfma_kernel<*, 4> (the filter fragments of a tap shared by four
    position fragments per wave) vs <*, 2>: same MFMAs per position — forward
    and data gradient, strided and valid layers"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(32, 2) + conv(64, 1) + conv(64, 2) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (8, 40, 40, 72, 2)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=True)
    assert 'gconv' in _kernels(ph) and 'gconv' in _kernels(ph, 'dgrad')
    xd = net.dev.to_device(x)

    def run():
        y = ph.forward(xd)
        dy = net.dev.to_device(np.ones(tuple(y.shape), np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        return y.cpu().numpy(), dx, net.grads
    switch('GCONV_MF4', 1)   # the adjoint too
    y4, dx4, g4 = run()
    switch('GCONV_MF4', None)
    switch('GCONV_MF2', 1)
    y2, dx2, g2 = run()
    switch('GCONV_MF2', None)
    np.testing.assert_array_equal(y4, y2)
    np.testing.assert_array_equal(dx4, dx2)
    for a, b in zip(g4, g2):
        np.testing.assert_array_equal(a, b)


def test_gather_mfma_split_contraction_matches_the_unsplit_walk(monkeypatch):
    """The 128 / 256-channel discriminator layers sit on a few thousand
    positions: gconv_mfma_kernel then splits the (tap, k-chunk) walk over
    blockIdx.z and gconv_splitk_epilogue sums the slices in fixed order.  Same
    bf16 products, different fp32 summation order: forward, data gradient and
    weight gradients agree with the unsplit launch to fp32 round-off amplified
    by the bf16 stores in between (a last-bit change of a pre-activation can
    move its bf16 rounding) — and the stack passes the per-op oracle check"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(128, 1) + conv(128, 2) + conv(256, 1) + conv(256, 2) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 15, 15, 35, 64)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=True)
    assert 'gconv' in _kernels(ph) and 'gconv' in _kernels(ph, 'dgrad')
    xd = net.dev.to_device(x)

    def run(ph):
        y = ph.forward(xd)
        dy = net.dev.to_device(np.ones(tuple(y.shape), np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        return y.cpu().numpy(), dx, [g.copy() for g in net.grads]
    ys, dxs, gs = run(ph)
    # (a plan keeps the options it was created with: the unsplit walk is a
    # second plan over the same parameter store)
    yu, dxu, gu = run(net.plan(shape, training=True,
                               options={'NO_GCONV_SPLITK': 1}))
    assert not np.array_equal(dxs, dxu) or not np.array_equal(ys, yu), \
        'the split path was not taken at this shape'
    assert rel_linf(ys, yu) < 1e-3
    assert rel_linf(dxs, dxu) < 1e-3
    # (a flipped bf16 rounding of one activation is 4e-3 of that element; the
    # deep layers sum over 60 ... 800 positions only)
    for a, b in zip(gs, gu):
        assert rel_linf(a, b) < 3e-2
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 5, 3e-2, 2e-2)


def test_valid_conv_on_the_halo_tile_kernel_with_chunked_data_gradient():
    """64 -> 128 valid stride-1 conv (the discriminator's fifth layer): forward
    on conv3_mfma_kernel (two C_out tiles, no padding), data gradient as two
    64-channel slices of dPre through the same kernel accumulated in place on
    x's own grid (no frame, no fold)"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(64, 2) + conv(128, 1) + conv(128, 2) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 23, 21, 43, 32)
    from sup3r_amd.engine import Network
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=True)
    fwd, dg = _kernels(ph), _kernels(ph, 'dgrad')
    assert 'mfma_tile' in fwd, fwd
    assert 'mfma_chunked' in dg, dg
    del ph
    net.clear_plans()
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 9, 3e-2, 2e-2)
    _fwd_bwd_vs_oracle(spec, shape, 'f32', 9, 1e-4, 1e-3)


def test_stride2_lds_halo_conv_matches_the_gather_kernel(monkeypatch):
    """conv_halo_s2_kernel (32 -> 32 stride-2 valid conv: de-interleaved LDS
    halo, filter in LDS, persistent workgroups) vs gconv_mfma_kernel on the
    same bf16 cells: same products per output, fp32 sums in the same tap order
    — bit-identical, ragged tiles included; and vs the oracle per op"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(32, 2) + conv(64, 1) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 25, 31, 75, 2)
    switch('HALO_S2_MIN_TILES', 1)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=True)
    assert 'halo_s2' in _kernels(ph), _kernels(ph)
    xd = net.dev.to_device(x)
    y1 = ph.forward(xd).cpu().numpy()
    switch('NO_HALO_S2', 1)
    net2 = Network(spec, precision='bf16')
    net2.build(shape, seed=0)
    ph2 = net2.plan(shape, training=True)
    assert 'halo_s2' not in _kernels(ph2)
    y2 = ph2.forward(net2.dev.to_device(x)).cpu().numpy()
    switch('NO_HALO_S2', None)
    np.testing.assert_array_equal(y1, y2)
    del ph, ph2
    net.clear_plans(); net2.clear_plans()
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 13, 3e-2, 2e-2)


def test_stride2_64_channel_conv_on_the_split_lds_halo_kernel():
    """conv_halo_s2_k64_kernel (round 3): the discriminator's 64 -> 64
    stride-2 valid conv with the contraction split over the two 32-channel
    halves of the input (half filter + de-interleaved halo of a 2 x 2 x 16
    tile resident in LDS, raw fp32 sums of pass 0 added by pass 1).  Against
    the gather kernel on the same bf16 cells — same products, another fp32
    summation order: the bf16 outputs differ on a handful of elements by one
    spacing — and against the oracle per op; ragged tiles in every axis."""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(64, 1) + conv(64, 2) + conv(64, 1) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 19, 22, 47, 2)
    switch('HALO_S2_MIN_TILES', 1)
    rng = np.random.default_rng(4)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=True)
    k = _kernels(ph)
    assert k[2] == 'halo_s2', k
    y1 = ph.forward(net.dev.to_device(x)).cpu().numpy()
    t1 = ph.tensor(ph.plan.ops[2]['out'])
    switch('NO_HALO_S2_K64', 1)
    net2 = Network(spec, precision='bf16')
    net2.build(shape, seed=0)
    ph2 = net2.plan(shape, training=True)
    assert _kernels(ph2)[2] != 'halo_s2'
    y2 = ph2.forward(net2.dev.to_device(x)).cpu().numpy()
    t2 = ph2.tensor(ph2.plan.ops[2]['out'])
    switch('NO_HALO_S2_K64', None)
    # the conv's own output: equal up to one bf16 spacing on a few elements
    diff = np.abs(t1 - t2)
    frac = float((diff > 0).mean())
    print(f'64 -> 64 s2: {frac:.2e} of the outputs differ from the gather '
          f'kernel, worst {float((diff / np.maximum(np.abs(t2), 1e-6)).max()):.2e}')
    assert frac < 2e-2
    assert float((diff / np.maximum(np.abs(t2), 1e-3)).max()) < 1.0 / 64
    assert rel_linf(y1, y2) < 2e-2
    del ph, ph2
    net.clear_plans(); net2.clear_plans()
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 14, 3e-2, 2e-2)


def test_stride2_kernels_on_same_padded_convs():
    """TF 'same' padding of a stride-2 conv on an EVEN extent is one zero cell
    past the end and none in front — the LDS-halo stride-2 kernels
    zero-fill what lies past the tensor anyway, so the reference's test
    discriminators (`config_disc_st_test.json` layout, C4) run on them too:
    32 -> 32 forward and data gradient bit-identical to the gather kernels, the
    64 -> 64 forward within a bf16 spacing, everything against the oracle."""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'same'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(32, 2) + conv(64, 1) + conv(64, 2) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 24, 20, 44, 2)
    switch('HALO_S2_MIN_TILES', 1)
    switch('DGRAD_S2_MIN_TILES', 1)
    rng = np.random.default_rng(6)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        kf, kd = _kernels(ph), _kernels(ph, 'dgrad')
        y = ph.forward(net.dev.to_device(x))
        t1 = ph.tensor(ph.plan.ops[1]['out'])
        dy = net.dev.to_device(np.ones(tuple(y.shape), np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        y = y.cpu().numpy()
        del ph
        net.clear_plans()
        return kf, kd, y, t1, dx, g
    kf, kd, y1, t1, dx1, g1 = run()
    assert kf[1] == 'halo_s2' and kf[3] == 'halo_s2', kf
    assert kd[1] == 's2', kd
    switch('NO_HALO_S2', 1)
    switch('NO_DGRAD_S2', 1)
    kf0, kd0, y0, t0, dx0, g0 = run()
    switch('NO_HALO_S2', None)
    switch('NO_DGRAD_S2', None)
    assert 'halo_s2' not in kf0 and 's2' not in kd0
    np.testing.assert_array_equal(t1, t0)            # 32 -> 32 s2 forward
    assert rel_linf(y1, y0) < 2e-2                   # (64 -> 64: another fp32 order)
    assert rel_max(dx1, dx0) < 2e-2
    for a, b in zip(g1, g0):
        assert rel_max(a, b) < 2e-2
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 15, 3e-2, 2e-2)


def test_activation_sign_bytes_for_the_stride2_data_gradient_change_nothing():
    """Round 3: the first discriminator layer's forward (gconv_fewch_halo)
    writes sign bytes next to its bf16 output — bit q of byte [position][kq] =
    channel 8 kq + q > 0 — and the stride-2 data gradient behind it reads
    those 4 B per position as its fused LeakyReLU mask instead of the 64-B row
    of y at a 128-B pitch (FETCH 1.77 -> GB).  Same mask, so dx and every
    weight gradient are bit-identical with the bytes switched off; the
    production shape class (hi-res disc_st, bf16-only dPre) is the one run."""
    spec = _load('disc_st.json')
    shape = (4, 64, 64, 112, 2)
    rng = np.random.default_rng(8)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        assert 's2' in _kernels(ph, 'dgrad'), _kernels(ph, 'dgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(
            np.random.default_rng(9).standard_normal(tuple(y.shape)).astype(np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        del ph
        net.clear_plans()
        return dx, g
    dx1, g1 = run()
    switch('NO_SIGN_BYTES', 1)
    dx0, g0 = run()
    switch('NO_SIGN_BYTES', None)
    assert np.abs(dx1).max() > 0
    np.testing.assert_array_equal(dx1, dx0)
    for a, b in zip(g1, g0):
        np.testing.assert_array_equal(a, b)


def test_bf16_side_copy_of_dpre_changes_nothing(monkeypatch):
    """The halo-tile data gradient rounds dPre to bf16 while staging it; in
    bf16 training plans the fold / mask pass that produces dPre leaves a bf16
    copy (same round-to-nearest-even) and the data gradient stages that at
    half the bytes.  Weight gradients and dx are bit-identical with the copy
    switched off"""
    spec = _load('gen_3x_4x_2f.json')
    shape = (2, 8, 8, 12, 2)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        assert 'mfma_frame' in _kernels(ph, 'dgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(
            np.random.default_rng(6).standard_normal(tuple(y.shape)).astype(np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        del ph
        net.clear_plans()
        return dx, g
    dx1, g1 = run()
    switch('NO_DPRE16', 1)
    dx0, g0 = run()
    switch('NO_DPRE16', None)
    np.testing.assert_array_equal(dx1, dx0)
    for a, b in zip(g1, g0):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('n_samples', [8, 6, 16])
def test_trunk_data_gradient_on_the_persistent_kernel_is_bit_identical(monkeypatch, n_samples):
    """conv3_mfma_persist_kernel<4, DG> — the padded frames of the samples
    stacked into a grid so that 4 x 8 tiles fit them exactly, zero-flagged
    separator rows, fp32 store — vs the halo-tile kernel over the same bf16
    dPre: same MFMA sequence per output, so dx and every weight gradient are
    bit-identical; the stat counter proves the path ran"""
    # (the C2 generator: trunk frames 18 x 18 x 62 = 9 x 9 x 4 tiles for the
    # 2 x 4 grid of 8 samples, 3 % overhang; 6 samples = 2 x 3 frames = 36 x 54
    # positions, tiles straddle frames and overhang; 16 = 4 x 4)
    spec = _load('gen_5x_12x_2f.json')
    shape = (n_samples, 16, 16, 5, 4)
    switch('PERSIST_DGRAD_MIN_TILES', 1)
    # (round 4: the persistent kernel stores its frame as bf16 by default; the
    # kernel-vs-kernel identity is that of the fp32 frame, the bf16 frame is
    # compared with it further down)
    switch('NO_FRAME16', 1)
    rng = np.random.default_rng(8)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        n0 = net.dev.stat('persist_dgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(
            np.random.default_rng(9).standard_normal(tuple(y.shape)).astype(np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        used = net.dev.stat('persist_dgrad') - n0
        del ph
        net.clear_plans()
        return dx, g, used
    dx1, g1, used1 = run()
    assert used1 > 0, 'no data gradient ran on the persistent kernel'
    switch('NO_PERSIST_DGRAD', 1)
    dx0, g0, used0 = run()
    switch('NO_PERSIST_DGRAD', None)
    assert used0 == 0
    np.testing.assert_array_equal(dx1, dx0)
    for a, b in zip(g1, g0):
        np.testing.assert_array_equal(a, b)
    # the bf16 frame (default): one more bf16 rounding per trunk data gradient —
    # the input gradient after 35 of them and every weight gradient stay within
    # 2e-2 of the fp32-frame run (measured ~3e-3)
    switch('NO_FRAME16', None)
    dx2, g2, used2 = run()
    assert used2 > 0 and np.abs(dx2 - dx1).max() > 0
    assert rel_max(dx2, dx1) < 2e-2, rel_max(dx2, dx1)
    gmax = max(float(np.abs(b).max()) for b in g1)
    for a, b in zip(g2, g1):
        assert np.abs(a - b).max() <= 2e-2 * max(float(np.abs(b).max()),
                                                 1e-3 * gmax)


def test_first_disc_layer_bf16_only_dpre_changes_only_the_bias_sum_order(monkeypatch):
    """conv_dgrad_s2_kernel<2, O16> stores the gradient of the first
    discriminator activation — dPre of the 2 -> 32 conv, mask fused — as bf16
    only; conv_wgrad_c2_kernel<.., DY16> / conv_dgrad_c2_kernel read it that
    way and the bias gradient comes from the channel sums of the store.  The
    fp32 route rounds the same values to bf16 in those kernels: dx and the
    filter gradients are bit-identical, the bias gradient differs by the
    order of its fp32 sum"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(32, 2) + conv(64, 1) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 27, 33, 77, 2)
    switch('DGRAD_S2_MIN_TILES', 1)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        assert _kernels(ph, 'dgrad')[1] == 's2' and _kernels(ph, 'wgrad')[0] == 'c2'
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(np.ones(tuple(y.shape), np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        del ph
        net.clear_plans()
        return dx, g
    dx1, g1 = run()
    switch('NO_DPRE16', 1)
    dx0, g0 = run()
    switch('NO_DPRE16', None)
    np.testing.assert_array_equal(dx1, dx0)
    for k, (a, b) in enumerate(zip(g1, g0)):
        if a.ndim > 1:
            np.testing.assert_array_equal(a, b)
        else:
            assert rel_linf(a, b) < 1e-5, k


def test_first_layer_data_gradient_on_the_sliding_window_kernel():
    """conv_dgrad_c2_slide_kernel (round 3): the data gradient of the 2 -> 32
    conv as a column walk along s0 with the three t-taps packed into the
    MFMA's M dimension — same bf16 operands as conv_dgrad_c2_kernel, another
    fp32 summation order: dx agrees to round-off, nothing else changes, and
    the stack passes the oracle check.  Ragged in s0 (2 segments would need
    > 40 rows: one short segment), s1 (33 = 2 x 16 + 1) and t (77 = 5 x 14 +
    7)."""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(32, 1) + conv(32, 2) + conv(64, 1) + \
        [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (2, 47, 33, 77, 2)
    switch('DGRAD_S2_MIN_TILES', 1)
    switch('DGRAD_C2_SLIDE_MIN_UNITS', 1)
    rng = np.random.default_rng(13)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network
    net = Network(spec, precision='bf16')
    net.build(shape, seed=0)
    xd = net.dev.to_device(x)

    def run(options):
        ph = net.plan(shape, training=True, options=options)
        assert _kernels(ph, 'dgrad')[0] == 'c2'
        y = ph.forward(xd)
        dy = net.dev.to_device(np.ones(tuple(y.shape), np.float32))
        before = net.dev.stat('dgrad_c2_slide')
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        return dx, [a.copy() for a in net.grads], \
            net.dev.stat('dgrad_c2_slide') - before
    dx1, g1, used1 = run(None)
    dx0, g0, used0 = run({'NO_DGRAD_C2_SLIDE': 1})
    assert used1 == 1 and used0 == 0, (used1, used0)
    assert np.abs(dx0).max() > 0
    err = rel_max(dx1, dx0)
    print(f'sliding-window first-layer data gradient vs the tile kernel: {err:.2e}')
    assert err < 1e-5, err
    for a, b in zip(g1, g0):
        np.testing.assert_array_equal(a, b)
    net.clear_plans()
    _fwd_bwd_vs_oracle(spec, shape, 'bf16', 13, 3e-2, 2e-2)


def test_halo_tile_kernel_with_two_n_fragments_is_bit_identical(monkeypatch):
    """data gradient of a valid 32 -> 64 conv = a 64 -> 32 conv on the
    halo-tile kernel: with C_out <= 32 only two of the four N fragments of the
    cout tile are computed (conv3_mfma_kernel<.., NFV = 2>); the other two
    were zero rows"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(64, 1) + [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    shape = (4, 16, 32, 64, 32)
    rng = np.random.default_rng(12)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        assert _kernels(ph, 'dgrad')[0] == 'mfma_valid', _kernels(ph, 'dgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(np.ones(tuple(y.shape), np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        del ph
        net.clear_plans()
        return dx
    dx2 = run()
    switch('NO_TILE_NF2', 1)
    dx4 = run()
    switch('NO_TILE_NF2', None)
    assert np.abs(dx2).max() > 0
    np.testing.assert_array_equal(dx2, dx4)


@pytest.mark.parametrize('shape', [(8, 14, 14, 30, 32), (6, 10, 13, 21, 32)])
def test_valid_conv_data_gradient_on_the_persistent_kernel_is_bit_identical(monkeypatch, shape):
    """data gradient of a valid 32 -> 64 conv: the full correlation of dPre
    with the flipped filter lands on x's own grid, so the samples' frames stack
    like the trunk's padded ones and conv3_mfma_persist_kernel<2, DG> (two N
    fragments) writes dx directly — bit-identical to the halo-tile kernel"""
    def conv(f, s):
        return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3,
                 'strides': s, 'padding': 'valid'},
                {'alpha': 0.2, 'class': 'LeakyReLU'}]
    spec = conv(64, 1) + [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
    switch('PERSIST_DGRAD_MIN_TILES', 1)
    rng = np.random.default_rng(15)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        n0 = net.dev.stat('persist_dgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(
            np.random.default_rng(16).standard_normal(tuple(y.shape)).astype(np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        used = net.dev.stat('persist_dgrad') - n0
        del ph
        net.clear_plans()
        return dx, g, used
    dx1, g1, used1 = run()
    assert used1 > 0, 'the data gradient did not run on the persistent kernel'
    switch('NO_PERSIST_DGRAD', 1)
    dx0, g0, used0 = run()
    switch('NO_PERSIST_DGRAD', None)
    assert used0 == 0 and np.abs(dx1).max() > 0
    np.testing.assert_array_equal(dx1, dx0)
    for a, b in zip(g1, g0):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('n_samples,extent', [(8, (6, 6, 30)), (3, (6, 6, 30)), (4, (8, 8, 32))])
def test_wide_conv_data_gradient_slices_on_the_persistent_kernel(monkeypatch, n_samples, extent):
    """data gradient of the 64 -> 200 expansion conv: the four 64-channel
    slices of its dPre (the last one 8 channels wide) go through
    conv3_mfma_persist_kernel<4, DG> reading the bf16 copy the depth-to-space
    mask pass leaves behind (cell stride 200, dead 16-B chunks zeroed), the
    later slices adding to the frame in the store — against the halo-tile
    kernel over fp32 slices: same bf16 operands, same order, bit-identical"""
    from sup3r_amd.configs.author_configs import pcc
    spec = pcc(3, 64) + pcc(3, 64) + pcc(3, 200, act=False) + \
        [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
         {'alpha': 0.2, 'class': 'LeakyReLU'}] + pcc(3, 2, act=False)
    # (8 x 8 x 32: tiles fit exactly — the expansion conv's weight gradient then
    # also runs on the wave-specialised kernel, its seventh cout tile moved
    # back to channels 168 .. 199)
    shape = (n_samples,) + extent + (4,)
    switch('PERSIST_DGRAD_MIN_TILES', 1)
    rng = np.random.default_rng(18)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        assert 'mfma_chunked' in _kernels(ph, 'dgrad')
        n0 = net.dev.stat('persist_dgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(
            np.random.default_rng(19).standard_normal(tuple(y.shape)).astype(np.float32))
        dx = ph.backward(dy, need_dx=True).cpu().numpy()
        g = [a.copy() for a in net.grads]
        used = net.dev.stat('persist_dgrad') - n0
        del ph
        net.clear_plans()
        return dx, g, used
    dx1, g1, used1 = run()
    switch('NO_CHUNKED_DY16', 1)
    dx0, g0, used0 = run()
    switch('NO_CHUNKED_DY16', None)
    assert used1 >= used0 + 4, (used1, used0)
    assert np.abs(dx1).max() > 0
    np.testing.assert_array_equal(dx1, dx0)
    for a, b in zip(g1, g0):
        if a.shape == (200,):
            # the expansion conv's bias gradient: with the bf16 copy its
            # channel sums ride along the depth-to-space mask pass (per
            # workgroup, then bias_grad_stage2), without it they are taken
            # from the fp32 dPre — the same numbers in another fp32 order
            assert rel_max(a, b) < 1e-5
            continue
        np.testing.assert_array_equal(a, b)


def test_wave_specialised_trunk_weight_gradient_is_bit_identical(monkeypatch):
    """conv3_wgrad_bf16_ws_kernel — 4 producer waves fill the other of two
    half-tile LDS buffers by LDS-DMA while 12 consumer waves run the k-steps
    of the current half — walks the same positions in the same k-step order as
    conv3_wgrad_bf16_kernel: every weight gradient bit-identical (three tiles
    per workgroup at this shape, both halves, both buffers)"""
    spec = _load('gen_5x_12x_2f.json')
    shape = (8, 16, 16, 4, 4)
    rng = np.random.default_rng(13)
    x = rng.standard_normal(shape).astype(np.float32)
    from sup3r_amd.engine import Network

    def run():
        net = Network(spec, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=True)
        assert 'bf16_trunk' in _kernels(ph, 'wgrad')
        y = ph.forward(net.dev.to_device(x))
        dy = net.dev.to_device(
            np.random.default_rng(14).standard_normal(tuple(y.shape)).astype(np.float32))
        ph.backward(dy)
        g = [a.copy() for a in net.grads]
        del ph
        net.clear_plans()
        return g
    g1 = run()
    switch('NO_WGRAD_WS', 1)
    g0 = run()
    switch('NO_WGRAD_WS', None)
    assert any(np.abs(a).max() > 0 for a in g1)
    for a, b in zip(g1, g0):
        np.testing.assert_array_equal(a, b)
