"""SolarCC (sup3r/models/solar_cc.py): the oracle restatement under the
reference's own custom-loss test procedure (tests/training/test_train_solar.py:
162-252) on CPU, and the HIP engine against the oracle on the GPU."""
import json
import os
import tempfile

import numpy as np
import pytest

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _load(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


def _day_slices(t_len, start=8, hours=8):
    return [slice(start + x, start + x + hours) for x in range(0, t_len, 24)]


def _oracle(loss, lr_shape=(2, 4, 4, 12, 2), seed=31):
    from oracle.gan import SolarCCOracle
    from oracle.network import Network as ONet
    rng = np.random.default_rng(seed)
    gspec, dspec = _load('test_gen_st_2x_4x_2f.json'), _load('test_disc_st_same.json')
    lr = rng.standard_normal(lr_shape).astype(np.float32)
    ogen, odisc = ONet(gspec), ONet(dspec)
    ogen.init_weights(lr, None, seed=4, bias_scale=0.05)
    hr0 = ogen.forward(lr)
    hr_true = rng.standard_normal(hr0.shape).astype(np.float32)
    odisc.init_weights(hr_true[:, :, :, :8], seed=5, bias_scale=0.05)
    return SolarCCOracle(ogen, odisc, loss=loss), gspec, dspec, lr, hr_true


def test_oracle_solar_custom_loss_reference_procedure():
    """test_solar_custom_loss: the content loss drops when the daylight hours
    of the synthetic field are replaced by the true ones; mismatched shapes
    raise RuntimeError; a time axis that is not whole days asserts."""
    rng = np.random.default_rng(0)
    oracle, *_ = _oracle('MeanAbsoluteError')
    shape = (1, 4, 4, 72, 2)
    gen = rng.uniform(0, 1, shape).astype(np.float32)
    true = rng.uniform(0, 1, shape).astype(np.float32)
    ts = (3, 30, 60)
    with pytest.raises(RuntimeError):
        oracle.loss_and_grads(None, rng.uniform(0, 1, (1, 5, 5, 24, 2)), 0.0,
                              hi_res_gen=rng.uniform(0, 1, (1, 10, 10, 24, 2)),
                              time_samples=(0,))
    with pytest.raises(AssertionError):
        oracle.loss_and_grads(None, rng.uniform(0, 1, (1, 5, 5, 20, 2)), 0.0,
                              hi_res_gen=rng.uniform(0, 1, (1, 5, 5, 20, 2)),
                              time_samples=(0,))
    loss1, det1, _ = oracle.loss_and_grads(None, true, 0.0, hi_res_gen=gen,
                                           time_samples=ts)
    gen2 = gen.copy()
    for sl in _day_slices(72):
        gen2[:, :, :, sl] = true[:, :, :, sl]
    loss2, det2, _ = oracle.loss_and_grads(None, true, 0.0, hi_res_gen=gen2,
                                           time_samples=ts)
    assert loss1 > loss2
    # the point-loss hours (11, 12) lie inside the daylight window: that part
    # vanishes, the 24-h mean part does not
    assert det2['c_sub_mean_absolute_error'] == 0
    assert det2['c_24h_mean_absolute_error'] > 0


def test_solar_cc_class_surface():
    from sup3r_amd import SolarCC
    assert (SolarCC.STARTING_HOUR, SolarCC.DAYLIGHT_HOURS,
            SolarCC.POINT_LOSS_HOURS) == (8, 8, 2)
    lo = np.zeros((1, 3, 3, 4, 1))
    hi = np.arange(30, dtype=np.float32).reshape(1, 1, 1, 30, 1) * np.ones((1, 3, 3, 1, 1))
    obj = SolarCC.__new__(SolarCC)
    obj._t_enhance = 8
    out = obj.temporal_pad(lo, hi)
    assert out.shape == (1, 3, 3, 32, 1)
    assert np.array_equal(out[0, 0, 0, :, 0],
                          np.pad(np.arange(30, dtype=np.float32), 1, mode='reflect'))


@pytest.mark.gpu
@pytest.mark.parametrize('loss', ['MeanAbsoluteError',
                                  {'MeanSquaredError': {}, 'MeanAbsoluteError': {},
                                   'term_weights': [0.6, 0.4]}])
def test_solar_cc_steps_vs_oracle(loss):
    """generator step, discriminator step and the public calc_loss of SolarCC
    on the HIP engine against the oracle, two days, fixed window draws (one
    overlapping the other); then the reference's custom-loss procedure, the
    error behaviour, generate's time padding and save / load."""
    from sup3r_amd import SolarCC
    oracle, gspec, dspec, lr, hr_true = _oracle(loss)
    assert hr_true.shape == (2, 8, 8, 48, 2)
    model = SolarCC(gspec, dspec, loss=loss, learning_rate=1e-3)
    model.generator.set_weights(oracle.gen.weights)
    model.discriminator.set_weights(oracle.disc.weights)
    model.init_weights(lr.shape, hr_true.shape)
    ts = (17, 21)
    model._compute.time_samples = ts
    w_adv = 0.05
    names = ['mean_absolute_error'] if isinstance(loss, str) else [
        'mean_squared_error', 'mean_absolute_error']

    _, det_ref, g_ref = oracle.loss_and_grads(
        lr, hr_true, w_adv, train_gen=True, compute_disc=True, time_samples=ts)
    g_ref = [g.copy() for g in g_ref]
    which, det = model.get_single_grad(
        lr, hr_true, weight_gen_advers=w_adv, train_gen=True,
        train_disc=False, compute_disc=True)
    assert which == 'gen'
    keys = ['loss_gen', 'loss_gen_content', 'loss_gen_advers', 'loss_disc']
    keys += [p + n for p in ('c_sub_', 'c_24h_') for n in names]
    for k in keys:
        assert abs(float(det[k]) - float(det_ref[k])) < 2e-4 * max(
            1.0, abs(float(det_ref[k]))), (k, det[k], det_ref[k])
    gmax = max(float(np.abs(r).max()) for r in g_ref)
    for g, gr in zip(model.generator.grads, g_ref):
        assert np.abs(g - gr).max() < 2e-3 * float(np.abs(gr).max()) + 2e-5 * gmax

    _, det_ref, g_ref = oracle.loss_and_grads(
        lr, hr_true, w_adv, train_gen=False, train_disc=True, time_samples=ts)
    which, det = model.get_single_grad(
        lr, hr_true, weight_gen_advers=w_adv, train_gen=False, train_disc=True)
    assert which == 'disc'
    assert abs(float(det['loss_disc']) - float(det_ref['loss_disc'])) < 2e-4
    gmax = max(float(np.abs(r).max()) for r in g_ref)
    for g, gr in zip(model.discriminator.grads, g_ref):
        assert np.abs(g - gr).max() < 2e-3 * float(np.abs(gr).max()) + 2e-5 * gmax

    # the reference's custom-loss procedure through the public calc_loss
    rng = np.random.default_rng(1)
    shape = (1, 8, 8, 72, 2)
    gen = rng.uniform(0, 1, shape).astype(np.float32)
    true = rng.uniform(0, 1, shape).astype(np.float32)
    model._compute.time_samples = None       # random windows, as in training
    with pytest.raises(RuntimeError):
        model.calc_loss(rng.uniform(0, 1, (1, 5, 5, 24, 2)).astype(np.float32),
                        rng.uniform(0, 1, (1, 10, 10, 24, 2)).astype(np.float32))
    with pytest.raises(AssertionError):
        model.calc_loss(rng.uniform(0, 1, (1, 8, 8, 20, 2)).astype(np.float32),
                        rng.uniform(0, 1, (1, 8, 8, 20, 2)).astype(np.float32))
    loss1, _ = model.calc_loss(true, gen, weight_gen_advers=0.0)
    gen2 = gen.copy()
    for sl in _day_slices(72):
        gen2[:, :, :, sl] = true[:, :, :, sl]
    loss2, _ = model.calc_loss(true, gen2, weight_gen_advers=0.0)
    assert float(loss1) > float(loss2)
    model._compute.time_samples = (3, 30, 60)
    ref, _, _ = oracle.loss_and_grads(None, true, 0.0, hi_res_gen=gen,
                                      time_samples=(3, 30, 60))
    assert abs(float(loss1) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))

    # generate pads the time axis to low_res * t_enhance; save / load keep the class
    model._t_enhance = 5
    y = model.generate(lr[:1], norm_in=False, un_norm_out=False)
    assert y.shape == (1, 8, 8, 60, 2)
    model._t_enhance = 4
    with tempfile.TemporaryDirectory() as td:
        model.save(os.path.join(td, 'cc_gan'))
        loaded = SolarCC.load(os.path.join(td, 'cc_gan'))
        assert model.meta['class'] == 'SolarCC' and loaded.meta['class'] == 'SolarCC'
        y2 = loaded.generate(lr[:1], norm_in=False, un_norm_out=False)
        assert np.array_equal(model.generate(lr[:1], norm_in=False, un_norm_out=False), y2)


@pytest.mark.gpu
def test_solar_cc_structured_content_term_runs():
    """a feature-map loss (CoarseMseLoss: 5-D point-loss windows, 4-D daily
    means) composes with the windows: against the oracle's loss functions
    applied to the same slices"""
    from oracle import losses as OL
    from sup3r_amd import SolarCC
    oracle, gspec, dspec, lr, hr_true = _oracle('MeanAbsoluteError')
    spec = {'CoarseMseLoss': {}, 'MeanAbsoluteError': {}, 'term_weights': [0.5, 0.5]}
    model = SolarCC(gspec, dspec, loss=spec, learning_rate=1e-3)
    model.generator.set_weights(oracle.gen.weights)
    model.discriminator.set_weights(oracle.disc.weights)
    model.init_weights(lr.shape, hr_true.shape)
    model._compute.time_samples = (0, 40)
    gen = oracle.gen.forward(lr)
    loss, det = model.calc_loss(hr_true, gen, weight_gen_advers=0.0)
    g64, t64 = gen.astype(np.float64), hr_true.astype(np.float64)
    ref = 0.0
    for d in (0, 24):
        p = slice(11 + d, 13 + d)
        ref += OL.multi_term_loss(spec, g64[:, :, :, p], t64[:, :, :, p])[0] / 2
        ref += OL.multi_term_loss(spec, g64[:, :, :, d:d + 24].mean(axis=3),
                                  t64[:, :, :, 8 + d:16 + d].mean(axis=3))[0] / 2
    assert abs(float(det['loss_gen_content']) - ref) < 2e-5 * max(1.0, abs(ref))
    which, det2 = model.get_single_grad(lr, hr_true, weight_gen_advers=0.0,
                                        train_gen=True, train_disc=False)
    assert which == 'gen' and all(np.isfinite(g).all() for g in model.generator.grads)


@pytest.mark.gpu
@pytest.mark.parametrize('hr_steps', (24, 48))
def test_solar_cc_train(hr_steps):
    """tests/training/test_train_solar.py::test_solar_cc_model on a synthetic
    batch handler: one and two days per sample, generator-only training with
    and without the adversarial term, checkpoint, class name in the meta,
    save / load, loss_fun is the configured content loss, generate's shape."""
    from sup3r_amd import SolarCC, Sup3rGan
    from tests.helpers import SyntheticBatchHandler
    Sup3rGan.seed()
    model = SolarCC(os.path.join(CFG, 'test_gen_st_2x_4x_2f.json'),
                    os.path.join(CFG, 'test_disc_st_same.json'),
                    learning_rate=1e-4, loss='MeanAbsoluteError')
    bh = SyntheticBatchHandler((8, 8, hr_steps), 2, 4, ['u', 'v'], batch_size=2,
                               n_batches=2)
    with tempfile.TemporaryDirectory() as td:
        model.train(bh, input_resolution={'spatial': '4km', 'temporal': '1440min'},
                    n_epoch=2, weight_gen_advers=0.0, train_gen=True,
                    train_disc=False, checkpoint_int=None,
                    out_dir=os.path.join(td, 'test_{epoch}'))
        assert 'test_1' in os.listdir(td)
        assert model.meta['class'] == 'SolarCC'
        assert model.meta['hr_out_features'] == ['u', 'v']
        for col in ('train_loss_gen', 'val_loss_gen', 'train_c_sub_mean_absolute_error',
                    'train_c_24h_mean_absolute_error'):
            assert col in model.history, (col, list(model.history.columns))
        assert np.isfinite(model.history['train_loss_gen'].values).all()
        # both networks, adversarial term on
        model.train(bh, input_resolution={'spatial': '4km', 'temporal': '1440min'},
                    n_epoch=1, weight_gen_advers=0.01, train_gen=True,
                    train_disc=True, checkpoint_int=None,
                    out_dir=os.path.join(td, 'gan_{epoch}'))
        assert np.isfinite(model.history['train_loss_disc'].values[-1])
        out_dir = os.path.join(td, 'cc_gan')
        model.save(out_dir)
        loaded = SolarCC.load(out_dir)
        assert loaded.meta['class'] == 'SolarCC'
    x = np.random.default_rng(0).uniform(0, 1, (1, 6, 6, hr_steps // 4, 2)).astype(np.float32)
    y = model.generate(x)
    assert y.shape == (1, 12, 12, hr_steps, 2)
    assert np.array_equal(loaded.generate(x), y)
