"""GPU tests (``-m gpu``) of the Sup3rGan surface on the HIP engine: one
generator step and one discriminator step against the oracle restatement of
``calc_loss`` + ``tape.gradient`` (base.py:830-911, abstract.py:1190-1238), and
the behavioural assertions of the reference's tests/training/test_train_gan.py
re-expressed on a synthetic batch handler."""
import json
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _cfg(name):
    return os.path.join(CFG, name)


def _load(name):
    with open(_cfg(name)) as f:
        return json.load(f)


def _cmp_grads(grads, refs):
    """relative L-inf per tensor, floored at 1e-3 of the largest gradient of
    the network (tensors whose gradient is numerically zero compare at fp32
    round-off of the big ones)."""
    gmax = max(float(np.abs(r).max()) for r in refs)
    assert gmax > 0
    for g, gr in zip(grads, refs):
        assert g.shape == gr.shape
        tol = 2e-3 * float(np.abs(gr).max()) + 2e-5 * gmax
        assert np.abs(g - gr).max() < tol


@pytest.mark.parametrize('gen_cfg,disc_cfg,lr_shape,exo', [
    ('test_gen_st_2x_4x_2f.json', 'test_disc_st_same.json',
     (3, 4, 4, 4, 2), ()),
    ('test_gen_st_3x_4x_2f_topo.json', 'test_disc_st_same.json',
     (2, 4, 4, 4, 2), ("topography",)),
    ('test_gen_s_2x_2f.json', 'test_disc_s_same.json', (4, 6, 6, 2), ()),
])
def test_gan_steps_vs_oracle(gen_cfg, disc_cfg, lr_shape, exo):
    from oracle.gan import GanOracle
    from oracle.network import Network as ONet
    from sup3r_amd import Sup3rGan

    rng = np.random.default_rng(21)
    gspec, dspec = _load(gen_cfg), _load(disc_cfg)
    lr = rng.standard_normal(lr_shape).astype(np.float32)
    ogen, odisc = ONet(gspec), ONet(dspec)
    # shapes: run the oracle generator once to learn the hi-res shape
    probe_exo = None
    if exo:
        s = 3
        hs = (lr_shape[0], lr_shape[1] * s, lr_shape[2] * s, lr_shape[3] * 4, 1)
        probe_exo = {exo[0]: rng.standard_normal(hs).astype(np.float32)}
    ogen.init_weights(lr, probe_exo, seed=4, bias_scale=0.05)
    hr_gen0 = ogen.forward(lr, probe_exo)
    hr_true = rng.standard_normal(
        hr_gen0.shape[:-1] + (hr_gen0.shape[-1] + len(exo),)).astype(
            np.float32)
    if exo:
        hr_true[..., -1:] = probe_exo[exo[0]]
    odisc.init_weights(hr_true, seed=5, bias_scale=0.05)
    oracle = GanOracle(ogen, odisc, loss={'MeanAbsoluteError': {},
                                          'MeanSquaredError': {},
                                          'term_weights': [0.7, 0.3]})

    model = Sup3rGan(gspec, dspec, loss={'MeanAbsoluteError': {},
                                         'MeanSquaredError': {},
                                         'term_weights': [0.7, 0.3]},
                     learning_rate=1e-3)
    model.generator.set_weights(ogen.weights)
    model.discriminator.set_weights(odisc.weights)
    model.init_weights(lr.shape, hr_true.shape)
    assert model.hr_exo_features == list(exo)

    w_adv = 0.05
    # --- generator step (compute_disc=True like _train_batch with both nets)
    _, det_ref, g_ref = oracle.loss_and_grads(
        lr, hr_true, w_adv, train_gen=True, train_disc=False,
        compute_disc=True, exo_names=exo)
    g_ref = [g.copy() for g in g_ref]
    which, det = model.get_single_grad(
        lr, hr_true, weight_gen_advers=w_adv, train_gen=True,
        train_disc=False, compute_disc=True)
    assert which == 'gen'
    for k in ('loss_gen', 'loss_gen_content', 'loss_gen_advers', 'loss_disc',
              'mean_absolute_error', 'mean_squared_error'):
        assert abs(float(det[k]) - float(det_ref[k])) < 2e-4 * max(
            1.0, abs(float(det_ref[k]))), (k, det[k], det_ref[k])
    _cmp_grads(model.generator.grads, g_ref)

    # --- discriminator step
    _, det_ref, g_ref = oracle.loss_and_grads(
        lr, hr_true, w_adv, train_gen=False, train_disc=True, exo_names=exo)
    which, det = model.get_single_grad(
        lr, hr_true, weight_gen_advers=w_adv, train_gen=False,
        train_disc=True)
    assert which == 'disc'
    assert abs(float(det['loss_disc']) - float(det_ref['loss_disc'])) < 2e-4
    _cmp_grads(model.discriminator.grads, g_ref)

    # --- calc_loss public form == loss from the gradient pass
    out = model._tf_generate(lr, None if not exo else
                             {exo[0]: hr_true[..., -1:]})
    loss, det2 = model.calc_loss(hr_true, out, weight_gen_advers=w_adv,
                                 train_gen=True, compute_disc=True)
    assert abs(loss.numpy() - float(det2['loss_gen'])) < 1e-7


@pytest.mark.parametrize('gen_cfg,disc_cfg,s,t,sample_shape', [
    ('test_gen_st_2x_4x_2f.json', 'test_disc_st_same.json', 2, 4,
     (8, 8, 16)),
    ('test_gen_s_2x_2f.json', 'test_disc_s_same.json', 2, 1, (10, 10, 1)),
])
def test_train_behaviour(gen_cfg, disc_cfg, s, t, sample_shape):
    """tests/training/test_train_gan.py::test_train assertions (:160-245)."""
    from sup3r_amd import Sup3rGan
    from tests.helpers import SyntheticBatchHandler

    Sup3rGan.seed()
    lr = 5e-4
    n_epoch = 4
    model = Sup3rGan(_cfg(gen_cfg), _cfg(disc_cfg), learning_rate=lr,
                     loss={'MeanAbsoluteError': {}, 'MeanSquaredError': {}})
    bh = SyntheticBatchHandler(sample_shape, s, t, ['u', 'v'], batch_size=4,
                               n_batches=4)
    with tempfile.TemporaryDirectory() as td:
        model.train(bh, input_resolution={'spatial': '8km',
                                          'temporal': '40min'},
                    n_epoch=n_epoch, weight_gen_advers=0, train_gen=True,
                    train_disc=False, checkpoint_int=1,
                    out_dir=os.path.join(td, 'test_{epoch}'))
        assert bh.stopped
        assert 'config_generator' in model.meta
        assert 'config_discriminator' in model.meta
        assert len(model.history) == n_epoch
        assert all(model.history['gen_train_frac'] == 1)
        assert all(model.history['disc_train_frac'] == 0)
        tl = model.history['train_loss_gen'].values
        vl = model.history['val_loss_gen'].values
        assert np.sum(np.diff(tl)) < 0
        assert np.sum(np.diff(vl)) < 0
        assert 'test_0' in os.listdir(td) and 'test_1' in os.listdir(td)
        assert 'model_gen.pkl' in os.listdir(td + '/test_1')
        assert 'model_disc.pkl' in os.listdir(td + '/test_1')
        for col in ('train_mean_absolute_error', 'train_mean_squared_error',
                    'val_mean_absolute_error', 'val_mean_squared_error',
                    'OptmGen/learning_rate', 'OptmDisc/learning_rate',
                    'elapsed_time', 'total_batches', 'weight_gen_advers',
                    'disc_loss_bound_0', 'disc_loss_bound_1'):
            assert col in model.history, col
        assert any(c.startswith('OptmGen/Adam/v')
                   for c in model.history.columns)
        out_dir = os.path.join(td, 'st_gan')
        model.save(out_dir)
        loaded = Sup3rGan.load(out_dir)
        with open(os.path.join(out_dir, 'model_params.json')) as f:
            params = json.load(f)
        assert np.allclose(params['optimizer']['learning_rate'], lr)
        assert np.allclose(params['optimizer_disc']['learning_rate'], lr)
        assert 'config_generator' in loaded.meta
        assert model.meta['class'] == 'Sup3rGan'
        dummy = Sup3rGan(_cfg(gen_cfg), _cfg(disc_cfg), learning_rate=lr,
                         loss={'MeanAbsoluteError': {},
                               'MeanSquaredError': {}})
        dummy.meta.update(hr_out_features=['u', 'v'])
        for batch in bh:
            out_og = model._tf_generate(batch.low_res)
            out_loaded = loaded._tf_generate(batch.low_res)
            dummy.init_weights(batch.low_res.shape, batch.high_res.shape)
            out_dummy = dummy._tf_generate(batch.low_res)
            # save -> load -> bit-identical generate (test_train_gan.py:221)
            assert (out_og == out_loaded).all().item()
            assert not (out_og == out_dummy).all().item()
            loss_og = model.calc_loss(batch.high_res, out_og)[0]
            loss_dummy = dummy.calc_loss(batch.high_res, out_dummy)[0]
            assert loss_og.numpy() < loss_dummy.numpy()
        # a new shape goes through the generator (:231-245)
        if model.is_5d:
            test_data = np.ones((3, 10, 10, 4, 2), dtype=np.float32)
            y = model._tf_generate(test_data)
            assert y.shape[3] == test_data.shape[3] * t
        else:
            test_data = np.ones((3, 10, 10, 2), dtype=np.float32)
            y = model._tf_generate(test_data)
        assert y.shape[0] == 3 and y.shape[1] == 10 * s and y.shape[2] == 10 * s
        assert y.shape[-1] == test_data.shape[-1]
        # public generate == un-normalised _tf_generate
        g = model.generate(test_data)
        np.testing.assert_allclose(g, y.cpu().numpy(), rtol=1e-6)


def test_train_disc_and_errors():
    """test_train_disc (:45-110): disc trains whenever its loss is outside the
    bounds; bad resolution raises (:389-422)."""
    from sup3r_amd import Sup3rGan
    from tests.helpers import SyntheticBatchHandler
    Sup3rGan.seed()
    model = Sup3rGan(_cfg('test_gen_st_2x_4x_2f.json'),
                     _cfg('test_disc_st_same.json'), learning_rate=5e-5,
                     loss='MeanAbsoluteError')
    bh = SyntheticBatchHandler((8, 8, 16), 2, 4, ['u', 'v'], batch_size=3,
                               n_batches=2)
    with tempfile.TemporaryDirectory() as td:
        kw = dict(input_resolution={'spatial': '8km', 'temporal': '40min'},
                  n_epoch=3, weight_gen_advers=0.0, train_gen=True,
                  train_disc=True, disc_loss_bounds=[-np.inf, 0.0],
                  checkpoint_int=1, out_dir=os.path.join(td, 'test_{epoch}'))
        model.train(bh, **kw)
        assert all(model.history['disc_train_frac'] == 1)
        model.save(os.path.join(td, 'gan'))
        loaded = Sup3rGan.load(os.path.join(td, 'gan'))
        loaded.train(bh, **kw)
        assert len(loaded.history) == 6
        assert list(loaded.history.index) == list(range(6))
        assert all(loaded.history['disc_train_frac'] == 1)
        bad = Sup3rGan(_cfg('test_gen_st_2x_4x_2f.json'),
                       _cfg('test_disc_st_same.json'))
        with pytest.raises(RuntimeError):
            bad.train(bh, input_resolution={'spatial': '7km',
                                            'temporal': '40min'}, n_epoch=1,
                      out_dir=os.path.join(td, 'x_{epoch}'))


def test_condmom_masked_mse_and_train():
    """Sup3rCondMom (BASELINE config C5 topology at test size): masked MSE and
    its gradient vs the oracle, then the reference's training behaviour
    (tests/training/test_train_conditional.py: loss decreases, save/load)."""
    from oracle.network import Network as ONet
    from sup3r_amd import Sup3rCondMom
    from tests.helpers import SyntheticMomBatchHandler
    rng = np.random.default_rng(3)
    gspec = _load('test_gen_st_2x_4x_2f.json')
    lr = rng.standard_normal((3, 4, 4, 4, 2)).astype(np.float32)
    ogen = ONet(gspec)
    ogen.init_weights(lr, seed=2, bias_scale=0.05)
    y = ogen.forward(lr)
    true = rng.standard_normal(y.shape).astype(np.float32)
    mask = (rng.uniform(size=y.shape) > 0.3).astype(np.float32)
    d = (y - true) * mask
    loss_ref = float((d.astype(np.float64) ** 2).mean())
    ogen.backward((2.0 * d * mask / d.size).astype(np.float32))
    g_ref = [g.copy() for g in ogen.grads]
    model = Sup3rCondMom(gspec, learning_rate=1e-3)
    model.generator.set_weights(ogen.weights)
    model.init_weights(lr.shape, true.shape)
    which, det = model.get_single_grad(lr, true, mask=mask)
    assert which == 'gen'
    assert set(det) == {'mean_squared_error', 'loss_gen'}
    assert abs(float(det['loss_gen']) - loss_ref) < 1e-5 * max(1, loss_ref)
    _cmp_grads(model.generator.grads, g_ref)
    out = model._tf_generate(lr)
    loss, det2 = model.calc_loss(true, out, mask)
    assert abs(loss.numpy() - loss_ref) < 1e-5 * max(1, loss_ref)

    Sup3rCondMom.seed()
    model = Sup3rCondMom(_cfg('test_gen_st_2x_4x_2f.json'), learning_rate=5e-4)
    bh = SyntheticMomBatchHandler((8, 8, 16), 2, 4, ['u', 'v'], batch_size=4,
                                  n_batches=4, pad=1)
    with tempfile.TemporaryDirectory() as td:
        model.train(bh, input_resolution={'spatial': '8km',
                                          'temporal': '40min'},
                    n_epoch=4, checkpoint_int=2,
                    out_dir=os.path.join(td, 'test_{epoch}'))
        assert len(model.history) == 4
        assert np.sum(np.diff(model.history['train_loss_gen'].values)) < 0
        assert 'val_loss_gen' in model.history
        assert 'learning_rate_gen' in model.history
        assert 'model_gen.pkl' in os.listdir(os.path.join(td, 'test_2'))
        assert 'model_disc.pkl' not in os.listdir(os.path.join(td, 'test_2'))
        out_dir = os.path.join(td, 'cm')
        model.save(out_dir)
        loaded = Sup3rCondMom.load(out_dir)
        b = bh.batches[0]
        assert (model._tf_generate(b.low_res)
                == loaded._tf_generate(b.low_res)).all().item()
        with open(os.path.join(out_dir, 'model_params.json')) as f:
            assert json.load(f)['num_par'] == sum(
                w.size for w in model.generator_weights)


def test_c2_training_steps_are_deterministic():
    """Race screen (tools/determinism_soak.py in short): the same four
    ``_train_batch`` steps of the C2 GAN (production generator + discriminator,
    batch 8: every production training kernel) twice from the same seed.
    Every reduction runs in a fixed order, so the weights must come out
    bit-identical; an operand consumed before its wait or an LDS hand-over
    without its barrier shows up as a difference."""
    import torch
    from sup3r_amd import Sup3rGan
    from sup3r_amd.engine import Device

    def run():
        Sup3rGan.seed(7)
        m = Sup3rGan(_cfg('gen_5x_12x_2f.json'), _cfg('disc_st.json'),
                     loss='MeanAbsoluteError', precision='bf16')
        dev = Device.get()
        rng = np.random.default_rng(3)
        lr_s, hr_s = (8, 16, 16, 24, 4), (8, 80, 80, 288, 2)
        m.init_weights(lr_s, hr_s)
        losses = []
        for _ in range(4):
            class Batch:
                low_res = dev.to_device(rng.standard_normal(lr_s).astype(np.float32))
                high_res = dev.to_device(rng.standard_normal(hr_s).astype(np.float32))
            d = m._train_batch(Batch, True, False, False, True, False, False, 1e-3)
            losses.append((d['loss_gen'], d['loss_disc']))
        torch.cuda.synchronize()
        w = [np.array(a) for a in m.generator.weights] + \
            [np.array(a) for a in m.discriminator.weights]
        del m
        torch.cuda.empty_cache()
        return w, losses
    w0, l0 = run()
    w1, l1 = run()
    assert all(np.isfinite(a).all() for a in w0)
    assert l0 == l1
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)
