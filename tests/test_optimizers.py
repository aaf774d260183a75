"""The optimizers ``Sup3rGan(optimizer=...)`` accepts by keras name
(sup3r/models/abstract.py:321-350, models/utilities.py:150-158): config
surface on the host and three fused device steps of each against the numpy
restatement of keras-2.15's ``update_step`` (float64)."""
import numpy as np
import pytest

CASES = [
    ('SGD', {'learning_rate': 0.05}),
    ('SGD', {'learning_rate': 0.05, 'momentum': 0.9}),
    ('SGD', {'learning_rate': 0.05, 'momentum': 0.8, 'nesterov': True}),
    ('RMSprop', {'learning_rate': 1e-2}),
    ('RMSprop', {'learning_rate': 1e-2, 'rho': 0.8, 'momentum': 0.5}),
    ('Adagrad', {'learning_rate': 0.1}),
    ('Adamax', {'learning_rate': 2e-2}),
    ('AdamW', {'learning_rate': 1e-2, 'weight_decay': 0.01}),
]


def test_optimizer_config_surface():
    from sup3r_amd.optimizers import (Adam, get_optimizer_class,
                                      init_optimizer)
    o = init_optimizer(None, 1e-4)
    assert isinstance(o, Adam) and o.learning_rate == 1e-4
    o = init_optimizer('SGD', 5e-3)
    assert o.get_config() == {'name': 'SGD', 'learning_rate': 5e-3,
                              'momentum': 0.0, 'nesterov': False}
    o = init_optimizer({'name': 'RMSprop', 'learning_rate': 1e-3,
                        'rho': 0.95}, None)
    assert o.rho == 0.95 and o.name == 'RMSprop'
    again = get_optimizer_class(o.get_config()).from_config(o.get_config())
    assert again.get_config() == o.get_config()
    for bad in ({'name': 'Nadam'}, {'name': 'Adam', 'amsgrad': True},
                {'name': 'RMSprop', 'centered': True},
                {'name': 'SGD', 'clipnorm': 1.0}):
        with pytest.raises(KeyError):
            init_optimizer(bad, None)


# what keras 2.15 `tf.keras.optimizers.<Name>().get_config()` returns — the
# dict abstract.py:544-557 writes into model_params.json and `Sup3rGan.load`
# hands back to `init_optimizer` (abstract.py:339-346)
_KERAS_BASE = {'weight_decay': None, 'clipnorm': None, 'global_clipnorm': None,
               'clipvalue': None, 'use_ema': False, 'ema_momentum': 0.99,
               'ema_overwrite_frequency': None, 'jit_compile': True,
               'is_legacy_optimizer': False}
KERAS_CONFIGS = [
    dict(_KERAS_BASE, name='Adam', learning_rate=1e-4, beta_1=0.9,
         beta_2=0.999, epsilon=1e-07, amsgrad=False),
    dict(_KERAS_BASE, name='SGD', learning_rate=0.01, momentum=0.0,
         nesterov=False),
    dict(_KERAS_BASE, name='RMSprop', learning_rate=0.001, rho=0.9,
         momentum=0.0, epsilon=1e-07, centered=False),
    dict(_KERAS_BASE, name='AdamW', learning_rate=0.001, weight_decay=0.004,
         beta_1=0.9, beta_2=0.999, epsilon=1e-07, amsgrad=False),
]


@pytest.mark.parametrize('conf', KERAS_CONFIGS, ids=lambda c: c['name'])
def test_verbatim_keras_get_config_loads(conf):
    """round 3 refused `ema_momentum=0.99` / `jit_compile=True`, which every
    keras config carries: no model_params.json of the reference loaded"""
    from sup3r_amd.optimizers import init_optimizer
    o = init_optimizer(dict(conf), None)
    assert type(o).__name__ == conf['name']
    assert o.learning_rate == conf['learning_rate']
    for k, v in o.get_config().items():
        assert conf[k] == v, k
    # switched ON, the same options are refused
    for k, v in (('use_ema', True), ('clipnorm', 1.0), ('clipvalue', 0.5),
                 ('global_clipnorm', 2.0)):
        with pytest.raises(KeyError):
            init_optimizer(dict(conf, **{k: v}), None)


def test_model_params_with_keras_optimizer_config_load(tmp_path):
    """`Sup3rGan.load` on a params file whose optimizer entries are keras
    configs (base.py:133-214)"""
    import json
    from sup3r_amd.optimizers import init_optimizer
    params = {'optimizer': KERAS_CONFIGS[0], 'optimizer_disc': KERAS_CONFIGS[1]}
    fp = tmp_path / 'model_params.json'
    fp.write_text(json.dumps(params))
    back = json.loads(fp.read_text())
    assert init_optimizer(back['optimizer'], None).beta_2 == 0.999
    assert init_optimizer(back['optimizer_disc'], None).momentum == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('name,kw', CASES)
def test_device_steps_vs_keras_restatement(name, kw):
    from oracle.gan import KerasOptimizer
    from sup3r_amd import _lib
    from sup3r_amd.engine import Network
    from sup3r_amd.optimizers import init_optimizer
    spec = [{'class': 'Conv2D', 'filters': 5, 'kernel_size': 3},
            {'class': 'Conv2D', 'filters': 3, 'kernel_size': 3}]
    net = Network(spec, precision='f32')
    net.build((1, 8, 8, 2), seed=1)
    opt = init_optimizer(dict(kw, name=name), None)
    ref = KerasOptimizer(name, **kw)
    w = [a.astype(np.float64) for a in net.weights]
    rng = np.random.default_rng(5)
    for step in range(3):
        g = [rng.standard_normal(a.shape).astype(np.float32) * 0.3 for a in w]
        net.set_weights(g, which=_lib.BUF_G)
        opt.iterations += 1
        net.optimizer_step(opt.KIND, opt.hyper(), opt.iterations)
        ref.apply_gradients([a.astype(np.float64) for a in g], w)
        for a, b in zip(net.weights, w):
            assert np.abs(a - b).max() < 2e-6 * max(1.0, np.abs(b).max()), \
                (name, kw, step)
    for a, b in zip(net.slots('v'), ref.v):
        assert np.abs(a - b).max() < 1e-5 * max(1e-3, np.abs(b).max())


@pytest.mark.gpu
def test_gan_trains_with_a_named_optimizer(tmp_path):
    """``Sup3rGan(optimizer='SGD' | {...})``: the step goes through the fused
    kernel of that optimizer, ``update_optimizer`` swaps settings and keeps
    the step count (base.py:326-348), ``history`` carries its slot columns"""
    import os
    from sup3r_amd import Sup3rGan
    from tests.helpers import SyntheticBatchHandler
    cfg = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd',
                       'configs')
    m = Sup3rGan(os.path.join(cfg, 'test_gen_s_2x_2f.json'),
                 os.path.join(cfg, 'test_disc_s_same.json'),
                 optimizer={'name': 'SGD', 'learning_rate': 1e-3,
                            'momentum': 0.9},
                 optimizer_disc='RMSprop', learning_rate_disc=1e-4,   # (config only)
                 loss='MeanAbsoluteError')
    bh = SyntheticBatchHandler((10, 10, 1), 2, 1, ['u', 'v'], batch_size=4,
                               n_batches=2)
    m.train(bh, {'spatial': '8km', 'temporal': '60min'}, 1,
            weight_gen_advers=1e-3, train_disc=False, checkpoint_int=None,
            out_dir=os.path.join(str(tmp_path), 'gan_{epoch}'))
    w0 = [w.copy() for w in m.generator_weights]
    n_it = m.optimizer.iterations
    assert n_it == 2 and m.optimizer.name == 'SGD'
    assert any(c.startswith('OptmGen/SGD/m/') for c in m.history.columns)
    m.update_optimizer('gen', learning_rate=5e-4)
    assert m.optimizer.learning_rate == 5e-4 and m.optimizer.iterations == n_it
    m.train(bh, {'spatial': '8km', 'temporal': '60min'}, 1,
            weight_gen_advers=1e-3, train_disc=False, checkpoint_int=None,
            out_dir=os.path.join(str(tmp_path), 'gan_{epoch}'))
    assert any(not np.array_equal(a, b)
               for a, b in zip(m.generator_weights, w0))
    assert m.model_params['optimizer']['name'] == 'SGD'
