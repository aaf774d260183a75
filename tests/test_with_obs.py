"""``Sup3rGanWithObs`` (sup3r/models/with_obs.py): observation masks, the
sparse-observation generator inputs, the extra loss terms and their gradient,
and the reference's own test procedure (tests/training/
test_train_conditioned_obs.py:23-107) on synthetic batches.

The ``Sup3rConcatObs`` layer itself lives in phygnn (not vendored): this
package's stated semantics — an un-observed cell carries 0 in normalised
units — is what the oracle restates here too; UNVERIFIED against phygnn."""
import os

import numpy as np
import pytest

from tests.helpers import SyntheticBatchHandler, emulate_plan

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _pcc(filters, act=True):
    out = [{'class': 'FlexiblePadding', 'mode': 'REFLECT',
            'paddings': [[0, 0], [3, 3], [3, 3], [0, 0]]},
           {'class': 'Conv2DTranspose', 'filters': filters, 'kernel_size': 3,
            'strides': 1, 'activation': 'relu' if act else None},
           {'class': 'Cropping2D', 'cropping': 4}]
    return out


def gen_config():
    """the shape of tests/conftest.py:gen_config_with_concat_masked: two
    ``Sup3rConcatObs`` layers between the hi-res convs"""
    return (_pcc(16) + _pcc(16) + [
        {'class': 'SpatialExpansion', 'spatial_mult': 2},
        {'class': 'Activation', 'activation': 'relu'}] + _pcc(2) + [
        {'class': 'Sup3rConcatObs', 'name': 'u_10m_obs'},
        {'class': 'Sup3rConcatObs', 'name': 'v_10m_obs'}] + _pcc(2, False))


def _model(**kw):
    from sup3r_amd import Sup3rGanWithObs
    Sup3rGanWithObs.seed(3)
    kw.setdefault('onshore_obs_frac', {'spatial': 0.1})
    kw.setdefault('loss_obs_weight', 0.1)
    m = Sup3rGanWithObs(gen_config(),
                        os.path.join(CFG, 'test_disc_s_same.json'),
                        loss='MeanAbsoluteError', learning_rate=1e-4, **kw)
    m.meta['hr_out_features'] = ['u_10m', 'v_10m']
    m.meta['lr_features'] = ['u_10m', 'v_10m']
    return m


@pytest.mark.gpu
def test_obs_masks_and_params():
    m = _model()
    assert m.obs_features == ['u_10m_obs', 'v_10m_obs']
    assert m.hr_exo_features == [] and m.obs_training_inds == [0, 1]
    mask = m._get_full_obs_mask(np.zeros((1, 20, 20, 1, 1)))
    frac = 1 - mask.sum() / mask.size
    # the reference's own acceptance bound (test_train_conditioned_obs.py:60)
    assert np.abs(0.1 - frac) < mask.size / (2 * np.sqrt(mask.size))
    big = m._get_full_obs_mask(np.zeros((6, 200, 200, 2)))
    assert big.shape == (6, 200, 200, 2) and big.dtype == bool
    assert abs((~big).mean() - 0.1) < 5e-3
    # one draw per spatial cell, shared by the features
    np.testing.assert_array_equal(big[..., 0], big[..., 1])
    p = m.model_params
    assert p['onshore_obs_frac'] == {'spatial': 0.1}
    assert p['loss_obs_weight'] == 0.1 and p['loss_obs'] == 'MeanAbsoluteError'
    # offshore cells (topography <= 0) use the sparser offshore fractions
    m2 = _model(offshore_obs_frac={'spatial': 0.0})
    m2.meta['hr_out_features'] = ['u_10m', 'v_10m']
    with pytest.raises(KeyError):
        _model(loss_obs='SpatialExtremesLoss')


@pytest.mark.gpu
def test_obs_loss_terms_and_gradients_vs_oracle():
    """loss_obs / loss_non_obs / obs_frac (with_obs.py:88-99,262-277) and the
    generator gradient of content + loss_obs_weight * loss_obs against the
    oracle fed the same sparse observation fields"""
    from oracle.network import Network as ONet
    m = _model(precision='f32')
    rng = np.random.default_rng(8)
    lr = rng.standard_normal((3, 10, 10, 2)).astype(np.float32)
    hr = rng.standard_normal((3, 20, 20, 2)).astype(np.float32)
    m.init_weights(lr.shape, hr.shape)
    og = ONet(gen_config())
    probe = {k: np.zeros((3, 20, 20, 1), np.float32)
             for k in ('u_10m_obs', 'v_10m_obs')}
    og.init_weights(lr, probe, seed=2, bias_scale=0.1)
    m.generator.set_weights(og.weights)
    # the mask the model will draw
    m._obs_rng = np.random.default_rng(42)
    mask = m._get_full_obs_mask(hr)
    m._obs_rng = np.random.default_rng(42)
    _, det = m.get_single_grad(lr, hr, weight_gen_advers=0.0, train_gen=True,
                               train_disc=False)
    exo = {n: np.where(mask[..., j:j + 1], 0, hr[..., j:j + 1]).astype(
        np.float32) for j, n in enumerate(m.obs_features)}
    y = og.forward(lr, exo)
    d = (y - hr).astype(np.float64)
    seen = ~mask
    l_obs, l_non = np.abs(d[seen]).mean(), np.abs(d[mask]).mean()
    content = np.abs(d).mean()
    assert abs(det['loss_obs'] - l_obs) < 1e-5 * max(1, l_obs)
    assert abs(det['loss_non_obs'] - l_non) < 1e-5 * max(1, l_non)
    assert abs(det['obs_frac'] - seen.mean()) < 1e-7
    assert abs(det['loss_gen_content'] - (content + 0.1 * l_obs)) < 1e-5
    assert abs(det['loss_gen'] - det['loss_gen_content']) < 1e-6
    ph = m.generator.plan(lr.shape, training=True)
    emulate_plan(og, ph, masks=True, rounding=False)
    g = np.sign(d) / d.size + 0.1 * seen * np.sign(d) / seen.sum()
    og.backward(g.astype(np.float32))
    gmax = max(float(np.abs(r).max()) for r in og.grads)
    worst = max(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3 * gmax))
                for a, b in zip(m.generator.grads, og.grads))
    print(f'Sup3rGanWithObs generator gradients vs oracle: {worst:.2e}')
    assert worst < 1e-3, worst
    # validation path: same terms, no gradient
    m._obs_rng = np.random.default_rng(42)
    loss, det_v, _, exo_v = m._get_hr_exo_and_loss(lr, hr,
                                                   weight_gen_advers=0.0)
    np.testing.assert_array_equal(exo_v['mask'], mask)
    assert abs(det_v['loss_obs'] - l_obs) < 1e-5 * max(1, l_obs)


@pytest.mark.gpu
def test_train_save_load_generate_like_the_reference_test(tmp_path):
    """tests/training/test_train_conditioned_obs.py:62-107 on synthetic data"""
    from sup3r_amd import Sup3rGanWithObs
    model = _model()
    bh = SyntheticBatchHandler((20, 20, 1), 2, 1, ['u_10m', 'v_10m'],
                               batch_size=2, n_batches=2)
    kw = {'input_resolution': {'spatial': '16km', 'temporal': '3600min'},
          'n_epoch': 2, 'weight_gen_advers': 0.0, 'train_gen': True,
          'train_disc': False, 'checkpoint_int': None,
          'out_dir': os.path.join(str(tmp_path), 'test_{epoch}')}
    model.train(bh, **kw)
    for col in ('train_loss_obs', 'train_loss_non_obs', 'train_obs_frac'):
        assert col in model.history.columns, list(model.history.columns)
    loaded = Sup3rGanWithObs.load(os.path.join(str(tmp_path), 'test_1'))
    assert loaded.onshore_obs_frac == {'spatial': 0.1}
    assert loaded.loss_obs_weight == 0.1
    loaded.train(bh, **kw)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (4, 30, 30, 2))
    obs = [rng.uniform(0, 1, (4, 60, 60, 1)) for _ in range(2)]
    gaps = rng.choice([True, False], (60, 60, 1), p=[0.9, 0.1])
    for o in obs:
        o[:, gaps] = np.nan
    with pytest.raises(RuntimeError):
        model.generate(x, exogenous_data=None)
    exo = {n: {'steps': [{'model': 0, 'combine_type': 'layer', 'data': o}]}
           for n, o in zip(('u_10m_obs', 'v_10m_obs'), obs)}
    y = model.generate(x, exogenous_data=exo)
    assert y.dtype == np.float32 and y.shape == (4, 60, 60, 2)
    assert np.isfinite(y).all()
    # an observed value matters, an un-observed one does not
    exo2 = {n: {'steps': [{'model': 0, 'combine_type': 'layer',
                           'data': np.where(np.isnan(o), np.nan, o + 1.0)}]}
            for n, o in zip(('u_10m_obs', 'v_10m_obs'), obs)}
    assert not np.array_equal(model.generate(x, exogenous_data=exo2), y)
