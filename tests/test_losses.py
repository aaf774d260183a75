"""Structured content losses.

CPU: the oracle restatement (oracle/losses.py) under the reference's own test
procedures (/root/reference/tests/utilities/test_loss_metrics.py:56-309).
GPU: Sup3rGan.calc_loss values against the oracle (fp32 reductions: rtol 2e-5)
and the gradient the kernels leave in the generator-output gradient buffer
against central differences of the float64 oracle along random directions
(rtol 2e-3: fp32 gradient vs O(eps^2) differences)."""
import numpy as np
import pytest

from oracle import losses as OL


# ----------------------------------------------------------------- CPU: oracle
def test_oracle_md_loss_reference_procedure():
    """test_md_loss: _compute_md against np.gradient"""
    rng = np.random.default_rng(0)
    x = rng.random((6, 10, 10, 8, 3))
    u_np = np.gradient(x[..., 0], axis=3) + x[..., 0] * np.gradient(
        x[..., 0], axis=1) + x[..., 1] * np.gradient(x[..., 0], axis=2)
    v_np = np.gradient(x[..., 1], axis=3) + x[..., 0] * np.gradient(
        x[..., 1], axis=1) + x[..., 1] * np.gradient(x[..., 1], axis=2)
    assert np.allclose(OL.compute_md(x, 0), u_np)
    assert np.allclose(OL.compute_md(x, 1), v_np)
    with pytest.raises(ValueError):
        OL.derivative(x, axis=0)
    with pytest.raises(AssertionError):
        OL.material_derivative_loss(x[..., 0], x[..., 0])


def test_oracle_multiterm_reference_procedure():
    """test_multiterm_loss"""
    rng = np.random.default_rng(1)
    x = rng.random((6, 10, 10, 8, 3))
    y = rng.random((6, 10, 10, 8, 3))
    loss, _ = OL.multi_term_loss({'MaterialDerivativeLoss': {},
                                  'MeanAbsoluteError': {},
                                  'term_weights': [0.2, 0.8]}, x, y)
    assert np.allclose(0.2 * OL.material_derivative_loss(x, y) +
                       0.8 * OL.mae(x, y), loss)


def test_oracle_coarse_mse_and_extremes_reference_procedure():
    """test_coarse_mse_loss, test_tex_loss, test_spex_loss"""
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, (6, 10, 10, 8, 3))
    y = rng.uniform(0, 1, (6, 10, 10, 8, 3))
    assert OL.mse(x, y) > 10 * OL.coarse_mse_loss(x, y)
    for sign in (1, -1):
        x = np.zeros((1, 1, 1, 72, 1))
        y = np.zeros((1, 1, 1, 72, 1))
        x[..., 24, 0] = sign * 20
        y[..., 25, 0] = sign * 25
        assert OL.temporal_extremes_loss(x, y) > 1.5
        x = np.zeros((1, 10, 10, 2, 1))
        y = np.zeros((1, 10, 10, 2, 1))
        x[:, 5, 5, :, 0] = sign * 20
        y[:, 5, 5, :, 0] = sign * 25
        assert OL.spatial_extremes_loss(x, y) > 1.5


def test_oracle_lr_loss_reference_procedure():
    """test_lr_loss: coarsen with the utilities, then the pointwise loss"""
    from oracle.transform import spatial_coarsening, temporal_coarsening
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (3, 10, 10, 48, 2))
    y = rng.uniform(-1, 1, (3, 10, 10, 48, 2))
    assert np.allclose(OL.low_res_loss(x, y), OL.mse(x, y))
    xl, yl = spatial_coarsening(x, 5), spatial_coarsening(y, 5)
    assert np.allclose(OL.low_res_loss(x, y, s_enhance=5), OL.mse(xl, yl))
    for meth in ('average', 'subsample'):
        xt = temporal_coarsening(xl, 12, meth)
        yt = temporal_coarsening(yl, 12, meth)
        assert np.allclose(OL.low_res_loss(x, y, s_enhance=5, t_enhance=12,
                                           t_method=meth), OL.mse(xt, yt))
    x4 = rng.uniform(-1, 1, (3, 10, 10, 2))
    y4 = rng.uniform(-1, 1, (3, 10, 10, 2))
    base = OL.low_res_loss(x4, y4, s_enhance=5)
    assert np.allclose(base, OL.mse(spatial_coarsening(x4, 5),
                                    spatial_coarsening(y4, 5)))
    assert OL.low_res_loss(x4, y4, s_enhance=5,
                           ex_loss='SpatialExtremesLoss') > base


def test_oracle_mmd_reference_procedure():
    """test_mmd_loss: adding MMD raises the loss for a pattern mismatch and
    lowers it when only the shift differs"""
    x = np.zeros((6, 10, 10, 8, 3))
    y = np.zeros((6, 10, 10, 8, 3))
    x[:, 7:9, 7:9, :, :] = 1
    y[:, 2:5, 2:5, :, :] = 1
    assert OL.mmd_loss(x, y) > 0
    assert abs(OL.mmd_loss(x, x)) < 1e-12


def test_loss_spec_errors():
    from sup3r_amd.compute import parse_loss_spec
    with pytest.raises(KeyError):
        parse_loss_spec('PerceptualLoss')
    with pytest.raises(TypeError):
        parse_loss_spec({'LowResLoss': {'bogus': 1}})
    with pytest.raises(KeyError):
        parse_loss_spec({'LowResLoss': {'ex_loss': 'MmdLoss'}})
    with pytest.raises(TypeError):
        parse_loss_spec({'CoarseMseLoss': {'x': 1}})


# ----------------------------------------------------------------- GPU: device
SPECS = [
    'ExpLoss', 'MmdLoss', 'MaterialDerivativeLoss', 'SpatialDerivativeLoss',
    'TemporalDerivativeLoss', 'CoarseMseLoss', 'SpatialExtremesLoss',
    'TemporalExtremesLoss',
    {'LowResLoss': {'s_enhance': 3, 't_enhance': 4, 't_method': 'average'}},
    {'LowResLoss': {'s_enhance': 2, 't_enhance': 2, 't_method': 'subsample',
                    'tf_loss': 'MeanAbsoluteError',
                    'ex_loss': 'TemporalExtremesLoss'}},
    {'LowResLoss': {'s_enhance': 1, 't_enhance': 1,
                    'ex_loss': 'SpatialExtremesLoss'}},
    {'MaterialDerivativeLoss': {}, 'MeanAbsoluteError': {},
     'SpatialExtremesLoss': {}, 'term_weights': [0.2, 0.7, 0.1]},
    'SpatiotemporalFftLoss',
]


def _device_loss_and_grad(spec, gen, true, n_exo=0):
    """calc_loss value and d loss / d gen through HipGanCompute"""
    import torch
    from sup3r_amd import _lib
    from sup3r_amd.compute import HipGanCompute, parse_loss_spec
    from sup3r_amd.utilities import camel_to_underscore
    terms = parse_loss_spec(spec)
    cp = HipGanCompute.__new__(HipGanCompute)
    from sup3r_amd.engine import Device
    cp.dev = Device.get()
    cp._scal = None
    dev = cp.dev
    g, t = dev.to_device(gen), dev.to_device(true)
    scal = cp._scalars()
    L = _lib.lib()
    L.s3_fill(dev.ctx, cp._ptr(scal), scal.numel(), 0.0)
    d = torch.zeros_like(g)
    c_used = gen.shape[-1] - n_exo
    from sup3r_amd.compute import SLOTS_PER_TERM
    total = 0.0
    coefs = []
    for i, (name, kind, w, kw) in enumerate(terms):
        slot = 4 + SLOTS_PER_TERM * i
        if isinstance(kind, str):
            coefs.append(cp._structured_term(name, kind, kw, g, t, c_used, w,
                                             scal, slot, d))
        else:
            coefs.append([1.0])
            rc = L.s3_loss_content(dev.ctx, kind, cp._ptr(g), g.shape[-1],
                                   cp._ptr(t), t.shape[-1], c_used,
                                   g.numel() // g.shape[-1], w,
                                   cp._ptr(scal, slot), cp._ptr(d), 1)
            _lib.check(rc, dev.ctx, 's3_loss_content')
    vals = scal.cpu().numpy()
    for i, (name, kind, w, kw) in enumerate(terms):
        slot = 4 + SLOTS_PER_TERM * i
        total += w * sum(cf * float(vals[slot + j])
                         for j, cf in enumerate(coefs[i]))
    return total, d.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('spec', SPECS, ids=[str(i) for i in range(len(SPECS))])
def test_device_losses_value_and_gradient_vs_oracle(spec):
    rng = np.random.default_rng(7)
    shape = (3, 12, 6, 8, 4)
    gen = rng.standard_normal(shape).astype(np.float32)
    true = rng.standard_normal(shape).astype(np.float32)
    loss, grad = _device_loss_and_grad(spec, gen, true)
    g64, t64 = gen.astype(np.float64), true.astype(np.float64)
    ref, _ = OL.multi_term_loss(spec, g64, t64)
    assert abs(loss - ref) <= 2e-5 * max(1.0, abs(ref)), (loss, ref)
    assert np.isfinite(grad).all()
    for k in range(3):
        v = rng.standard_normal(shape)
        eps = 1e-5
        fd = (OL.multi_term_loss(spec, g64 + eps * v, t64)[0] -
              OL.multi_term_loss(spec, g64 - eps * v, t64)[0]) / (2 * eps)
        an = float((grad.astype(np.float64) * v).sum())
        assert abs(an - fd) <= 2e-3 * max(abs(fd), 1e-3), (spec, an, fd)


@pytest.mark.gpu
def test_device_losses_4d_and_exo_channels():
    """4-D batches (spatial models) and trailing exo channels that take no
    part in the content loss (calc_loss_gen_content, base.py:478-503)"""
    rng = np.random.default_rng(8)
    gen = rng.standard_normal((4, 10, 15, 3)).astype(np.float32)
    true = rng.standard_normal((4, 10, 15, 3)).astype(np.float32)
    for spec in ('SpatialDerivativeLoss', 'CoarseMseLoss', 'SpatialExtremesLoss',
                 {'LowResLoss': {'s_enhance': 5}}):
        loss, grad = _device_loss_and_grad(spec, gen, true, n_exo=1)
        ref, _ = OL.multi_term_loss(spec, gen[..., :2].astype(np.float64),
                                    true[..., :2].astype(np.float64))
        assert abs(loss - ref) <= 2e-5 * max(1.0, abs(ref)), spec
        assert np.all(grad[..., 2] == 0)
    with pytest.raises(AssertionError):
        _device_loss_and_grad('TemporalDerivativeLoss', gen, true)


def test_oracle_st_fft_loss_reference_procedure():
    """test_st_fft_loss: the 1/4 (FFT + spatial extremes + temporal extremes +
    MAE) combination on all-zero fields with one spike is > 1"""
    def loss_obj(x, y):
        return 0.25 * (OL.spatiotemporal_fft_loss(x, y) +
                       OL.spatial_extremes_loss(x, y) +
                       OL.temporal_extremes_loss(x, y) + OL.mae(x, y))
    for sign in (1, -1):
        x = np.zeros((1, 10, 10, 5, 1))
        y = np.zeros((1, 10, 10, 5, 1))
        assert OL.spatiotemporal_fft_loss(x, y) == 0
        x[:, 5, 5, 2, 0] = sign * 100
        y[:, 5, 5, 2, 0] = sign * 150
        assert loss_obj(x, y) > 1.0
        assert OL.spatiotemporal_fft_loss(x, y) > 0.2


@pytest.mark.gpu
def test_device_spatial_fft_loss_4d_and_dft_axis():
    """s3_dft_axis against numpy's fft (1e-4 of the largest bin: fp32 direct
    sums of up to 37 terms) and SpatialFftLoss on a 4-D batch with a trailing
    exo channel"""
    import torch
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    import ctypes as C
    rng = np.random.default_rng(11)
    dev = Device.get()
    L = _lib.lib()
    x = rng.standard_normal((3, 37, 20)).astype(np.float32)
    xi = rng.standard_normal((3, 37, 20)).astype(np.float32)
    d = [dev.to_device(v) for v in (x, xi)]
    ore, oim = torch.empty_like(d[0]), torch.empty_like(d[0])
    for sign in (-1, 1):
        rc = L.s3_dft_axis(dev.ctx, C.c_void_p(d[0].data_ptr()),
                           C.c_void_p(d[1].data_ptr()),
                           C.c_void_p(ore.data_ptr()), C.c_void_p(oim.data_ptr()),
                           3, 37, 20, sign)
        _lib.check(rc, dev.ctx, 's3_dft_axis')
        z = x.astype(np.float64) + 1j * xi
        ref = np.fft.fft(z, axis=1) if sign < 0 else np.fft.ifft(z, axis=1) * 37
        got = ore.cpu().numpy() + 1j * oim.cpu().numpy()
        assert np.abs(got - ref).max() < 1e-4 * np.abs(ref).max()
    gen = rng.standard_normal((4, 10, 15, 3)).astype(np.float32)
    true = rng.standard_normal((4, 10, 15, 3)).astype(np.float32)
    loss, grad = _device_loss_and_grad('SpatialFftLoss', gen, true, n_exo=1)
    ref = OL.spatial_fft_loss(gen[..., :2].astype(np.float64),
                              true[..., :2].astype(np.float64))
    assert abs(loss - ref) <= 1e-4 * max(1.0, abs(ref))
    assert np.all(grad[..., 2] == 0) and np.abs(grad[..., :2]).max() > 0
    g64, t64 = gen[..., :2].astype(np.float64), true[..., :2].astype(np.float64)
    v = rng.standard_normal(g64.shape)
    eps = 1e-5
    fd = (OL.spatial_fft_loss(g64 + eps * v, t64) -
          OL.spatial_fft_loss(g64 - eps * v, t64)) / (2 * eps)
    an = float((grad[..., :2].astype(np.float64) * v).sum())
    assert abs(an - fd) <= 5e-3 * max(abs(fd), 1e-3), (an, fd)


@pytest.mark.gpu
def test_device_extremes_reference_procedure_and_ties():
    """test_tex_loss / test_spex_loss inputs (all-zero fields with one spike:
    every other element ties for the opposite extremum — the gradient is
    shared equally, as tf.reduce_min / reduce_max do)"""
    for sign in (1, -1):
        x = np.zeros((1, 1, 1, 72, 1), np.float32)
        y = np.zeros((1, 1, 1, 72, 1), np.float32)
        x[..., 24, 0] = sign * 20
        y[..., 25, 0] = sign * 25
        loss, grad = _device_loss_and_grad('TemporalExtremesLoss', x, y)
        assert loss > 1.5 and abs(loss - OL.temporal_extremes_loss(x, y)) < 1e-5
        # spike: d|20 - 25| / 2 = -sign/2; the 71 tied zeros vs the 71 zeros of y: 0
        assert abs(grad[0, 0, 0, 24, 0] + sign * 0.5) < 1e-6
        x = np.zeros((1, 10, 10, 2, 1), np.float32)
        y = np.zeros((1, 10, 10, 2, 1), np.float32)
        x[:, 5, 5, :, 0] = sign * 20
        y[:, 5, 5, :, 0] = sign * 25
        loss, _ = _device_loss_and_grad('SpatialExtremesLoss', x, y)
        assert loss > 1.5 and abs(loss - OL.spatial_extremes_loss(x, y)) < 1e-5


@pytest.mark.gpu
def test_gan_trains_with_structured_content_loss():
    """Sup3rGan with a multi-term structured content loss: calc_loss details
    and one generator step (the gradient reaches the weights)"""
    import os
    from sup3r_amd import Sup3rGan
    cfg = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')
    spec = {'MeanAbsoluteError': {}, 'MaterialDerivativeLoss': {},
            'LowResLoss': {'s_enhance': 2, 't_enhance': 4},
            'term_weights': [0.6, 0.2, 0.2]}
    model = Sup3rGan(os.path.join(cfg, 'test_gen_st_2x_4x_2f.json'),
                     os.path.join(cfg, 'test_disc_st_same.json'), loss=spec,
                     learning_rate=1e-3)
    rng = np.random.default_rng(9)
    lr = rng.standard_normal((4, 5, 6, 4, 3)).astype(np.float32)
    hr = rng.standard_normal((4, 10, 12, 16, 2)).astype(np.float32)
    model.init_weights(lr.shape, hr.shape)
    gen = model._tf_generate(lr)
    gen = gen.cpu().numpy() if hasattr(gen, 'cpu') else np.asarray(gen)
    loss, details = model.calc_loss(hr, gen, weight_gen_advers=0.0,
                                    train_gen=True, train_disc=False)
    ref, parts = OL.multi_term_loss(spec, gen.astype(np.float64),
                                    hr.astype(np.float64))
    assert abs(float(details['loss_gen_content']) - ref) < 2e-5 * max(1, ref)
    assert abs(float(details['material_derivative_loss']) -
               parts['MaterialDerivativeLoss']) < 2e-5
    w0 = [np.array(w) for w in model.generator_weights]

    class B:
        low_res, high_res = lr, hr
    d = model._train_batch(B, True, False, False, True, False, False, 0.0)
    assert np.isfinite(d['loss_gen'])
    assert any(np.abs(np.array(a) - b).max() > 0
               for a, b in zip(model.generator_weights, w0))


@pytest.mark.gpu
def test_sup3r_gan_dc_validation_updates_sampling_weights():
    """Sup3rGanDC.calc_val_loss (sup3r/models/dc.py:64-116): bin losses ->
    normalised spatial / temporal weights handed to the batch handler."""
    import os
    from sup3r_amd import Sup3rGanDC
    cfg = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')
    model = Sup3rGanDC(os.path.join(cfg, 'test_gen_st_2x_4x_2f.json'),
                       os.path.join(cfg, 'test_disc_st_same.json'),
                       loss='MeanAbsoluteError')
    rng = np.random.default_rng(10)

    class Batch:
        def __init__(self, scale):
            self.low_res = rng.standard_normal((2, 5, 6, 4, 3)).astype(np.float32)
            self.high_res = scale * rng.standard_normal(
                (2, 10, 12, 16, 2)).astype(np.float32)

    class Handler:
        n_space_bins, n_time_bins = 2, 3
        spatial_weights = [0.5, 0.5]
        temporal_weights = [1 / 3] * 3
        val_data = [Batch(s) for s in (1, 1, 1, 1, 1, 8)]

        def update_weights(self, spatial_weights, temporal_weights):
            self.spatial_weights = spatial_weights
            self.temporal_weights = temporal_weights
    bh = Handler()
    model.init_weights((2, 5, 6, 4, 3), (2, 10, 12, 16, 2))
    total, content = model.calc_val_loss_gen(bh, 0.0)
    assert total.shape == content.shape == (2, 3)
    assert total[1, 2] == total.max()          # the bin with 8x larger truth
    details = model.calc_val_loss(bh, 0.0)
    assert set(details) == {'mean_val_loss_gen', 'mean_val_loss_gen_content'}
    assert abs(np.sum(bh.spatial_weights) - 1) < 1e-6
    assert abs(np.sum(bh.temporal_weights) - 1) < 1e-6
    assert np.argmax(bh.spatial_weights) == 1
    assert np.argmax(bh.temporal_weights) == 2


def test_oracle_sliced_wasserstein_properties():
    """SlicedWassersteinLoss has no test in the reference; the restatement is
    checked on what its definition implies: zero for identical fields and for
    a permutation of the observations' positions that the directions cannot
    see after sorting only when the fields agree; positive and growing with a
    shift; invariant to the scale of the raw directions (l2-normalised)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 6, 4, 3))
    proj = rng.standard_normal((64, 5 * 6 * 4))
    assert OL.sliced_wasserstein_loss(x, x, proj) == 0.0
    l1 = OL.sliced_wasserstein_loss(x, x + 0.5, proj)
    l2 = OL.sliced_wasserstein_loss(x, x + 1.0, proj)
    assert 0 < l1 < l2
    assert abs(OL.sliced_wasserstein_loss(x, x + 0.5, 7.0 * proj) - l1) < 1e-12
    x4 = rng.standard_normal((2, 5, 6, 3))
    p4 = rng.standard_normal((16, 30))
    assert OL.sliced_wasserstein_loss(x4, x4 * 1.1, p4) > 0


def test_parse_sliced_wasserstein_spec():
    from sup3r_amd.compute import parse_loss_spec
    t = parse_loss_spec({'SlicedWassersteinLoss': {'n_projections': 64},
                         'MeanAbsoluteError': {}, 'term_weights': [0.3, 0.7]})
    assert t[0][:3] == ('SlicedWassersteinLoss', 'sw', 0.3)
    assert t[0][3] == {'n_projections': 64}
    with pytest.raises(TypeError):
        parse_loss_spec({'SlicedWassersteinLoss': {'sigma': 1.0}})


def _sw_device(gen, true, n_proj, seed, want_grad=True, n_exo=0):
    import torch
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev = Device.get()
    L = _lib.lib()
    g, t = dev.to_device(gen), dev.to_device(true)
    c = gen.shape[-1]
    npos = int(np.prod(gen.shape[1:-1]))
    out = dev.empty((4,))
    L.s3_fill(dev.ctx, out.data_ptr(), 4, 0.0)
    d = torch.zeros_like(g)
    rc = L.s3_loss_sliced_wasserstein(
        dev.ctx, g.data_ptr(), c, t.data_ptr(), c, gen.shape[0], npos,
        c - n_exo, n_proj, seed, 1.0, out.data_ptr(),
        d.data_ptr() if want_grad else None)
    _lib.check(rc, dev.ctx, 's3_loss_sliced_wasserstein')
    proj = dev.empty((n_proj, npos))
    rc = L.s3_sw_directions(dev.ctx, seed, n_proj, npos, proj.data_ptr())
    _lib.check(rc, dev.ctx, 's3_sw_directions')
    return float(out.cpu().numpy()[0]), d.cpu().numpy(), proj.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('shape,n_proj', [((3, 7, 6, 9, 2), 50),
                                          ((2, 9, 5, 3), 128),
                                          ((5, 6, 7, 11, 5), 33)])
def test_device_sliced_wasserstein_vs_oracle(shape, n_proj):
    """s3_loss_sliced_wasserstein against the oracle fed with the directions
    the device drew (s3_sw_directions): value to 2e-5, gradient against central
    differences of the float64 oracle; ragged position counts (not a multiple
    of 4 / 256), n_proj not a power of two, > 16 (observation, feature)
    columns (two column passes); the directions are standard normal draws."""
    rng = np.random.default_rng(21)
    gen = rng.standard_normal(shape).astype(np.float32)
    true = (rng.standard_normal(shape) * 1.3 + 0.2).astype(np.float32)
    seed = 0x1234_5678_9ABC_DEF0
    loss, grad, proj = _sw_device(gen, true, n_proj, seed)
    assert abs(proj.mean()) < 5 / np.sqrt(proj.size)
    assert abs(proj.std() - 1) < 5 / np.sqrt(proj.size)
    g64, t64 = gen.astype(np.float64), true.astype(np.float64)
    ref = OL.sliced_wasserstein_loss(g64, t64, proj)
    assert abs(loss - ref) <= 2e-5 * max(1.0, abs(ref)), (loss, ref)
    for k in range(3):
        v = rng.standard_normal(shape)
        eps = 1e-6
        fd = (OL.sliced_wasserstein_loss(g64 + eps * v, t64, proj) -
              OL.sliced_wasserstein_loss(g64 - eps * v, t64, proj)) / (2 * eps)
        an = float((grad.astype(np.float64) * v).sum())
        assert abs(an - fd) <= 2e-3 * max(abs(fd), 1e-3), (an, fd)
    # another seed: other directions, the same seed: the same bits
    loss2, _, proj2 = _sw_device(gen, true, n_proj, seed + 1, want_grad=False)
    assert np.abs(proj2 - proj).max() > 0.1 and loss2 != loss
    loss3, grad3, _ = _sw_device(gen, true, n_proj, seed)
    assert loss3 == loss and np.array_equal(grad3, grad)


@pytest.mark.gpu
def test_device_sliced_wasserstein_through_calc_loss_spec():
    """as a ``get_loss_fun`` term next to MAE, with an exo channel that takes
    no part; identical fields -> 0 and zero gradient"""
    rng = np.random.default_rng(22)
    shape = (2, 8, 8, 6, 3)
    gen = rng.standard_normal(shape).astype(np.float32)
    true = rng.standard_normal(shape).astype(np.float32)
    spec = {'SlicedWassersteinLoss': {'n_projections': 64},
            'MeanAbsoluteError': {}, 'term_weights': [0.4, 0.6]}
    loss, grad = _device_loss_and_grad(spec, gen, true, n_exo=1)
    mae = OL.mae(gen[..., :2].astype(np.float64), true[..., :2].astype(np.float64))
    assert loss > 0.6 * mae and np.isfinite(grad).all()
    assert np.all(grad[..., 2] == 0)
    l0, g0, _ = _sw_device(gen, gen.copy(), 64, 5)
    assert l0 == 0.0 and not g0.any()
