"""CPU restatement of the structured content losses — TEST INFRASTRUCTURE ONLY
(imported by tests/ only; see oracle/__init__.py).

Follows sup3r/utilities/loss_metrics.py: ``_derivative`` (:12-59), ExpLoss
(:98-118), gaussian_kernel / MmdLoss (:62-147), MaterialDerivativeLoss
(:150-225), SpatialDerivativeLoss (:228-260), TemporalDerivativeLoss (:263-294),
CoarseMseLoss (:297-322), SpatialExtremesLoss (:325-357), TemporalExtremesLoss
(:360-392), SpatialFftLoss / SpatiotemporalFftLoss (:395-485), LowResLoss
(:488-638), with keras MeanAbsoluteError /
MeanSquaredError = global means for equal shapes.  Pinned by
the reference's own test procedures (tests/utilities/test_loss_metrics.py:
``test_md_loss`` against np.gradient, ``test_lr_loss`` against the coarsening
utilities, the extremes / coarse-MSE inequalities) in tests/test_losses.py.
"""
import numpy as np

from .transform import spatial_coarsening, temporal_coarsening


def mae(a, b):
    return float(np.mean(np.abs(a - b)))


def mse(a, b):
    return float(np.mean((a - b) ** 2))


def derivative(x, axis=1):
    """central differences inside, one-sided at the two ends (= np.gradient)"""
    if axis not in (1, 2, 3):
        raise ValueError(f'_derivative received axis={axis}.')
    x = np.moveaxis(x, axis, 0)
    out = np.concatenate([x[1:2] - x[0:1], (x[2:] - x[:-2]) / 2,
                          x[-1:] - x[-2:-1]], axis=0)
    return np.moveaxis(out, 0, axis)


def compute_md(x, fidx):
    uidx = 2 * (fidx // 2)
    vidx = uidx + 1
    out = derivative(x[..., fidx], axis=3)
    out = out + x[..., uidx] * derivative(x[..., fidx], axis=1)
    out = out + x[..., vidx] * derivative(x[..., fidx], axis=2)
    return out


def material_derivative_loss(x1, x2):
    assert x1.ndim == 5 and x2.ndim == 5
    hub = x1.shape[-1] // 2
    d1 = np.stack([compute_md(x1, i) for i in range(0, 2 * hub, 2)])
    d2 = np.stack([compute_md(x2, i) for i in range(0, 2 * hub, 2)])
    return mae(d1, d2)


def spatial_derivative_loss(x1, x2):
    assert x1.ndim >= 4 and x2.ndim >= 4
    return mae(derivative(x1, 1) + derivative(x1, 2),
               derivative(x2, 1) + derivative(x2, 2))


def temporal_derivative_loss(x1, x2):
    assert x1.ndim == 5 and x2.ndim == 5
    return mae(derivative(x1, 3), derivative(x2, 3))


def coarse_mse_loss(x1, x2):
    return mse(x1.mean(axis=(1, 2)), x2.mean(axis=(1, 2)))


def spatial_extremes_loss(x1, x2):
    return (mae(x1.min(axis=(1, 2)), x2.min(axis=(1, 2))) +
            mae(x1.max(axis=(1, 2)), x2.max(axis=(1, 2)))) / 2


def temporal_extremes_loss(x1, x2):
    return (mae(x1.min(axis=3), x2.min(axis=3)) +
            mae(x1.max(axis=3), x2.max(axis=3))) / 2


def exp_loss(x1, x2):
    return float(np.mean(1 - np.exp(-(x1 - x2) ** 2)))


def gaussian_kernel(x1, x2, sigma=1.0):
    return np.exp(-0.5 * np.sum((np.expand_dims(x1, axis=1) - x2) ** 2,
                                axis=-1) / sigma ** 2)


def mmd_loss(x1, x2, sigma=1.0):
    return float(np.mean(gaussian_kernel(x1, x1, sigma)) +
                 np.mean(gaussian_kernel(x2, x2, sigma)) -
                 np.mean(2 * gaussian_kernel(x1, x2, sigma)))


def sliced_wasserstein_loss(x1, x2, proj):
    """SlicedWassersteinLoss.__call__ (loss_metrics.py:743-789) with the random
    directions given: ``proj`` = the raw normal draws (n_projections, H*W*T)
    that tf.random.normal produces at :777 (l2-normalised here as at :778).
    (B, H, W[, T], C) -> (B, HWT, C); proj @ x -> (B, n_proj, C); sort along
    the projection axis (:786-787); mean squared difference."""
    assert x1.ndim in (4, 5) and x1.shape == x2.shape
    b, c = x1.shape[0], x1.shape[-1]
    f1 = x1.reshape(b, -1, c).astype(np.float64)
    f2 = x2.reshape(b, -1, c).astype(np.float64)
    pr = np.asarray(proj, dtype=np.float64)
    assert pr.shape[1] == f1.shape[1]
    pr = pr / np.sqrt((pr ** 2).sum(axis=-1, keepdims=True))
    p1 = np.sort(pr @ f1, axis=1)
    p2 = np.sort(pr @ f2, axis=1)
    return float(np.mean((p1 - p2) ** 2))


def _fft_map(x, axes):
    """log(1 + w |fftn(x)|), w = product of the squared un-wrapped frequency
    indices of the transformed axes (loss_metrics.py:399-417, :445-465; numpy's
    fftn is tf.signal.fft2d / fft3d up to complex64 round-off)"""
    xh = np.abs(np.fft.fftn(x, axes=axes))
    w = np.ones([1] * x.ndim)
    for a in axes:
        shape = [1] * x.ndim
        shape[a] = x.shape[a]
        w = w * (np.arange(x.shape[a], dtype=np.float64) ** 2).reshape(shape)
    return np.log(1 + w * xh)


def spatial_fft_loss(x1, x2):
    assert x1.ndim == 4 and x2.ndim == 4
    return mae(_fft_map(x1, (1, 2)), _fft_map(x2, (1, 2)))


def spatiotemporal_fft_loss(x1, x2):
    assert x1.ndim == 5 and x2.ndim == 5
    return mae(_fft_map(x1, (1, 2, 3)), _fft_map(x2, (1, 2, 3)))


def low_res_loss(x1, x2, s_enhance=1, t_enhance=1, t_method='average',
                 tf_loss='MeanSquaredError', ex_loss=None):
    assert x1.shape == x2.shape
    ex = 0.0
    if ex_loss is not None:
        ex = {'SpatialExtremesLoss': spatial_extremes_loss,
              'TemporalExtremesLoss': temporal_extremes_loss}[ex_loss](x1, x2)
    if s_enhance > 1:
        x1 = spatial_coarsening(x1, s_enhance)
        x2 = spatial_coarsening(x2, s_enhance)
    t_method = str(t_method).casefold()
    if t_enhance > 1 and t_method in ('average', 'subsample'):
        assert x1.ndim == 5
        x1 = temporal_coarsening(x1, t_enhance, t_method)
        x2 = temporal_coarsening(x2, t_enhance, t_method)
    fun = {'MeanSquaredError': mse, 'MeanAbsoluteError': mae}[tf_loss]
    return fun(x1, x2) + ex


LOSSES = {
    'MeanAbsoluteError': mae, 'MeanSquaredError': mse, 'ExpLoss': exp_loss,
    'MmdLoss': mmd_loss, 'MaterialDerivativeLoss': material_derivative_loss,
    'SpatialDerivativeLoss': spatial_derivative_loss,
    'TemporalDerivativeLoss': temporal_derivative_loss,
    'CoarseMseLoss': coarse_mse_loss,
    'SpatialExtremesLoss': spatial_extremes_loss,
    'TemporalExtremesLoss': temporal_extremes_loss,
    'LowResLoss': low_res_loss,
    'SpatialFftLoss': spatial_fft_loss,
    'SpatiotemporalFftLoss': spatiotemporal_fft_loss,
}


def multi_term_loss(spec, x1, x2):
    """get_loss_fun (sup3r/models/abstract.py:461-502): weighted sum of terms"""
    spec = {spec: {}} if isinstance(spec, str) else dict(spec)
    names = [k for k in spec if k != 'term_weights']
    weights = spec.get('term_weights', [1.0] * len(names))
    details = {n: LOSSES[n](x1, x2, **(spec[n] or {})) for n in names}
    return sum(w * details[n] for n, w in zip(names, weights)), details
