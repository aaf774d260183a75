"""Batch transform on the device against the oracle
restatement of SingleBatchQueue.transform, whose gaussian filter IS the
reference's scipy call.  Tolerance: fp32 round-off of sums of <= 25 + 65 terms
(1e-5 on O(1) data)."""
import numpy as np
import pytest


def test_gaussian_taps_match_scipy():
    """the 1-D taps handed to the kernel are scipy's (``_gaussian_kernel1d``)"""
    from scipy.ndimage import gaussian_filter1d
    from sup3r_amd.batch_transform import gaussian_taps
    for sigma in (0.6, 1.0, 2.5):
        taps, radius = gaussian_taps(sigma)
        imp = np.zeros(4 * radius + 1)
        imp[2 * radius] = 1.0
        ref = gaussian_filter1d(imp, sigma, mode='nearest')
        np.testing.assert_allclose(ref[radius:3 * radius + 1], taps,
                                   rtol=0, atol=1e-15)


def test_oracle_transform_shapes_and_errors():
    from oracle.transform import transform
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 12, 8, 8, 3))
    lr, hr = transform(x, 2, 4, ['a', 'b', 'c'], [0, 1], smoothing=1.0,
                       smoothing_ignore=['c'],
                       temporal_coarsening_method='average')
    assert lr.shape == (2, 6, 4, 2, 3) and hr.shape == (2, 12, 8, 8, 2)
    with pytest.raises(ValueError):
        transform(x, 5, 4, ['a', 'b', 'c'], [0, 1])
    with pytest.raises(KeyError):
        transform(x, 2, 4, ['a', 'b', 'c'], [0, 1],
                  temporal_coarsening_method='median')


@pytest.mark.gpu
@pytest.mark.parametrize('shape,s,t,method,smoothing,ignore', [
    ((3, 12, 10, 8, 3), 2, 4, 'subsample', None, None),
    ((2, 15, 10, 12, 2), 5, 3, 'average', 0.8, None),
    ((2, 12, 12, 6, 3), 3, 2, 'total', 1.3, ['v']),
    ((2, 8, 8, 4, 2), 2, 2, 'max', None, None),
    ((2, 8, 8, 4, 2), 2, 2, 'min', 2.0, ['u']),
    ((4, 20, 16, 3), 4, 1, 'subsample', 1.0, None),        # 4-D batch
])
def test_device_transform_vs_oracle(shape, s, t, method, smoothing, ignore):
    from oracle.transform import transform
    from sup3r_amd.batch_transform import DeviceBatchTransform
    rng = np.random.default_rng(1)
    feats = ['u', 'v', 'w'][:shape[-1]]
    hr_ind = list(range(shape[-1] - 1)) if shape[-1] > 2 else [0, 1]
    x = rng.standard_normal(shape).astype(np.float32)
    lr_ref, hr_ref = transform(x.astype(np.float64), s, t, feats, hr_ind,
                               smoothing, ignore, method)
    tr = DeviceBatchTransform(s, t, feats, hr_ind)
    lr, hr = tr.transform(x, smoothing=smoothing, smoothing_ignore=ignore,
                          temporal_coarsening_method=method)
    assert tuple(lr.shape) == lr_ref.shape and tuple(hr.shape) == hr_ref.shape
    np.testing.assert_allclose(lr.cpu().numpy(), lr_ref, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(hr.cpu().numpy(), hr_ref.astype(np.float32))


@pytest.mark.gpu
def test_device_transform_errors():
    from sup3r_amd.batch_transform import DeviceBatchTransform
    tr = DeviceBatchTransform(5, 4, ['u', 'v'])
    with pytest.raises(ValueError):
        tr.transform(np.zeros((1, 12, 10, 8, 2), np.float32))
    tr = DeviceBatchTransform(2, 4, ['u', 'v'])
    with pytest.raises(KeyError):
        tr.transform(np.zeros((1, 12, 10, 8, 2), np.float32),
                     temporal_coarsening_method='median')
