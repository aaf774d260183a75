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


# --- the reference's own test procedure for the coarsening helpers
# (/root/reference/tests/utilities/test_utilities.py:225-357), run on the
# oracle (CPU) and on the device transform (GPU): inputs are arange arrays, the
# expectation is the block mean / window mean|sum|first computed by slicing.

_S_COARSEN_CASES = [
    # shape, obs_axis (test_s_coarsen_5D / _4D / _4D_no_obs / _3D)
    ((2, 20, 20, 12, 3), True),
    ((2, 20, 20, 3), True),
    ((20, 20, 12, 3), False),
    ((20, 20, 3), False),
]


def _block_mean_checks(arr, coarse, s, obs_axis):
    a = 1 if obs_axis else 0
    lead = (slice(None),) * a
    for o in ([0, 1] if obs_axis else [None]):
        for i in range(coarse.shape[a]):
            for j in range(coarse.shape[a + 1]):
                blk = arr[lead + (slice(i * s, (i + 1) * s),
                                  slice(j * s, (j + 1) * s))]
                got = coarse[lead + (i, j)]
                if obs_axis:
                    blk, got = blk[o], got[o]
                np.testing.assert_allclose(got, blk.mean(axis=(0, 1)),
                                           rtol=1e-6)


@pytest.mark.parametrize('shape,obs_axis', _S_COARSEN_CASES)
def test_oracle_s_coarsen_reference_procedure(shape, obs_axis):
    from oracle.transform import spatial_coarsening
    arr = np.arange(int(np.prod(shape))).reshape(shape).astype(float)
    for s in (1, 2, 4, 5):
        coarse = spatial_coarsening(arr, s_enhance=s, obs_axis=obs_axis)
        a = 1 if obs_axis else 0
        assert coarse.shape[a] == shape[a] // s
        assert coarse.shape[a + 1] == shape[a + 1] // s
        _block_mean_checks(arr, coarse, s, obs_axis)


def test_oracle_s_coarsen_errors_reference_procedure():
    """test_s_coarsen_errors: 3, 7 and 40 do not divide a 20 x 20 grid"""
    from oracle.transform import spatial_coarsening
    arr = np.arange(28800).reshape((2, 20, 20, 12, 3))
    for s in (3, 7, 40):
        with pytest.raises(ValueError):
            spatial_coarsening(arr, s_enhance=s)
    with pytest.raises(ValueError):
        spatial_coarsening(np.arange(10), s_enhance=2, obs_axis=False)
    with pytest.raises(ValueError):
        spatial_coarsening(np.zeros((4, 4)), s_enhance=2, obs_axis=True)


def _t_coarsen_expect(arr, t, method):
    win = arr.reshape(arr.shape[:3] + (-1, t, arr.shape[4]))
    return {'average': win.mean(axis=4), 'total': win.sum(axis=4),
            'subsample': win[:, :, :, :, 0]}[method]


def test_oracle_t_coarsen_reference_procedure():
    """test_t_coarsen: (3, 10, 10, 48, 2) arange, t_enhance 4"""
    from oracle.transform import temporal_coarsening
    arr = np.arange(3 * 10 * 10 * 48 * 2).reshape((3, 10, 10, 48, 2))
    arr = arr.astype(float)
    for method in ('average', 'total', 'subsample'):
        out = temporal_coarsening(arr, t_enhance=4, method=method)
        assert out.shape == (3, 10, 10, 12, 2)
        np.testing.assert_allclose(out, _t_coarsen_expect(arr, 4, method))


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 20, 20, 12, 3), (2, 20, 20, 3)])
def test_device_s_coarsen_reference_procedure(shape):
    """arange values reach 28799 < 2**24 and block sums of <= 25 of them stay
    below 2**24 as well, so the fp32 device sums are exact; the division is
    one rounding (rtol 1e-6)"""
    from sup3r_amd.batch_transform import DeviceBatchTransform
    arr = np.arange(int(np.prod(shape))).reshape(shape).astype(np.float32)
    feats = ['u', 'v', 'w']
    for s in (1, 2, 4, 5):
        tr = DeviceBatchTransform(s, 1, feats, [0, 1, 2])
        lr, hr = tr.transform(arr)
        coarse = lr.cpu().numpy()
        assert coarse.shape[1] == 20 // s and coarse.shape[2] == 20 // s
        _block_mean_checks(arr.astype(np.float64), coarse, s, True)
        np.testing.assert_array_equal(hr.cpu().numpy(), arr)
    for s in (3, 7, 40):
        with pytest.raises(ValueError):
            DeviceBatchTransform(s, 1, feats, [0, 1, 2]).transform(arr)


@pytest.mark.gpu
def test_device_t_coarsen_reference_procedure():
    from sup3r_amd.batch_transform import DeviceBatchTransform
    arr = np.arange(3 * 10 * 10 * 48 * 2).reshape((3, 10, 10, 48, 2))
    arr = arr.astype(np.float32)
    for method in ('average', 'total', 'subsample'):
        tr = DeviceBatchTransform(1, 4, ['u', 'v'], [0, 1])
        lr, _ = tr.transform(arr, temporal_coarsening_method=method)
        assert tuple(lr.shape) == (3, 10, 10, 12, 2)
        np.testing.assert_allclose(
            lr.cpu().numpy(), _t_coarsen_expect(arr.astype(np.float64), 4,
                                                method), rtol=1e-6)
