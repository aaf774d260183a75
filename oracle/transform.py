"""CPU restatement of SingleBatchQueue.transform — TEST INFRASTRUCTURE ONLY
(imported by tests/ only; see oracle/__init__.py).

Follows sup3r/preprocessing/batch_queues/base.py:32-87, with
spatial_coarsening (sup3r/utilities/utilities.py:406-523), temporal_coarsening
(:345-403) and smooth_data (batch_queues/utilities.py:57-103).  The gaussian
filter is the reference's own dependency, scipy.ndimage.gaussian_filter, called
exactly as the reference calls it — this part of the oracle is pinned.
"""
import numpy as np
from scipy.ndimage import gaussian_filter


def spatial_coarsening(data, s_enhance=2, obs_axis=True):
    """utilities.py:406-523 — block mean over the two spatial axes; these are
    axes (1, 2) with an observation axis and (0, 1) without one"""
    nd = len(data.shape)
    if nd < 2 or (obs_axis and nd < 3):
        raise ValueError('Data must be 2D-5D (3D-5D with obs_axis) to do '
                         f'spatial coarsening, but received: {data.shape}')
    if s_enhance is None or s_enhance <= 1:
        return data
    a = 1 if obs_axis else 0
    if data.shape[a] % s_enhance or data.shape[a + 1] % s_enhance:
        raise ValueError('s_enhance must evenly divide grid size.')
    data = np.reshape(data, (*data.shape[:a], data.shape[a] // s_enhance,
                             s_enhance, data.shape[a + 1] // s_enhance,
                             s_enhance, *data.shape[a + 2:]))
    return data.sum(axis=(a + 1, a + 3)) / s_enhance ** 2


def temporal_coarsening(data, t_enhance=4, method='subsample'):
    if t_enhance is None or len(data.shape) != 5:
        return data
    new_shape = (data.shape[0], data.shape[1], data.shape[2], -1, t_enhance,
                 data.shape[4])
    if method == 'subsample':
        return data[:, :, :, ::t_enhance, :]
    if method == 'average':
        return np.nansum(np.reshape(data, new_shape), axis=4) / t_enhance
    if method == 'max':
        return np.max(np.reshape(data, new_shape), axis=4)
    if method == 'min':
        return np.min(np.reshape(data, new_shape), axis=4)
    if method == 'total':
        return np.nansum(np.reshape(data, new_shape), axis=4)
    raise KeyError(method)


def smooth_data(low_res, training_features, smoothing_ignore, smoothing=None):
    if smoothing is None:
        return low_res
    low_res = np.array(low_res, copy=True)
    feat_iter = [j for j in range(low_res.shape[-1])
                 if training_features[j] not in smoothing_ignore]
    for i in range(low_res.shape[0]):
        for j in feat_iter:
            if len(low_res.shape) == 5:
                for t in range(low_res.shape[-2]):
                    low_res[i, ..., t, j] = gaussian_filter(
                        low_res[i, ..., t, j], smoothing, mode='nearest')
            else:
                low_res[i, ..., j] = gaussian_filter(
                    low_res[i, ..., j], smoothing, mode='nearest')
    return low_res


def transform(samples, s_enhance, t_enhance, features, hr_features_ind,
              smoothing=None, smoothing_ignore=None,
              temporal_coarsening_method='subsample'):
    low_res = spatial_coarsening(samples, s_enhance)
    if t_enhance != 1:
        low_res = temporal_coarsening(low_res, t_enhance,
                                      temporal_coarsening_method)
    low_res = smooth_data(low_res, features, smoothing_ignore or [], smoothing)
    return low_res, samples[..., hr_features_ind]
