"""Low-res batch derivation on the MI355X.

``DeviceBatchTransform.transform`` mirrors ``SingleBatchQueue.transform``
(sup3r/preprocessing/batch_queues/base.py:32-87): spatial block-mean
coarsening (sup3r/utilities/utilities.py:406-523), temporal coarsening
(:345-403), gaussian smoothing (batch_queues/utilities.py:57-103) and the
hi-res feature selection — on the device, so a training batch never visits the
host between the sampler and ``Sup3rGan._train_batch``.
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import Device


def gaussian_taps(sigma, truncate=4.0):
    """The normalised 1-D kernel scipy.ndimage.gaussian_filter builds
    (``_gaussian_kernel1d``, order 0): radius = int(truncate * sigma + 0.5)."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return (phi / phi.sum()), radius


class DeviceBatchTransform:
    """``transform(samples) -> (low_res, high_res)`` device tensors."""

    def __init__(self, s_enhance, t_enhance, features, hr_features_ind=None,
                 device=None):
        self.s_enhance, self.t_enhance = int(s_enhance), int(t_enhance)
        self.features = list(features)
        self.hr_features_ind = (list(range(len(self.features)))
                                if hr_features_ind is None
                                else list(hr_features_ind))
        self.dev = device or Device.get()

    def _ptr(self, t):
        return C.c_void_p(t.data_ptr())

    def transform(self, samples, smoothing=None, smoothing_ignore=None,
                  temporal_coarsening_method='subsample'):
        L, dev = _lib.lib(), self.dev
        hr = dev.to_device(samples)
        is_5d = hr.dim() == 5
        if hr.dim() not in (4, 5):
            raise ValueError('Data must be 4D or 5D to do spatial coarsening, '
                             f'but received: {tuple(hr.shape)}')
        n, s1, s2 = (int(v) for v in hr.shape[:3])
        t = int(hr.shape[3]) if is_5d else 1
        c = int(hr.shape[-1])
        s = self.s_enhance
        if s1 % s or s2 % s:
            raise ValueError('s_enhance must evenly divide grid size. '
                             f'Received s_enhance: {s} with data shape: '
                             f'{tuple(hr.shape)}')
        te = self.t_enhance if is_5d else 1
        if temporal_coarsening_method not in _lib.TC_METHODS:
            raise KeyError(
                'Did not recognize temporal_coarsening method '
                f'"{temporal_coarsening_method}", can only accept one of: '
                '[subsample, average, total, max, min]')
        ot = t // te if te > 1 else t
        lr_shape = (n, s1 // s, s2 // s) + ((ot,) if is_5d else ()) + (c,)
        lr = dev.empty(lr_shape)
        rc = L.s3_coarsen(dev.ctx, self._ptr(hr), n, s1, s2, t, c, s, te,
                          _lib.TC_METHODS[temporal_coarsening_method],
                          self._ptr(lr))
        _lib.check(rc, dev.ctx, 's3_coarsen')
        if smoothing is not None:
            ignore = smoothing_ignore if smoothing_ignore is not None else []
            mask = 0
            for j in range(c):
                if self.features[j] not in ignore:
                    mask |= 1 << j
            taps, radius = gaussian_taps(smoothing)
            w = np.ascontiguousarray(taps, dtype=np.float32)
            tmp, out = dev.empty(lr_shape), dev.empty(lr_shape)
            rc = L.s3_gaussian_smooth(
                dev.ctx, self._ptr(lr), n, s1 // s, s2 // s, ot, c,
                w.ctypes.data_as(C.POINTER(C.c_float)), radius, mask,
                self._ptr(tmp), self._ptr(out))
            _lib.check(rc, dev.ctx, 's3_gaussian_smooth')
            lr = out
        ind = self.hr_features_ind
        if ind == list(range(c)):
            high_res = hr
        else:
            high_res = dev.empty(tuple(hr.shape[:-1]) + (len(ind),))
            n_pos = hr.numel() // c
            for k, j in enumerate(ind):
                rc = L.s3_copy_channels(dev.ctx, self._ptr(hr), c, j,
                                        self._ptr(high_res), len(ind), k, 1,
                                        n_pos, 0)
                _lib.check(rc, dev.ctx, 's3_copy_channels')
        return lr, high_res
