"""Forward-pass output epilogue on the MI355X.

``DeviceOutputTransform.transform_output`` mirrors
``OutputHandler._transform_output`` (sup3r/writers/base.py:304-345): the
optional u/v -> windspeed / winddirection inversion
(``invert_uv_features`` :233-302, ``invert_uv`` in
sup3r/preprocessing/derivers/utilities.py:204-258) and ``enforce_limits``
with clipping (sup3r/utilities/utilities.py:155-220) run on the hi-res chunk
while it is still on the device, so only final values cross PCIe.  The grid
angle theta(s1, s2) is computed on the host from ``lat_lon`` exactly as the
reference does (a small 2-D array) and uploaded as cos / sin tables.
``nn_fill=True`` (utilities.py:208-215 + ``nn_fill_array`` :55-75): the
out-of-range / NaN mask of a feature is built on the device, the index map of
the nearest valid cell comes from the reference's own
``scipy.ndimage.distance_transform_edt(mask, return_indices=True)`` call on
that boolean mask (host, only for features that leave their range), and the
refill is a device gather — the field itself never leaves the device.
"""
import ctypes as C
import logging
import re
from warnings import warn

import numpy as np

from . import _lib
from .engine import Device

logger = logging.getLogger(__name__)

# min / max columns of sup3r/utilities/output_attrs.json (data, not code)
OUTPUT_LIMITS = {
    'u': (-120, 120), 'v': (-120, 120), 'windspeed': (0, 120),
    'winddirection': (0, 360), 'clearsky_ratio': (0, 1), 'dhi': (0, 1350),
    'dni': (0, 1350), 'ghi': (0, 1350), 'rsds': (0, 1350),
    'temperature': (-200, 100), 'temperature_min': (-200, 100),
    'temperature_max': (-200, 100), 'relativehumidity': (0, 100),
    'relativehumidity_min': (0, 100), 'relativehumidity_max': (0, 100),
    'pressure': (0, 150000), 'pr': (0, np.inf), 'srl': (0, np.inf),
}


def get_feature_basename(feature):
    """Feature name without its height / pressure suffix (utilities.py:78-92)."""
    height = re.findall(r'_\d+m', feature)
    press = re.findall(r'_\d+pa', feature)
    if height:
        return feature.replace(height[0], '')
    if press:
        return feature.replace(press[0], '')
    return feature


def get_renamed_features(features):
    """u_*m / v_*m -> windspeed_*m / winddirection_*m (writers/base.py:201-231)."""
    out = list(features)
    for f in features:
        m = re.match(r'u_(.*?)m$', f.lower())
        if m:
            h = m.group(1)
            out[features.index(f'u_{h}m')] = f'windspeed_{h}m'
            out[features.index(f'v_{h}m')] = f'winddirection_{h}m'
    return out


def grid_theta(lat_lon):
    """theta(s1, s2) of invert_uv (derivers/utilities.py:230-244) in the
    caller's row order (the reference flips rows, rotates, flips back)."""
    lat_lon = np.asarray(lat_lon)
    flip = lat_lon[-1, 0, 0] > lat_lon[0, 0, 0]
    if flip:
        lat_lon = lat_lon[::-1]
    dy = lat_lon[:, :, 0] - np.roll(lat_lon[:, :, 0], 1, axis=0)
    dx = lat_lon[:, :, 1] - np.roll(lat_lon[:, :, 1], 1, axis=0)
    dy = (dy + 90) % 180 - 90
    dx = (dx + 180) % 360 - 180
    theta = (np.pi / 2) - np.arctan2(dy, dx)
    if len(theta) > 1:
        theta[0] = theta[1]
    return theta[::-1] if flip else theta


class DeviceOutputTransform:
    """``transform_output(data, features, lat_lon, invert_uv)`` on device
    tensors of shape (s1, s2, t, features)."""

    def __init__(self, device=None):
        self.dev = device or Device.get()

    def transform_output(self, data, features, lat_lon, invert_uv=False,
                         nn_fill=False):
        L, dev = _lib.lib(), self.dev
        x = dev.to_device(data)
        if x.dim() != 4:
            raise ValueError('expected (spatial_1, spatial_2, temporal, '
                             f'features), got {tuple(x.shape)}')
        s1, s2, t, c = (int(v) for v in x.shape)
        features = list(features)
        assert len(features) == c, (features, c)
        if invert_uv and any(re.match(r'[uv]_(.*?)m$', f.lower())
                             for f in features):
            theta = grid_theta(lat_lon)
            cs = dev.to_device(np.cos(theta).astype(np.float32))
            sn = dev.to_device(np.sin(theta).astype(np.float32))
            for f in list(features):
                m = re.match(r'u_(.*?)m$', f.lower())
                if not m:
                    continue
                h = m.group(1)
                ui, vi = features.index(f'u_{h}m'), features.index(f'v_{h}m')
                rc = L.s3_invert_uv(dev.ctx, C.c_void_p(x.data_ptr()),
                                    s1 * s2, t, c, ui, vi,
                                    C.c_void_p(cs.data_ptr()),
                                    C.c_void_p(sn.data_ptr()))
                _lib.check(rc, dev.ctx, 's3_invert_uv')
            features = get_renamed_features(features)
        lo = np.empty(c, np.float32)
        hi = np.empty(c, np.float32)
        for i, fn in enumerate(features):
            name = get_feature_basename(fn)
            if name not in OUTPUT_LIMITS:
                msg = f'Could not find "{name}" in OUTPUT_ATTRS dict!'
                logger.error(msg)
                raise KeyError(msg)
            lo[i], hi[i] = OUTPUT_LIMITS[name]
        # the reference warns when a feature leaves its range (utilities.py:
        # 186-206); the extrema come from the device reduction
        st = dev.empty((1, 64, c, 3))
        rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(x.data_ptr()), 1,
                              s1 * s2 * t, c, C.c_void_p(st.data_ptr()))
        _lib.check(rc, dev.ctx, 's3_chunk_stats')
        sth = st.cpu().numpy()[0]
        for i, fn in enumerate(features):
            f_min, f_max = sth[:, i, 0].min(), sth[:, i, 1].max()
            bad = f_max > hi[i] or f_min < lo[i]
            if bad:
                msg = (f'{fn} has a range of ({f_min}, {f_max}) outside '
                       f'({lo[i]}, {hi[i]}). Enforcing the range with '
                       + ('nearest neighbor interpolation.' if nn_fill
                          else 'clipping.'))
                logger.warning(msg)
                warn(msg)
            # NaNs are excluded from the device extrema (counted in slot 2):
            # the reference fills them whatever the range (utilities.py:
            # 208-215 -> nn_fill_array)
            n_nan = float(sth[:, i, 2].sum())
            if nn_fill and (bad or n_nan > 0
                            or not np.isfinite([f_min, f_max]).all()):
                self._nn_fill_channel(x, i, float(lo[i]), float(hi[i]))
        if nn_fill:
            return x, features
        pf = C.POINTER(C.c_float)
        rc = L.s3_clip_channels(dev.ctx, C.c_void_p(x.data_ptr()), c,
                                s1 * s2 * t, lo.ctypes.data_as(pf),
                                hi.ctypes.data_as(pf))
        _lib.check(rc, dev.ctx, 's3_clip_channels')
        return x, features


    def _nn_fill_channel(self, x, ch, lo, hi):
        """``data[..., ch] = nn_fill_array(where(out of range, nan, data[..., ch]))``
        (utilities.py:208-215, :55-75) on the device tensor ``x``."""
        import torch
        from scipy import ndimage as nd
        L, dev = _lib.lib(), self.dev
        s1, s2, t, c = (int(v) for v in x.shape)
        n_pos = s1 * s2 * t
        mask = torch.empty(n_pos, dtype=torch.uint8, device=x.device)
        rc = L.s3_range_mask(dev.ctx, C.c_void_p(x.data_ptr()), c, ch, n_pos,
                             lo, hi, C.c_void_p(mask.data_ptr()))
        _lib.check(rc, dev.ctx, 's3_range_mask')
        m = mask.cpu().numpy().astype(bool).reshape(s1, s2, t)
        if not m.any():
            return
        if m.all():
            raise ValueError('nn_fill: no value of the feature is inside its '
                             'range')
        # the reference's own call (utilities.py:70-72) on the boolean mask
        idx = nd.distance_transform_edt(m, return_distances=False,
                                        return_indices=True)
        flat = np.ravel_multi_index(tuple(idx), (s1, s2, t)).astype(np.int32)
        src = torch.from_numpy(np.ascontiguousarray(flat.ravel())).to(x.device)
        rc = L.s3_fill_indexed(dev.ctx, C.c_void_p(x.data_ptr()), c, ch, n_pos,
                               C.c_void_p(mask.data_ptr()),
                               C.c_void_p(src.data_ptr()))
        _lib.check(rc, dev.ctx, 's3_fill_indexed')
