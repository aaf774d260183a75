"""CPU restatement of the forward-pass output epilogue — TEST INFRASTRUCTURE
ONLY (imported by tests/ only; see oracle/__init__.py).

Follows OutputHandler._transform_output (sup3r/writers/base.py:304-345):
invert_uv_features / invert_uv_single_pair (:233-302) with invert_uv and
transform_rotate_wind of sup3r/preprocessing/derivers/utilities.py:146-258,
then enforce_limits(nn_fill=False) of sup3r/utilities/utilities.py:155-220 with
the min / max columns of sup3r/utilities/output_attrs.json.
"""
import re

import numpy as np

# min / max of sup3r/utilities/output_attrs.json (data, not code)
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
    """utilities.py:78-92"""
    height = re.findall(r'_\d+m', feature)
    press = re.findall(r'_\d+pa', feature)
    if height:
        return feature.replace(height[0], '')
    if press:
        return feature.replace(press[0], '')
    return feature


def grid_theta(lat_lon):
    """angle of the grid's s1 axis from the meridian (derivers/utilities.py:
    236-244), on the possibly flipped grid"""
    dy = lat_lon[:, :, 0] - np.roll(lat_lon[:, :, 0], 1, axis=0)
    dx = lat_lon[:, :, 1] - np.roll(lat_lon[:, :, 1], 1, axis=0)
    dy = (dy + 90) % 180 - 90
    dx = (dx + 180) % 360 - 180
    theta = (np.pi / 2) - np.arctan2(dy, dx)
    if len(theta) > 1:
        theta[0] = theta[1]
    return theta


def transform_rotate_wind(ws, wd, lat_lon):
    """derivers/utilities.py:146-201"""
    invert_lat = False
    if lat_lon[-1, 0, 0] > lat_lon[0, 0, 0]:
        invert_lat = True
        lat_lon, ws, wd = lat_lon[::-1], ws[::-1], wd[::-1]
    theta = grid_theta(lat_lon)
    wd = np.radians(wd)
    u_rot = np.cos(theta)[:, :, np.newaxis] * ws * np.sin(wd)
    u_rot += np.sin(theta)[:, :, np.newaxis] * ws * np.cos(wd)
    v_rot = -np.sin(theta)[:, :, np.newaxis] * ws * np.sin(wd)
    v_rot += np.cos(theta)[:, :, np.newaxis] * ws * np.cos(wd)
    if invert_lat:
        u_rot, v_rot = u_rot[::-1], v_rot[::-1]
    return u_rot, v_rot


def invert_uv(u, v, lat_lon):
    """derivers/utilities.py:204-258"""
    invert_lat = False
    if lat_lon[-1, 0, 0] > lat_lon[0, 0, 0]:
        invert_lat = True
        lat_lon, u, v = lat_lon[::-1], u[::-1], v[::-1]
    theta = grid_theta(lat_lon)
    u_rot = np.cos(theta)[:, :, np.newaxis] * u
    u_rot -= np.sin(theta)[:, :, np.newaxis] * v
    v_rot = np.sin(theta)[:, :, np.newaxis] * u
    v_rot += np.cos(theta)[:, :, np.newaxis] * v
    ws = np.hypot(u_rot, v_rot)
    wd = (np.degrees(np.arctan2(u_rot, v_rot)) + 360) % 360
    if invert_lat:
        ws, wd = ws[::-1], wd[::-1]
    return ws, wd


def get_renamed_features(features):
    """writers/base.py:201-231"""
    out = list(features)
    for f in features:
        m = re.match(r'u_(.*?)m$', f.lower())
        if m:
            h = m.group(1)
            out[features.index(f'u_{h}m')] = f'windspeed_{h}m'
            out[features.index(f'v_{h}m')] = f'winddirection_{h}m'
    return out


def nn_fill_array(array):
    """utilities.py:55-75: NaNs take the value of the nearest non-NaN cell
    (scipy's Euclidean distance transform over ALL axes of the array)."""
    from scipy import ndimage as nd
    nan_mask = np.isnan(array)
    indices = nd.distance_transform_edt(nan_mask, return_distances=False,
                                        return_indices=True)
    return array[tuple(indices)]


def enforce_limits(features, data, nn_fill=False):
    """utilities.py:155-220"""
    data = np.array(data, copy=True)
    for fidx, fn in enumerate(features):
        name = get_feature_basename(fn)
        if name not in OUTPUT_LIMITS:
            raise KeyError(f'Could not find "{name}" in OUTPUT_ATTRS dict!')
        lo, hi = OUTPUT_LIMITS[name]
        if nn_fill:
            data[..., fidx] = np.where(data[..., fidx] > hi, np.nan,
                                       data[..., fidx])
            data[..., fidx] = np.where(data[..., fidx] < lo, np.nan,
                                       data[..., fidx])
            data[..., fidx] = nn_fill_array(data[..., fidx])
        else:
            data[..., fidx] = np.maximum(data[..., fidx], lo)
            data[..., fidx] = np.minimum(data[..., fidx], hi)
    return data.astype(np.float32)


def transform_output(data, features, lat_lon, invert_uv_flag=False,
                     nn_fill=False):
    """writers/base.py:304-345; returns (data, features)"""
    data = np.array(data, copy=True)
    features = list(features)
    if invert_uv_flag and any(re.match(r'[uv]_(.*?)m$', f.lower())
                              for f in features):
        for f in list(features):
            m = re.match(r'u_(.*?)m$', f.lower())
            if m:
                h = m.group(1)
                ui, vi = features.index(f'u_{h}m'), features.index(f'v_{h}m')
                ws, wd = invert_uv(data[..., ui], data[..., vi], lat_lon)
                data[..., ui], data[..., vi] = ws, wd
        features = get_renamed_features(features)
    return enforce_limits(features, data, nn_fill=nn_fill), features
