"""Output epilogue: oracle properties on CPU and the device
kernels against the oracle on GPU.  Tolerance: fp32 trig (1e-4 m/s on the
speed, 1e-2 degree on the direction away from calm, wrap-aware)."""
import warnings

import numpy as np
import pytest


def _lat_lon(s1, s2, rng, ascending=False):
    lat = np.linspace(41.0, 39.0, s1)[:, None] + 0.03 * rng.standard_normal((s1, s2))
    lon = np.linspace(-105.0, -103.0, s2)[None, :] + 0.03 * rng.standard_normal((s1, s2))
    ll = np.stack([lat + 0 * lon, lon + 0 * lat], -1)
    return ll[::-1].copy() if ascending else ll


@pytest.mark.parametrize('ascending', [False, True])
def test_oracle_uv_roundtrip(ascending):
    """invert_uv undoes transform_rotate_wind (the pair the reference's
    derivers / writers apply on the way in and out)."""
    from oracle.output import invert_uv, transform_rotate_wind
    rng = np.random.default_rng(0)
    ll = _lat_lon(9, 7, rng, ascending)
    ws = rng.uniform(0.5, 30, (9, 7, 5))
    wd = rng.uniform(0, 360, (9, 7, 5))
    u, v = transform_rotate_wind(ws, wd, ll)
    ws2, wd2 = invert_uv(u, v, ll)
    np.testing.assert_allclose(ws2, ws, atol=1e-10)
    d = np.abs(wd2 - wd)
    assert np.minimum(d, 360 - d).max() < 1e-9


def test_oracle_limits_and_renaming():
    from oracle.output import transform_output
    rng = np.random.default_rng(1)
    data = rng.standard_normal((4, 5, 3, 3)) * 100
    feats = ['u_100m', 'v_100m', 'temperature_2m']
    out, names = transform_output(data, feats, _lat_lon(4, 5, rng), True)
    assert names == ['windspeed_100m', 'winddirection_100m', 'temperature_2m']
    assert out.dtype == np.float32
    assert out[..., 0].min() >= 0 and out[..., 0].max() <= 120
    assert out[..., 1].min() >= 0 and out[..., 1].max() <= 360
    assert out[..., 2].max() <= 100 and out[..., 2].min() >= -200
    with pytest.raises(KeyError):
        transform_output(data, ['foo', 'bar', 'baz'], _lat_lon(4, 5, rng))


@pytest.mark.gpu
@pytest.mark.parametrize('ascending', [False, True])
def test_device_output_transform_vs_oracle(ascending):
    from oracle.output import transform_output
    from sup3r_amd.output_transform import DeviceOutputTransform
    rng = np.random.default_rng(2)
    s1, s2, t = 20, 17, 24
    ll = _lat_lon(s1, s2, rng, ascending)
    feats = ['u_10m', 'v_10m', 'temperature_2m', 'u_100m', 'v_100m']
    data = (rng.standard_normal((s1, s2, t, 5)) * np.array(
        [8, 8, 90, 60, 60])).astype(np.float32)
    ref, names_ref = transform_output(data.astype(np.float64), feats, ll, True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out, names = DeviceOutputTransform().transform_output(
            data, feats, ll, invert_uv=True)
    assert names == names_ref
    assert any('temperature_2m' in str(x.message) for x in w)
    out = out.cpu().numpy()
    for i, n in enumerate(names):
        if n.startswith('winddirection'):
            d = np.abs(out[..., i] - ref[..., i])
            d = np.minimum(d, 360 - d)
            calm = ref[..., i - 1] < 0.05          # direction of a calm is noise
            assert d[~calm].max() < 1e-2
        else:
            np.testing.assert_allclose(out[..., i], ref[..., i], rtol=0,
                                       atol=2e-4)
    # without the inversion: clipping only, bit-exact
    ref2, names2 = transform_output(data.astype(np.float64), feats, ll, False)
    out2, n2 = DeviceOutputTransform().transform_output(data, feats, ll)
    assert n2 == names2 == feats
    np.testing.assert_array_equal(out2.cpu().numpy(), ref2)
    with pytest.raises(KeyError):
        DeviceOutputTransform().transform_output(data, ['a'] * 5, ll)
