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


def _spiked(rng, s1=14, s2=11, t=9):
    """(s1, s2, t, 3) field with isolated and clustered out-of-range values
    and a few NaNs in the temperature / humidity channels"""
    data = np.stack([rng.uniform(-60, 60, (s1, s2, t)),
                     rng.uniform(-150, 90, (s1, s2, t)),
                     rng.uniform(5, 95, (s1, s2, t))], -1).astype(np.float32)
    for ch, bad in ((1, 300.0), (1, -900.0), (2, 180.0), (2, -3.0)):
        idx = rng.integers(0, [s1, s2, t], size=(12, 3))
        data[idx[:, 0], idx[:, 1], idx[:, 2], ch] = bad
    data[2:5, 3:6, 4:6, 1] = 555.0          # a cluster: nearest valid cell is 2+ away
    data[0, 0, 0, 2] = np.nan
    return data


def test_oracle_nn_fill_takes_the_nearest_valid_value():
    """enforce_limits(nn_fill=True) (utilities.py:208-215): every out-of-range
    / NaN value is replaced by an in-range value of the same feature whose
    index distance is minimal; in-range values are untouched"""
    from oracle.output import enforce_limits
    rng = np.random.default_rng(5)
    data = _spiked(rng)
    feats = ['u_10m', 'temperature_2m', 'relativehumidity_2m']
    out = enforce_limits(feats, data.astype(np.float64), nn_fill=True)
    lims = [(-120, 120), (-200, 100), (0, 100)]
    for ch, (lo, hi) in enumerate(lims):
        v = data[..., ch]
        ok = (v >= lo) & (v <= hi)
        np.testing.assert_array_equal(out[..., ch][ok], v[ok])
        assert np.isfinite(out[..., ch]).all()
        assert out[..., ch].min() >= lo and out[..., ch].max() <= hi
        good = np.argwhere(ok)
        for p in np.argwhere(~ok)[:40]:
            d2 = ((good - p) ** 2).sum(1)
            nearest_vals = v[tuple(good[d2 == d2.min()].T)]
            assert out[tuple(p) + (ch,)] in nearest_vals.astype(np.float32)


@pytest.mark.gpu
def test_device_nn_fill_vs_oracle():
    """transform_output(nn_fill=True): device mask + the reference's own EDT
    call on it + device gather = the oracle's enforce_limits(nn_fill=True),
    bit for bit (no arithmetic is involved)"""
    from oracle.output import transform_output
    from sup3r_amd.output_transform import DeviceOutputTransform
    rng = np.random.default_rng(5)
    data = _spiked(rng, 20, 17, 24)
    feats = ['u_10m', 'temperature_2m', 'relativehumidity_2m']
    ll = _lat_lon(20, 17, rng)
    ref, names_ref = transform_output(data.astype(np.float64), feats, ll, False,
                                      nn_fill=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out, names = DeviceOutputTransform().transform_output(
            data, feats, ll, invert_uv=False, nn_fill=True)
    assert names == names_ref == feats
    assert any('nearest neighbor' in str(x.message) for x in w)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # nothing out of range: nothing changes
    clean = np.clip(data, [-100, -150, 5], [100, 90, 95])
    clean[0, 0, 0, 2] = 50.0
    out2, _ = DeviceOutputTransform().transform_output(clean, feats, ll,
                                                       nn_fill=True)
    np.testing.assert_array_equal(out2.cpu().numpy(), clean)
    # NaNs in a channel that is otherwise INSIDE its range are filled as well
    # (nn_fill_array runs for every feature, utilities.py:208-215): the
    # device extrema skip NaNs, so the range alone would not trigger it
    holes = clean.copy()
    holes[3, 4, 5, 1] = np.nan
    holes[10:12, 7, 2, 1] = np.nan
    ref3, _ = transform_output(holes.astype(np.float64), feats, ll, False,
                               nn_fill=True)
    out3, _ = DeviceOutputTransform().transform_output(holes, feats, ll,
                                                       nn_fill=True)
    assert np.isfinite(out3.cpu().numpy()).all()
    np.testing.assert_array_equal(out3.cpu().numpy(), ref3)


# --- the reference's own known-answer tests for the wind rotation pair
# (/root/reference/tests/utilities/test_utilities.py:360-452 test_transform_rotate,
# /root/reference/tests/output/test_output_handling.py:60-91 test_invert_uv),
# run on the oracle (CPU) and the device kernel (GPU).

def _ref_lat_lon():
    lats = np.array([[1, 1, 1], [0, 0, 0]])
    lons = np.array([[-120, -100, -80], [-120, -100, -80]])
    return np.stack([lats, lons], axis=-1).astype(np.float64)


_KNOWN_UV = [(0, 0, -1), (90, -1, 0), (270, 1, 0), (180, 0, 1),
             (45, -1 / np.sqrt(2), -1 / np.sqrt(2))]


@pytest.mark.parametrize('wd,u_t,v_t', _KNOWN_UV)
def test_oracle_transform_rotate_known_answers(wd, u_t, v_t):
    from oracle.output import invert_uv, transform_rotate_wind
    ll = _ref_lat_lon()
    ws = np.ones((2, 3, 1), np.float32)
    u, v = transform_rotate_wind(ws, np.full((2, 3, 1), wd, np.float32), ll)
    assert np.allclose(u, u_t, atol=1e-5) and np.allclose(v, v_t, atol=1e-5)
    ws2, wd2 = invert_uv(u, v, ll)
    assert np.allclose(ws2, 1, atol=1e-5)
    assert np.allclose(wd2 % 360, wd, atol=1e-3)


@pytest.mark.parametrize('flip', [False, True])
def test_oracle_invert_uv_reference_procedure(flip):
    from oracle.output import invert_uv, transform_rotate_wind
    rng = np.random.default_rng(3)
    ll = _ref_lat_lon()[::-1] if flip else _ref_lat_lon()
    ws = rng.random((2, 3, 5))
    wd = 360 * rng.random((2, 3, 5))
    u, v = transform_rotate_wind(ws.astype(np.float32), wd.astype(np.float32),
                                 ll)
    ws2, wd2 = invert_uv(u, v, ll)
    assert np.allclose(ws, ws2) and np.allclose(wd, wd2)


@pytest.mark.gpu
@pytest.mark.parametrize('flip', [False, True])
def test_device_invert_uv_known_answers(flip):
    """device inversion of the reference's known (u, v) pairs gives speed 1
    and the stated direction; random pairs round-trip (test_invert_uv)"""
    from oracle.output import transform_rotate_wind
    from sup3r_amd.output_transform import DeviceOutputTransform
    ll = _ref_lat_lon()[::-1].copy() if flip else _ref_lat_lon()
    feats = ['u_100m', 'v_100m']
    tr = DeviceOutputTransform()
    for wd, _, _ in _KNOWN_UV:
        u, v = transform_rotate_wind(np.ones((2, 3, 1)),
                                     np.full((2, 3, 1), float(wd)), ll)
        data = np.stack([u, v], axis=-1).astype(np.float32)
        out, names = tr.transform_output(data, feats, ll, invert_uv=True)
        out = out.cpu().numpy()
        assert names == ['windspeed_100m', 'winddirection_100m']
        assert np.allclose(out[..., 0], 1, atol=1e-5)
        d = np.abs(out[..., 1] - wd)
        assert np.minimum(d, 360 - d).max() < 1e-2
    rng = np.random.default_rng(4)
    ws = rng.random((2, 3, 5)) + 0.05
    wd = 360 * rng.random((2, 3, 5))
    u, v = transform_rotate_wind(ws, wd, ll)
    data = np.stack([u, v], axis=-1).astype(np.float32)
    out = tr.transform_output(data, feats, ll, invert_uv=True)[0].cpu().numpy()
    assert np.allclose(out[..., 0], ws, atol=1e-5)
    d = np.abs(out[..., 1] - wd)
    assert np.minimum(d, 360 - d).max() < 1e-2
