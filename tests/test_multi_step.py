"""GPU test (``-m gpu``) of ``MultiStepGan``: the chain
semantics of sup3r/models/multi_step.py:128-276 as the reference's own
tests/forward_pass + tests/training use it — a spatial-only step feeding a
spatiotemporal step (4-D -> 5-D transposition), per-step normalisation,
feature matching, save -> ``MultiStepGan.load`` round trip."""
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _model(gen, disc, feats, s, t, seed, in_shape):
    from sup3r_amd import Sup3rGan
    Sup3rGan.seed(seed)
    # np.float32 statistics, as the batch handlers supply them
    # (abstract.py:160-161) and as load_saved_params restores them
    means = {f: np.float32(0.3 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.5 + 0.25 * i) for i, f in enumerate(feats)}
    m = Sup3rGan(os.path.join(CFG, gen), os.path.join(CFG, disc),
                 means=means, stdevs=stds)
    m.set_model_params(lr_features=list(feats), hr_out_features=list(feats),
                       s_enhance=s, t_enhance=t)
    hr = (in_shape[0],) + tuple(
        d * (s if i < 2 else t) for i, d in enumerate(in_shape[1:-1])) + (
        len(feats),)
    m.init_weights(in_shape, hr)
    return m


def test_multi_step_chain_matches_manual_and_roundtrips():
    from sup3r_amd import MultiStepGan
    rng = np.random.default_rng(3)
    feats = ['u_10m', 'v_10m']
    m_s = _model('test_gen_s_2x_2f.json', 'test_disc_s_same.json', feats, 2, 1,
                 1, (4, 6, 5, 2))
    m_st = _model('test_gen_st_2x_4x_2f.json', 'test_disc_st_same.json', feats,
                  2, 4, 2, (1, 12, 10, 4, 2))
    ms = MultiStepGan([m_s, m_st])
    assert len(ms) == 2 and ms.s_enhance == 4 and ms.t_enhance == 4
    assert ms.lr_features == feats and ms.hr_out_features == feats
    x = (rng.standard_normal((4, 6, 5, 2)) * 2 + 1).astype(np.float32)
    y = ms.generate(x)
    assert y.shape == (1, 24, 20, 16, 2)
    # manual chain: spatial step on (t, s1, s2, f), then (1, s1, s2, t, f)
    y1 = m_s.generate(x)
    y1 = np.transpose(y1, (1, 2, 0, 3))[np.newaxis]
    y2 = m_st.generate(y1)
    np.testing.assert_array_equal(y, y2)
    # norm flags: first step un-normalised input, last step normalised output
    y_nn = ms.generate(x, norm_in=False, un_norm_out=False)
    y1n = np.transpose(m_s.generate(x, norm_in=False), (1, 2, 0, 3))[None]
    np.testing.assert_array_equal(
        y_nn, m_st.generate(y1n, un_norm_out=False))
    with tempfile.TemporaryDirectory() as td:
        dirs = [os.path.join(td, 'a'), os.path.join(td, 'b')]
        m_s.save(dirs[0])
        m_st.save(dirs[1])
        loaded = MultiStepGan.load(dirs)
        np.testing.assert_array_equal(loaded.generate(x), y)
    # feature mismatch between steps is an error (multi_step.py:183-191)
    m_bad = _model('test_gen_st_2x_4x_2f.json', 'test_disc_st_same.json',
                   ['a', 'b'], 2, 4, 2, (1, 12, 10, 4, 2))
    with pytest.raises(RuntimeError):
        MultiStepGan([m_s, m_bad]).generate(x)
