"""GPU tests (``-m gpu``) of the ForwardPass executors: the chunked result must
equal the un-chunked generator where the receptive field allows
(test_forward_pass.py:411-558 of the reference, re-run on this generator) and
the device-pipelined ``run_batched`` must reproduce the chunk-by-chunk ``run``
bit for bit — ragged edge chunks, halo padding, normalisation, rank sharding."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _model():
    from sup3r_amd import Sup3rGan
    feats = ['u_10m', 'v_10m']
    Sup3rGan.seed(5)
    means = {f: np.float32(0.2 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.25 + 0.5 * i) for i, f in enumerate(feats)}
    m = Sup3rGan(os.path.join(CFG, 'test_gen_st_2x_4x_2f.json'),
                 os.path.join(CFG, 'test_disc_st_same.json'), means=means,
                 stdevs=stds)
    m.set_model_params(lr_features=feats, hr_out_features=feats, s_enhance=2,
                       t_enhance=4)
    m.init_weights((1, 8, 8, 6, 2), (1, 16, 16, 24, 2))
    return m


@pytest.mark.parametrize('batch', [1, 3, 8])
def test_run_batched_equals_run(batch):
    from sup3r_amd import ChunkSlicer, ForwardPass
    model = _model()
    rng = np.random.default_rng(2)
    domain = (rng.standard_normal((14, 11, 13, 2)) * 2 + 0.5).astype(np.float32)
    slicer = ChunkSlicer((14, 11), 13, 2, 4, (6, 5, 6), spatial_pad=2,
                         temporal_pad=2)
    fwp = ForwardPass(model, slicer)
    ref = np.zeros(slicer.hr_shape + (2,), np.float32)
    got = np.full(slicer.hr_shape + (2,), np.nan, np.float32)
    n_ref = fwp.run_chunks(domain, out=ref)
    n = fwp.run_batched(domain, out=got, batch=batch)
    assert n == n_ref == slicer.n_chunks
    np.testing.assert_array_equal(got, ref)
    # rank sharding: two ranks fill disjoint windows of one output
    both = np.full_like(ref, np.nan)
    for r in range(2):
        ForwardPass(model, slicer, rank=r, nranks=2).run_batched(
            domain, out=both, batch=batch)
    np.testing.assert_array_equal(both, ref)
    # writer callback path
    seen = {}
    fwp.run_batched(domain, writer=lambda i, s_, d: seen.__setitem__(i, (s_, d)),
                    batch=batch)
    assert sorted(seen) == list(range(slicer.n_chunks))
    for i, (s_, d) in seen.items():
        np.testing.assert_array_equal(d, ref[s_])


def test_run_batched_errors():
    from sup3r_amd import ChunkSlicer, ForwardPass
    model = _model()
    slicer = ChunkSlicer((8, 8), 8, 2, 4, (4, 4, 4), spatial_pad=1,
                         temporal_pad=1)
    fwp = ForwardPass(model, slicer)
    domain = np.random.default_rng(0).standard_normal((8, 8, 8, 2)).astype(
        np.float32)
    bad = domain.copy()
    bad[6, 6, 6, 0] = np.nan
    with pytest.raises(ValueError, match='NaN'):
        fwp.run_batched(bad, out=np.zeros(slicer.hr_shape + (2,), np.float32))
    # a dead generator (all-zero weights) gives constant output channels ->
    # MemoryError from the output check, in both executors
    model.generator.set_weights(
        [np.zeros_like(w) for w in model.generator.weights])
    for runner in (fwp.run_chunks, fwp.run_batched, fwp.run):
        with pytest.raises(MemoryError):
            runner(domain, out=np.zeros(slicer.hr_shape + (2,), np.float32))


def test_chunk_stats_kernel():
    """s3_chunk_stats against numpy: min / max / NaN count per chunk and
    channel after folding the 64 slabs."""
    import ctypes as C
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev, L = Device.get(), _lib.lib()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 5000, 2)).astype(np.float32)
    x[1, :, 1] = 0.75                   # constant channel
    x[2, 17, 0] = np.nan
    xd = dev.to_device(x)
    st = dev.empty((3, 64, 2, 3))
    rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(xd.data_ptr()), 3, 5000, 2,
                          C.c_void_p(st.data_ptr()))
    _lib.check(rc, dev.ctx, 's3_chunk_stats')
    s_ = st.cpu().numpy()
    mn, mx, nn = s_[..., 0].min(1), s_[..., 1].max(1), s_[..., 2].sum(1)
    np.testing.assert_array_equal(mn, np.nanmin(x, axis=1))
    np.testing.assert_array_equal(mx, np.nanmax(x, axis=1))
    np.testing.assert_array_equal(nn, np.isnan(x).sum(1))
    assert mn[1, 1] == mx[1, 1] == np.float32(0.75)
