"""GPU tests (``-m gpu``) of the ForwardPass executors: the chunked result must
equal the un-chunked generator where the receptive field allows
(test_forward_pass.py:411-558 of the reference, re-run on this generator) and
the device-pipelined ``run_batched`` must reproduce the chunk-by-chunk ``run``
bit for bit — ragged edge chunks, halo padding, normalisation, rank sharding."""
import json
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
    for runner in (fwp.run_chunks, fwp.run_batched, fwp.run_domain):
        with pytest.raises(MemoryError):
            runner(domain, out=np.zeros(slicer.hr_shape + (2,), np.float32))


@pytest.mark.parametrize('c,npos', [(2, 5000), (3, 5000), (4, 70001), (2, 4999)])
def test_chunk_stats_kernel(c, npos):
    """s3_chunk_stats against numpy: min / max / NaN count per chunk and
    channel after folding the 64 slabs — the float4 walk (c | 1024, 16-byte
    chunks) and the plain per-channel kernel (c = 3; odd element counts)."""
    import ctypes as C
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev, L = Device.get(), _lib.lib()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, npos, c)).astype(np.float32)
    x[1, :, 1] = 0.75                   # constant channel
    x[2, 17, 0] = np.nan
    x[2, npos - 1, c - 1] = np.nan
    xd = dev.to_device(x)
    st = dev.empty((3, 64, c, 3))
    rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(xd.data_ptr()), 3, npos, c,
                          C.c_void_p(st.data_ptr()))
    _lib.check(rc, dev.ctx, 's3_chunk_stats')
    s_ = st.cpu().numpy()
    mn, mx, nn = s_[..., 0].min(1), s_[..., 1].max(1), s_[..., 2].sum(1)
    np.testing.assert_array_equal(mn, np.nanmin(x, axis=1))
    np.testing.assert_array_equal(mx, np.nanmax(x, axis=1))
    np.testing.assert_array_equal(nn, np.isnan(x).sum(1))
    assert mn[1, 1] == mx[1, 1] == np.float32(0.75)


@pytest.mark.parametrize('affine', [True, False])
def test_fused_chunk_epilogue_equals_the_three_kernels(affine):
    """s3_chunk_epilogue (un-norm + halo crop + output-check statistics in one
    pass) against s3_affine_channels -> s3_copy_block -> s3_chunk_stats: the
    cropped batch bit for bit, the folded statistics exactly — with NaNs, a
    constant channel and a negative zero in the window."""
    import ctypes as C
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev, L = Device.get(), _lib.lib()
    rng = np.random.default_rng(11)
    n, dims, c = 3, (13, 11, 20), 2
    lo, cn = (2, 1, 4), (9, 8, 14)
    y = rng.standard_normal((n,) + dims + (c,)).astype(np.float32)
    y[1, ..., 1] = 0.5
    y[2, 5, 5, 9, 0] = np.nan
    y[0, 3, 3, 6, 1] = -0.0
    scale = np.array([1.75, 0.5], np.float32)
    shift = np.array([0.3, -2.0], np.float32)
    pf, i64x3 = C.POINTER(C.c_float), C.c_int64 * 3
    yd = dev.to_device(y)
    yc = dev.empty((n,) + cn + (c,))
    st = dev.empty((n, 64, c, 3))
    rc = L.s3_chunk_epilogue(
        dev.ctx, C.c_void_p(yd.data_ptr()), n, i64x3(*dims), i64x3(*lo),
        i64x3(*cn), c, scale.ctypes.data_as(pf) if affine else None,
        shift.ctypes.data_as(pf) if affine else None,
        C.c_void_p(yc.data_ptr()), C.c_void_p(st.data_ptr()))
    _lib.check(rc, dev.ctx, 's3_chunk_epilogue')
    got, gs = yc.cpu().numpy(), st.cpu().numpy()
    # the separate kernels
    y2 = dev.to_device(y)
    if affine:
        rc = L.s3_affine_channels(
            dev.ctx, C.c_void_p(y2.data_ptr()), C.c_void_p(y2.data_ptr()), c,
            y2.numel() // c, scale.ctypes.data_as(pf), shift.ctypes.data_as(pf))
        _lib.check(rc, dev.ctx, 's3_affine_channels')
    ref = dev.empty((n,) + cn + (c,))
    for k in range(n):
        src = y2[k].data_ptr() + 4 * c * (
            (lo[0] * dims[1] + lo[1]) * dims[2] + lo[2])
        rc = L.s3_copy_block(
            dev.ctx, C.c_void_p(src), C.c_void_p(ref[k].data_ptr()), cn[0],
            cn[1], cn[2] * c, dims[1] * dims[2] * c, dims[2] * c,
            cn[1] * cn[2] * c, cn[2] * c)
        _lib.check(rc, dev.ctx, 's3_copy_block')
    st2 = dev.empty((n, 64, c, 3))
    rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(ref.data_ptr()), n,
                          int(np.prod(cn)), c, C.c_void_p(st2.data_ptr()))
    _lib.check(rc, dev.ctx, 's3_chunk_stats')
    want, ws = ref.cpu().numpy(), st2.cpu().numpy()
    assert got.tobytes() == want.tobytes()
    for g, w in ((gs, ws),):
        np.testing.assert_array_equal(g[..., 0].min(1), w[..., 0].min(1))
        np.testing.assert_array_equal(g[..., 1].max(1), w[..., 1].max(1))
        np.testing.assert_array_equal(g[..., 2].sum(1), w[..., 2].sum(1))
    assert gs[..., 2].sum() == 1 and \
        gs[1, :, 1, 0].min() == gs[1, :, 1, 1].max()
    # rows that are not 16-byte aligned are refused (the executor then takes
    # the three kernels)
    rc = L.s3_chunk_epilogue(
        dev.ctx, C.c_void_p(yd.data_ptr()), n, i64x3(*dims), i64x3(2, 1, 3),
        i64x3(9, 8, 14), c, None, None, C.c_void_p(yc.data_ptr()),
        C.c_void_p(st.data_ptr()))
    assert rc == -1


def test_sdma_delivery_round_trip():
    """s3_host_alloc + s3_dma_d2h_begin / s3_dma_wait (ROCr SDMA copy): the
    bytes of a device buffer arrive in the pinned host buffer; a buffer ROCr
    does not know is refused."""
    import ctypes as C
    import torch
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev, L = Device.get(), _lib.lib()
    n = 3 * 1024 * 1024 + 5
    x = np.random.default_rng(3).standard_normal(n).astype(np.float32)
    xd = dev.to_device(x)
    torch.cuda.synchronize()
    hp = C.c_void_p()
    _lib.check(L.s3_host_alloc(dev.ctx, n * 4, 0, C.byref(hp)), dev.ctx,
               's3_host_alloc')
    try:
        arr = np.ctypeslib.as_array((C.c_float * n).from_address(hp.value))
        arr[:] = 0
        t = C.c_uint64()
        _lib.check(L.s3_dma_d2h_begin(dev.ctx, C.c_void_p(xd.data_ptr()), hp,
                                      n * 4, C.byref(t)), dev.ctx, 'begin')
        _lib.check(L.s3_dma_wait(dev.ctx, t, 10000), dev.ctx, 'wait')
        np.testing.assert_array_equal(arr, x)
        plain = np.zeros(16, np.float32)
        rc = L.s3_dma_d2h_begin(dev.ctx, C.c_void_p(xd.data_ptr()),
                                plain.ctypes.data_as(C.c_void_p), 64,
                                C.byref(t))
        assert rc < 0
    finally:
        _lib.check(L.s3_host_free(dev.ctx, hp), dev.ctx, 's3_host_free')


def test_delivery_falls_back_to_hipmemcpy_and_the_ring_can_be_released():
    """``options={'sdma_delivery': False}`` (what a refused ROCr copy switches
    every later batch to): the same chunks through hipMemcpyAsync on the copy stream; then the
    pinned rings are given back"""
    from sup3r_amd import ForwardPass
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    model = _model()
    domain = np.random.default_rng(8).standard_normal((12, 12, 16, 2)).astype(
        np.float32)
    register_model('Sup3rGan', {'model_dir': 'fallback-test'}, model)
    st = ArrayStrategy(domain, {'model_dir': 'fallback-test'}, (6, 6, 4),
                       spatial_pad=1, temporal_pad=1, max_nodes=1, model=model)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]

    def run(**options):
        return [np.array(d) for _, failed, d in ForwardPass.iter_chunks(
            (fwp.get_input_chunk(i) for i in ids), model, batch=3,
            options=options)]
    ref = run()
    got = run(sdma_delivery=False)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    assert ForwardPass._delivery
    ForwardPass.release_delivery_buffers()
    assert not ForwardPass._delivery
    again = run()                      # rings come back on demand
    for a, b in zip(again, ref):
        np.testing.assert_array_equal(a, b)


def test_iter_chunks_views_stay_valid_for_two_batches():
    """the delivery ring: a yielded array is still intact after two further
    batches have been yielded (the documented life time)"""
    from sup3r_amd import ForwardPass
    from sup3r_amd.strategy import ArrayStrategy
    model = _model()
    rng = np.random.default_rng(4)
    domain = (rng.standard_normal((12, 12, 40, 2))).astype(np.float32)
    from sup3r_amd.forward_pass import register_model
    register_model('Sup3rGan', {'model_dir': 'ring-test'}, model)
    st = ArrayStrategy(domain, {'model_dir': 'ring-test'}, (6, 6, 4),
                       spatial_pad=1, temporal_pad=1, max_nodes=1, model=model)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]
    assert len(ids) >= 12
    ref = {}
    for i in ids:
        (c, failed, d), = ForwardPass.iter_chunks(
            [fwp.get_input_chunk(i)], model, batch=1)
        ref[i] = np.array(d)
    held = []
    for c, failed, d in ForwardPass.iter_chunks(
            (fwp.get_input_chunk(i) for i in ids), model, batch=2):
        assert not failed
        held.append((c.index, d))
        batch_now = (len(held) - 1) // 2
        for pos, (idx, arr) in enumerate(held):
            if pos // 2 >= batch_now - 2:   # this batch + the two before it
                np.testing.assert_array_equal(arr, ref[idx])


def test_interleaved_generators_do_not_share_a_delivery_ring():
    """two ``iter_chunks`` generators advanced in lock step over the same
    output shape (the rings were class-level, keyed by shape
    only): each keeps the documented two-batch life time, because each owns a
    lane of the ring table; a finished generator gives its lane back"""
    from sup3r_amd import ForwardPass
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    model = _model()
    rng = np.random.default_rng(14)
    doms = [rng.standard_normal((12, 12, 40, 2)).astype(np.float32)
            for _ in range(2)]
    register_model('Sup3rGan', {'model_dir': 'lane-test'}, model)
    fwps, ids = [], None
    for dom in doms:
        st = ArrayStrategy(dom, {'model_dir': 'lane-test'}, (6, 6, 4),
                           spatial_pad=1, temporal_pad=1, max_nodes=1,
                           model=model)
        fwps.append(ForwardPass(st, 0))
        ids = [int(i) for i in st.node_chunks[0]]
    refs = []
    for fwp in fwps:
        refs.append({c.index: np.array(d) for c, _, d in
                     ForwardPass.iter_chunks(
                         (fwp.get_input_chunk(i) for i in ids), model,
                         batch=2)})
    assert not ForwardPass._lanes
    def chunks_of(fwp):           # (binds fwp now, not when the generator runs)
        return (fwp.get_input_chunk(i) for i in ids)
    gens = [ForwardPass.iter_chunks(chunks_of(fwp), model, batch=2)
            for fwp in fwps]
    held = [[], []]
    for step in range(len(ids)):
        for g in range(2):
            c, failed, d = next(gens[g])
            held[g].append((c.index, d))
            now = (len(held[g]) - 1) // 2
            for pos, (idx, arr) in enumerate(held[g]):
                if pos // 2 >= now - 2:
                    np.testing.assert_array_equal(arr, refs[g][idx])
        if step == 0:
            assert ForwardPass._lanes == {0, 1}
    for g in gens:
        assert next(g, None) is None
    assert not ForwardPass._lanes
    ForwardPass.release_delivery_buffers()


def test_chunk_time_first_and_last_kernels_vs_numpy():
    """s3_chunk_time_first / s3_chunk_time_last against the numpy statements
    they replace (forward_pass.py:274-337 transposes, abstract.py:197-275
    norm_input / un_norm_output), bit for bit: fp32 statistics, fp64
    statistics (numpy computes in fp64, the result is rounded to fp32), none"""
    import ctypes as C

    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev, L = Device.get(), _lib.lib()
    rng = np.random.default_rng(33)
    n, h, w, t, c = 3, 7, 5, 6, 3
    x = (rng.standard_normal((n, h, w, t, c)) * 3 + 1).astype(np.float32)
    xd = dev.to_device(x)
    pd = C.POINTER(C.c_double)
    i64x3 = C.c_int64 * 3
    for mode in ('f32', 'f64', 'none'):
        mu = rng.standard_normal(c) + 0.5
        sd = rng.uniform(0.5, 2.0, c)
        if mode == 'f32':
            mu, sd = mu.astype(np.float32), sd.astype(np.float32)
        want = np.concatenate([np.transpose(x[k], (2, 0, 1, 3))
                               for k in range(n)], axis=0)
        if mode != 'none':
            want = ((want.copy() - mu) / sd).astype(np.float32)
        out = dev.empty((n * t, h, w, c))
        m64 = np.ascontiguousarray(mu, np.float64)
        s64 = np.ascontiguousarray(sd, np.float64)
        rc = L.s3_chunk_time_first(
            dev.ctx, C.c_void_p(xd.data_ptr()), n, i64x3(h, w, t), c,
            m64.ctypes.data_as(pd) if mode != 'none' else None,
            s64.ctypes.data_as(pd) if mode != 'none' else None,
            int(mode == 'f32'), C.c_void_p(out.data_ptr()))
        _lib.check(rc, dev.ctx, 's3_chunk_time_first')
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    # the way out: (n t, H, W, c) -> cropped (n, H', W', t', c), un-normalised
    y = (rng.standard_normal((n * t, h, w, c)) * 2).astype(np.float32)
    yd = dev.to_device(y)
    sc = rng.uniform(0.5, 2.0, c).astype(np.float32)
    sh = rng.standard_normal(c).astype(np.float32)
    lo, cn = (1, 0, 2), (5, 4, 3)
    pf = C.POINTER(C.c_float)
    for affine in (True, False):
        yc = dev.empty((n,) + cn + (c,))
        rc = L.s3_chunk_time_last(
            dev.ctx, C.c_void_p(yd.data_ptr()), n, i64x3(t, h, w),
            i64x3(*lo), i64x3(*cn), c,
            sc.ctypes.data_as(pf) if affine else None,
            sh.ctypes.data_as(pf) if affine else None,
            C.c_void_p(yc.data_ptr()))
        _lib.check(rc, dev.ctx, 's3_chunk_time_last')
        for k in range(n):
            hi = np.transpose(y[k * t:(k + 1) * t], (1, 2, 0, 3))
            if affine:
                hi = hi * sc + sh
            np.testing.assert_array_equal(
                yc[k].cpu().numpy(),
                hi[lo[0]:lo[0] + cn[0], lo[1]:lo[1] + cn[1],
                   lo[2]:lo[2] + cn[2]])
    # a crop window that leaves the chunk is refused
    rc = L.s3_chunk_time_last(dev.ctx, C.c_void_p(yd.data_ptr()), n,
                              i64x3(t, h, w), i64x3(0, 0, 4), i64x3(2, 2, 3),
                              c, None, None, C.c_void_p(yc.data_ptr()))
    assert rc < 0


@pytest.mark.parametrize('cfg,precision', [
    ('test_gen_s_2x_2f.json', 'f32'),
    ('sup3r/spatial/gen_2x_2f.json', 'bf16'),     # conv2d_ws / logical-axes kernels
])
def test_spatial_model_chunks_on_the_device_equal_the_generate_path(cfg,
                                                                    precision):
    """a 2-D (spatial) model through ``iter_chunks``: its chunks' time steps
    are the batch axis (forward_pass.py:274-337).  Round 5: batches of such
    chunks run on the device (transpose back to (s1, s2, t), halo crop,
    un-normalisation in ``s3_chunk_time_last``) — bit-identical to the
    chunk-by-chunk ``run_generator`` -> ``model.generate`` path, ragged edge
    chunks and temporal padding included"""
    from sup3r_amd import ForwardPass, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_10m', 'v_10m']
    Sup3rGan.seed(9)
    means = {f: np.float32(0.3 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.5 + 0.25 * i) for i, f in enumerate(feats)}
    m = Sup3rGan(os.path.join(CFG, cfg),
                 os.path.join(CFG, 'test_disc_s_same.json'), means=means,
                 stdevs=stds, precision=precision)
    m.set_model_params(lr_features=feats, hr_out_features=feats, s_enhance=2,
                       t_enhance=1)
    m.init_weights((1, 16, 16, 2), (1, 32, 32, 2))
    assert m.is_4d
    rng = np.random.default_rng(21)
    domain = (rng.standard_normal((44, 37, 21, 2)) * 2 + 0.4).astype(
        np.float32)
    register_model('Sup3rGan', {'model_dir': 'fwp-4d'}, m)
    # (44 = 2 x 22, 37 = 19 + 18: every chunk image has >= 256 positions, the
    # size from which a bf16 plan's kernels do not depend on the batch — below
    # it the few-position kernels are chosen by the batch's TOTAL positions
    # and the two paths agree to bf16 accumulation-order noise only, see the
    # last lines)
    st = ArrayStrategy(domain, {'model_dir': 'fwp-4d'}, (22, 19, 8),
                       spatial_pad=2, temporal_pad=3, max_nodes=1, model=m)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]
    assert len(ids) >= 12

    def run(batch, **options):
        return {c.index: np.array(d) for c, failed, d in
                ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids),
                                        m, batch=batch, options=options)
                if not failed}
    ref = run(1, device_chunks_4d=False)
    assert len(ref) == len(ids)
    c0 = fwp.get_input_chunk(ids[0])
    assert ForwardPass._device_path(m, c0)
    for batch, dev_norm in ((1, True), (3, True), (3, False)):
        # (dev_norm: transpose to time-major + norm_input in
        # s3_chunk_time_first, numpy's fp32 arithmetic; False: host numpy)
        got = run(batch, device_norm_4d=dev_norm)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape and got[k].ndim == 4
            np.testing.assert_array_equal(got[k], ref[k])
    # ragged 4-row edge chunks (8 x 22 = 176 positions per image)
    st2 = ArrayStrategy(domain, {'model_dir': 'fwp-4d'}, (20, 18, 8),
                        spatial_pad=2, temporal_pad=3, max_nodes=1, model=m)
    fwp = ForwardPass(st2, 0)
    ids = [int(i) for i in st2.node_chunks[0]]
    ref = run(1, device_chunks_4d=False)
    got = run(3)
    for k in ref:
        if precision == 'f32':
            np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=1e-4)
        else:
            assert np.abs(got[k] - ref[k]).max() < 3e-2 * np.abs(ref[k]).max()


def test_spatial_model_with_exo_chunks_on_the_device():
    """the sup3rcc spatial step (``sup3rcc/gen_wind_5x_1x_6f.json``): lo-res
    topography concatenated at the input ('input' combine type) and hi-res
    topography through a mid-network ``Sup3rConcat`` ('layer'), on a 4-D model
    whose batch axis is the chunk's time axis — exo fields move their time
    axis to the batch too (forward_pass.py:303-337).  Batches of such chunks
    on the device == chunk by chunk through ``run_generator`` ->
    ``model.generate``, bit for bit."""
    from sup3r_amd import ForwardPass, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_10m', 'v_10m', 'u_100m', 'v_100m', 'u_200m', 'v_200m']
    Sup3rGan.seed(13)
    means = {f: np.float32(0.2 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.5 + 0.2 * i) for i, f in enumerate(feats)}
    means['topography'] = np.float32(300.0)
    stds['topography'] = np.float32(150.0)
    m = Sup3rGan(os.path.join(CFG, 'sup3r/sup3rcc/gen_wind_5x_1x_6f.json'),
                 os.path.join(CFG, 'test_disc_s_same.json'), means=means,
                 stdevs=stds, precision='bf16')
    m.set_model_params(lr_features=feats + ['topography'],
                       hr_out_features=feats, hr_exo_features=['topography'],
                       s_enhance=5, t_enhance=1)
    m.init_weights((1, 18, 17, 7), (1, 90, 85, 7))
    assert m.is_4d
    rng = np.random.default_rng(23)
    domain = (rng.standard_normal((32, 30, 9, 6)) * 2 + 0.4).astype(np.float32)
    topo_hr = (300 + 150 * rng.standard_normal((160, 150, 1))).astype(
        np.float32)
    topo_lr = topo_hr.reshape(32, 5, 30, 5, 1).mean(axis=(1, 3)).astype(
        np.float32)
    exo = {'topography': {'steps': [
        {'model': 0, 'combine_type': 'input', 'data': topo_lr,
         's_enhance': 1, 't_enhance': 1},
        {'model': 0, 'combine_type': 'layer', 'data': topo_hr,
         's_enhance': 5, 't_enhance': 1}]}}
    register_model('Sup3rGan', {'model_dir': 'fwp-4d-exo'}, m)
    st = ArrayStrategy(domain, {'model_dir': 'fwp-4d-exo'}, (16, 15, 6),
                       spatial_pad=1, temporal_pad=1, exo_data=exo,
                       max_nodes=1, model=m)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]
    assert len(ids) == 8

    def run(batch, **options):
        return {c.index: np.array(d) for c, failed, d in
                ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids),
                                        m, batch=batch, options=options)
                if not failed}
    c0 = fwp.get_input_chunk(ids[0])
    assert c0.exo_data['topography']['steps'][1]['data'].ndim == 4
    assert ForwardPass._device_path(m, c0)
    assert not ForwardPass._device_path(m, c0, {'device_chunks_4d': False})
    ref = run(1, device_chunks_4d=False)
    assert len(ref) == len(ids)
    for batch in (1, 4):
        got = run(batch)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape and got[k].ndim == 4
            assert got[k].shape[-1] == 6 and np.isfinite(got[k]).all()
            np.testing.assert_array_equal(got[k], ref[k])
    # the topography matters (a different field, different winds)
    exo2 = {'topography': {'steps': [dict(st_, data=st_['data'][::-1].copy())
                                     for st_ in exo['topography']['steps']]}}
    st2 = ArrayStrategy(domain, {'model_dir': 'fwp-4d-exo'}, (16, 15, 6),
                        spatial_pad=1, temporal_pad=1, exo_data=exo2,
                        max_nodes=1, model=m)
    fwp2 = ForwardPass(st2, 0)
    other = next(np.array(d) for c, failed, d in ForwardPass.iter_chunks(
        [fwp2.get_input_chunk(ids[0])], m, batch=1))
    assert np.abs(other - ref[ids[0]]).max() > 1e-3


# ------------------------------------------- the reference's entry points
def _topo_model(tmp_path=None):
    """a topography-conditioned 3x / 4x generator (Sup3rConcat mid-network)"""
    from sup3r_amd import Sup3rGan
    feats, exo = ['u_10m', 'v_10m'], ['topography']
    Sup3rGan.seed(7)
    means = {'u_10m': np.float32(0.2), 'v_10m': np.float32(-0.4),
             'topography': np.float32(300.0)}
    stds = {'u_10m': np.float32(1.25), 'v_10m': np.float32(1.75),
            'topography': np.float32(150.0)}
    m = Sup3rGan(os.path.join(CFG, 'test_gen_st_3x_4x_2f_topo.json'),
                 os.path.join(CFG, 'test_disc_st_same.json'), means=means,
                 stdevs=stds)
    m.set_model_params(lr_features=feats, hr_out_features=feats,
                       hr_exo_features=exo, s_enhance=3, t_enhance=4)
    m.init_weights((1, 7, 7, 6, 2), (1, 21, 21, 24, 3))
    return m


def _lat_lon(n1, n2):
    lat = np.linspace(41, 40, n1)[:, None] + np.zeros((1, n2))
    lon = np.linspace(-105, -104, n2)[None] + np.zeros((n1, 1))
    return np.stack([lat, lon], -1)


def test_strategy_chunks_with_exo_on_the_device(tmp_path):
    """``ForwardPass.run(strategy, node_index)`` / ``run_chunk(chunk, ...)``
    (forward_pass.py:427-500,582-673) over ``ForwardPassChunk`` structures
    carrying hi-res topography (strategy.py:520-581) through a
    ``Sup3rConcat`` generator: the batched device executor is bit-identical
    to the chunk-by-chunk ``model.generate`` path for every batch size, the
    oracle agrees on a chunk, and the file output is the device-transformed
    field"""
    from oracle.gan import norm_input, un_norm_output
    from oracle.network import Network as ONet
    from sup3r_amd.forward_pass import (ForwardPass, ForwardPassChunk,
                                        register_model)
    from sup3r_amd.strategy import ArrayStrategy
    import json
    model = _topo_model()
    register_model('Sup3rGan', {'model_dir': 'topo-test'}, model)
    rng = np.random.default_rng(4)
    domain = (rng.standard_normal((14, 10, 14, 2)) * 2 + 0.3).astype(
        np.float32)
    topo = (300 + 150 * rng.standard_normal((42, 30, 1))).astype(np.float32)
    exo = {'topography': {'steps': [
        {'model': 0, 'combine_type': 'layer', 'data': topo, 's_enhance': 3,
         't_enhance': 4}]}}

    def strategy(**kw):
        return ArrayStrategy(domain, {'model_dir': 'topo-test'}, (6, 5, 6),
                             spatial_pad=1, temporal_pad=1, exo_data=exo,
                             lat_lon=_lat_lon(42, 30), **kw)
    st = strategy()
    sl = st.fwp_slicer
    # reference-shaped path, chunk by chunk through model.generate
    fwp = ForwardPass(st, 0)
    ref = np.full(sl.hr_shape + (2,), np.nan, np.float32)
    for i in range(sl.n_chunks):
        c = fwp.get_input_chunk(i)
        assert isinstance(c, ForwardPassChunk)
        ref[sl.chunks[i]['hr_slice']] = ForwardPass.run_generator(
            c.input_data, c.hr_crop_slice, model, s_enhance=3, t_enhance=4,
            exo_data=c.exo_data)
    assert np.isfinite(ref).all()
    # the device executor, ragged edge chunks in separate shape groups
    for batch in (1, 4, 8):
        got = np.full_like(ref, np.nan)
        done, kept = ForwardPass.run(strategy(), 0, batch=batch,
                                     return_data=True)
        assert done == sl.n_chunks
        for idx, data in kept:
            got[sl.chunks[idx]['hr_slice']] = data
        np.testing.assert_array_equal(got, ref)
    # two nodes (= two GPUs) split the chunk list, nothing else is shared
    got = np.full_like(ref, np.nan)
    st2 = strategy(max_nodes=2)
    for node in range(2):
        for idx, data in ForwardPass.run(st2, node, return_data=True)[1]:
            got[sl.chunks[idx]['hr_slice']] = data
    np.testing.assert_array_equal(got, ref)
    # run_chunk on one structure + the fp32 oracle on its padded input
    c = fwp.get_input_chunk(3)
    failed, data = ForwardPass.run_chunk(c, {'model_dir': 'topo-test'},
                                         'Sup3rGan', False)
    assert not failed
    np.testing.assert_array_equal(data, ref[sl.chunks[3]['hr_slice']])
    with open(os.path.join(CFG, 'test_gen_st_3x_4x_2f_topo.json')) as f:
        og = ONet(json.load(f))
    c = fwp.get_input_chunk(3)
    x = norm_input(c.input_data[None], [0.2, -0.4], [1.25, 1.75]).astype(
        np.float32)
    t = ((c.exo_data['topography']['steps'][0]['data'][None] - np.float32(
        300.0)) / np.float32(150.0)).astype(np.float32)
    og.forward(x, {'topography': t})
    og.set_weights(model.generator_weights)
    y = un_norm_output(og.forward(x, {'topography': t}), [0.2, -0.4],
                       [1.25, 1.75])
    y = y[0][tuple(c.hr_crop_slice)]
    err = np.abs(y - data).max() / max(1.0, np.abs(y).max())
    print(f'topography chunk through run_chunk vs oracle: {err:.2e}')
    assert err < 1e-4, err
    # file output: u/v inverted on the device, written by the npz handler
    from oracle.output import transform_output
    pat = os.path.join(str(tmp_path), 'out_{file_id}.npz')
    st3 = strategy(out_pattern=pat, invert_uv=True, nn_fill=False)
    assert ForwardPass.run(st3, 0) == sl.n_chunks
    assert st3.node_finished(0) and ForwardPass.run(st3, 0) == 0
    f3 = np.load(st3.out_files[3], allow_pickle=False)
    assert list(f3['features']) == ['windspeed_10m', 'winddirection_10m']
    hs = sl.chunks[3]['hr_slice']
    want, _ = transform_output(
        ref[hs].astype(np.float64), ['u_10m', 'v_10m'],
        _lat_lon(42, 30)[hs[0], hs[1]], True, nn_fill=False)
    assert np.abs(f3['data'][..., 0] - want[..., 0]).max() < 1e-4
    d = np.abs(f3['data'][..., 1] - want[..., 1])
    assert np.minimum(d, 360 - d).max() < 5e-2
    assert f3['gids'].shape == (hs[0].stop - hs[0].start,
                                hs[1].stop - hs[1].start)


def test_multi_step_model_with_per_step_exo_through_run_chunk():
    """The production arrangement of the reference's forward-pass configs:
    ``model_class: MultiStepGan`` = a spatial (4-D) step followed by a
    spatio-temporal one, hi-res topography entering the SECOND step
    (``exo_data[feature]['steps'][k]['model'] == 1``, exo.py:108-130).  The
    chunk goes through ``ForwardPass.run_chunk`` (forward_pass.py:582-673 —
    ``_reshape_data_chunk`` :303-337 moves time to the batch axis for the
    spatial step) and must equal the manual chain on the same padded input."""
    from sup3r_amd import MultiStepGan, Sup3rGan
    from sup3r_amd.forward_pass import (ForwardPass, ForwardPassChunk,
                                        register_model)
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_10m', 'v_10m']
    Sup3rGan.seed(11)
    m_s = Sup3rGan(os.path.join(CFG, 'test_gen_s_2x_2f.json'),
                   os.path.join(CFG, 'test_disc_s_same.json'),
                   means={f: np.float32(0.1) for f in feats},
                   stdevs={f: np.float32(2.0) for f in feats})
    m_s.set_model_params(lr_features=feats, hr_out_features=feats,
                         s_enhance=2, t_enhance=1)
    m_s.init_weights((4, 6, 6, 2), (4, 12, 12, 2))
    m_st = _topo_model()
    ms = MultiStepGan([m_s, m_st])
    assert ms.s_enhance == 6 and ms.t_enhance == 4
    register_model('MultiStepGan', {'model_dirs': ['s', 'st-topo']}, ms)
    rng = np.random.default_rng(21)
    domain = (rng.standard_normal((10, 9, 8, 2)) * 2 + 0.3).astype(np.float32)
    # topography at the resolution the second step's Sup3rConcat sees:
    # lo-res x 2 (step 0) x 3 (step 1)
    topo = (300 + 150 * rng.standard_normal((60, 54, 1))).astype(np.float32)
    exo = {'topography': {'steps': [
        {'model': 1, 'combine_type': 'layer', 'data': topo, 's_enhance': 6,
         't_enhance': 4}]}}
    st = ArrayStrategy(domain, {'model_dirs': ['s', 'st-topo']}, (5, 5, 4),
                       spatial_pad=1, temporal_pad=1, exo_data=exo,
                       model_class='MultiStepGan', model=ms)
    fwp = ForwardPass(st, 0)
    sl = st.fwp_slicer
    assert sl.n_chunks == 8
    for i in (0, 3, 7):
        c = fwp.get_input_chunk(i)
        assert isinstance(c, ForwardPassChunk)
        failed, data = ForwardPass.run_chunk(
            c, st.model_kwargs, 'MultiStepGan', False)
        assert not failed
        want = tuple((s_.stop - s_.start) for s_ in sl.chunks[i]['hr_slice'])
        assert data.shape == want + (2,), (data.shape, want)
        # the manual chain on the same padded lo-res window (a fresh chunk:
        # run_chunk re-lays the exo entries of the one it was given in place,
        # as the reference does)
        c = fwp.get_input_chunk(i)
        x = c.input_data                                   # (s1, s2, t, f)
        y1 = m_s.generate(np.transpose(x, (2, 0, 1, 3)))   # time as batch
        y1 = np.transpose(y1, (1, 2, 0, 3))[None]
        t = c.exo_data['topography']['steps'][0]['data'][None]
        y2 = m_st.generate(y1, exogenous_data={'topography': {'steps': [
            {'model': 0, 'combine_type': 'layer', 'data': t}]}})
        np.testing.assert_array_equal(
            data, y2[0][tuple(c.hr_crop_slice)])
    # the node runner takes the same route for every chunk
    done, kept = ForwardPass.run(st, 0, return_data=True)
    assert done == 8 and len(kept) == 8


def test_step_handover_kernel_vs_numpy():
    """``s3_step_handover``: un_norm_output of step i, the channel selection
    of ``_match_model_input``, the trailing 'input' exo channels and
    norm_input of step i + 1 (multi_step.py:233-259) in one pass — the bits of
    the numpy chain in fp32"""
    import ctypes as C

    from sup3r_amd import _lib
    from sup3r_amd.engine import Device
    dev, L = Device.get(), _lib.lib()
    rng = np.random.default_rng(31)
    pf = C.POINTER(C.c_float)
    for (n_pos, c_src, cmap, n_exo, un, nrm) in (
            (10007, 5, [3, 0, 4], 2, True, True),
            (4096, 2, [0, 1], 0, True, True),
            (333, 14, list(range(14)), 1, True, False),
            (1000, 3, [2], 0, False, True),
            (77, 4, [1, 1, 0], 3, False, False)):
        y = (rng.standard_normal((n_pos, c_src)) * 3).astype(np.float32)
        exo = (300 + 150 * rng.standard_normal((n_pos, max(n_exo, 1)))).astype(
            np.float32)
        sc = (0.5 + rng.random(c_src)).astype(np.float32)
        sh = rng.standard_normal(c_src).astype(np.float32)
        c_dst = len(cmap) + n_exo
        mu = rng.standard_normal(c_dst).astype(np.float32)
        sd = (0.5 + rng.random(c_dst)).astype(np.float32)
        u = y * sc + sh if un else y
        want = u[:, cmap]
        if n_exo:
            want = np.concatenate([want, exo[:, :n_exo]], axis=-1)
        if nrm:
            want = (want - mu) / sd
        assert want.dtype == np.float32
        yd, ed = dev.to_device(y), dev.to_device(exo[:, :max(n_exo, 1)].copy())
        xd = dev.empty((n_pos, c_dst))
        rc = L.s3_step_handover(
            dev.ctx, C.c_void_p(yd.data_ptr()), n_pos, c_src,
            (C.c_int32 * len(cmap))(*cmap), len(cmap),
            sc.ctypes.data_as(pf) if un else None,
            sh.ctypes.data_as(pf) if un else None,
            C.c_void_p(ed.data_ptr()) if n_exo else None, n_exo,
            mu.ctypes.data_as(pf) if nrm else None,
            sd.ctypes.data_as(pf) if nrm else None, C.c_void_p(xd.data_ptr()))
        _lib.check(rc, dev.ctx, 's3_step_handover')
        np.testing.assert_array_equal(xd.cpu().numpy(), want)
    # argument checks
    rc = L.s3_step_handover(dev.ctx, C.c_void_p(yd.data_ptr()), 10, 4,
                            (C.c_int32 * 1)(7), 1, None, None, None, 0, None,
                            None, C.c_void_p(xd.data_ptr()))
    assert rc != 0 and 'map' in _lib.last_error(dev.ctx)


def test_multi_step_chain_of_spatial_steps_on_the_device():
    """MultiStepGan([spatial 2x, spatial 5x + topography]) through
    ``iter_chunks``: plans, hand-over (s3_step_handover: un-normalise, pick
    the next step's channels, append its lo-res topography, normalise), exo
    uploads, crop and delivery on the device — bit-identical to the chain of
    ``generate`` calls through host numpy (multi_step.py:233-259)"""
    from sup3r_amd import ForwardPass, MultiStepGan, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    Sup3rGan.seed(19)
    f6 = ['u_10m', 'v_10m', 'u_100m', 'v_100m', 'u_200m', 'v_200m']
    st = {f: (0.1 * (i + 1), 1.0 + 0.25 * i) for i, f in enumerate(f6)}
    st['topography'] = (300.0, 150.0)
    means = {k: np.float32(v[0]) for k, v in st.items()}
    stds = {k: np.float32(v[1]) for k, v in st.items()}
    # step 1: 6 features 2x (no exo); step 2: the same 6 in ANOTHER order +
    # lo-res topography at the input, hi-res topography mid-network, 5x
    spec1 = json.load(open(os.path.join(CFG, 'sup3r/spatial/gen_2x_2f.json')))
    for layer in (spec1['hidden_layers'] if isinstance(spec1, dict)
                  else spec1):
        if layer.get('filters') == 2:
            layer['filters'] = 6
    m1 = Sup3rGan(spec1, os.path.join(CFG, 'test_disc_s_same.json'),
                  means=means, stdevs=stds, precision='bf16')
    m1.set_model_params(lr_features=f6, hr_out_features=f6, s_enhance=2,
                        t_enhance=1)
    m1.init_weights((1, 18, 17, 6), (1, 36, 34, 6))
    order = ['v_10m', 'u_10m', 'u_200m', 'v_200m', 'u_100m', 'v_100m']
    means2 = {k: np.float32(v + 0.05) for k, v in means.items()}
    m2 = Sup3rGan(os.path.join(CFG, 'sup3r/sup3rcc/gen_wind_5x_1x_6f.json'),
                  os.path.join(CFG, 'test_disc_s_same.json'), means=means2,
                  stdevs=stds, precision='bf16')
    m2.set_model_params(lr_features=order + ['topography'],
                        hr_out_features=order, hr_exo_features=['topography'],
                        s_enhance=5, t_enhance=1)
    m2.init_weights((1, 36, 34, 7), (1, 180, 170, 7))
    ms = MultiStepGan([m1, m2])
    assert ms.s_enhance == 10 and ms.t_enhance == 1 and ms.is_4d
    register_model('MultiStepGan', {'model_dirs': ['s1', 's2']}, ms)
    rng = np.random.default_rng(29)
    domain = (rng.standard_normal((32, 30, 7, 6)) * 2 + 0.4).astype(np.float32)
    topo_hr = (300 + 150 * rng.standard_normal((320, 300, 1))).astype(
        np.float32)
    topo_mid = topo_hr.reshape(64, 5, 60, 5, 1).mean(axis=(1, 3)).astype(
        np.float32)
    exo = {'topography': {'steps': [
        {'model': 1, 'combine_type': 'input', 'data': topo_mid,
         's_enhance': 2, 't_enhance': 1},
        {'model': 1, 'combine_type': 'layer', 'data': topo_hr,
         's_enhance': 10, 't_enhance': 1}]}}
    stg = ArrayStrategy(domain, {'model_dirs': ['s1', 's2']}, (16, 15, 4),
                        spatial_pad=1, temporal_pad=1, exo_data=exo,
                        model_class='MultiStepGan', max_nodes=1, model=ms)
    fwp = ForwardPass(stg, 0)
    ids = [int(i) for i in stg.node_chunks[0]]
    assert len(ids) == 8

    def run(batch, **options):
        return {c.index: np.array(d) for c, failed, d in
                ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids),
                                        ms, batch=batch, options=options)
                if not failed}
    c0 = fwp.get_input_chunk(ids[0])
    assert ForwardPass._device_path(ms, c0)
    assert not ForwardPass._device_path(ms, c0, {'device_chains': False})
    ref = run(1, device_chains=False)
    assert len(ref) == len(ids)
    for batch in (1, 3):
        got = run(batch)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape and got[k].shape[-1] == 6
            assert np.isfinite(got[k]).all()
            np.testing.assert_array_equal(got[k], ref[k])
    # ... and the host chain is the manual chain of generate calls
    c = fwp.get_input_chunk(ids[1])
    y1 = m1.generate(np.transpose(c.input_data, (2, 0, 1, 3)))
    sel = y1[..., [f6.index(f) for f in order]]
    e = c.exo_data['topography']['steps']
    y2 = m2.generate(sel, exogenous_data={'topography': {'steps': [
        {'model': 0, 'combine_type': 'input',
         'data': np.transpose(e[0]['data'], (2, 0, 1, 3))},
        {'model': 0, 'combine_type': 'layer',
         'data': np.transpose(e[1]['data'], (2, 0, 1, 3))}]}})
    want = np.transpose(y2, (1, 2, 0, 3))[tuple(c.hr_crop_slice)]
    np.testing.assert_array_equal(ref[ids[1]], want)
    # a MultiStepGan of ONE step (config_fwp_temporal.json) is a chain too
    one = MultiStepGan([m1])
    register_model('MultiStepGan', {'model_dirs': ['s1']}, one)
    st1 = ArrayStrategy(domain, {'model_dirs': ['s1']}, (16, 15, 4),
                        spatial_pad=1, temporal_pad=1,
                        model_class='MultiStepGan', max_nodes=1, model=one)
    f1 = ForwardPass(st1, 0)
    ca = f1.get_input_chunk(0)
    assert ForwardPass._device_path(one, ca)
    a = next(np.array(d) for _, _, d in ForwardPass.iter_chunks(
        [ca], one, batch=1))
    b = next(np.array(d) for _, _, d in ForwardPass.iter_chunks(
        [f1.get_input_chunk(0)], m1, batch=1))
    np.testing.assert_array_equal(a, b)


def test_multi_step_chain_spatial_then_temporal_on_the_device():
    """MultiStepGan([spatial 2x, spatio-temporal 3x / 4x + topography]): the
    hand-over also moves the time axis from the batch to axis 3
    (_transpose_model_input, multi_step.py:107-146); device chain == host
    chain, batches of 1 and 3 chunks"""
    from sup3r_amd import ForwardPass, MultiStepGan, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_10m', 'v_10m']
    Sup3rGan.seed(11)
    m_s = Sup3rGan(os.path.join(CFG, 'test_gen_s_2x_2f.json'),
                   os.path.join(CFG, 'test_disc_s_same.json'),
                   means={f: np.float32(0.1) for f in feats},
                   stdevs={f: np.float32(2.0) for f in feats})
    m_s.set_model_params(lr_features=feats, hr_out_features=feats,
                         s_enhance=2, t_enhance=1)
    m_s.init_weights((4, 16, 16, 2), (4, 32, 32, 2))
    m_st = _topo_model()
    ms = MultiStepGan([m_s, m_st])
    register_model('MultiStepGan', {'model_dirs': ['s', 'st-topo-b']}, ms)
    rng = np.random.default_rng(37)
    domain = (rng.standard_normal((28, 26, 8, 2)) * 2 + 0.3).astype(np.float32)
    topo = (300 + 150 * rng.standard_normal((168, 156, 1))).astype(np.float32)
    exo = {'topography': {'steps': [
        {'model': 1, 'combine_type': 'layer', 'data': topo, 's_enhance': 6,
         't_enhance': 4}]}}
    stg = ArrayStrategy(domain, {'model_dirs': ['s', 'st-topo-b']},
                        (14, 13, 4), spatial_pad=1, temporal_pad=1,
                        exo_data=exo, model_class='MultiStepGan',
                        max_nodes=1, model=ms)
    fwp = ForwardPass(stg, 0)
    ids = [int(i) for i in stg.node_chunks[0]]
    assert len(ids) == 8

    def run(batch, **options):
        return {c.index: np.array(d) for c, failed, d in
                ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids),
                                        ms, batch=batch, options=options)
                if not failed}
    assert ForwardPass._device_path(ms, fwp.get_input_chunk(ids[0]))
    ref = run(1, device_chains=False)
    assert len(ref) == 8
    for batch in (1, 3):
        got = run(batch)
        for k in ref:
            assert got[k].shape == ref[k].shape and got[k].ndim == 4
            np.testing.assert_array_equal(got[k], ref[k])


def test_residency_is_explicit():
    """``run_batched`` never serves a stale upload: a second array of the same
    shape (CPython may even give it the same ``id``), or the same array
    updated in place, is uploaded again; an explicit ``upload_domain`` handle
    is reused only while the model's statistics are the ones it was
    normalised with"""
    from sup3r_amd import ChunkSlicer, ForwardPass
    model = _model()
    slicer = ChunkSlicer((8, 8), 8, 2, 4, (4, 4, 4), spatial_pad=1,
                         temporal_pad=1)
    fwp = ForwardPass(model, slicer)
    rng = np.random.default_rng(0)

    def run(d):
        out = np.zeros(slicer.hr_shape + (2,), np.float32)
        fwp.run_batched(d, out=out)
        return out
    outs = []
    for i in range(3):
        d = rng.standard_normal((8, 8, 8, 2)).astype(np.float32)
        outs.append((d.copy(), run(d)))
        del d
    for d, o in outs:
        np.testing.assert_array_equal(run(d), o)
    assert not np.array_equal(outs[0][1], outs[1][1])
    d = outs[0][0].copy()
    o0 = run(d)
    d[:] = outs[1][0]                       # in-place refill
    np.testing.assert_array_equal(run(d), outs[1][1])
    assert not np.array_equal(o0, outs[1][1])
    h = fwp.upload_domain(outs[2][0])
    np.testing.assert_array_equal(run(h), outs[2][1])
    model.set_norm_stats({'u_10m': 0.0, 'v_10m': 0.0},
                         {'u_10m': 1.0, 'v_10m': 1.0})
    with pytest.raises(RuntimeError, match='statistics'):
        run(h)


@pytest.mark.parametrize('temporal_pad', [2, 1])
def test_window_forward_equals_full_forward_then_crop(temporal_pad):
    """``s3_plan_forward_window``: the executor's halo crop + un-normalisation
    inside the generator's tail conv (only the chunk's own positions of the
    last conv are computed, no full-size output, no epilogue pass) against the
    full forward + ``s3_chunk_epilogue``.  With the temporal halo a multiple of
    8 hi-res steps (2 x 12) every output runs the identical MFMA sequence: the
    same bits; otherwise the banded tail groups a position's taps differently
    (fp32 summation order).  Edge chunks (no halo on the domain border: the
    crop starts at 0) included.  Reference: forward_pass.py:384-425
    (``_run_generator`` + ``hr_crop_slice``), abstract.py:243-275
    (``un_norm_output``)."""
    from sup3r_amd import ForwardPass, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_100m', 'v_100m', 'temperature_100m', 'pressure_0m']
    Sup3rGan.seed(3)
    means = {f: np.float32(0.3 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.5 + 0.25 * i) for i, f in enumerate(feats)}
    m = Sup3rGan(os.path.join(CFG, 'gen_5x_12x_2f.json'),
                 os.path.join(CFG, 'test_disc_st_same.json'),
                 precision='bf16', means=means, stdevs=stds)
    m.set_model_params(lr_features=feats, hr_out_features=feats[:2],
                       s_enhance=5, t_enhance=12)
    tp = temporal_pad
    m.init_weights((1, 8, 8, 4 + 2 * tp, 4), (1, 40, 40, 12 * (4 + 2 * tp), 2))
    rng = np.random.default_rng(9)
    domain = rng.standard_normal((18, 12, 12, 4)).astype(np.float32)
    register_model('Sup3rGan', {'model_dir': 'win-test'}, m)
    st = ArrayStrategy(domain, {'model_dir': 'win-test'}, (6, 6, 4),
                       spatial_pad=1, temporal_pad=tp, max_nodes=1, model=m)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]

    def run(window):
        out = {}
        for c, failed, d in ForwardPass.iter_chunks(
                (fwp.get_input_chunk(i) for i in ids), m, batch=3,
                options={'window_forward': window}):
            assert not failed
            out[c.index] = np.array(d)
        return out
    ph = m._gen.plan((3, 8, 8, 4 + 2 * tp, 4), training=False)
    assert ph.supports_window
    got, ref = run(True), run(False)
    assert len(got) == len(ids) == 18
    for i in ids:
        assert got[i].shape == ref[i].shape == (30, 30, 48, 2)
        if tp == 2:
            np.testing.assert_array_equal(got[i], ref[i])
        else:
            scale = np.abs(ref[i]).max()
            assert np.abs(got[i] - ref[i]).max() < 2e-6 * scale
    assert np.abs(ref[ids[0]]).max() > 0
