"""Golden-vector tests.  CPU: the oracle reproduces the committed vectors
(freezes the checker).  GPU (-m gpu): the HIP path reproduces them through the
C-ABI (box-independent expected values)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
CFG = os.path.join(HERE, '..', 'sup3r_amd', 'configs')

NETS = [('gen_st_2x_4x_2f.npz', 'test_gen_st_2x_4x_2f.json'),
        ('gen_st_3x_4x_2f_topo.npz', 'test_gen_st_3x_4x_2f_topo.json'),
        ('gen_s_2x_2f.npz', 'test_gen_s_2x_2f.json'),
        ('disc_st_same.npz', 'test_disc_st_same.json'),
        ('disc_st_valid.npz', 'test_disc_st_valid.json')]


def _load_cfg(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


def _weights(d):
    n = sum(1 for k in d.files if k.startswith('w') and k[1:].isdigit())
    return [d[f'w{i}'] for i in range(n)], [d[f'g{i}'] for i in range(n)]


@pytest.mark.parametrize('gold,cfg', NETS)
def test_oracle_reproduces_golden(gold, cfg):
    from oracle.network import Network
    d = np.load(os.path.join(GOLD, gold))
    ws, gs = _weights(d)
    exo = {'topography': d['exo']} if 'exo' in d.files else None
    net = Network(_load_cfg(cfg))
    net.forward(d['x'], exo)
    net.set_weights(ws)
    y = net.forward(d['x'], exo)
    np.testing.assert_allclose(y, d['y'], rtol=0, atol=1e-5)
    dx = net.backward(d['dy'])
    np.testing.assert_allclose(dx, d['dx'], rtol=0,
                               atol=1e-4 * np.abs(d['dx']).max())
    for g, gr in zip(net.grads, gs):
        np.testing.assert_allclose(g, gr, rtol=0,
                                   atol=1e-4 * max(1e-6, np.abs(gr).max()))


def test_permutation_goldens_exact():
    from oracle import layers as L
    d = np.load(os.path.join(GOLD, 'permutation_ops.npz'))
    for b in (2, 3, 5):
        np.testing.assert_array_equal(
            L.depth_to_space(d[f'd2s_x_b{b}'], b), d[f'd2s_y_b{b}'])
    for m in (2, 3):
        y = L.SpatioTemporalExpansion(temporal_mult=m).forward(
            d[f'trepeat_x_m{m}'])
        np.testing.assert_array_equal(y, d[f'trepeat_y_m{m}'])


@pytest.mark.gpu
@pytest.mark.parametrize('gold,cfg', NETS)
def test_hip_reproduces_golden(gold, cfg):
    from sup3r_amd.engine import Network
    d = np.load(os.path.join(GOLD, gold))
    ws, gs = _weights(d)
    net = Network(_load_cfg(cfg), precision='f32')
    net.set_weights(ws)
    dev = net.dev
    exo = {'topography': dev.to_device(d['exo'])} if 'exo' in d.files else {}
    ph = net.plan(d['x'].shape, training=True)
    y = ph.forward(dev.to_device(d['x']), exo).cpu().numpy()
    assert np.abs(y - d['y']).max() < 1e-4 * max(1.0, np.abs(d['y']).max())
    dx = ph.backward(dev.to_device(d['dy']), need_dx=True).cpu().numpy()
    assert np.abs(dx.reshape(d['dx'].shape) - d['dx']).max() < \
        1e-3 * np.abs(d['dx']).max()
    gmax = max(float(np.abs(g).max()) for g in gs)
    for g, gr in zip(net.grads, gs):
        assert np.abs(g - gr).max() < 2e-3 * np.abs(gr).max() + 2e-5 * gmax


@pytest.mark.gpu
def test_hip_permutation_ops_exact():
    """depth-to-space (DCR) and temporal nearest repeat are pure index
    permutations: bit-exact against the integer goldens."""
    from sup3r_amd.engine import Network
    d = np.load(os.path.join(GOLD, 'permutation_ops.npz'))
    for b in (2, 3, 5):
        net = Network([{'class': 'SpatialExpansion', 'spatial_mult': b}])
        # no weights: an empty parameter store
        y = net(d[f'd2s_x_b{b}']).cpu().numpy()
        np.testing.assert_array_equal(y, d[f'd2s_y_b{b}'])
    for m in (2, 3):
        net = Network([{'class': 'SpatioTemporalExpansion',
                        'temporal_mult': m, 'temporal_method': 'nearest'}])
        y = net(d[f'trepeat_x_m{m}']).cpu().numpy()
        np.testing.assert_array_equal(y, d[f'trepeat_y_m{m}'])


@pytest.mark.gpu
def test_forward_pass_executor_on_gpu():
    """Chunk executor with the HIP generator: single chunk == direct generate
    (test_fwp_nochunking), 2-rank sharding == 1 rank bit-exactly, chunked with
    overlap >= receptive radius == un-chunked in the interior."""
    from sup3r_amd import Sup3rGan
    from sup3r_amd.forward_pass import ChunkSlicer, ForwardPass
    Sup3rGan.seed(3)
    model = Sup3rGan(os.path.join(CFG, 'test_gen_st_2x_4x_2f.json'),
                     os.path.join(CFG, 'test_disc_st_same.json'))
    model.meta.update(lr_features=['u', 'v'], hr_out_features=['u', 'v'])
    model.set_norm_stats({'u': 0.3, 'v': -0.2}, {'u': 1.5, 'v': 0.7})
    rng = np.random.default_rng(1)
    domain = rng.standard_normal((12, 12, 8, 2)).astype(np.float32)
    full = model.generate(domain[None])[0]
    assert full.shape == (24, 24, 32, 2)
    s1 = ChunkSlicer((12, 12), 8, 2, 4, (12, 12, 8))
    out1 = np.zeros(s1.hr_shape + (2,), np.float32)
    ForwardPass(model, s1).run_domain(domain, out=out1)
    np.testing.assert_array_equal(out1, full)
    s = ChunkSlicer((12, 12), 8, 2, 4, (6, 6, 4), spatial_pad=3,
                    temporal_pad=2)
    a = np.zeros(s.hr_shape + (2,), np.float32)
    ForwardPass(model, s).run_domain(domain, out=a)
    b = np.zeros_like(a)
    n0 = ForwardPass(model, s, rank=0, nranks=2).run_domain(domain, out=b)
    n1 = ForwardPass(model, s, rank=1, nranks=2).run_domain(domain, out=b)
    assert n0 + n1 == s.n_chunks
    np.testing.assert_array_equal(a, b)
    # overlap reduces the chunk-edge error (the generator's receptive field,
    # ~10 lo-res cells, exceeds any overlap this tiny domain allows)
    s0 = ChunkSlicer((12, 12), 8, 2, 4, (6, 6, 4))
    c = np.zeros_like(a)
    ForwardPass(model, s0).run_domain(domain, out=c)
    assert np.abs(a - full).mean() < np.abs(c - full).mean()


@pytest.mark.gpu
def test_rccl_single_rank_allreduce():
    """The RCCL entry points resolve and run (1-rank communicator): SUM
    all-reduce over the flat gradient buffer is the identity."""
    import ctypes as C
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device, Network
    L = _lib.lib()
    dev = Device.get()
    uid = (C.c_char * 128)()
    assert L.s3_comm_unique_id(uid) == 0
    dev.init_comm(0, 1, uid)
    net = Network([{'class': 'Conv2D', 'filters': 4, 'kernel_size': 3}])
    net.build((1, 6, 6, 2), seed=0)
    gs = [np.random.default_rng(0).standard_normal(w.shape).astype(np.float32)
          for w in net.weights]
    net.set_weights(gs, which=_lib.BUF_G)
    net.allreduce_grads()
    dev.sync()
    for a, b in zip(net.grads, gs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_collective_watchdog_deadline_and_abort():
    """``Device.wait`` (``s3_comm_wait``): the bounded host wait behind the
    loss read-back of a data-parallel step.  Nothing pending -> returns; a
    deadline shorter than the queued work -> ``TimeoutError`` naming what was
    pending, the communicator aborted (``ncclCommAbort``) and the context back
    in single-rank state, the device still usable afterwards."""
    import ctypes as C
    from sup3r_amd import _lib
    from sup3r_amd.engine import Device, Network
    L = _lib.lib()
    dev = Device.get()
    dev.sync()
    dev.wait(timeout_s=30)                       # idle: immediate
    uid = (C.c_char * 128)()
    assert L.s3_comm_unique_id(uid) == 0
    dev.init_comm(0, 1, uid)
    net = Network([{'class': 'Conv3D', 'filters': 64, 'kernel_size': 3,
                    'padding': 'same'}] * 6, precision='f32')
    shape = (8, 24, 24, 48, 64)
    net.build(shape, seed=0)
    ph = net.plan(shape, training=False)
    x = dev.to_device(np.zeros(shape, np.float32))
    ph.forward(x)
    net.set_weights([np.ones_like(w) for w in net.weights], which=_lib.BUF_G)
    net.allreduce_grads()
    dev.wait(timeout_s=120)                      # completes: counters reset
    # a queue of exact-fp32 forwards (tens of ms) + a collective behind it,
    # and a 0 ms deadline
    for _ in range(6):
        ph.forward(x)
    net.allreduce_grads()
    with pytest.raises(TimeoutError) as ei:
        dev.wait(timeout_s=0.0)
    msg = str(ei.value)
    assert 'not finished' in msg and '1 collective' in msg and \
        'communicator aborted' in msg, msg
    assert dev.nranks == 1
    dev.sync()                                   # the queued work still drains
    net.allreduce_grads()                        # single rank again: identity
    dev.wait(timeout_s=60)
    for g in net.grads:
        assert np.all(g == 1.0)


@pytest.mark.gpu
def test_bucketed_allreduce_under_the_backward_pass_single_rank():
    """``s3_params_arm_allreduce``: the backward pass hands the finished tail
    of the gradient buffer to RCCL bucket by bucket on the comm stream.  With
    a 1-rank communicator the SUM is the identity, so the armed step must
    leave exactly the gradients (and, after Adam, the weights) of the plain
    step — and the buckets must cover the buffer exactly once."""
    import ctypes as C
    import os
    from sup3r_amd import Sup3rGan, _lib
    from sup3r_amd.engine import Device
    L = _lib.lib()
    dev = Device.get()
    uid = (C.c_char * 128)()
    assert L.s3_comm_unique_id(uid) == 0
    dev.init_comm(0, 1, uid)
    cfg = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd',
                       'configs')
    rng = np.random.default_rng(3)
    lr = rng.standard_normal((4, 4, 4, 4, 2)).astype(np.float32)
    hr = rng.standard_normal((4, 8, 8, 16, 2)).astype(np.float32)

    def run(bucket):
        Sup3rGan.seed(4)
        m = Sup3rGan(os.path.join(cfg, 'test_gen_st_2x_4x_2f.json'),
                     os.path.join(cfg, 'test_disc_st_same.json'),
                     loss='MeanAbsoluteError', learning_rate=1e-3)
        m.init_weights(lr.shape, hr.shape)
        out = {}
        for which, kw in (('gen', dict(train_gen=True, train_disc=False)),
                          ('disc', dict(train_gen=False, train_disc=True))):
            net = m.generator if which == 'gen' else m.discriminator
            # (the flat store pads every tensor to whole float4s)
            total = int(L.s3_params_total(net.params))
            before = dev.stat('bucket_elems')
            m._compute.loss_and_grads(
                lr, hr, m._loss_terms, weight_gen_advers=1e-2,
                overlap_bucket=bucket, **kw)
            m._compute.allreduce_grads(which)
            dev.sync()
            if bucket:
                assert dev.stat('bucket_elems') - before == total, which
            out[which] = net.grads
            m._compute.apply(which, m.optimizer if which == 'gen'
                             else m.optimizer_disc)
        return out, m.weights
    g0, w0 = run(None)
    for bucket in (4096, 1 << 30):          # many small buckets / one
        g1, w1 = run(bucket)
        for which in g0:
            for a, b in zip(g1[which], g0[which]):
                np.testing.assert_array_equal(a, b)
        for a, b in zip(w1, w0):
            np.testing.assert_array_equal(a, b)
    # the arming belongs to ONE backward pass: consumed by it,
    # removable by the caller, never set without a communicator
    Sup3rGan.seed(4)
    m = Sup3rGan(os.path.join(cfg, 'test_gen_st_2x_4x_2f.json'),
                 os.path.join(cfg, 'test_disc_st_same.json'),
                 loss='MeanAbsoluteError', learning_rate=1e-3)
    m.init_weights(lr.shape, hr.shape)
    kw = dict(train_gen=True, train_disc=False)
    m._compute.loss_and_grads(lr, hr, m._loss_terms, weight_gen_advers=1e-2,
                              overlap_bucket=4096, **kw)
    before = dev.stat('bucket_elems')
    # (no allreduce_grads in between — as after an exception in the caller)
    m._compute.loss_and_grads(lr, hr, m._loss_terms, weight_gen_advers=1e-2,
                              **kw)
    dev.sync()
    assert dev.stat('bucket_elems') == before
    m.generator.arm_allreduce(4096)
    m.generator.arm_allreduce(-1)
    m._compute.loss_and_grads(lr, hr, m._loss_terms, weight_gen_advers=1e-2,
                              **kw)
    m._compute.allreduce_grads('gen')        # nothing armed: the plain SUM
    dev.sync()
    assert dev.stat('bucket_elems') == before
    L.s3_comm_destroy(dev.ctx)
    dev.rank, dev.nranks = 0, 1
    m.generator.arm_allreduce(4096)          # no communicator: a no-op
    m._compute.loss_and_grads(lr, hr, m._loss_terms, weight_gen_advers=1e-2,
                              **kw)
    m._compute.allreduce_grads('gen')
    dev.sync()
    assert dev.stat('bucket_elems') == before
