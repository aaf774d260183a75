"""CPU tests of the chunk executor's host logic (slice algebra, padding,
placement, rank sharding) with a stand-in model whose ``generate`` is a pure
numpy local operator — the reference's own self-consistency checks
(tests/forward_pass/test_forward_pass.py:411-558: chunked == un-chunked)
re-expressed without files.  Includes the world_size-2 gloo run of the
chunk-sharded path."""
import os
import socket
import sys

import numpy as np
import pytest

from sup3r_amd.forward_pass import ChunkSlicer, ForwardPass, chunk_slices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LocalModel:
    """generate = nearest-neighbour enhancement of a 3x3x3 box mean, i.e. a
    local operator with receptive radius 1 (so overlap >= 1 makes chunking
    exact away from domain edges, and reflect padding makes it exact AT them
    only if the un-chunked run pads the same way)."""

    s_enhance, t_enhance = 2, 3
    is_4d, is_5d = False, True
    hr_out_features = ['a', 'b']

    def generate(self, x, exogenous_data=None):
        x = np.asarray(x, np.float64)
        xp = np.pad(x, [(0, 0), (1, 1), (1, 1), (1, 1), (0, 0)],
                    mode='reflect')
        acc = 0
        for a in range(3):
            for b in range(3):
                for c in range(3):
                    acc = acc + xp[:, a:a + x.shape[1], b:b + x.shape[2],
                                   c:c + x.shape[3]]
        y = acc / 27.0
        y = np.repeat(np.repeat(np.repeat(y, 2, 1), 2, 2), 3, 3)
        return y


def test_chunk_slices():
    assert chunk_slices(10, 4) == [slice(0, 4), slice(4, 8), slice(8, 10)]
    assert chunk_slices(4, 4) == [slice(0, 4)]


def test_slicer_covers_domain_once():
    s = ChunkSlicer((20, 17), 50, 2, 3, (8, 8, 16), spatial_pad=2,
                    temporal_pad=3)
    assert s.n_chunks == 3 * 3 * 4
    cover = np.zeros(s.hr_shape, np.int32)
    for c in s.chunks:
        cover[c['hr_slice']] += 1
        # padded shape is chunk + 2*pad on every side after edge padding
        for d in range(3):
            n = c['lr_pad_slice'][d].stop - c['lr_pad_slice'][d].start
            n += sum(c['pad_width'][d])
            core = c['lr_slice'][d].stop - c['lr_slice'][d].start
            assert n == core + 2 * (2 if d < 2 else 3)
    assert (cover == 1).all()
    # chunk index ordering: spatial fastest, then time (slicer.py:668-673)
    assert s.get_chunk_indices(10) == (1, 1)
    all_ids = sorted(sum((s.rank_chunks(r, 4) for r in range(4)), []))
    assert all_ids == list(range(s.n_chunks))
    blk = [s.rank_chunks(r, 4, 'block') for r in range(4)]
    assert sorted(sum(blk, [])) == list(range(s.n_chunks))


def test_chunked_equals_unchunked():
    rng = np.random.default_rng(0)
    domain = rng.standard_normal((20, 17, 25, 2))
    model = LocalModel()
    full = model.generate(domain[None])[0]
    s = ChunkSlicer((20, 17), 25, 2, 3, (8, 8, 10), spatial_pad=1,
                    temporal_pad=1)
    out = np.zeros(s.hr_shape + (2,))
    n = ForwardPass(model, s).run_domain(domain, out=out)
    assert n == s.n_chunks
    np.testing.assert_allclose(out, full, atol=1e-12)
    # single chunk == direct generate (test_fwp_nochunking)
    s1 = ChunkSlicer((20, 17), 25, 2, 3, (20, 17, 25))
    out1 = np.zeros(s1.hr_shape + (2,))
    ForwardPass(model, s1).run_domain(domain, out=out1)
    np.testing.assert_array_equal(out1, full)


def test_models_without_a_device_chunk_path_take_the_generate_loop():
    """a model that carries an engine network (``_gen``) but
    overrides ``generate`` (``SolarCC``: ``supports_device_chunks = False``)
    must reach ``run_chunks`` from ``run_batched`` and ``run_domain`` too — the
    device path would bypass the override."""
    class Overriding(LocalModel):
        _gen = object()                      # "has a device network"
        supports_device_chunks = False
        lr_features = ['a', 'b']
        hr_exo_features = []
        calls = 0

        def generate(self, x, exogenous_data=None):
            type(self).calls += 1
            return super().generate(x, exogenous_data)
    rng = np.random.default_rng(1)
    domain = rng.standard_normal((10, 9, 8, 2))
    model = Overriding()
    s = ChunkSlicer((10, 9), 8, 2, 3, (5, 5, 4), spatial_pad=1,
                    temporal_pad=1)
    for runner in ('run_domain', 'run_batched'):
        Overriding.calls = 0
        out = np.zeros(s.hr_shape + (2,))
        n = getattr(ForwardPass(model, s), runner)(domain, out=out)
        assert n == s.n_chunks == Overriding.calls, runner
        np.testing.assert_allclose(out, model.generate(domain[None])[0],
                                   atol=1e-12)


def test_output_check_and_errors():
    assert ForwardPass._output_check(np.full((4, 4, 4, 1), np.nan))
    assert ForwardPass._output_check(np.ones((4, 4, 4, 2)))
    assert not ForwardPass._output_check(
        np.random.default_rng(0).standard_normal((4, 4, 4, 2)))
    model = LocalModel()
    s = ChunkSlicer((8, 8), 8, 2, 3, (4, 4, 4))
    fp = ForwardPass(model, s)
    bad = np.zeros((8, 8, 8, 2))
    bad[0, 0, 0, 0] = np.nan
    with pytest.raises(ValueError):
        fp.run_domain_chunk(bad, 0)
    with pytest.raises(MemoryError):
        fp.run_domain_chunk(np.ones((8, 8, 8, 2)), 0)
    s_bad = ChunkSlicer((8, 8), 8, 3, 3, (4, 4, 4))
    with pytest.raises(RuntimeError):
        ForwardPass(model, s_bad)


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["S3_ROOT"])
import numpy as np, torch, torch.distributed as dist
from sup3r_amd.forward_pass import ChunkSlicer, ForwardPass
from sup3r_amd.distributed import env_rank, shard_batch, sum_over_ranks_host
from tests.test_forward_pass_cpu import LocalModel
dist.init_process_group("gloo")
rank, _, world = env_rank()
rng = np.random.default_rng(0)
domain = rng.standard_normal((12, 10, 14, 2))
model = LocalModel()
s = ChunkSlicer((12, 10), 14, 2, 3, (5, 5, 6), spatial_pad=1, temporal_pad=1)
out = np.zeros(s.hr_shape + (2,))
fp = ForwardPass(model, s, rank=rank, nranks=world)
n = fp.run_domain(domain, out=out)
# ranks wrote disjoint windows: SUM over ranks assembles the domain
total = sum_over_ranks_host([out])[0]
full = model.generate(domain[None])[0]
assert np.allclose(total, full, atol=1e-12), "sharded != unsharded"
cnt = sum_over_ranks_host([np.array([float(n)])])[0]
assert cnt[0] == s.n_chunks
# training-side semantics: equal batch shards, gradients SUMMED over ranks
batch = np.arange(8 * 3, dtype=np.float64).reshape(8, 3)
mine = shard_batch(batch, rank, world)
g = sum_over_ranks_host([mine.sum(axis=0)])[0]
assert np.allclose(g, batch.sum(axis=0))
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_world_size_2_gloo(tmp_path):
    import subprocess
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, S3_ROOT=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    # both workers ran every assertion (their stdout lines may interleave)
    assert res.stdout.count('ok') >= 2, res.stdout


# ------------------------------------------- the reference's entry points
import dataclasses  # noqa: E402


@dataclasses.dataclass
class RefChunk:
    """replica of sup3r.pipeline.strategy.ForwardPassChunk (strategy.py:
    37-54): the executor only duck-types the structure"""
    input_data: np.ndarray
    exo_data: dict
    hr_crop_slice: tuple
    lr_pad_slice: tuple
    hr_lat_lon: np.ndarray
    hr_times: object
    gids: np.ndarray
    out_file: str
    pad_width: tuple
    index: int

    def __post_init__(self):
        self.shape = self.input_data.shape


class ExoLocalModel(LocalModel):
    """LocalModel + a hi-res 'layer' exo field added to both outputs"""
    meta, model_params = {}, {}
    s_enhancements, t_enhancements = [2], [3]
    lr_features = ['a', 'b']

    def generate(self, x, exogenous_data=None):
        y = super().generate(x)
        if exogenous_data is not None:
            topo = exogenous_data['topography']['steps'][0]['data']
            assert topo.shape[:4] == y.shape[:4], (topo.shape, y.shape)
            y = y + topo
        return y


def _array_strategy(domain, topo, max_nodes=1, **kw):
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    model = register_model('ExoLocalModel', {'model_dir': 'mem'},
                           ExoLocalModel())
    exo = None if topo is None else {'topography': {'steps': [
        {'model': 0, 'combine_type': 'layer', 'data': topo, 's_enhance': 2,
         't_enhance': 3}]}}
    return ArrayStrategy(domain, {'model_dir': 'mem'}, (8, 8, 10),
                         spatial_pad=1, temporal_pad=1,
                         model_class='ExoLocalModel', exo_data=exo,
                         max_nodes=max_nodes, model=model, **kw), model


def test_strategy_chunks_with_exo_through_the_reference_entry_points():
    """``ForwardPass.run(strategy, node_index)`` over ``ForwardPassChunk``s
    that carry hi-res exo data (strategy.py:520-581, forward_pass.py:66-72,
    427-500): chunked == un-chunked, every node's share placed once, and the
    classmethod ``run_chunk(chunk, model_kwargs, model_class, allowed_const,
    ...)`` on a replica of the reference's chunk structure"""
    rng = np.random.default_rng(1)
    domain = rng.standard_normal((20, 17, 25, 2))
    topo = rng.standard_normal((40, 34, 1))          # constant in time
    strategy, model = _array_strategy(domain, topo, max_nodes=3)
    sl = strategy.fwp_slicer
    full = model.generate(domain[None])[0] + topo[:, :, None, :]
    out = np.full(sl.hr_shape + (2,), np.nan)
    n = 0
    assert len(strategy.node_chunks) == 3
    for node in range(3):
        done, kept = ForwardPass.run(strategy, node, return_data=True)
        n += done
        for idx, data in kept:
            assert np.isnan(out[sl.chunks[idx]['hr_slice']]).all()
            out[sl.chunks[idx]['hr_slice']] = data
    assert n == sl.n_chunks and strategy.node_finished(0)
    # interior cells: exact; the un-chunked run reflect-pads the same way at
    # the domain edges (LocalModel), so everything is
    np.testing.assert_allclose(out, full, rtol=0, atol=1e-12)
    # a finished node is skipped (forward_pass.py:441)
    assert ForwardPass.run(strategy, 0) == 0
    # --- run_chunk on a replica of the reference's structure
    strategy2, _ = _array_strategy(domain, topo)
    fwp = ForwardPass(strategy2, 0)
    c = fwp.get_input_chunk(4)
    rc = RefChunk(**{f.name: getattr(c, f.name)
                     for f in dataclasses.fields(RefChunk)})
    assert rc.shape == (10, 10, 12, 2)
    # 'layer' exo arrives edge-padded, expanded in time, at hi-res
    assert rc.exo_data['topography']['steps'][0]['data'].shape == \
        (20, 20, 36, 1)
    failed, data = ForwardPass.run_chunk(
        rc, {'model_dir': 'mem'}, 'ExoLocalModel', False, invert_uv=False,
        meta=fwp.meta, nn_fill=True, output_workers=None)
    assert not failed
    np.testing.assert_allclose(data, full[sl.chunks[4]['hr_slice']],
                               rtol=0, atol=1e-12)
    assert set(fwp.meta) == {'node_index', 'creation_date', 'model_meta',
                             'gan_params', 'strategy_meta'}
    # NaN in the input -> RuntimeError naming the feature (:640-645) on the
    # device path; the host path reports through model.generate
    with pytest.raises(KeyError):
        ForwardPass.run_chunk(rc, 'nowhere', 'NoSuchModel', False)


def test_failed_chunk_raises_memory_error():
    class Flat(ExoLocalModel):
        def generate(self, x, exogenous_data=None):
            return np.zeros_like(super().generate(x))
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    m = register_model('Flat', {'model_dir': 'flat'}, Flat())
    domain = np.random.default_rng(2).standard_normal((8, 8, 10, 2))
    st = ArrayStrategy(domain, {'model_dir': 'flat'}, (8, 8, 10), 1, 1,
                       model_class='Flat', model=m)
    with pytest.raises(MemoryError):
        ForwardPass.run(st, 0)
    st = ArrayStrategy(domain, {'model_dir': 'flat'}, (8, 8, 10), 1, 1,
                       model_class='Flat', model=m, allowed_const=[0])
    assert ForwardPass.run(st, 0) == 1


def test_reshape_data_chunk_lays_exo_out_per_model_step():
    """forward_pass.py:303-337: an exo entry takes the layout of the model
    step that consumes it — (t, s1, s2, f) for a spatial step, a leading
    batch axis for a spatio-temporal one — not that of the first step"""
    from sup3r_amd.forward_pass import ForwardPass

    class Step:
        def __init__(self, is_4d):
            self.is_4d = is_4d

    class Chain:
        models = [Step(True), Step(False)]
        is_4d = True                     # the first step is spatial

    x = np.arange(5 * 6 * 4 * 2, dtype=np.float32).reshape(5, 6, 4, 2)
    e0 = np.arange(5 * 6 * 4, dtype=np.float32).reshape(5, 6, 4, 1)
    e1 = np.arange(10 * 12, dtype=np.float32).reshape(10, 12, 1)
    exo = {'topography': {'steps': [
        {'model': 0, 'combine_type': 'input', 'data': e0},
        {'model': 1, 'combine_type': 'layer', 'data': e1}]}}
    d, out, i_t, i_s = ForwardPass._reshape_data_chunk(Chain(), x, exo)
    assert d.shape == (4, 5, 6, 2) and (i_t, i_s) == (0, 1)
    np.testing.assert_array_equal(d, np.transpose(x, (2, 0, 1, 3)))
    s0, s1 = out['topography']['steps']
    np.testing.assert_array_equal(s0['data'], np.transpose(e0, (2, 0, 1, 3)))
    assert s1['data'].shape == (1, 10, 12, 1)
    bad = {'topography': {'steps': [{'model': 2, 'combine_type': 'layer',
                                     'data': e1}]}}
    with pytest.raises(AssertionError):
        ForwardPass._reshape_data_chunk(Chain(), x, bad)


@pytest.mark.parametrize('mode', ['reflect', 'symmetric', 'edge', 'wrap',
                                  'constant'])
def test_pad_source_data_time_invariant_exo_is_the_repeated_field(mode):
    """``pad_source_data`` (forward_pass.py:122-186) repeats a 3-D exo field
    along time and pads the result.  For the padding modes that keep a
    time-constant field time-constant the build hands out a zero-stride view
    of the spatially padded field instead — the same VALUES and shape as
    ``np.pad(np.repeat(...))``, which the device executor uploads once per
    chunk; 'constant' (zero) padding along time keeps the materialised form"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, 5, 4, 2)).astype(np.float32)
    topo = rng.standard_normal((18, 15, 1)).astype(np.float32)
    tvar = rng.standard_normal((18, 15, 8, 1)).astype(np.float32)
    pad = ((1, 2), (2, 1), (1, 1))
    exo = {'topography': {'steps': [
        {'model': 0, 'combine_type': 'layer', 'data': topo},
        {'model': 0, 'combine_type': 'input', 'data': tvar}]}}
    enh = {'topography': [(3, 2), (3, 2)]}
    out, got = ForwardPass.pad_source_data(x, pad, exo, mode=mode,
                                           enhancements=enh)
    np.testing.assert_array_equal(out, np.pad(x, (*pad, (0, 0)), mode=mode))
    ew = ((3, 6), (6, 3), (2, 2), (0, 0))
    want0 = np.pad(np.repeat(topo[:, :, None, :], 2 * 4, axis=2), ew,
                   mode=mode)
    g0 = got['topography']['steps'][0]['data']
    assert g0.shape == want0.shape
    np.testing.assert_array_equal(g0, want0)
    assert (g0.strides[2] == 0) == (mode != 'constant')
    # a field WITH a time axis is padded as it is
    g1 = got['topography']['steps'][1]['data']
    np.testing.assert_array_equal(g1, np.pad(tvar, ew, mode=mode))
    assert g1.strides[2] != 0


def test_device_chain_predicate_on_duck_typed_steps():
    """which ``MultiStepGan`` chains the executor keeps on the device
    (``ForwardPass._device_chain``): spatial steps before spatio-temporal
    ones, every step on this engine with the base class's normalisation, fp32
    statistics, no 'output' exo, fp32 'input' exo for the later steps"""
    import types

    from sup3r_amd.gan import Sup3rGan

    dev = object()

    def step(rank4, dtype=np.float32, own_norm=False, on_device=True,
             own_combine=False):
        cls = type('Step', (), {
            'norm_input': (lambda self, x: x) if own_norm
            else Sup3rGan.norm_input,
            'un_norm_output': Sup3rGan.un_norm_output,
            'generate': Sup3rGan.generate,
            '_combine_fwp_input': (lambda self, x, e=None: x) if own_combine
            else Sup3rGan._combine_fwp_input,
            'supports_device_chunks': True})
        m = cls()
        m._gen = types.SimpleNamespace(dev=dev) if on_device else None
        m.is_4d, m.is_5d = rank4, not rank4
        m.lr_features, m.hr_out_features = ['u', 'v'], ['u', 'v']
        m._means = {'u': dtype(0), 'v': dtype(1)}
        m._stats_for = lambda feats: (np.array([m._means[f] for f in feats]),
                                      np.array([dtype(1)] * len(feats)))
        return m

    def chain(*steps):
        return types.SimpleNamespace(models=list(steps))
    chunk = types.SimpleNamespace(exo_data=None)
    ok = ForwardPass._device_chain
    assert ok(chain(step(True), step(True)), chunk)
    assert ok(chain(step(True), step(False)), chunk)
    assert ok(chain(step(False)), chunk)
    assert not ok(chain(step(False), step(True)), chunk)        # 5-D then 4-D
    assert not ok(chain(step(True), step(True, np.float64)), chunk)
    assert not ok(chain(step(True, own_norm=True), step(True)), chunk)
    # (the device hand-over re-implements the base class's input combination)
    assert not ok(chain(step(True), step(True, own_combine=True)), chunk)
    assert not ok(chain(step(True), step(True, on_device=False)), chunk)
    assert not ok(chain(step(True), step(True)), chunk,
                  {'device_chains': False})
    # a single spatial model whose feature count exceeds what the time-major
    # transposes carry (s3_chunk_time_first / _last: 16 channels) goes chunk
    # by chunk through model.generate; 16 is still on the device path
    single = step(True)
    assert ForwardPass._device_path(single, chunk)
    assert not ForwardPass._device_path(single, chunk,
                                        {'device_chunks_4d': False})
    single.hr_out_features = [f'f{i}' for i in range(16)]
    assert ForwardPass._device_path(single, chunk)
    single.hr_out_features = [f'f{i}' for i in range(17)]
    assert not ForwardPass._device_path(single, chunk)
    single.hr_out_features, single.lr_features = ['u'], \
        [f'f{i}' for i in range(17)]
    assert not ForwardPass._device_path(single, chunk)
    wide5 = step(False)
    wide5.lr_features = [f'f{i}' for i in range(17)]
    assert ForwardPass._device_path(wide5, chunk)        # 5-D: no such limit
    e32 = np.zeros((4, 4, 2, 1), np.float32)
    for ctype, data, want in (('layer', e32, True), ('input', e32, True),
                              ('input', e32.astype(np.float64), False),
                              ('output', e32, False)):
        chunk = types.SimpleNamespace(exo_data={'topography': {'steps': [
            {'model': 1, 'combine_type': ctype, 'data': data}]}})
        assert ok(chain(step(True), step(True)), chunk) == want, ctype
    # the public predicate routes chains there
    assert ForwardPass._device_path(chain(step(True), step(True)),
                                    types.SimpleNamespace(exo_data=None))


def test_chunk_read_ahead_keeps_order_propagates_errors_and_stops():
    """``forward_pass._read_ahead`` (the helper thread behind
    ``ChunkPathOptions.input_prefetch``): items in order, an exception of the
    source surfaces at the consumer where it happened, and a consumer that goes
    away stops the helper instead of leaving it parked on a full queue"""
    import threading
    import time

    from sup3r_amd.forward_pass import _read_ahead
    assert list(_read_ahead(iter(range(50)), 4)) == list(range(50))
    assert list(_read_ahead(iter(()), 3)) == []

    def boom():
        yield 1
        yield 2
        raise KeyError('source failed')
    got = []
    try:
        for v in _read_ahead(boom(), 2):
            got.append(v)
    except KeyError as e:
        assert 'source failed' in str(e)
    else:
        raise AssertionError('the source error was swallowed')
    assert got == [1, 2]
    # early exit: the generator is closed after two of a long stream
    pulled = []

    def slow():
        for i in range(10 ** 6):
            pulled.append(i)
            yield i
    n0 = threading.active_count()
    gen = _read_ahead(slow(), 3)
    assert next(gen) == 0 and next(gen) == 1
    gen.close()
    time.sleep(0.3)
    n_pulled = len(pulled)
    time.sleep(0.2)
    assert len(pulled) == n_pulled and n_pulled <= 8       # (depth 3 + in flight)
    assert threading.active_count() <= n0
