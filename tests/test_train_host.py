"""CPU tests of the HOST logic of ``Sup3rGan.train``: loss futures, running
windows, history schema, the train / skip gating, checkpoint round trips, and
the sharded multi-GPU step — single process with virtual shards against the
oracle's per-shard SUM, and a real world_size-2 run over gloo with DIFFERENT
initial weights and a gating threshold placed where ranks would disagree if
they looked at their own shard's loss.  The arithmetic behind the model is the
numpy oracle (``tests/cpu_compute.CpuGanCompute`` installed as the compute
factory): these tests say nothing about the HIP kernels — the ``-m gpu`` tests
do — only about the control flow the reference keeps in
sup3r/models/base.py:944-1191 and abstract.py:785-914."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')
GEN = os.path.join(CFG, 'test_gen_st_2x_4x_2f.json')
DISC = os.path.join(CFG, 'test_disc_st_same.json')


def _model(**kw):
    from sup3r_amd import Sup3rGan
    from tests.cpu_compute import CpuGanCompute

    class CpuGan(Sup3rGan):
        _compute_factory = CpuGanCompute
    kw.setdefault('learning_rate', 1e-3)
    kw.setdefault('loss', 'MeanAbsoluteError')
    return CpuGan(GEN, DISC, **kw)


def _handler(**kw):
    from tests.helpers import SyntheticBatchHandler
    kw.setdefault('batch_size', 4)
    kw.setdefault('n_batches', 3)
    return SyntheticBatchHandler((8, 8, 16), 2, 4, ['u', 'v'], **kw)


RES = {'spatial': '8km', 'temporal': '40min'}


def test_loss_window_semantics():
    """running means over the last n mini-batches; a key a batch did not
    compute is carried at its previous mean (base.py:1065-1072)"""
    from sup3r_amd.ledger import LossWindow
    w = LossWindow('train_', 3)
    w.push({'loss_gen': 1.0, 'loss_disc': 4.0, 'gen_train_frac': 1.0})
    w.push({'loss_gen': 3.0, 'gen_train_frac': 1.0}, carry=w.means())
    m = w.means()
    assert m['train_loss_gen'] == 2.0 and m['train_loss_disc'] == 4.0
    assert m['gen_train_frac'] == 1.0
    w.push({'loss_gen': 5.0}, carry=m)
    w.push({'loss_gen': 7.0}, carry=w.means())       # first row drops out
    assert w.means()['train_loss_gen'] == 5.0
    assert len(w) == 3
    w.resize(2)
    assert w.means()['train_loss_gen'] == 6.0
    w.resize(4)
    w.push({'loss_gen': 2.0})
    assert w.means()['train_loss_gen'] == pytest.approx((5 + 7 + 2) / 3)


def test_history_table_and_early_stop():
    from sup3r_amd import Sup3rGan
    from sup3r_amd.ledger import History
    h = History()
    assert h.next_epochs(2) == [0, 1]
    for e, v in enumerate([1.0, 0.5, 0.499, 0.4985, 0.498, 0.4979, 0.4978,
                           0.4977]):
        h.write(e, {'elapsed_time': float(e), 'train_loss_gen': v})
        h.write(e, {'OptmGen/learning_rate': 1e-4})
    assert list(h.frame.index) == list(range(8))
    assert h.frame.index.name == 'epoch'
    assert h.next_epochs(2) == [8, 9]
    assert Sup3rGan.early_stop(h.frame, 'train_loss_gen', 0.005, 5)
    assert not Sup3rGan.early_stop(h.frame, 'train_loss_gen', 1e-5, 5)
    assert not Sup3rGan.early_stop(h.frame.iloc[:6], 'train_loss_gen', 0.005,
                                   5)      # needs more than n_epoch + 1 rows
    with tempfile.TemporaryDirectory() as td:
        fp = os.path.join(td, 'history.csv')
        h.frame.to_csv(fp)
        again = History(fp)
        assert again.next_epochs(1) == [8]
        assert again.last_row()['train_loss_gen'] == pytest.approx(0.4977)


def test_train_history_gating_and_resume():
    """history schema of test_train_gan.py:174-199; the discriminator sits out
    while its running loss is under the lower bound and the generator while it
    is over the upper one (base.py:1161-1164); a loaded model continues the
    epoch numbering (base.py:739-743)."""
    m = _model()
    bh = _handler()
    with tempfile.TemporaryDirectory() as td:
        kw = dict(input_resolution=RES, n_epoch=2, weight_gen_advers=1e-3,
                  checkpoint_int=1, out_dir=os.path.join(td, 'e_{epoch}'))
        # bounds no loss can leave: both train on every batch
        m.train(bh, disc_loss_bounds=(-np.inf, np.inf), **kw)
        h = m.history
        assert list(h.index) == [0, 1] and bh.stopped
        assert (h['gen_train_frac'] == 1).all()
        # first batch: running disc loss is 0 = "too good" only if lower
        # bound >= 0; with -inf every batch trains the discriminator
        assert (h['disc_train_frac'] == 1).all()
        for col in ('elapsed_time', 'train_loss_gen', 'train_loss_disc',
                    'train_loss_gen_content', 'train_loss_gen_advers',
                    'train_mean_absolute_error', 'val_loss_gen',
                    'val_loss_gen_advers', 'total_batches', 'weight_gen_advers',
                    'disc_loss_bound_0', 'disc_loss_bound_1',
                    'OptmGen/learning_rate', 'OptmDisc/learning_rate',
                    'OptmGen/iteration'):
            assert col in h, col
        assert any(c.startswith('OptmGen/Adam/v/') for c in h.columns)
        assert h['total_batches'].iloc[-1] == 6
        assert h['OptmGen/iteration'].iloc[-1] == 6
        trace = m._compute.log
        assert trace.count(('adam', 'gen')) == 6
        assert trace.count(('adam', 'disc')) == 6
        assert os.path.exists(os.path.join(td, 'e_1', 'model_disc.pkl'))
        # disc loss ~0.69 > upper bound 0.1: the generator must sit out,
        # except on the very first batch of a fresh window (running mean 0 is
        # "disc too good": only the generator trains)
        m2 = _model()
        m2.train(_handler(), disc_loss_bounds=(0.05, 0.1), **kw)
        log2 = m2._compute.log
        assert log2[0] == ('adam', 'gen')
        assert log2.count(('adam', 'gen')) == 1
        assert log2.count(('adam', 'disc')) == 5
        assert m2.history['gen_train_frac'].iloc[0] == pytest.approx(1 / 3)
        # resume
        m.save(os.path.join(td, 'ckpt'))
        m3 = type(m).load(os.path.join(td, 'ckpt'))
        m3.train(_handler(), disc_loss_bounds=(-np.inf, np.inf), **kw)
        assert list(m3.history.index) == [0, 1, 2, 3]
        with open(os.path.join(td, 'ckpt', 'model_params.json')) as f:
            params = json.load(f)
        assert params['meta']['class'] == 'CpuGan'
        assert params['meta']['s_enhance'] == 2
        with pytest.raises(RuntimeError):
            _model().train(_handler(), input_resolution={
                'spatial': '7km', 'temporal': '40min'}, n_epoch=1,
                out_dir=os.path.join(td, 'x_{epoch}'))


def test_one_network_epoch_reads_back_once():
    """only_gen: nothing is resolved until the epoch's last batch is enqueued"""
    from sup3r_amd.compute import LossFuture
    m = _model()
    bh = _handler(n_batches=4)
    resolved = []
    orig = LossFuture.resolve

    def spy(self):
        resolved.append(len(m._compute.log))
        return orig(self)
    LossFuture.resolve = spy
    try:
        with tempfile.TemporaryDirectory() as td:
            m.train(bh, input_resolution=RES, n_epoch=1, train_disc=False,
                    weight_gen_advers=0.0,
                    out_dir=os.path.join(td, 'e_{epoch}'))
    finally:
        LossFuture.resolve = orig
    # the 4 training futures are read after all 4 Adam steps were issued
    assert resolved[:4] == [4, 4, 4, 4]


def test_adaptive_weight_and_update_optimizer():
    """test_train_gan.py:338-386 re-expressed"""
    m = _model()
    assert m.get_weight_update_fraction({'disc_train_frac': 0.2},
                                        'disc_train_frac', (0.5, 0.95),
                                        0.1) == pytest.approx(1.1)
    assert m.get_weight_update_fraction({'disc_train_frac': [0.1, 0.99]},
                                        'disc_train_frac', (0.5, 0.95),
                                        0.1) == pytest.approx(1 / 1.1)
    assert m.get_weight_update_fraction({'disc_train_frac': 0.7},
                                        'disc_train_frac', (0.5, 0.95),
                                        0.1) == 1
    w = m.update_adversarial_weights({'disc_train_frac': 0.2}, 0.05,
                                     (0.9, 0.99), 1e-3, True)
    assert w == pytest.approx(1.05e-3)
    assert m.update_adversarial_weights({'disc_train_frac': 0.2}, 0.05,
                                        (0.9, 0.99), 1e-3, False) == 1e-3
    with tempfile.TemporaryDirectory() as td:
        m.train(_handler(), input_resolution=RES, n_epoch=3,
                weight_gen_advers=1e-3, disc_loss_bounds=(0.0, 0.1),
                adaptive_update_fraction=0.05,
                adaptive_update_bounds=(0.9, 0.99),
                out_dir=os.path.join(td, 'e_{epoch}'))
    # epoch 0: the discriminator sat out the first batch (running loss 0 is at
    # the lower bound): frac 2/3 < 0.9 -> the weight grows; epoch 1: it trained
    # on every batch: frac 1 > 0.99 -> the weight shrinks again
    frac = m.history['disc_train_frac'].values
    wts = m.history['weight_gen_advers'].values
    assert frac[0] == pytest.approx(2 / 3) and frac[1] == 1
    assert wts[1] == pytest.approx(wts[0] * 1.05)
    assert wts[2] == pytest.approx(wts[1] / 1.05)
    it = m.optimizer.iterations
    m.update_optimizer('generator', learning_rate=123.0)
    assert m.optimizer.learning_rate == 123.0
    assert m.optimizer_disc.learning_rate != 123.0
    assert m.optimizer.iterations == it
    m.update_optimizer('all', learning_rate=7.0)
    assert m.optimizer_disc.learning_rate == 7.0


def test_checkpoint_roundtrip_without_forward():
    """a network loaded from disk and saved again without ever running keeps
    its weights (engine.Network holds them until its store is built); an empty
    one refuses to overwrite a checkpoint"""
    from sup3r_amd.engine import Network
    with open(GEN) as f:
        spec = json.load(f)
    net = Network(spec, name='generator')
    from sup3r_amd import spec as S
    table = S.build_plan(net.layers, (1, 4, 4, 4, 2)).params
    rng = np.random.default_rng(0)
    ws = [rng.standard_normal(p['shape']).astype(np.float32) for p in table]
    net.set_weights(ws)
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, 'a.pkl'), os.path.join(td, 'b.pkl')
        net.save(a)
        loaded = Network.load(a)
        assert len(loaded.weights) == len(ws)
        loaded.save(b)
        again = Network.load(b)
        for x, y in zip(again.weights, ws):
            np.testing.assert_array_equal(x, y)
        empty = Network.load(b)
        empty._pending = None
        with pytest.raises(RuntimeError, match='refusing'):
            empty.save(os.path.join(td, 'c.pkl'))


def test_virtual_shards_sum_gradients_like_the_reference():
    """abstract.py:785-841: the batch is split on axis 0, per-shard gradients
    are SUMMED (not averaged), one optimizer step.  Reported details: the
    mean over shards (documented deviation from 'the last GPU's')."""
    from oracle.gan import GanOracle
    from oracle.network import Network as ONet
    m = _model()
    bh = _handler()
    m.set_norm_stats(bh.means, bh.stds)
    m.init_weights(*bh.shapes)
    batch = bh.batches[0]
    w0 = [w.copy() for w in m.generator_weights]
    with open(GEN) as f:
        gspec = json.load(f)
    with open(DISC) as f:
        dspec = json.load(f)
    og, od = ONet(gspec), ONet(dspec)
    og.forward(batch.low_res)
    od.forward(batch.high_res)
    og.set_weights(w0)
    od.set_weights(m.discriminator_weights)
    orc = GanOracle(og, od, loss='MeanAbsoluteError')
    total, losses = None, []
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        loss, _, g = orc.loss_and_grads(batch.low_res[sl], batch.high_res[sl],
                                        1e-3, train_gen=True)
        g = [np.array(x, np.float64) for x in g]
        total = g if total is None else [a + b for a, b in zip(total, g)]
        losses.append(float(loss))
    m.virtual_gpus = 2
    det = m.run_gradient_descent(batch.low_res, batch.high_res,
                                 weight_gen_advers=1e-3, train_gen=True,
                                 train_disc=False, multi_gpu=True)
    for a, b in zip(m._gen.grads_acc, total):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-9)
    assert float(det['loss_gen']) == pytest.approx(np.mean(losses), rel=1e-6)
    # and it is NOT the gradient of the whole batch (that would be the mean)
    _, _, g_full = orc.loss_and_grads(batch.low_res, batch.high_res, 1e-3,
                                      train_gen=True)
    ratio = np.abs(total[0]).sum() / np.abs(g_full[0]).sum()
    assert ratio > 1.5
    with pytest.raises(ValueError, match='does not divide'):
        m.virtual_gpus = 3
        m.run_gradient_descent(batch.low_res, batch.high_res,
                               weight_gen_advers=1e-3, multi_gpu=True)


WORKER = r'''
import os, sys, json, tempfile
sys.path.insert(0, os.environ["S3_ROOT"])
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from sup3r_amd import Sup3rGan
from tests.test_train_host import _model, _handler, RES
from tests.cpu_compute import CpuGanCompute

# every rank draws DIFFERENT initial weights (no seed agreement) ...
Sup3rGan.seed(100 + rank)
m = _model()
bh = _handler(n_batches=4)          # ... but sees the same batches
# a gating threshold inside the spread of the per-shard disc losses: ranks
# deciding from their own shard would take different branches and dead-lock
# on mismatched collectives
td = tempfile.mkdtemp()
m.train(bh, input_resolution=RES, n_epoch=2, weight_gen_advers=1e-2,
        disc_loss_bounds=(0.6931, 0.6935), multi_gpu=True,
        out_dir=os.path.join(td, "r%d_{epoch}" % rank))
log = m._compute.log
assert log[0] == ("broadcast", 0), log[:3]
flat = np.concatenate([w.ravel() for w in m.weights]).astype(np.float64)
t = torch.from_numpy(flat.copy())
gathered = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
for g in gathered:
    assert torch.equal(g, gathered[0]), "replicas diverged"
# identical sequence of collectives / steps on every rank
codes = {"broadcast": 0, "allreduce": 1, "adam": 2, "gen": 0, "disc": 1, 0: 0}
seq = torch.tensor([codes[a] * 2 + codes[b] for a, b in log], dtype=torch.int64)
n = torch.tensor([len(seq)])
ns = [torch.zeros_like(n) for _ in range(world)]
dist.all_gather(ns, n)
assert all(int(x) == int(ns[0]) for x in ns), ns
seqs = [torch.zeros_like(seq) for _ in range(world)]
dist.all_gather(seqs, seq)
assert all(torch.equal(s, seqs[0]) for s in seqs)
hist = m.history[["train_loss_gen", "train_loss_disc", "gen_train_frac",
                  "disc_train_frac"]].values.astype(np.float64)
h = torch.from_numpy(hist.copy())
hs = [torch.zeros_like(h) for _ in range(world)]
dist.all_gather(hs, h)
assert all(torch.equal(x, hs[0]) for x in hs), "ranks report different losses"
if rank == 0:
    # the same run in ONE process walking both shards reaches the same weights
    Sup3rGan.seed(100)
    ref = _model()
    ref.virtual_gpus = 2
    ref.train(_handler(n_batches=4), input_resolution=RES, n_epoch=2,
              weight_gen_advers=1e-2, disc_loss_bounds=(0.6931, 0.6935),
              multi_gpu=True, out_dir=os.path.join(td, "ref_{epoch}"))
    rflat = np.concatenate([w.ravel() for w in ref.weights])
    assert np.allclose(rflat, flat, rtol=1e-4, atol=1e-6), \
        float(np.abs(rflat - flat).max())
    np.testing.assert_allclose(
        ref.history["train_loss_disc"].values.astype(float), hist[:, 1],
        rtol=1e-5)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_world_size_2_training_over_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, S3_ROOT=ROOT, OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=240)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count('ok') >= 2, res.stdout


def test_loss_window_keeps_inf_like_pandas():
    """LossWindow.means drops NaN only (pandas skipna): a diverged batch with
    an inf loss stays visible to the train / skip gating (base.py:1161-1164)"""
    from sup3r_amd.ledger import LossWindow
    w = LossWindow('train_')
    w.resize(4)
    w.push({'loss_disc': 0.5, 'loss_gen': 1.0})
    w.push({'loss_disc': float('inf'), 'loss_gen': float('nan')})
    m = w.means()
    assert m['train_loss_disc'] == float('inf')
    assert m['train_loss_gen'] == 1.0
