"""Recorded training steps (sup3r_amd/captured.py): a replayed hipGraph of
``Sup3rGan._train_batch`` (sup3r/models/base.py:944-1031) must be the eager
step — same weights bit for bit, same loss details, same optimizer counters —
on the reference's own test shape (tests/training/test_train_gan.py:45-114:
gen_2x_2f + disc_s_same, batch 15 of 5 x 5 -> 10 x 10)."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')
LR, HR = (15, 5, 5, 2), (15, 10, 10, 2)

# (train_gen, only_gen, gen_too_good, train_disc, only_disc, disc_too_good):
# both networks; the generator alone while the discriminator sits out; both
# again (the discriminator's packed filters are then one step behind the
# generator-only batches unless the replay re-packs them)
BOTH = (True, False, False, True, False, False)
GEN = (True, False, False, True, False, True)
DISC = (True, False, True, True, False, False)
SCHEDULE = [BOTH] * 4 + [GEN] * 3 + [BOTH] * 2 + [DISC] * 3 + [BOTH, GEN, BOTH]


def _model(optimizer=None):
    from sup3r_amd import Sup3rGan
    Sup3rGan.seed(11)
    kw = {}
    if optimizer:
        kw = {'optimizer': optimizer, 'optimizer_disc': optimizer}
    m = Sup3rGan(os.path.join(CFG, 'gen_2x_2f.json'),
                 os.path.join(CFG, 'disc_s_same.json'),
                 loss='MeanAbsoluteError', precision='bf16',
                 learning_rate=1e-3, **kw)
    m.init_weights(LR, HR)
    return m


def _run(capture, schedule=SCHEDULE, optimizer=None, host_batches=False,
         generate_at=()):
    import torch
    from sup3r_amd.engine import Device
    m = _model(optimizer)
    m.capture_steps = capture
    dev = Device.get()
    rng = np.random.default_rng(5)
    details, gens = [], []
    for i, flags in enumerate(schedule):
        lo = rng.standard_normal(LR).astype(np.float32)
        hi = rng.standard_normal(HR).astype(np.float32)

        class Batch:
            low_res = lo if host_batches else dev.to_device(lo)
            high_res = hi if host_batches else dev.to_device(hi)
        d = m._train_batch(Batch, *flags, 1e-2)
        details.append({k: float(v) for k, v in d.items()})
        if i in generate_at:      # an inference between two replays
            gens.append(np.asarray(m.generate(lo, norm_in=False,
                                              un_norm_out=False)))
    torch.cuda.synchronize()
    w = [np.array(a) for a in m.generator.weights] + \
        [np.array(a) for a in m.discriminator.weights]
    its = (int(m.optimizer.iterations), int(m.optimizer_disc.iterations))
    rec = getattr(m, '_recorder', None)
    return w, details, its, gens, rec


def test_replayed_steps_are_the_eager_steps_bit_for_bit():
    w0, d0, it0, g0, rec0 = _run(False, generate_at=(5, 13))
    with warnings.catch_warnings():
        warnings.simplefilter('error')          # a failed capture warns
        w1, d1, it1, g1, rec1 = _run(True, generate_at=(5, 13))
    assert rec0 is None and rec1 is not None
    # three keys (both / generator only / discriminator only), two eager runs
    # each before the record
    assert rec1.replays == len(SCHEDULE) - 6, rec1.replays
    nodes = sorted(e['rec'].nodes for e in rec1._entries.values())
    assert len(nodes) == 3 and nodes[0] > 30, nodes   # (round 4: one launch per fewpos conv pass)
    assert it0 == it1 == (sum(f != DISC for f in SCHEDULE),
                          sum(f != GEN for f in SCHEDULE))
    assert all(np.isfinite(a).all() for a in w0)
    assert d0 == d1
    for a, b in zip(g0, g1):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)


def test_auto_mode_records_small_batches_from_host_arrays():
    """'auto' (the default) records the C1 shape; numpy batches go straight
    into the recorded input buffers."""
    w0, d0, _, _, _ = _run(False, schedule=[BOTH] * 6)
    w1, d1, _, _, rec = _run('auto', schedule=[BOTH] * 6, host_batches=True)
    assert rec is not None and rec.replays == 4
    assert d0 == d1
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('name', ['SGD', 'RMSprop', 'Adagrad', 'Adamax',
                                  'AdamW'])
def test_every_optimizer_replays(name):
    """the step scalars of every keras optimizer (models/utilities.py:150-158)
    reach the recorded update launch through ``s3_optimizer_stage``"""
    opt = {'name': name, 'learning_rate': 1e-3}
    if name == 'SGD':
        opt['momentum'] = 0.9
    w0, d0, it0, _, _ = _run(False, schedule=[BOTH] * 5, optimizer=opt)
    w1, d1, it1, _, rec = _run(True, schedule=[BOTH] * 5, optimizer=opt)
    assert rec.replays == 3 and it0 == it1 == (5, 5)
    assert d0 == d1
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)


def test_large_batches_and_structured_losses_stay_eager():
    from sup3r_amd import Sup3rGan
    from sup3r_amd.captured import StepRecorder
    m = _model()
    rec = StepRecorder(m._compute)

    class Small:
        low_res = np.zeros(LR, np.float32)
        high_res = np.zeros(HR, np.float32)

    class Large:
        low_res = np.zeros((8, 16, 16, 24, 4), np.float32)
        high_res = np.zeros((8, 80, 80, 288, 2), np.float32)
    assert rec.eligible(m, Small, 'auto', False)
    assert not rec.eligible(m, Large, 'auto', False)
    assert rec.eligible(m, Large, True, False)
    assert not rec.eligible(m, Small, 'auto', True)        # multi_gpu
    assert not rec.eligible(m, Small, False, False)
    Sup3rGan.seed(1)
    m2 = Sup3rGan(os.path.join(CFG, 'gen_2x_2f.json'),
                  os.path.join(CFG, 'disc_s_same.json'),
                  loss={'SpatialExtremesLoss': {}}, precision='bf16')
    assert not StepRecorder(m2._compute).eligible(m2, Small, True, False)


def test_condmom_recorded_steps_are_the_eager_steps_bit_for_bit():
    """``Sup3rCondMom._train_step`` (the loop body of
    /root/reference/sup3r/models/conditional.py:363-489) at BASELINE.md's C5
    shape — lr (4, 4, 4, 4, 2) -> (4, 12, 12, 16, 2), masked MSE: replayed
    hipGraph steps == eager steps (weights bit for bit, loss values, optimizer
    counter), the batch's three fields (low_res / output / mask) each through
    its recorded input buffer"""
    import types
    import torch
    from sup3r_amd import Sup3rCondMom
    from sup3r_amd.engine import Device
    lr_s, hr_s = (4, 4, 4, 4, 2), (4, 12, 12, 16, 2)

    def run(capture):
        Sup3rCondMom.seed(7)
        m = Sup3rCondMom(os.path.join(CFG, 'gen_3x_4x_2f.json'),
                         precision='bf16', learning_rate=1e-3)
        m.capture_steps = capture
        m.init_weights(lr_s, hr_s)
        dev = Device.get()
        rng = np.random.default_rng(9)
        losses = []
        for i in range(7):
            b = types.SimpleNamespace(
                low_res=dev.to_device(rng.standard_normal(lr_s).astype(np.float32)),
                output=dev.to_device(rng.standard_normal(hr_s).astype(np.float32)),
                mask=dev.to_device((rng.uniform(size=hr_s) > 0.3).astype(np.float32)))
            d = m._train_step(b).resolve()
            losses.append(float(d['loss_gen']))
        torch.cuda.synchronize()
        return ([np.array(a) for a in m.generator.weights], losses,
                int(m.optimizer.iterations), getattr(m, '_recorder', None))
    w0, l0, it0, rec0 = run(False)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        w1, l1, it1, rec1 = run(True)
    # (two eager runs of the key, then the record + replays)
    assert rec0 is None and rec1 is not None and rec1.replays == 7 - 2, \
        (rec1 and rec1.replays)
    assert it0 == it1 == 7
    assert l0 == l1, (l0, l1)
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)
