"""GPU parity tests (``-m gpu``) added in round 3: the TRAINING step that
``bench.py --mode train`` times, at its size, against the oracle.

* ``gen_5x_12x_2f`` backward in bf16 at batch 8 (the benched geometry): per-op
  forward, data gradient and every weight gradient against the oracle that
  rounds where the device rounds and uses the device's LeakyReLU masks;
  kernel selection asserted (persistent data gradient, wave-specialised trunk
  weight gradient, chunked 64 -> 200 data gradient);
* one C2 ``Sup3rGan._train_batch`` — generator step, then discriminator step
  with the UPDATED generator — in (i) exact fp32 at batch 1 under the device's
  masks (1e-3) and (ii) **bf16 at batch 8** (one sample replicated; the
  oracle walks that sample teacher-forced, see ``helpers.
  teacher_forced_check``): gradients of both steps, loss scalars, Adam ``m`` /
  ``v`` slots after the step, and the weights after the step against the keras
  update applied to the device's own gradient;
* a bf16 ``Sup3rCondMom`` step (C5).

Reference: sup3r/models/base.py:944-1031 (``_train_batch``),
sup3r/models/abstract.py:1190-1238 (``get_single_grad``), :843-914
(``run_gradient_descent``), sup3r/models/conditional.py:221-283.

Every tolerance is stated where it is asserted.
"""
import json
import os

import numpy as np
import pytest

from tests.helpers import emulate_plan, rel_linf, teacher_forced_check

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _load(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


def _conv_kernels(ph, field):
    from sup3r_amd import spec as S
    return [ph.op_info(i)[field] for i, op in enumerate(ph.plan.ops)
            if op['kind'] == S.OP_CONV]


# ----------------------------------------------------------------- snapshots
class Snapshot:
    """Host copy of what a training plan holds after a pass: every op's
    output tensor (batch entries ``sample`` only), its storage type and the
    kernel selection — the subset of the ``PlanHandle`` interface that
    ``helpers.emulate_plan`` / ``teacher_forced_check`` use.  Needed because
    the discriminator plan of the generated field is overwritten by the
    discriminator step of the same ``_train_batch``."""

    def __init__(self, ph, sample=slice(0, 1)):
        from sup3r_amd import spec as S
        self.plan, self.precision = ph.plan, ph.precision
        self._t, self._is16, self._info = {}, {}, {}
        for i, op in enumerate(ph.plan.ops):
            tid = op['out']
            self._t[tid] = np.array(ph.tensor(tid)[sample])
            self._is16[tid] = ph.tensor_is_bf16(tid)
            if op['kind'] == S.OP_CONV:
                self._info[i] = ph.op_info(i)

    def tensor(self, tid):
        return self._t[tid]

    def tensor_is_bf16(self, tid):
        return self._is16[tid]

    def op_info(self, i):
        return self._info[i]

    @property
    def output(self):
        return self._t[self.plan.output]


class Stacked(Snapshot):
    """Several snapshots of plans of the SAME network seen as one batch (the
    reference runs the discriminator on hi_res_true and hi_res_gen in one
    ``calc_loss``; the device keeps one plan per field)."""

    def __init__(self, parts):
        first = parts[0]
        self.plan, self.precision = first.plan, first.precision
        self._is16, self._info = first._is16, first._info
        self._t = {tid: np.concatenate([p._t[tid] for p in parts], axis=0)
                   for tid in first._t}


def _assert_per_op(stats, what, frac16=1e-2, frac32=1e-3):
    """tests/test_parity_r02.py::_assert_per_op"""
    worst = max(stats, key=lambda d: d['frac'])
    print(f'{what}: {len(stats)} ops, worst mismatch fraction '
          f'{worst["frac"]:.2e} (op {worst["op"]})')
    for d in stats:
        assert d['excess'] <= (d['noise'] if d['bf16'] else 5 * d['noise']), d
        assert d['frac'] < (frac16 if d['bf16'] else frac32), d


def _grad_errors(got, ref, scale=1.0):
    """max |g - g_ref| per tensor relative to max |g_ref| (tensors whose
    gradient is numerically nothing compare at the round-off of the large
    ones)"""
    ref = [np.asarray(r, np.float64) * scale for r in ref]
    gmax = max(float(np.abs(r).max()) for r in ref)
    return [float(np.abs(np.asarray(g, np.float64) - r).max()
                  / max(float(np.abs(r).max()), 1e-3 * gmax))
            for g, r in zip(got, ref)]


# ------------------------------------------- (a) C2 generator backward, bf16
def test_c2_generator_bf16_backward_batch8_under_device_masks():
    """``gen_5x_12x_2f`` at lo-res (8, 16, 16, 24, 4), bf16: the backward pass
    of the benched training step.  Forward per op (teacher forced) + 3e-2 end
    to end, then the data gradient and all 76 weight gradients against the
    oracle on the device's activations and masks, rounding where the device
    rounds: <= 2e-2 of each tensor's largest gradient.  The production
    kernels are the ones under test."""
    from tests.test_parity_r02 import _fwd_bwd_vs_oracle
    spec = _load('gen_5x_12x_2f.json')
    ph = _fwd_bwd_vs_oracle(spec, (8, 16, 16, 24, 4), 'bf16', 21, 3e-2, 2e-2,
                            replicate=True)
    fwd, wg, dg = (_conv_kernels(ph, f) for f in ('fwd', 'wgrad', 'dgrad'))
    print('fwd', sorted(set(fwd)), 'wgrad', sorted(set(wg)), 'dgrad',
          sorted(set(dg)))
    assert fwd.count('mfma_persist') >= 34, fwd
    # 33 body convs + the T = 96 head conv on the wave-specialised /
    # transpose-read trunk weight gradient
    assert wg.count('bf16_trunk') >= 34, wg
    assert 'mfma_chunked' in dg, dg                 # 64 -> 200
    assert dg.count('mfma_frame') >= 33, dg
    # the data gradients ran on the persistent kernel (a run-time choice)
    assert ph.dev.stat('persist_dgrad') >= 33, ph.dev.stat('persist_dgrad')


# ------------------------------------------------ (b, c) one C2 _train_batch
def _train_batch_vs_oracle(precision, batch, tol_g, tol_loss):
    """One ``_train_batch`` of the C2 GAN (``gen_5x_12x_2f`` + ``disc_st``,
    MeanAbsoluteError content loss, weight_gen_advers 1e-2) on the device
    against ``GanOracle.loss_and_grads`` + keras Adam.

    The device batch is ``batch`` copies of ONE sample; the oracle runs that
    sample (the losses are batch means and the relativistic terms use batch
    means of the logits, so the scalars AND the gradients of the replicated
    batch equal those of the single sample).  A hook on the optimizer step
    snapshots what the generator step left on the device before the
    discriminator step overwrites it.

    fp32: the oracle runs its own forward passes; the device's masks are
    installed before its backward passes.  bf16: the oracle's forward is the
    teacher-forced walk (checked per op), so its backward passes run over the
    device's activations with the device's roundings."""
    from oracle.gan import GanOracle
    from oracle.network import Network as ONet
    from sup3r_amd import Sup3rGan
    rng = np.random.default_rng(77)
    gspec, dspec = _load('gen_5x_12x_2f.json'), _load('disc_st.json')
    lr1 = rng.standard_normal((1, 16, 16, 24, 4)).astype(np.float32)
    hr1 = rng.standard_normal((1, 80, 80, 288, 2)).astype(np.float32)
    lr, hr = np.repeat(lr1, batch, 0), np.repeat(hr1, batch, 0)
    og, od = ONet(gspec), ONet(dspec)
    og.init_weights(lr1[:, :6, :6, :6], seed=5, bias_scale=0.05)
    od.init_weights(hr1, seed=6, bias_scale=0.05)
    step, w_adv = 1e-4, 1e-2
    m = Sup3rGan(os.path.join(CFG, 'gen_5x_12x_2f.json'),
                 os.path.join(CFG, 'disc_st.json'), loss='MeanAbsoluteError',
                 learning_rate=step, precision=precision)
    m.init_weights(lr.shape, hr.shape)
    m.generator.set_weights(og.weights)
    m.discriminator.set_weights(od.weights)
    w0 = {'gen': [w.copy() for w in og.weights],
          'disc': [w.copy() for w in od.weights]}
    gph = m.generator.plan(lr.shape, training=True)
    dph_t = m.discriminator.plan(hr.shape, training=True, slot=0)
    dph_g = m.discriminator.plan(hr.shape, training=True, slot=1)
    bf16 = precision == 'bf16'
    if bf16:
        # the benched kernels are the ones under test
        assert _conv_kernels(gph, 'fwd').count('mfma_persist') >= 34
        assert _conv_kernels(gph, 'wgrad').count('bf16_trunk') >= 34
        assert 'c2' in _conv_kernels(dph_g, 'dgrad')
        assert 's2' in _conv_kernels(dph_g, 'dgrad')

    snap = {}
    apply = m._compute.apply

    def spy(which, optimizer):
        # (enqueued after the step's backward pass: reading syncs the stream)
        net = m.generator if which == 'gen' else m.discriminator
        snap[which + '_grads'] = net.grads
        if which == 'gen':
            snap['G'] = Snapshot(gph)
            snap['D_gen1'] = Snapshot(dph_g)
        return apply(which, optimizer)
    m._compute.apply = spy

    class B:
        low_res, high_res = lr, hr
    got = m._train_batch(B, True, False, False, True, False, False, w_adv)
    D_true, D_gen2 = Snapshot(dph_t), Snapshot(dph_g)
    # the generated field of the discriminator step (inference plan, updated
    # generator) — deterministic, so running it again gives the same bits
    hr_gen2 = m.generator.plan(lr.shape, training=False).forward(
        m.generator.dev.to_device(lr)).cpu().numpy()[:1]
    np.testing.assert_array_equal(
        m.generator.plan(lr.shape, training=False).forward(
            m.generator.dev.to_device(lr)).cpu().numpy()[:1], hr_gen2)

    orc = GanOracle(og, od, loss='MeanAbsoluteError', learning_rate=step)
    walks = {}

    def forced(ref, snapshot, what):
        def run(x, *exo):
            if walks.setdefault(what, 0) == 0:
                emulate_plan(ref, snapshot, masks=False)
            walks[what] += 1
            stats = teacher_forced_check(ref, snapshot, x)
            _assert_per_op(stats, f'{precision} {what}, per op')
            return snapshot.output.reshape(x.shape[0], -1).astype(np.float32) \
                if what.startswith('disc') else \
                snapshot.output.astype(np.float32)
        return run

    def masks(*pairs):
        def install():
            for ref, snapshot in pairs:
                emulate_plan(ref, snapshot, masks=True, rounding=False)
        return install

    # ---- generator step (train_gen, compute_disc = train_disc: base.py:1003)
    D1 = Stacked([D_true, snap['D_gen1']])
    if bf16:
        orc.gen_forward = forced(og, snap['G'], 'generator step: gen')
        orc.disc_forward = forced(od, D1, 'disc on [true; gen]')
    else:
        def own_forward_device_output(x, *exo):
            # MeanAbsoluteError's gradient is sign(gen - true) / n: wherever
            # the two fields agree to round-off the sign is as arbitrary as a
            # LeakyReLU mask at zero (~20 of 3.7 M positions here; the last
            # bias gradient is a SUM of those signs, so a handful of flips is
            # 2e-3 of it).  The oracle's own forward is checked, then the loss
            # and the discriminator see the device's field.
            y = og.forward(x, *exo)
            y_dev = snap['G'].output
            e = rel_linf(y_dev, y)
            print(f'generator forward vs oracle: {e:.2e}')
            assert e < 1e-4, e
            return y_dev.astype(np.float32)
        orc.gen_forward = own_forward_device_output
    orc.pre_backward = masks((og, snap['G']), (od, D1))
    _, det1, g_ref = orc.loss_and_grads(lr1, hr1, w_adv, train_gen=True,
                                        train_disc=False, compute_disc=True)
    g_ref = [np.array(g) for g in g_ref]
    errs = _grad_errors(snap['gen_grads'], g_ref)
    print(f'C2 {precision} batch {batch}, generator step: worst gradient '
          f'error {max(errs):.2e} (tensor {int(np.argmax(errs))})')
    assert max(errs) < tol_g, errs
    for k in ('loss_gen', 'loss_gen_content', 'loss_gen_advers', 'loss_disc'):
        ref = float(det1[k])
        if k == 'loss_disc':
            continue          # reported from the discriminator step below
        assert abs(got[k] - ref) < tol_loss * max(1.0, abs(ref)), \
            (k, got[k], ref)
    orc.opt.apply_gradients(g_ref, og.weights)
    _check_adam(m.generator, m.optimizer, orc.opt, snap['gen_grads'],
                w0['gen'], tol_g, 'gen')

    # ---- discriminator step: the generator is the UPDATED one on both sides.
    # The oracle continues from the device's post-step generator weights: for
    # a first Adam step delta w ~ lr * sign(g), so the two weight sets differ
    # wherever a ~0 gradient has another sign — that is the optimizer's
    # conditioning, checked above through m / v and the update formula, not
    # something the discriminator step should inherit.
    og.set_weights(m.generator.weights)
    D2 = Stacked([D_true, D_gen2])
    if bf16:
        orc.gen_forward = lambda x, *exo: hr_gen2
        orc.disc_forward = forced(od, D2, 'disc on [true; gen updated]')
    else:
        y2 = og.forward(lr1)
        e2 = rel_linf(hr_gen2, y2)
        print(f'updated generator, forward vs oracle: {e2:.2e}')
        assert e2 < 1e-4, e2
        orc.gen_forward = lambda x, *exo: y2
    orc.pre_backward = masks((od, D2))
    _, det2, d_ref = orc.loss_and_grads(lr1, hr1, w_adv, train_gen=False,
                                        train_disc=True)
    d_ref = [np.array(g) for g in d_ref]
    errs = _grad_errors(snap['disc_grads'], d_ref)
    print(f'C2 {precision} batch {batch}, discriminator step: worst gradient '
          f'error {max(errs):.2e} (tensor {int(np.argmax(errs))})')
    assert max(errs) < tol_g, errs
    ref = float(det2['loss_disc'])
    assert abs(got['loss_disc'] - ref) < tol_loss * max(1.0, abs(ref)), \
        (got['loss_disc'], ref)
    orc.opt_disc.apply_gradients(d_ref, od.weights)
    _check_adam(m.discriminator, m.optimizer_disc, orc.opt_disc,
                snap['disc_grads'], w0['disc'], tol_g, 'disc')
    return got


def _check_adam(net, opt, opt_ref, dev_grads, w_before, tol, name):
    """Adam after one step: the slots against the oracle's (m is linear, v
    quadratic in the gradient: ``tol`` / 2 ``tol``), and the weights against
    the keras update (abstract.py:899,912; keras-2.15 ``Adam.update_step``)
    evaluated in float64 on the DEVICE's gradient — a first step moves every
    weight by ~lr * sign(g), so comparing against the oracle's weights would
    only compare signs."""
    assert opt.iterations == opt_ref.iterations == 1
    em = _grad_errors(net.slots('m'), opt_ref.m)
    assert max(em) < tol, (name, 'm', em)
    ev = _grad_errors(net.slots('v'), opt_ref.v)
    assert max(ev) < 2.5 * tol, (name, 'v', ev)
    cfg = opt.get_config()
    b1, b2, eps, lr = (cfg['beta_1'], cfg['beta_2'], cfg['epsilon'],
                       cfg['learning_rate'])
    alpha = lr * np.sqrt(1 - b2) / (1 - b1)
    worst = 0.0
    for w, w0, g in zip(net.weights, w_before, dev_grads):
        g = np.asarray(g, np.float64)
        want = w0 - alpha * ((1 - b1) * g) / (np.sqrt((1 - b2) * g * g) + eps)
        worst = max(worst, float(np.abs(w - want).max()))
    print(f'{name}: Adam slots m {max(em):.2e} v {max(ev):.2e}; weights vs '
          f'the keras update of the device gradient: {worst:.2e} (lr {lr})')
    # fp32 evaluation of m / (sqrt(v) + eps): a few ulp of lr, plus one ulp
    # of the weight itself (|w| < 1)
    assert worst < 1e-3 * lr + 2e-7, (name, worst)


def test_c2_train_batch_fp32_under_device_masks():
    """exact-fp32 mode, batch 1: gradients of both steps 1e-4 (measured
    2e-6 / 5e-6; round 2 had 2e-2 with own masks on each side), loss scalars
    1e-4"""
    _train_batch_vs_oracle('f32', 1, 1e-4, 1e-4)


def test_c2_train_batch_bf16x3_under_device_masks():
    """the fp32-class training mode (round 3): one C2 ``_train_batch`` in
    ``precision='bf16x3'`` — trunk forward / data gradient / weight gradient,
    the discriminator's gather-MFMA convs and the few-channel hi-res convs all
    split-bf16 (hi*hi + hi*lo + lo*hi) on the bf16 matrix cores — against the
    fp32 oracle under the device's masks: gradients and loss scalars 1e-4"""
    _train_batch_vs_oracle('bf16x3', 1, 1e-4, 1e-4)


def test_c2_train_batch_bf16_batch8_vs_emulating_oracle():
    """the benched training step (``bench.py --mode train``: bf16, batch 8):
    per-op forward of the generator and of the discriminator on both fields,
    gradients of both steps 1e-2 (measured 1.1e-3 / 3.6e-3), loss scalars
    1e-3, Adam slots"""
    _train_batch_vs_oracle('bf16', 8, 1e-2, 1e-3)


# --------------------------------------------------------- (e) C5 in bf16
def test_condmom_bf16_step_on_the_3x_4x_body():
    """C5 in the throughput mode: ``Sup3rCondMom`` over ``gen_3x_4x_2f`` at
    lo-res (8, 16, 16, 24, 2) (the bf16 MFMA kernels), masked MSE: loss value
    against the teacher-forced oracle output 1e-3, gradients 1e-2
    (conditional.py:221-283)"""
    from oracle.network import Network as ONet
    from sup3r_amd import Sup3rCondMom
    rng = np.random.default_rng(12)
    spec = _load('gen_3x_4x_2f.json')
    nb = 8
    lr1 = rng.standard_normal((1, 16, 16, 24, 2)).astype(np.float32)
    out1 = rng.standard_normal((1, 48, 48, 96, 2)).astype(np.float32)
    mask1 = (rng.uniform(size=out1.shape) > 0.3).astype(np.float32)
    lr, out, mask = (np.repeat(a, nb, 0) for a in (lr1, out1, mask1))
    og = ONet(spec)
    og.init_weights(lr1[:, :6, :6, :6], seed=4, bias_scale=0.1)
    m = Sup3rCondMom(os.path.join(CFG, 'gen_3x_4x_2f.json'), precision='bf16')
    m.init_weights(lr.shape, out.shape)
    m.generator.set_weights(og.weights)
    _, det = m.get_single_grad(lr, out, mask=mask)
    ph = m.generator.plan(lr.shape, training=True)
    assert 'bf16_trunk' in _conv_kernels(ph, 'wgrad')
    emulate_plan(og, ph, masks=False)
    stats = teacher_forced_check(og, ph, lr1, sample=slice(0, 1))
    _assert_per_op(stats, 'CondMom bf16, per op')
    y = ph.tensor(ph.plan.output)[:1]
    d = (y * mask1 - out1 * mask1).astype(np.float64)
    loss_ref = float((d * d).mean())
    assert abs(float(det['loss_gen']) - loss_ref) < 1e-3 * max(1, loss_ref)
    # ... and the exact fp32 oracle agrees at the accuracy of the mode
    y32 = ONet(spec)
    y32.forward(lr1[:, :6, :6, :6])
    y32.set_weights(og.weights)
    assert rel_linf(y, y32.forward(lr1)) < 3e-2
    emulate_plan(og, ph, masks=True, rounding=False, sample=slice(0, 1))
    og.backward((2.0 * d * mask1 / d.size).astype(np.float32))
    errs = _grad_errors(m.generator.grads, og.grads)
    print(f'CondMom bf16 on gen_3x_4x_2f: worst gradient error '
          f'{max(errs):.2e}')
    assert max(errs) < 1e-2, errs          # measured 1.6e-3


# ------------------------------------- (f) the C4 discriminator as benched
def test_c4_discriminator_at_the_benched_shape():
    """``disc_st_same`` (the reference's test discriminator layout with TF
    'same' padding) at the hi-res shape of ``bench.py --mode train --config
    c4``, batch 8 of a replicated sample, bf16: per-op forward, gradients
    under the device masks 2e-2, and the kernels the round-3 widening put
    under it — the stride-2 LDS-halo forward / data gradient on 'same'
    padded convs, transpose-read weight gradients down to 6-long t extents"""
    from tests.test_parity_r02 import _fwd_bwd_vs_oracle
    spec = _load('disc_st_same.json')
    ph = _fwd_bwd_vs_oracle(spec, (8, 48, 48, 96, 2), 'bf16', 23, 3e-2, 2e-2,
                            replicate=True)
    fwd, wg, dg = (_conv_kernels(ph, f) for f in ('fwd', 'wgrad', 'dgrad'))
    print('fwd', fwd, 'wgrad', wg, 'dgrad', dg)
    # (the 64 -> 64 stride-2 layer has too few tiles for its LDS-halo kernel at
    # batch 8; at the benched batch 32 it is on it too)
    assert fwd.count('halo_s2') >= 1 and 'halo32' in fwd, fwd
    assert 's2' in dg, dg
    # (only the last layer — 432 output positions at batch 8 — stays on the
    # exact-fp32 weight gradient)
    assert wg.count('f32_gen') <= 1 and wg.count('bf16_gen') >= 5, wg
