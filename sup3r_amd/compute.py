"""Device-side compute core of Sup3rGan on MI355X: generator + discriminator
networks (``engine.Network``), losses, reverse pass and optimizer steps — every
arithmetic step is a C-ABI call into libsup3r_hip.so.

Replaces, for one (low_res, hi_res_true) mini-batch, the TensorFlow work inside
``AbstractSingleModel.get_single_grad`` (sup3r/models/abstract.py:1190-1238):
``_tf_generate`` -> ``calc_loss`` (base.py:830-911: two ``_tf_discriminate``
calls, content loss, relativistic BCE) -> ``tape.gradient`` — and
``optimizer.apply_gradients`` (abstract.py:899,912).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .engine import Device, Network
from .utilities import LossValue, camel_to_underscore

CONTENT_KINDS = {'MeanAbsoluteError': _lib.LOSS_MAE,
                 'MeanSquaredError': _lib.LOSS_MSE,
                 'ExpLoss': _lib.LOSS_EXP}

# structured content losses of sup3r/utilities/loss_metrics.py that have
# MI355X kernels (kernels_loss.hip): name -> (feature map, metric on the map)
STRUCTURED_KINDS = {
    'SpatialDerivativeLoss': ('deriv_s', _lib.LOSS_MAE),
    'TemporalDerivativeLoss': ('deriv_t', _lib.LOSS_MAE),
    'MaterialDerivativeLoss': ('material', _lib.LOSS_MAE),
    'CoarseMseLoss': ('mean_s', _lib.LOSS_MSE),
    'SpatialExtremesLoss': ('ext_s', _lib.LOSS_MAE),
    'TemporalExtremesLoss': ('ext_t', _lib.LOSS_MAE),
    'LowResLoss': ('lowres', None),
    'MmdLoss': ('mmd', None),
    'SlicedWassersteinLoss': ('sw', None),
    'SpatialFftLoss': ('fft_s', _lib.LOSS_MAE),
    'SpatiotemporalFftLoss': ('fft_st', _lib.LOSS_MAE),
}
LMAP = {'deriv_s': 0, 'deriv_t': 1, 'material': 2, 'mean_s': 3, 'ext_s': 4,
        'ext_t': 5, 'coarsen': 6}
SLOTS_PER_TERM = 4          # scalar slots a term may use (value = sum coef * slot)
MAX_TERMS = 14


def _lowres_kwargs(kw):
    """LowResLoss.__init__ (loss_metrics.py:500-540)"""
    out = {'s_enhance': 1, 't_enhance': 1, 't_method': 'average',
           'tf_loss': 'MeanSquaredError', 'ex_loss': None}
    for k, v in kw.items():
        if k not in out:
            raise TypeError(f"LowResLoss got an unexpected keyword '{k}'")
        out[k] = v
    out['t_method'] = str(out['t_method']).casefold()
    if out['tf_loss'] not in ('MeanSquaredError', 'MeanAbsoluteError'):
        raise KeyError(f"LowResLoss tf_loss \"{out['tf_loss']}\" has no MI355X "
                       "kernel (MeanSquaredError | MeanAbsoluteError)")
    if out['ex_loss'] not in (None, 'SpatialExtremesLoss',
                              'TemporalExtremesLoss'):
        raise KeyError(out['ex_loss'])       # EX_LOSS_METRICS lookup
    return out


def parse_loss_spec(loss):
    """``get_loss_fun`` spec handling (abstract.py:461-502): str | dict with
    optional ``term_weights``.  Returns [(name, kind, weight, kwargs), ...];
    kind is a pointwise S3_LOSS_* code or the name of a feature map."""
    spec = {loss: {}} if isinstance(loss, str) else dict(loss)
    names = [k for k in spec if k != 'term_weights']
    weights = spec.get('term_weights', [1.0] * len(names))
    if len(names) > MAX_TERMS:
        raise ValueError(f'at most {MAX_TERMS} loss terms')
    terms = []
    for n, w in zip(names, weights):
        kw = dict(spec[n] or {})
        if n in CONTENT_KINDS:
            if kw:
                raise TypeError(f'loss "{n}" takes no kwargs: {kw}')
            terms.append((n, CONTENT_KINDS[n], float(w), {}))
        elif n in STRUCTURED_KINDS:
            if n == 'LowResLoss':
                kw = _lowres_kwargs(kw)
            elif n == 'MmdLoss':
                if set(kw) - {'sigma'}:
                    raise TypeError(f'MmdLoss kwargs: {kw}')
            elif n == 'SlicedWassersteinLoss':
                if set(kw) - {'n_projections'}:
                    raise TypeError(f'SlicedWassersteinLoss kwargs: {kw}')
            elif kw:
                raise TypeError(f'loss "{n}" takes no kwargs: {kw}')
            terms.append((n, STRUCTURED_KINDS[n][0], float(w), kw))
        else:
            raise KeyError(
                'Could not find requested loss function "{}" among the '
                'content losses with an MI355X kernel ({}).'.format(
                    n, list(CONTENT_KINDS) + list(STRUCTURED_KINDS)))
    return terms


def details_from_scalars(vals, loss_terms, coefs, with_disc, train_gen,
                         with_advers, weight_gen_advers):
    """``loss_details`` of ``Sup3rGan.calc_loss`` (base.py:903-911) from the
    loss-scalar buffer: [0] loss_disc, [1] adversarial term, [4 + 4 i ...] the
    slots of content term i (value = sum coef * slot)."""
    details = {}
    if with_disc:
        details['loss_disc'] = LossValue(vals[0])
    if train_gen:
        content = 0.0
        for i, (name, kind, w, kw) in enumerate(loss_terms):
            slot = 4 + SLOTS_PER_TERM * i
            val = sum(cf * float(vals[slot + j])
                      for j, cf in enumerate(coefs[i]))
            details[camel_to_underscore(name)] = LossValue(val)
            content += w * val
        advers = float(vals[1]) if with_advers else 0.0
        details['loss_gen_content'] = LossValue(content)
        details['loss_gen_advers'] = LossValue(advers)
        details['loss_gen'] = LossValue(content + weight_gen_advers * advers)
    return details


class LossFuture:
    """Loss scalars of one ``loss_and_grads`` call that are still on the
    device.  ``resolve()`` does the one device -> host read (a stream sync) and
    returns the ``loss_details`` dict; until then the host keeps enqueuing the
    all-reduce, the Adam step and the next network's work."""

    def __init__(self, scal, recipe, scale=1.0, dev=None):
        self._scal, self._recipe, self._scale = scal, recipe, float(scale)
        self._details = None
        self._dev = dev

    def resolve(self):
        if self._details is None:
            # with replicas the step behind these scalars went through RCCL: a
            # peer that never arrives must not park this host in a blocking
            # copy for ever (Device.wait: deadline + communicator abort)
            if self._dev is not None and self._dev.nranks > 1:
                self._dev.wait()
            vals = self._scal.cpu().numpy().astype(np.float64) * self._scale
            self._details = self._recipe(vals)
            self._scal = None
        return self._details


class HipGanCompute:
    """Generator / discriminator pair on one GPU."""

    supports_defer = True
    # tests switch the sharing of D(hi_res_true) inside a mini-batch off to
    # compare against recomputing it
    share_dtrue_allowed = True

    def __init__(self, gen_layers, disc_layers, device=None, precision=None):
        self.dev = device or Device.get()
        self.gen = Network(gen_layers, 'generator', self.dev, precision)
        self.disc = None if disc_layers is None else Network(
            disc_layers, 'discriminator', self.dev, precision)
        self._scal = None

    # ---------------------------------------------------------------- utils
    def _zero(self, scal):
        rc = _lib.lib().s3_fill(self.dev.ctx, self._ptr(scal), scal.numel(),
                                0.0)
        _lib.check(rc, self.dev.ctx, 's3_fill')
        return scal

    def _scalars(self):
        """the shared, zeroed loss-scalar buffer of synchronous calls"""
        if self._scal is None:
            self._scal = self.dev.empty((4 + SLOTS_PER_TERM * MAX_TERMS,))
        return self._zero(self._scal)

    def new_scalars(self):
        """a private, zeroed loss-scalar buffer (deferred reads outlive the
        call; the shards of a split mini-batch accumulate into one)"""
        return self._zero(self.dev.empty((4 + SLOTS_PER_TERM * MAX_TERMS,)))

    def add_scalars(self, dst, src):
        """dst += src on the device (shards of a split mini-batch)"""
        self._copy_channels(src, 0, dst, 0, 1, accumulate=True)

    def allreduce_scalars(self, scal):
        """SUM of a loss-scalar buffer over the data-parallel ranks"""
        rc = _lib.lib().s3_allreduce_sum(self.dev.ctx, self._ptr(scal),
                                         scal.numel())
        _lib.check(rc, self.dev.ctx, 's3_allreduce_sum')

    def _ptr(self, t, offset=0):
        return C.c_void_p(t.data_ptr() + 4 * offset)

    def _copy_channels(self, src, c0_src, dst, c0_dst, nc, accumulate=False):
        if src.dim() == 1:           # flat buffers: one "channel" per element
            src, dst = src.view(-1, 1), dst.view(-1, 1)
        n_pos = src.numel() // src.shape[-1]
        rc = _lib.lib().s3_copy_channels(
            self.dev.ctx, self._ptr(src), src.shape[-1], c0_src,
            self._ptr(dst), dst.shape[-1], c0_dst, nc, n_pos,
            int(accumulate))
        _lib.check(rc, self.dev.ctx, 's3_copy_channels')

    def exo_from_true(self, hi_res_true, exo_names):
        """get_hr_exo_input (abstract.py:415-436): the trailing channels of the
        truth are the hi-res exogenous features, one per exo layer."""
        exo = {}
        k = len(exo_names)
        c = hi_res_true.shape[-1]
        for i, name in enumerate(exo_names):
            e = self.dev.empty(tuple(hi_res_true.shape[:-1]) + (1,))
            self._copy_channels(hi_res_true, c - k + i, e, 0, 1)
            exo[name] = e
        return exo

    # -------------------------------------------------------------- forward
    def tf_generate(self, low_res, hi_res_exo=None, training=False):
        x = self.dev.to_device(low_res)
        exo = {k: self.dev.to_device(v) for k, v in (hi_res_exo or {}).items()}
        ph = self.gen.plan(tuple(x.shape), training=training)
        return ph.forward(x, exo)

    def tf_discriminate(self, hi_res, training=False, slot=0):
        x = self.dev.to_device(hi_res)
        ph = self.disc.plan(tuple(x.shape), training=training, slot=slot)
        return ph.forward(x)

    def _disc_true(self, dph, hr_true, training):
        """D(hi_res_true).  Within one ``_train_batch`` the generator step and
        the discriminator step evaluate it on the same tensor with the same
        discriminator weights (only the generator is updated in between,
        base.py:1001-1025): the second evaluation — and, in a training plan,
        its saved activations — is the first one, bit for bit, so it is
        reused.  Only inside one ``Sup3rGan._launch_batch`` (``share_dtrue``;
        the entry is dropped when the mini-batch ends, so a producer that
        refills a pre-allocated hi-res tensor through raw pointers between
        batches can never be served stale activations), and there keyed on
        the tensor's storage, torch's in-place version counter, the weights'
        version and the plan; anything else recomputes."""
        key = (hr_true.data_ptr(), getattr(hr_true, '_version', None),
               tuple(hr_true.shape), self.disc.weights_version, id(dph),
               bool(training))
        share = getattr(self, 'share_dtrue', False) and \
            self.share_dtrue_allowed
        hit = getattr(self, '_dtrue', None)
        if share and hit is not None and hit[0] == key:
            return hit[1]
        out = dph.forward(hr_true)
        # (the input tensor is kept alive so its address cannot be recycled)
        self._dtrue = (key, out, hr_true) if share else None
        return out

    def batch_scope(self, active):
        """``True`` at the start, ``False`` at the end of one mini-batch: the
        window inside which D(hi_res_true) may be shared between the
        generator step and the discriminator step"""
        self.share_dtrue = bool(active)
        self._dtrue = None

    # --------------------------------------------------- loss (+ gradients)
    def loss_and_grads(self, low_res, hi_res_true, loss_terms,
                       weight_gen_advers=0.001, train_gen=True,
                       train_disc=False, compute_disc=False, exo_names=(),
                       backward=True, hi_res_gen=None, mask=None,
                       accumulate_wgrad=False, scal=None, defer=False,
                       extra_exo=None, obs=None, overlap_bucket=None):
        """One ``_get_hr_exo_and_loss`` + ``tape.gradient``.

        ``overlap_bucket`` (bytes): the trained network's gradients are final
        after this call — its last backward pass hands them to RCCL bucket by
        bucket while it runs (``s3_params_arm_allreduce``); the caller's
        ``allreduce_grads`` then only joins the streams.

        ``extra_exo``: further named generator inputs (the sparse observation
        fields of ``Sup3rGanWithObs``).  ``obs`` = (observed-cell mask as a
        0 / 1 host array over the output features, kind of the observation
        loss, its weight): adds ``loss_obs`` / ``loss_non_obs`` /
        ``obs_frac`` and, weighted, the observation term to the generator loss
        and its gradient (with_obs.py:248-279).

        ``backward=False`` evaluates ``calc_loss`` only (validation).  When
        ``hi_res_gen`` is given (public ``calc_loss(hi_res_true, hi_res_gen)``)
        the generator forward is skipped.  Gradients of the trained network are
        left in its device gradient buffer (``accumulate_wgrad``: ADDED to what
        is there — the next shard of a split mini-batch, abstract.py:785-805).
        ``scal``: device scalar buffer to accumulate the loss values into (not
        zeroed here); ``defer``: return a ``LossFuture`` in place of the
        details and do not synchronise.  Returns (loss, details, hr_gen).
        """
        L = _lib.lib()
        dev = self.dev
        hr_true = dev.to_device(hi_res_true)
        n_exo = len(exo_names)
        c_true = hr_true.shape[-1]
        gen_train = bool(backward and train_gen)
        disc_train = bool(backward and train_disc and not train_gen)
        obs_info = None
        if hi_res_gen is None:
            lr = dev.to_device(low_res)
            exo = self.exo_from_true(hr_true, list(exo_names))
            for name, arr in (extra_exo or {}).items():
                exo[name] = dev.to_device(arr)
            gph = self.gen.plan(tuple(lr.shape), training=gen_train)
            hr_gen = gph.forward(lr, exo)
        else:
            gph = None
            hr_gen = dev.to_device(hi_res_gen)
        c_gen = hr_gen.shape[-1]
        # _combine_loss_input (abstract.py:438-459)
        if c_true > c_gen:
            gen_full = dev.empty(tuple(hr_true.shape))
            self._copy_channels(hr_gen, 0, gen_full, 0, c_gen)
            self._copy_channels(hr_true, c_gen, gen_full, c_gen,
                                c_true - c_gen)
        else:
            gen_full = hr_gen
        if tuple(gen_full.shape) != tuple(hr_true.shape):
            raise RuntimeError(
                'The tensor shapes of the synthetic output {} and true high '
                'res {} did not have matching shape! Check the '
                'spatiotemporal enhancement multipliers in your your model '
                'config and data handlers.'.format(tuple(gen_full.shape),
                                                   tuple(hr_true.shape)))
        c_used = c_true - n_exo
        mask_d = None
        if mask is not None:
            # Sup3rCondMom: loss(gen * mask, true * mask) over ALL channels
            # (conditional.py:221-241)
            mask_d = dev.to_device(mask)
            c_used = c_true
        if scal is None:
            scal = self.new_scalars() if defer else self._scalars()
        need_disc = self.disc is not None
        dph_t = dph_g = None
        if need_disc:
            # the reference always runs the disc on both (base.py:884-885)
            tr = gen_train or disc_train
            dph_t = self.disc.plan(tuple(hr_true.shape), training=tr, slot=0)
            dph_g = self.disc.plan(tuple(hr_true.shape), training=tr, slot=1)
            d_true = self._disc_true(dph_t, hr_true, tr)
            d_gen = dph_g.forward(gen_full)
            nb = d_true.numel()
        if need_disc and (compute_disc or train_disc):
            g_t = dev.empty((nb,)) if disc_train else None
            g_g = dev.empty((nb,)) if disc_train else None
            rc = L.s3_loss_rel_bce(
                dev.ctx, self._ptr(d_true), self._ptr(d_gen), nb, 1.0,
                self._ptr(scal, 0),
                self._ptr(g_t) if disc_train else None,
                self._ptr(g_g) if disc_train else None)
            _lib.check(rc, dev.ctx, 's3_loss_rel_bce')
        loss_key = None
        if train_gen:
            d_gen_full = dev.empty(tuple(gen_full.shape)) if gen_train else None
            # the first term of a dense content loss WRITES its gradient (no
            # zero fill of the hi-res gradient tensor: 118 MB at C2 batch 8)
            first_writes = bool(
                gen_train and loss_terms and mask_d is None and
                not isinstance(loss_terms[0][1], str) and c_used == c_true)
            if gen_train and not first_writes:
                L.s3_fill(dev.ctx, self._ptr(d_gen_full), d_gen_full.numel(),
                          0.0)
            n_pos = gen_full.numel() // c_true
            term_coefs = []
            for i, (name, kind, w, kw) in enumerate(loss_terms):
                slot = 4 + SLOTS_PER_TERM * i
                dg = d_gen_full if gen_train else None
                if isinstance(kind, str):
                    if mask_d is not None:
                        raise RuntimeError(
                            f'{name} has no masked (Sup3rCondMom) form')
                    term_coefs.append(self._structured_term(
                        name, kind, kw, gen_full, hr_true, c_used, w, scal,
                        slot, dg))
                    continue
                term_coefs.append([1.0])
                if mask_d is None:
                    rc = L.s3_loss_content(
                        dev.ctx, kind, self._ptr(gen_full), c_true,
                        self._ptr(hr_true), c_true, c_used, n_pos, w,
                        self._ptr(scal, slot),
                        self._ptr(dg) if gen_train else None,
                        0 if (first_writes and i == 0) else 1)
                else:
                    rc = L.s3_loss_content_masked(
                        dev.ctx, kind, self._ptr(gen_full), c_true,
                        self._ptr(hr_true), c_true, self._ptr(mask_d),
                        mask_d.shape[-1], c_used, n_pos, w,
                        self._ptr(scal, slot),
                        self._ptr(dg) if gen_train else None, 1)
                _lib.check(rc, dev.ctx, 's3_loss_content')
            obs_info = None
            if obs is not None:
                # loss over the observed / the un-observed cells of the output
                # features (with_obs.py:88-99): the masked kernel averages
                # |m (gen - true)| over ALL cells, the reference over the
                # selected ones -> x n_total / n_selected; the weighted
                # observed term joins the generator loss and its gradient
                # (with_obs.py:271-275)
                m_host, okind, w_obs = obs
                m_host = np.asarray(m_host, np.float32)
                m_obs = dev.to_device(m_host)
                m_non = dev.to_device(1.0 - m_host)
                n_tot = int(m_host.size)
                n_obs = int(m_host.sum())
                if len(loss_terms) + 2 > MAX_TERMS:
                    raise ValueError('too many loss terms for the obs terms')
                c_o = int(m_obs.shape[-1])
                base = 4 + SLOTS_PER_TERM * len(loss_terms)
                w_eff = float(w_obs) * n_tot / n_obs if (w_obs and n_obs) \
                    else 0.0
                for q, (mk, wq) in enumerate(((m_obs, w_eff), (m_non, 0.0))):
                    dgq = d_gen_full if (gen_train and wq) else None
                    rc = L.s3_loss_content_masked(
                        dev.ctx, okind, self._ptr(gen_full), c_true,
                        self._ptr(hr_true), c_true, self._ptr(mk), c_o, c_o,
                        n_pos, wq, self._ptr(scal, base + SLOTS_PER_TERM * q),
                        self._ptr(dgq) if dgq is not None else None, 1)
                    _lib.check(rc, dev.ctx, 's3_loss_content_masked')
                # the counts travel WITH the sums (two spare slots of the
                # observed term): under sharded / multi-GPU steps the loss
                # scalars of the shards are added (and all-reduced), and every
                # shard draws its own mask — the ratio n_tot / n_obs of the
                # details must be the one of the summed cells, not the last
                # shard's
                for q, cnt in ((1, n_obs), (2, n_tot)):
                    rc = L.s3_fill(dev.ctx, self._ptr(scal, base + q), 1,
                                   float(cnt))
                    _lib.check(rc, dev.ctx, 's3_fill')
                obs_info = (base, float(w_obs or 0.0))
            if need_disc:
                # adversarial term: roles swapped (base.py:899-901); only
                # D(gen) depends on the generator
                g_adv = dev.empty((nb,)) if gen_train else None
                rc = L.s3_loss_rel_bce(
                    dev.ctx, self._ptr(d_gen), self._ptr(d_true), nb,
                    float(weight_gen_advers), self._ptr(scal, 1),
                    self._ptr(g_adv) if gen_train else None, None)
                _lib.check(rc, dev.ctx, 's3_loss_rel_bce')
            if gen_train:
                if need_disc and weight_gen_advers != 0:
                    dx = dph_g.backward(g_adv, need_dx=True, need_wgrad=False)
                    self._copy_channels(dx, 0, d_gen_full, 0, c_gen,
                                        accumulate=True)
                if c_true > c_gen:
                    d_hr_gen = dev.empty(tuple(hr_gen.shape))
                    self._copy_channels(d_gen_full, 0, d_hr_gen, 0, c_gen)
                else:
                    d_hr_gen = d_gen_full
                if overlap_bucket:
                    self.gen.arm_allreduce(overlap_bucket)
                try:
                    gph.backward(d_hr_gen, need_wgrad=True,
                                 accumulate_wgrad=accumulate_wgrad)
                except BaseException:
                    self.gen.arm_allreduce(-1)          # disarm
                    raise
            loss_key = 'loss_gen'
        elif train_disc:
            if disc_train:
                dph_t.backward(g_t, need_wgrad=True,
                               accumulate_wgrad=accumulate_wgrad)
                if overlap_bucket:
                    self.disc.arm_allreduce(overlap_bucket)
                try:
                    dph_g.backward(g_g, need_wgrad=True,
                                   accumulate_wgrad=True)
                except BaseException:
                    self.disc.arm_allreduce(-1)         # disarm
                    raise
            loss_key = 'loss_disc'
        with_disc = need_disc and (compute_disc or train_disc)
        coefs = term_coefs if train_gen else None


        def recipe(vals):
            det = details_from_scalars(vals, loss_terms, coefs, with_disc,
                                       train_gen, need_disc,
                                       weight_gen_advers)
            if obs_info is not None:
                base, w_obs = obs_info
                # (summed over the shards like the losses: their ratio is what
                # matters, any common scale of the future cancels)
                n_obs, n_tot = float(vals[base + 1]), float(vals[base + 2])
                n_non = n_tot - n_obs
                l_obs = float(vals[base]) * n_tot / n_obs if n_obs else \
                    float('nan')
                l_non = float(vals[base + SLOTS_PER_TERM]) * n_tot / n_non \
                    if n_non else float('nan')
                det['loss_obs'] = LossValue(l_obs)
                det['loss_non_obs'] = LossValue(l_non)
                det['obs_frac'] = LossValue(n_obs / n_tot)
                if w_obs and n_obs:
                    for k in ('loss_gen', 'loss_gen_content'):
                        det[k] = LossValue(float(det[k]) + w_obs * l_obs)
            return det
        if defer:
            return None, LossFuture(scal, recipe, dev=self.dev), hr_gen
        details = recipe(scal.cpu().numpy())   # one sync per mini-batch
        loss = details.get(loss_key) if loss_key else None
        return loss, details, hr_gen

    # ------------------------------------------------- structured content losses
    def _structured_term(self, name, kind, kw, gen, true, c_used, w, scal,
                         slot, d_gen):
        """One M(F(gen), F(true)) term of sup3r/utilities/loss_metrics.py on
        the device: feature maps (s3_lossmap_fwd / s3_coarsen), keras MAE / MSE
        on the maps (s3_loss_content), adjoint back into ``d_gen``
        (s3_lossmap_bwd).  Returns the coefficients of the scalar slots whose
        weighted sum is the term's value."""
        L, dev = _lib.lib(), self.dev
        is_5d = gen.dim() == 5
        n, s1, s2 = (int(v) for v in gen.shape[:3])
        t = int(gen.shape[3]) if is_5d else 1
        c = int(gen.shape[-1])
        dims = (n, s1, s2, t, c, c_used)
        cls = name
        if kind in ('deriv_t', 'material', 'ext_t'):
            assert is_5d, (f'The {cls} is meant to be used on spatiotemporal '
                           'data only. Received tensor(s) that are not 5D')

        def fmap(code, x, shape, work=None):
            out = dev.empty(shape)
            rc = L.s3_lossmap_fwd(dev.ctx, code, self._ptr(x), *dims, 0, 0, 0,
                                  self._ptr(out),
                                  self._ptr(work) if work is not None else None)
            _lib.check(rc, dev.ctx, 's3_lossmap_fwd')
            return out

        def metric(m, fa, fb, cf, cu, npos, weight, sl, off=0, d_out=None):
            """loss value -> scal[sl]; returns d loss / d fa (or None),
            written at ``off`` into ``d_out`` when the caller provides it"""
            d_fa, d_off = d_out, off
            if d_gen is None:
                d_fa = None
            elif d_fa is None:
                d_fa, d_off = dev.empty((npos * cf,)), 0
            if d_fa is not None:
                L.s3_fill(dev.ctx, self._ptr(d_fa, d_off), npos * cf, 0.0)
            rc = L.s3_loss_content(
                dev.ctx, m, self._ptr(fa, off), cf, self._ptr(fb, off), cf, cu,
                npos, weight, self._ptr(scal, sl),
                self._ptr(d_fa, d_off) if d_fa is not None else None, 0)
            _lib.check(rc, dev.ctx, 's3_loss_content')
            return d_fa

        def fbwd(code, fx, g_out, p=(0, 0, 0), work=None):
            rc = L.s3_lossmap_bwd(
                dev.ctx, code, self._ptr(gen),
                self._ptr(fx) if fx is not None else None, self._ptr(g_out),
                *dims, *p, self._ptr(d_gen),
                self._ptr(work) if work is not None else None)
            _lib.check(rc, dev.ctx, 's3_lossmap_bwd')

        def extremes(spatial, weight, sl):
            """(MAE(min) + MAE(max)) / 2 (loss_metrics.py:325-392)"""
            code = LMAP['ext_s' if spatial else 'ext_t']
            ne = n * t * c_used if spatial else n * s1 * s2 * c_used
            slab = n * 64 * t * c_used
            work = dev.empty((2 * ne + slab,))
            fa = fmap(code, gen, (2 * ne,), work)
            fb = fmap(code, true, (2 * ne,), work)
            # both gradients side by side in one buffer: [d min | d max]
            g = dev.empty((2 * ne,)) if d_gen is not None else None
            for q in range(2):
                metric(_lib.LOSS_MAE, fa, fb, c_used, c_used, ne // c_used,
                       0.5 * weight, sl + q, off=q * ne, d_out=g)
            if d_gen is not None:
                fbwd(code, fa, g, work=work)
            return [0.5, 0.5]

        if kind in ('deriv_s', 'deriv_t'):
            npos = n * s1 * s2 * t
            fa = fmap(LMAP[kind], gen, (npos * c_used,))
            fb = fmap(LMAP[kind], true, (npos * c_used,))
            g = metric(_lib.LOSS_MAE, fa, fb, c_used, c_used, npos, w, slot)
            if g is not None:
                fbwd(LMAP[kind], None, g)
            return [1.0]
        if kind == 'material':
            hub = c_used // 2
            npos = n * s1 * s2 * t
            fa = fmap(LMAP[kind], gen, (npos * hub,))
            fb = fmap(LMAP[kind], true, (npos * hub,))
            g = metric(_lib.LOSS_MAE, fa, fb, hub, hub, npos, w, slot)
            if g is not None:
                fbwd(LMAP[kind], None, g)
            return [1.0]
        if kind == 'mean_s':
            work = dev.empty((n * 64 * t * c_used,))
            fa = fmap(LMAP[kind], gen, (n * t * c_used,), work)
            fb = fmap(LMAP[kind], true, (n * t * c_used,), work)
            g = metric(_lib.LOSS_MSE, fa, fb, c_used, c_used, n * t, w, slot)
            if g is not None:
                fbwd(LMAP[kind], None, g)
            return [1.0]
        if kind in ('ext_s', 'ext_t'):
            return extremes(kind == 'ext_s', w, slot)
        if kind == 'lowres':
            s_e, t_e = int(kw['s_enhance']), int(kw['t_enhance'])
            coefs = [1.0]
            if kw['ex_loss'] is not None:
                if kw['ex_loss'] == 'TemporalExtremesLoss':
                    assert is_5d
                extremes(kw['ex_loss'] == 'SpatialExtremesLoss', w, slot + 1)
                coefs = [1.0, 0.5, 0.5]
            if t_e > 1 and kw['t_method'] in ('average', 'subsample'):
                assert is_5d
            else:
                t_e = 1
            if s1 % s_e or s2 % s_e or t % t_e:
                raise ValueError('LowResLoss enhancement factors must evenly '
                                 f'divide the grid {tuple(gen.shape)}')
            m = CONTENT_KINDS[kw['tf_loss']]
            meth = _lib.TC_METHODS[kw['t_method'] if t_e > 1 else 'subsample']
            if s_e <= 1 and t_e <= 1:
                rc = L.s3_loss_content(
                    dev.ctx, m, self._ptr(gen), c, self._ptr(true), c, c_used,
                    n * s1 * s2 * t, w, self._ptr(scal, slot),
                    self._ptr(d_gen) if d_gen is not None else None, 1)
                _lib.check(rc, dev.ctx, 's3_loss_content')
                return coefs
            o = (n, s1 // s_e, s2 // s_e, t // t_e, c)
            lo = []
            for x in (gen, true):
                y = dev.empty(o)
                rc = L.s3_coarsen(dev.ctx, self._ptr(x), n, s1, s2, t, c, s_e,
                                  t_e, meth, self._ptr(y))
                _lib.check(rc, dev.ctx, 's3_coarsen')
                lo.append(y)
            g = metric(m, lo[0], lo[1], c, c_used, o[0] * o[1] * o[2] * o[3],
                       w, slot)
            if g is not None:
                fbwd(LMAP['coarsen'], None, g, p=(s_e, t_e, meth))
            return coefs
        if kind in ('fft_s', 'fft_st'):
            # log(1 + w |FFT|) of both fields, MAE, adjoint DFT of the spectral
            # gradient (loss_metrics.py:395-485); tf.signal.fft2d needs 4-D,
            # fft3d 5-D input
            mode3d = kind == 'fft_st'
            assert is_5d == mode3d, (f'{cls} expects '
                                     f'{"5D" if mode3d else "4D"} tensors')
            tot = n * s1 * s2 * t * c
            axes = [(n, s1, s2 * t * c), (n * s1, s2, t * c)]
            if mode3d:
                axes.append((n * s1 * s2, t, c))

            def dft(re, im, sign):
                for outer, ln, inner in axes:
                    ore, oim = dev.empty((tot,)), dev.empty((tot,))
                    rc = L.s3_dft_axis(
                        dev.ctx, self._ptr(re),
                        self._ptr(im) if im is not None else None,
                        self._ptr(ore), self._ptr(oim), outer, ln, inner, sign)
                    _lib.check(rc, dev.ctx, 's3_dft_axis')
                    re, im = ore, oim
                return re, im
            ys, spec = [], None
            for x in (gen, true):
                re, im = dft(x, None, -1)
                y = dev.empty((tot,))
                rc = L.s3_specmap(dev.ctx, 0, self._ptr(re), self._ptr(im),
                                  None, n, s1, s2, t, c, int(mode3d),
                                  self._ptr(y), None)
                _lib.check(rc, dev.ctx, 's3_specmap')
                ys.append(y)
                if x is gen:
                    spec = (re, im)
            g = metric(_lib.LOSS_MAE, ys[0], ys[1], c, c_used,
                       n * s1 * s2 * t, w, slot)
            if g is not None:
                gre, gim = dev.empty((tot,)), dev.empty((tot,))
                rc = L.s3_specmap(dev.ctx, 1, self._ptr(spec[0]),
                                  self._ptr(spec[1]), self._ptr(g), n, s1, s2,
                                  t, c, int(mode3d), self._ptr(gre),
                                  self._ptr(gim))
                _lib.check(rc, dev.ctx, 's3_specmap')
                dre, _ = dft(gre, gim, +1)
                rc = L.s3_copy_channels(dev.ctx, self._ptr(dre), c, 0,
                                        self._ptr(d_gen), c, 0, c_used,
                                        n * s1 * s2 * t, 1)
                _lib.check(rc, dev.ctx, 's3_copy_channels')
            return [1.0]
        if kind == 'mmd':
            if c_used > 8:
                raise ValueError('MmdLoss kernel handles at most 8 features')
            rc = L.s3_loss_mmd(
                dev.ctx, self._ptr(gen), c, self._ptr(true), c, n,
                s1 * s2 * t, c_used, float(kw.get('sigma', 1.0)), w,
                self._ptr(scal, slot),
                self._ptr(d_gen) if d_gen is not None else None)
            _lib.check(rc, dev.ctx, 's3_loss_mmd')
            return [1.0]
        if kind == 'sw':
            # new random directions per call (tf.random.normal in the
            # reference, loss_metrics.py:777): a fresh Philox seed each time
            seed = getattr(self, 'sw_seed', None)
            if seed is None:
                seed = int.from_bytes(os.urandom(8), 'little')
            self.sw_seed = (int(seed) + 0x9E3779B97F4A7C15) % (1 << 64)
            rc = L.s3_loss_sliced_wasserstein(
                dev.ctx, self._ptr(gen), c, self._ptr(true), c, n,
                s1 * s2 * t, c_used, int(kw.get('n_projections', 1024)),
                int(seed) % (1 << 64), w, self._ptr(scal, slot),
                self._ptr(d_gen) if d_gen is not None else None)
            _lib.check(rc, dev.ctx, 's3_loss_sliced_wasserstein')
            return [1.0]
        raise KeyError(kind)

    # ------------------------------------------------------------ optimizer
    def apply(self, which, optimizer):
        """keras ``apply_gradients`` on the whole store of one network: one
        fused launch, whatever the optimizer (``optimizers.py``)."""
        net = self.gen if which == 'gen' else self.disc
        kind = getattr(optimizer, 'KIND', None)
        if kind is None or not hasattr(optimizer, 'hyper'):
            raise KeyError(f'optimizer "{optimizer}" has no MI355X kernel')
        rec = getattr(self, '_recording', None)
        if rec is not None:
            # a step being recorded (captured.py): the update launch reads the
            # step's scalars from the device; the replay stages them and
            # counts the iteration
            rec.append((net, optimizer))
            net.optimizer_step_staged(kind)
            return
        optimizer.iterations += 1
        net.optimizer_step(kind, optimizer.hyper(), optimizer.iterations)

    def allreduce_grads(self, which):
        net = self.gen if which == 'gen' else self.disc
        net.allreduce_grads()

    def broadcast_state(self, root=0):
        """Every rank starts from rank ``root``'s weights and Adam slots (the
        reference's towers read one set of variables, abstract.py:827-841)."""
        for net in (self.gen, self.disc):
            if net is not None and net.built:
                for which in (_lib.BUF_W, _lib.BUF_M, _lib.BUF_V):
                    net.broadcast(which, root)
        # the first collectives of the job: a rank that never joined shows up
        # here, as a TimeoutError instead of a hang
        if self.dev.nranks > 1:
            self.dev.wait()
