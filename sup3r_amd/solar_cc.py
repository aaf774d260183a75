"""``SolarCC`` on the MI355X engine — the solar climate-change GAN whose loss
looks at daylight hours only.

API and semantics of sup3r/models/solar_cc.py: pointwise content loss on the
centre POINT_LOSS_HOURS of every day plus the content loss between the
24-hour mean of the synthetic day and the daylight mean of the true day
(:207-232); the discriminator sees the centre DAYLIGHT_HOURS of every true day
and ``n_days`` randomly placed DAYLIGHT_HOURS windows of the synthetic field
(:176-193); ``generate`` pads the time axis to ``low_res * t_enhance``
(:253-298).

On the device the windows are gathered into one (n_days * B, s1, s2,
DAYLIGHT_HOURS, C) tensor per side (``s3_time_window``), so each side is ONE
discriminator pass — the reference runs one pass per window and concatenates,
and no layer couples observations; window means are ``s3_time_mean``; every
gradient is scattered back into the full hi-res gradient by the adjoint mode of
the same two kernels before the generator's backward pass.
"""
import logging

import numpy as np

from . import _lib
from .compute import SLOTS_PER_TERM, HipGanCompute
from .gan import Sup3rGan
from .utilities import LossValue, camel_to_underscore

logger = logging.getLogger(__name__)


class SolarCompute(HipGanCompute):
    """``HipGanCompute`` with SolarCC's windowed ``calc_loss``."""

    supports_defer = False   # its loss_and_grads reads the scalars back itself

    # (STARTING_HOUR, DAYLIGHT_HOURS, POINT_LOSS_HOURS), set by the model
    hours = (8, 8, 2)
    # fixed window starts for tests; None = a fresh uniform draw per call
    # (tf.random.categorical over equal logits, solar_cc.py:185-186)
    time_samples = None

    def _window(self, full, t0, length, window, adjoint=False, scale=1.0,
                offset=0):
        outer = int(np.prod(full.shape[:3]))
        rc = _lib.lib().s3_time_window(
            self.dev.ctx, self._ptr(full), outer, int(full.shape[3]),
            int(full.shape[4]), int(t0), int(length),
            self._ptr(window, offset), int(adjoint), float(scale))
        _lib.check(rc, self.dev.ctx, 's3_time_window')

    def _mean(self, full, t0, length, mean, adjoint=False, scale=1.0):
        outer = int(np.prod(full.shape[:3]))
        rc = _lib.lib().s3_time_mean(
            self.dev.ctx, self._ptr(full), outer, int(full.shape[3]),
            int(full.shape[4]), int(t0), int(length), self._ptr(mean),
            int(adjoint), float(scale))
        _lib.check(rc, self.dev.ctx, 's3_time_mean')

    def _content_terms(self, gen, true, loss_terms, c_used, scal, slot0, d_gen,
                       wscale):
        """every term of the content loss on one (gen, true) pair; values go
        to ``scal`` slots, wscale * weight * gradients are added to d_gen"""
        L, dev = _lib.lib(), self.dev
        c = int(gen.shape[-1])
        n_pos = gen.numel() // c
        coefs = []
        for i, (name, kind, w, kw) in enumerate(loss_terms):
            slot = slot0 + SLOTS_PER_TERM * i
            if isinstance(kind, str):
                coefs.append(self._structured_term(
                    name, kind, kw, gen, true, c_used, w * wscale, scal, slot,
                    d_gen))
                continue
            coefs.append([1.0])
            rc = L.s3_loss_content(
                dev.ctx, kind, self._ptr(gen), c, self._ptr(true), c, c_used,
                n_pos, w * wscale, self._ptr(scal, slot),
                self._ptr(d_gen) if d_gen is not None else None, 1)
            _lib.check(rc, dev.ctx, 's3_loss_content')
        return coefs

    def loss_and_grads(self, low_res, hi_res_true, loss_terms,
                       weight_gen_advers=0.001, train_gen=True,
                       train_disc=False, compute_disc=False, exo_names=(),
                       backward=True, hi_res_gen=None, mask=None):
        """solar_cc.py:93-251 (+ ``tape.gradient`` when ``backward``)."""
        if mask is not None:
            raise RuntimeError('SolarCC has no masked loss')
        L, dev = _lib.lib(), self.dev
        start, daylight, point = self.hours
        hr_true = dev.to_device(hi_res_true)
        n_exo = len(exo_names)
        c_true = int(hr_true.shape[-1])
        gen_train = bool(backward and train_gen)
        disc_train = bool(backward and train_disc and not train_gen)
        if hi_res_gen is None:
            lr = dev.to_device(low_res)
            exo = self.exo_from_true(hr_true, list(exo_names))
            gph = self.gen.plan(tuple(lr.shape), training=gen_train)
            hr_gen = gph.forward(lr, exo)
        else:
            gph = None
            hr_gen = dev.to_device(hi_res_gen)
        c_gen = int(hr_gen.shape[-1])
        if c_true > c_gen and tuple(hr_gen.shape[:-1]) == tuple(
                hr_true.shape[:-1]):
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
        assert hr_true.dim() == 5 and hr_true.shape[3] % 24 == 0, (
            'Special SolarCC model can only accept multi-day hourly (multiple '
            'of 24) true / synthetic high res data in the axis=3 position but '
            'received shape {}'.format(tuple(hr_true.shape)))
        nb, s1, s2, t_len = (int(v) for v in hr_true.shape[:4])
        n_days = t_len // 24
        days = [24 * i for i in range(n_days)]
        sub0 = [start + d for d in days]                       # daylight
        pnt0 = [(24 - point) // 2 + d for d in days]           # point loss
        ts = self.time_samples
        if ts is None:
            ts = np.random.randint(0, t_len - daylight + 1, size=n_days)
        ts = [int(v) for v in np.asarray(ts).reshape(-1)]
        assert len(ts) == n_days and all(
            0 <= v <= t_len - daylight for v in ts)
        c_used = c_true - n_exo
        n_terms = len(loss_terms)
        scal = dev.empty((4 + SLOTS_PER_TERM * n_terms * 2 * n_days,))
        L.s3_fill(dev.ctx, self._ptr(scal), scal.numel(), 0.0)
        details = {}

        # ---- discriminator on the windows, one pass per side
        need_disc = self.disc is not None
        wshape = (n_days * nb, s1, s2, daylight, c_true)
        wsize = nb * s1 * s2 * daylight * c_true
        if need_disc:
            win_t, win_g = dev.empty(wshape), dev.empty(wshape)
            for i in range(n_days):
                self._window(hr_true, sub0[i], daylight, win_t,
                             offset=i * wsize)
                self._window(gen_full, ts[i], daylight, win_g,
                             offset=i * wsize)
            tr = gen_train or disc_train
            dph_t = self.disc.plan(wshape, training=tr, slot=0)
            dph_g = self.disc.plan(wshape, training=tr, slot=1)
            d_true = dph_t.forward(win_t)
            d_gen = dph_g.forward(win_g)
            n_out = d_true.numel()
        if need_disc and (compute_disc or train_disc):
            g_t = dev.empty((n_out,)) if disc_train else None
            g_g = dev.empty((n_out,)) if disc_train else None
            rc = L.s3_loss_rel_bce(
                dev.ctx, self._ptr(d_true), self._ptr(d_gen), n_out, 1.0,
                self._ptr(scal, 0),
                self._ptr(g_t) if disc_train else None,
                self._ptr(g_g) if disc_train else None)
            _lib.check(rc, dev.ctx, 's3_loss_rel_bce')

        loss_key = None
        coefs = {}
        if train_gen:
            d_full = None
            if gen_train:
                d_full = dev.empty(tuple(gen_full.shape))
                L.s3_fill(dev.ctx, self._ptr(d_full), d_full.numel(), 0.0)
            pshape = (nb, s1, s2, point, c_true)
            mshape = (nb, s1, s2, c_true)
            for i in range(n_days):
                # centre hours, pointwise (hr_*_ploss, :209-216)
                gen_p, true_p = dev.empty(pshape), dev.empty(pshape)
                self._window(gen_full, pnt0[i], point, gen_p)
                self._window(hr_true, pnt0[i], point, true_p)
                d_p = None
                if gen_train:
                    d_p = dev.empty(pshape)
                    L.s3_fill(dev.ctx, self._ptr(d_p), d_p.numel(), 0.0)
                slot = 4 + SLOTS_PER_TERM * n_terms * (2 * i)
                coefs[(i, 0)] = self._content_terms(
                    gen_p, true_p, loss_terms, c_used, scal, slot, d_p,
                    1.0 / n_days)
                if gen_train:
                    self._window(d_full, pnt0[i], point, d_p, adjoint=True)
                # 24-h mean of the synthetic day vs daylight mean of the true
                # day (:213-221)
                gen_m, true_m = dev.empty(mshape), dev.empty(mshape)
                self._mean(gen_full, days[i], 24, gen_m)
                self._mean(hr_true, sub0[i], daylight, true_m)
                d_m = None
                if gen_train:
                    d_m = dev.empty(mshape)
                    L.s3_fill(dev.ctx, self._ptr(d_m), d_m.numel(), 0.0)
                slot = 4 + SLOTS_PER_TERM * n_terms * (2 * i + 1)
                coefs[(i, 1)] = self._content_terms(
                    gen_m, true_m, loss_terms, c_used, scal, slot, d_m,
                    1.0 / n_days)
                if gen_train:
                    self._mean(d_full, days[i], 24, d_m, adjoint=True)
            if need_disc:
                # adversarial term: roles swapped (:234-236)
                g_adv = dev.empty((n_out,)) if gen_train else None
                rc = L.s3_loss_rel_bce(
                    dev.ctx, self._ptr(d_gen), self._ptr(d_true), n_out,
                    float(weight_gen_advers), self._ptr(scal, 1),
                    self._ptr(g_adv) if gen_train else None, None)
                _lib.check(rc, dev.ctx, 's3_loss_rel_bce')
            if gen_train:
                if need_disc and weight_gen_advers != 0:
                    dx = dph_g.backward(g_adv, need_dx=True, need_wgrad=False)
                    for i in range(n_days):
                        self._window(d_full, ts[i], daylight, dx,
                                     adjoint=True, offset=i * wsize)
                if c_true > c_gen:
                    d_hr_gen = dev.empty(tuple(hr_gen.shape))
                    self._copy_channels(d_full, 0, d_hr_gen, 0, c_gen)
                else:
                    d_hr_gen = d_full
                gph.backward(d_hr_gen, need_wgrad=True)
            loss_key = 'loss_gen'
        elif train_disc:
            if disc_train:
                dph_t.backward(g_t, need_wgrad=True, accumulate_wgrad=False)
                dph_g.backward(g_g, need_wgrad=True, accumulate_wgrad=True)
            loss_key = 'loss_disc'

        vals = scal.cpu().numpy()          # one sync per mini-batch
        if need_disc and (compute_disc or train_disc):
            details['loss_disc'] = LossValue(vals[0])
        if train_gen:
            content = 0.0
            for (i, part), cf in coefs.items():
                prefix = 'c_sub_' if part == 0 else 'c_24h_'
                base = 4 + SLOTS_PER_TERM * n_terms * (2 * i + part)
                for k, (name, kind, w, kw) in enumerate(loss_terms):
                    slot = base + SLOTS_PER_TERM * k
                    val = sum(c * float(vals[slot + j])
                              for j, c in enumerate(cf[k]))
                    key = prefix + camel_to_underscore(name)
                    details[key] = LossValue(
                        float(details.get(key, 0.0)) + val / n_days)
                    content += w * val / n_days
            advers = float(vals[1]) if need_disc else 0.0
            details['loss_gen_content'] = LossValue(content)
            details['loss_gen_advers'] = LossValue(advers)
            details['loss_gen'] = LossValue(
                content + weight_gen_advers * advers)
        loss = details.get(loss_key) if loss_key else None
        return loss, details, hr_gen


class SolarCC(Sup3rGan):
    """Solar climate change model (sup3r/models/solar_cc.py:13-324)."""

    STARTING_HOUR = 8
    DAYLIGHT_HOURS = 8
    POINT_LOSS_HOURS = 2

    _compute_factory = SolarCompute

    def __init__(self, *args, t_enhance=None, **kwargs):
        """``t_enhance`` fixes the temporal enhancement the forward pass
        expects: ``generate`` pads its output to ``low_res * t_enhance`` steps
        (:45-64)."""
        super().__init__(*args, **kwargs)
        self._t_enhance = t_enhance or self.t_enhance
        self.meta['t_enhance'] = self._t_enhance
        self._compute.hours = (self.STARTING_HOUR, self.DAYLIGHT_HOURS,
                               self.POINT_LOSS_HOURS)

    def init_weights(self, lr_shape, hr_shape, device=None):
        """the discriminator only ever sees DAYLIGHT_HOURS steps (:66-91)"""
        if hr_shape[3] != self.DAYLIGHT_HOURS:
            hr_shape = (tuple(hr_shape[0:3]) + (self.DAYLIGHT_HOURS,)
                        + tuple(hr_shape[-1:]))
        super().init_weights(lr_shape, hr_shape, device=device)

    def temporal_pad(self, low_res, hi_res, mode='reflect'):
        """pad the time axis of the generated array symmetrically to
        ``low_res.shape[-2] * t_enhance`` steps (:253-286)"""
        t_shape = low_res.shape[-2] * self._t_enhance
        t_pad = int((t_shape - hi_res.shape[-2]) / 2)
        pad_width = ((0, 0), (0, 0), (0, 0), (t_pad, t_pad), (0, 0))
        prepad_shape = hi_res.shape
        hi_res = np.pad(hi_res, pad_width, mode=mode)
        logger.debug('Padded hi_res output from %s to %s', prepad_shape,
                     hi_res.shape)
        return hi_res

    supports_device_chunks = False     # generate() is overridden below

    def generate(self, low_res, **kwargs):
        hi_res = self.temporal_pad(
            low_res, super().generate(low_res=low_res, **kwargs))
        logger.debug('Final SolarCC output has shape: %s', hi_res.shape)
        return hi_res

    @classmethod
    def load(cls, model_dir, t_enhance=None, verbose=True):
        fp_gen, fp_disc, params = cls._load(model_dir, verbose=verbose)
        return cls(fp_gen, fp_disc, t_enhance=t_enhance, **params)
