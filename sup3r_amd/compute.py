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

import numpy as np

from . import _lib
from .engine import Device, Network
from .utilities import LossValue, camel_to_underscore

CONTENT_KINDS = {'MeanAbsoluteError': _lib.LOSS_MAE,
                 'MeanSquaredError': _lib.LOSS_MSE}


def parse_loss_spec(loss):
    """``get_loss_fun`` spec handling (abstract.py:461-502): str | dict with
    optional ``term_weights``.  Returns [(name, kind, weight), ...]."""
    spec = {loss: {}} if isinstance(loss, str) else dict(loss)
    names = [k for k in spec if k != 'term_weights']
    weights = spec.get('term_weights', [1.0] * len(names))
    terms = []
    for n, w in zip(names, weights):
        if n not in CONTENT_KINDS:
            raise KeyError(
                'Could not find requested loss function "{}" among the '
                'content losses with an MI355X kernel ({}).'.format(
                    n, list(CONTENT_KINDS)))
        if spec[n]:
            raise KeyError(f'loss "{n}" takes no kwargs here: {spec[n]}')
        terms.append((n, CONTENT_KINDS[n], float(w)))
    return terms


class HipGanCompute:
    """Generator / discriminator pair on one GPU."""

    def __init__(self, gen_layers, disc_layers, device=None, precision=None):
        self.dev = device or Device.get()
        self.gen = Network(gen_layers, 'generator', self.dev, precision)
        self.disc = None if disc_layers is None else Network(
            disc_layers, 'discriminator', self.dev, precision)
        self._scal = None

    # ---------------------------------------------------------------- utils
    def _scalars(self):
        if self._scal is None:
            self._scal = self.dev.empty((16,))
        return self._scal

    def _ptr(self, t, offset=0):
        return C.c_void_p(t.data_ptr() + 4 * offset)

    def _copy_channels(self, src, c0_src, dst, c0_dst, nc, accumulate=False):
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

    # --------------------------------------------------- loss (+ gradients)
    def loss_and_grads(self, low_res, hi_res_true, loss_terms,
                       weight_gen_advers=0.001, train_gen=True,
                       train_disc=False, compute_disc=False, exo_names=(),
                       backward=True, hi_res_gen=None, mask=None):
        """One ``_get_hr_exo_and_loss`` + ``tape.gradient``.

        ``backward=False`` evaluates ``calc_loss`` only (validation).  When
        ``hi_res_gen`` is given (public ``calc_loss(hi_res_true, hi_res_gen)``)
        the generator forward is skipped.  Gradients of the trained network are
        left in its device gradient buffer.  Returns (loss, details, hr_gen).
        """
        L = _lib.lib()
        dev = self.dev
        hr_true = dev.to_device(hi_res_true)
        n_exo = len(exo_names)
        c_true = hr_true.shape[-1]
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
        scal = self._scalars()
        L.s3_fill(dev.ctx, self._ptr(scal), 16, 0.0)
        details = {}
        need_disc = self.disc is not None
        dph_t = dph_g = None
        if need_disc:
            # the reference always runs the disc on both (base.py:884-885)
            tr = gen_train or disc_train
            dph_t = self.disc.plan(tuple(hr_true.shape), training=tr, slot=0)
            dph_g = self.disc.plan(tuple(hr_true.shape), training=tr, slot=1)
            d_true = dph_t.forward(hr_true)
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
            if gen_train:
                L.s3_fill(dev.ctx, self._ptr(d_gen_full), d_gen_full.numel(),
                          0.0)
            n_pos = gen_full.numel() // c_true
            for i, (name, kind, w) in enumerate(loss_terms):
                if mask_d is None:
                    rc = L.s3_loss_content(
                        dev.ctx, kind, self._ptr(gen_full), c_true,
                        self._ptr(hr_true), c_true, c_used, n_pos, w,
                        self._ptr(scal, 4 + i),
                        self._ptr(d_gen_full) if gen_train else None, 1)
                else:
                    rc = L.s3_loss_content_masked(
                        dev.ctx, kind, self._ptr(gen_full), c_true,
                        self._ptr(hr_true), c_true, self._ptr(mask_d),
                        mask_d.shape[-1], c_used, n_pos, w,
                        self._ptr(scal, 4 + i),
                        self._ptr(d_gen_full) if gen_train else None, 1)
                _lib.check(rc, dev.ctx, 's3_loss_content')
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
            for i, (name, kind, w) in enumerate(loss_terms):
                details[camel_to_underscore(name)] = LossValue(vals[4 + i])
                content += w * float(vals[4 + i])
            advers = float(vals[1]) if need_disc else 0.0
            details['loss_gen_content'] = LossValue(content)
            details['loss_gen_advers'] = LossValue(advers)
            details['loss_gen'] = LossValue(
                content + weight_gen_advers * advers)
        loss = details.get(loss_key) if loss_key else None
        return loss, details, hr_gen

    # ------------------------------------------------------------ optimizer
    def apply(self, which, optimizer):
        """keras Adam ``apply_gradients`` on the whole store of one network."""
        net = self.gen if which == 'gen' else self.disc
        cfg = optimizer.get_config()
        if cfg['name'].lower() != 'adam':
            raise KeyError(f'optimizer "{cfg["name"]}" has no MI355X kernel')
        optimizer.iterations += 1
        net.adam_step(cfg['learning_rate'], cfg['beta_1'], cfg['beta_2'],
                      cfg['epsilon'], optimizer.iterations)

    def allreduce_grads(self, which):
        net = self.gen if which == 'gen' else self.disc
        net.allreduce_grads()
