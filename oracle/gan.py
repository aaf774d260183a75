"""Loss algebra, optimizer and one train batch of Sup3rGan restated in numpy
(TEST INFRASTRUCTURE).

Follows: ``Sup3rGan.calc_loss`` (sup3r/models/base.py:830-911),
``calc_loss_gen_content`` (:478-503), ``calc_loss_disc`` relativistic BCE
(:505-549), ``_train_batch`` (:944-1031), ``get_single_grad`` /
``run_gradient_descent`` (sup3r/models/abstract.py:843-914,1190-1238),
multi-term loss factory (:461-502), ``norm_input`` / ``un_norm_output``
(:197-275).  keras-2.15 semantics restated: MeanAbsoluteError /
MeanSquaredError (global mean for equal shapes),
``tf.nn.sigmoid_cross_entropy_with_logits`` stable form, ``Adam.update_step``.
"""
import numpy as np


# ---------------------------------------------------------------- losses
def mae(a, b):
    """keras MeanAbsoluteError()(a, b) -> scalar, and d/da, d/db."""
    d = a - b
    n = d.size
    loss = np.abs(d).mean()
    g = np.sign(d) / n
    return loss, g, -g


def mse(a, b):
    d = a - b
    n = d.size
    loss = (d * d).mean()
    g = 2.0 * d / n
    return loss, g, -g


CONTENT_LOSSES = {'MeanAbsoluteError': mae, 'MeanSquaredError': mse}


def sigmoid_xent(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z +
    log1p(exp(-|x|)); d/dx = sigmoid(x) - z."""
    loss = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    sig = np.where(x >= 0, 1.0 / (1.0 + np.exp(-np.abs(x))),
                   np.exp(-np.abs(x)) / (1.0 + np.exp(-np.abs(x))))
    return loss, sig - z


def rel_bce(disc_out_true, disc_out_gen):
    """calc_loss_disc (base.py:540-549).  Returns (loss, dL/d_true,
    dL/d_gen).  logits = [D_t - mean(D_g); D_g - mean(D_t)], labels = [1; 0],
    loss = mean(sigmoid_xent)."""
    dt, dg = disc_out_true, disc_out_gen
    lt = dt - dg.mean()
    lf = dg - dt.mean()
    xt, gt = sigmoid_xent(lt, np.ones_like(lt))
    xf, gf = sigmoid_xent(lf, np.zeros_like(lf))
    n = lt.size + lf.size
    loss = (xt.sum() + xf.sum()) / n
    gt = gt / n
    gf = gf / n
    d_true = gt - gf.sum() / dt.size
    d_gen = gf - gt.sum() / dg.size
    return loss, d_true.astype(dt.dtype), d_gen.astype(dg.dtype)


def content_loss(loss_spec, hi_res_gen, hi_res_true):
    """``get_loss_fun`` multi-term weighted sum (abstract.py:461-502); called
    gen-first as in base.py:503.  Returns (loss, details, dL/d_gen)."""
    spec = {loss_spec: {}} if isinstance(loss_spec, str) else dict(loss_spec)
    names = [k for k in spec if k != 'term_weights']
    weights = spec.get('term_weights', [1.0] * len(names))
    total, details = 0.0, {}
    dgen = np.zeros_like(hi_res_gen)
    for w, name in zip(weights, names):
        val, g_a, _ = CONTENT_LOSSES[name](hi_res_gen, hi_res_true)
        details[camel_to_underscore(name)] = val
        total = total + w * val
        dgen = dgen + w * g_a
    return total, details, dgen


def camel_to_underscore(name):
    """sup3r.utilities.utilities.camel_to_underscore."""
    import re
    s1 = re.sub('(.)([A-Z][a-z]+)', r'\1_\2', name)
    return re.sub('([a-z0-9])([A-Z])', r'\1_\2', s1).lower()


# ---------------------------------------------------------------- optimizer
class Adam:
    """keras-2.15 ``Adam.update_step`` (K14): alpha = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= m*alpha/(sqrt(v)+eps)."""

    def __init__(self, learning_rate=1e-4, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7):
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0
        self.m = None
        self.v = None

    def apply_gradients(self, grads, weights):
        if self.m is None:
            self.m = [np.zeros_like(w) for w in weights]
            self.v = [np.zeros_like(w) for w in weights]
        t = self.iterations + 1
        dt = weights[0].dtype.type
        b1p = dt(self.beta_1) ** t
        b2p = dt(self.beta_2) ** t
        alpha = dt(self.learning_rate) * np.sqrt(dt(1) - b2p) / (dt(1) - b1p)
        for w, g, m, v in zip(weights, grads, self.m, self.v):
            g = g.astype(w.dtype)
            m += (g - m) * dt(1 - self.beta_1)
            v += (g * g - v) * dt(1 - self.beta_2)
            w -= (m * alpha) / (np.sqrt(v) + dt(self.epsilon))
        self.iterations = t


class KerasOptimizer:
    """keras-2.15 ``update_step`` of the other optimizers sup3r may be given
    by name (abstract.py:321-350), restated from keras' published source
    (keras is not installed here): SGD, RMSprop (not centered), Adagrad,
    Adamax, AdamW.  Two slot lists ``m`` / ``v`` like the device store."""

    def __init__(self, name, **kw):
        self.name, self.kw = name, kw
        self.iterations = 0
        self.m = self.v = None

    def apply_gradients(self, grads, weights):
        k, name = self.kw, self.name
        if self.m is None:
            self.m = [np.zeros_like(w) for w in weights]
            init = k.get('initial_accumulator_value', 0.1) \
                if name == 'Adagrad' else 0.0
            self.v = [np.full_like(w, init) for w in weights]
        t = self.iterations + 1
        lr = k.get('learning_rate')
        for w, g, m, v in zip(weights, grads, self.m, self.v):
            g = g.astype(w.dtype)
            if name == 'SGD':
                mom = k.get('momentum', 0.0)
                if mom:
                    m[...] = -g * lr + m * mom
                    w += (-g * lr + m * mom) if k.get('nesterov') else m
                else:
                    w += -g * lr
            elif name == 'RMSprop':
                rho, eps = k.get('rho', 0.9), k.get('epsilon', 1e-7)
                v[...] = rho * v + (1 - rho) * g * g
                inc = lr * g / np.sqrt(v + eps)
                if k.get('momentum', 0.0) > 0:
                    m[...] = k['momentum'] * m + inc
                    w -= m
                else:
                    w -= inc
            elif name == 'Adagrad':
                v += g * g
                w -= lr * g / np.sqrt(v + k.get('epsilon', 1e-7))
            elif name == 'Adamax':
                b1, b2 = k.get('beta_1', 0.9), k.get('beta_2', 0.999)
                m += (g - m) * (1 - b1)
                v[...] = np.maximum(b2 * v, np.abs(g))
                w -= (lr * m) / ((1 - b1 ** t) * (v + k.get('epsilon', 1e-7)))
            elif name == 'AdamW':
                b1, b2 = k.get('beta_1', 0.9), k.get('beta_2', 0.999)
                w -= w * k.get('weight_decay', 0.004) * lr
                alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
                m += (g - m) * (1 - b1)
                v += (g * g - v) * (1 - b2)
                w -= (m * alpha) / (np.sqrt(v) + k.get('epsilon', 1e-7))
            else:
                raise KeyError(name)
        self.iterations = t


# ---------------------------------------------------------------- GAN step
class GanOracle:
    """Sup3rGan compute core: generator + discriminator ``oracle.network``
    objects, content loss spec, two Adam instances."""

    def __init__(self, gen, disc, loss='MeanSquaredError', learning_rate=1e-4,
                 learning_rate_disc=None, n_exo=0):
        self.gen, self.disc = gen, disc
        self.loss = loss
        self.opt = Adam(learning_rate)
        self.opt_disc = Adam(learning_rate_disc or learning_rate)
        self.n_exo = n_exo
        # test hooks (tests/test_parity_r03.py): replacements of the two
        # forward passes — a teacher-forced walk that leaves the DEVICE's
        # activations in the layers' caches — and a callable run between the
        # forward and the backward pass (installs the device's masks)
        self.gen_forward = None
        self.disc_forward = None
        self.pre_backward = None

    def _exo_from_true(self, hi_res_true, exo_names):
        """get_hr_exo_input (abstract.py:415-436)."""
        if not exo_names:
            return None
        k = len(exo_names)
        return {nm: hi_res_true[..., hi_res_true.shape[-1] - k + i:
                                hi_res_true.shape[-1] - k + i + 1]
                for i, nm in enumerate(exo_names)}

    def loss_and_grads(self, low_res, hi_res_true, weight_gen_advers=0.001,
                       train_gen=True, train_disc=False, compute_disc=False,
                       exo_names=()):
        """calc_loss + tape.gradient w.r.t. the trained network's weights.
        Returns (loss, details, grads list in keras order)."""
        exo = self._exo_from_true(hi_res_true, list(exo_names))
        hr_gen = (self.gen_forward or self.gen.forward)(low_res, exo)
        n_exo = len(exo_names)
        if n_exo:
            gen_full = np.concatenate(
                (hr_gen, hi_res_true[..., -n_exo:]), axis=-1)
        else:
            gen_full = hr_gen
        if gen_full.shape != hi_res_true.shape:
            raise RuntimeError('shape mismatch {} vs {}'.format(
                gen_full.shape, hi_res_true.shape))
        nb = hi_res_true.shape[0]
        d_both = (self.disc_forward or self.disc.forward)(
            np.concatenate((hi_res_true, gen_full), axis=0))
        d_true, d_gen = d_both[:nb], d_both[nb:]
        if self.pre_backward is not None:
            self.pre_backward()
        details = {}
        if compute_disc or train_disc:
            ld, g_dt, g_dg = rel_bce(d_true, d_gen)
            details['loss_disc'] = ld
        if train_gen:
            sl = slice(0, None) if n_exo == 0 else slice(0, -n_exo)
            lc, cdet, g_content = content_loss(
                self.loss, gen_full[..., sl], hi_res_true[..., sl])
            # adversarial: roles swapped (base.py:899-901)
            la, g_as_true, _ = rel_bce(d_gen, d_true)
            loss = lc + weight_gen_advers * la
            details.update(loss_gen=loss, loss_gen_content=lc,
                           loss_gen_advers=la)
            details.update(cdet)
            # d loss / d D(gen) only (D(true) does not depend on gen weights)
            d_d = np.concatenate((np.zeros_like(d_true),
                                  weight_gen_advers * g_as_true), axis=0)
            d_in = self.disc.backward(d_d.astype(d_both.dtype))[nb:]
            d_gen_full = d_in
            d_hr_gen = d_gen_full[..., sl] + g_content if n_exo == 0 else (
                d_gen_full[..., :hr_gen.shape[-1]] + g_content)
            self.gen.backward(d_hr_gen.astype(hr_gen.dtype))
            return loss, details, self.gen.grads
        if train_disc:
            d_d = np.concatenate((g_dt, g_dg), axis=0)
            self.disc.backward(d_d.astype(d_both.dtype))
            return details['loss_disc'], details, self.disc.grads
        return None, details, None

    def train_batch(self, low_res, hi_res_true, weight_gen_advers=0.001,
                    train_gen=True, train_disc=True, exo_names=()):
        """_train_batch with both networks active (base.py:999-1026): gen step
        (compute_disc=train_disc) then disc step with UPDATED gen weights."""
        details = {}
        if train_gen:
            _, det, grads = self.loss_and_grads(
                low_res, hi_res_true, weight_gen_advers, train_gen=True,
                train_disc=False, compute_disc=train_disc,
                exo_names=exo_names)
            self.opt.apply_gradients(grads, self.gen.weights)
            details.update(det)
        if train_disc:
            _, det, grads = self.loss_and_grads(
                low_res, hi_res_true, weight_gen_advers, train_gen=False,
                train_disc=True, exo_names=exo_names)
            self.opt_disc.apply_gradients(grads, self.disc.weights)
            details.update(det)
        return {k: float(v) for k, v in details.items()}


def norm_input(low_res, means, stdevs):
    """abstract.py:229-236 (means/stdevs np.float32 per feature)."""
    means = np.array([np.float32(m) for m in means])
    stdevs = np.array([np.float32(s) for s in stdevs])
    stdevs = np.where(stdevs == 0, 1, stdevs)
    return (low_res.copy() - means) / stdevs


def un_norm_output(output, means, stdevs):
    """abstract.py:268-273."""
    means = np.array([np.float32(m) for m in means])
    stdevs = np.array([np.float32(s) for s in stdevs])
    return (output * stdevs) + means


class SolarCCOracle(GanOracle):
    """SolarCC.calc_loss + tape.gradient (sup3r/models/solar_cc.py:93-251):
    discriminator on the centre DAYLIGHT_HOURS of every true day and on
    ``time_samples`` windows of the synthetic field (:176-199, the random draw
    of :185-186 is an argument here), content loss on the centre
    POINT_LOSS_HOURS plus on the 24-h synthetic mean against the daylight true
    mean, averaged over the days (:207-232)."""

    STARTING_HOUR, DAYLIGHT_HOURS, POINT_LOSS_HOURS = 8, 8, 2

    def loss_and_grads(self, low_res, hi_res_true, weight_gen_advers=0.001,
                       train_gen=True, train_disc=False, compute_disc=False,
                       time_samples=(), hi_res_gen=None):
        hr_gen = self.gen.forward(low_res) if hi_res_gen is None else hi_res_gen
        if hr_gen.shape != hi_res_true.shape:
            raise RuntimeError('shape mismatch {} vs {}'.format(
                hr_gen.shape, hi_res_true.shape))
        assert hi_res_true.shape[3] % 24 == 0
        t_len = hi_res_true.shape[3]
        n_days = t_len // 24
        dl, s0, pl = self.DAYLIGHT_HOURS, self.STARTING_HOUR, self.POINT_LOSS_HOURS
        day = [slice(x, x + 24) for x in range(0, 24 * n_days, 24)]
        sub = [slice(s0 + x, s0 + x + dl) for x in range(0, 24 * n_days, 24)]
        pnt = [slice((24 - pl) // 2 + x, (24 - pl) // 2 + x + pl)
               for x in range(0, 24 * n_days, 24)]
        ts = [int(v) for v in time_samples]
        assert len(ts) == n_days
        nb = hi_res_true.shape[0]
        win_g = np.concatenate([hr_gen[:, :, :, t0:t0 + dl] for t0 in ts], axis=0)
        win_t = np.concatenate([hi_res_true[:, :, :, s] for s in sub], axis=0)
        nw = win_t.shape[0]
        d_both = self.disc.forward(np.concatenate((win_t, win_g), axis=0))
        d_true, d_gen = d_both[:nw], d_both[nw:]
        details = {}
        if compute_disc or train_disc:
            ld, g_dt, g_dg = rel_bce(d_true, d_gen)
            details['loss_disc'] = ld
        if train_gen:
            g = np.zeros(hr_gen.shape, dtype=np.float64)
            lc = 0.0
            for i in range(n_days):
                l1, det1, g1 = content_loss(self.loss, hr_gen[:, :, :, pnt[i]],
                                            hi_res_true[:, :, :, pnt[i]])
                g[:, :, :, pnt[i]] += g1 / n_days
                t_mean = hi_res_true[:, :, :, sub[i]].mean(axis=3)
                g_mean = hr_gen[:, :, :, day[i]].mean(axis=3)
                l2, det2, g2 = content_loss(self.loss, g_mean, t_mean)
                g[:, :, :, day[i]] += g2[:, :, :, None, :] / 24 / n_days
                lc += (l1 + l2) / n_days
                for k, v in det1.items():
                    details['c_sub_' + k] = details.get('c_sub_' + k, 0) + v / n_days
                for k, v in det2.items():
                    details['c_24h_' + k] = details.get('c_24h_' + k, 0) + v / n_days
            la, g_as_true, _ = rel_bce(d_gen, d_true)
            loss = lc + weight_gen_advers * la
            details.update(loss_gen=loss, loss_gen_content=lc, loss_gen_advers=la)
            if hi_res_gen is not None:
                return loss, details, None
            d_d = np.concatenate((np.zeros_like(d_true),
                                  weight_gen_advers * g_as_true), axis=0)
            d_in = self.disc.backward(d_d.astype(d_both.dtype))[nw:]
            for i, t0 in enumerate(ts):
                g[:, :, :, t0:t0 + dl] += d_in[i * nb:(i + 1) * nb]
            self.gen.backward(g.astype(hr_gen.dtype))
            return loss, details, self.gen.grads
        if train_disc:
            d_d = np.concatenate((g_dt, g_dg), axis=0)
            self.disc.backward(d_d.astype(d_both.dtype))
            return details['loss_disc'], details, self.disc.grads
        return None, details, None
