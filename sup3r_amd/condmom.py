"""``Sup3rCondMom`` (conditional-moments model) on the MI355X engine.

Generator only; the loss is ``MSE(gen * mask, true * mask)`` and the batches
carry ``.output`` (the moment to learn) and ``.mask`` next to ``.low_res``.
Behaviour follows sup3r/models/conditional.py (constructor :23-99,
``calc_loss_cond_mom`` :221-241, ``calc_loss`` :243-283, ``calc_val_loss``
:285-313, ``_train_epoch`` :315-361, ``train`` :363-489).  The conv stack, the
reverse pass, Adam, the loss futures and the sharded multi-GPU step are
``Sup3rGan``'s; this class only swaps the loss inputs and the checkpoint
contents.
"""
import logging
import os
import time

from .compute import LossFuture
from .gan import Sup3rGan

logger = logging.getLogger(__name__)

_REPORTED = ('mean_squared_error', 'loss_gen')    # conditional.py:280-282


class _Reported(LossFuture):
    """A loss future narrowed to the two keys the reference reports."""

    def __init__(self, inner):
        self._inner = inner

    def resolve(self):
        full = self._inner.resolve()
        return {k: full[k] for k in _REPORTED}

    @property
    def _recipe(self):
        # (a recorded step rebuilds its futures from (scal, recipe, scale):
        # captured.py)
        inner = self._inner._recipe

        def narrowed(vals):
            full = inner(vals)
            return {k: full[k] for k in _REPORTED}
        return narrowed

    @property
    def _scale(self):
        return self._inner._scale

    @_scale.setter
    def _scale(self, v):
        self._inner._scale = v

    @property
    def _scal(self):
        return self._inner._scal

    @_scal.setter
    def _scal(self, v):
        self._inner._scal = v


class Sup3rCondMom(Sup3rGan):
    """Basic Sup3r conditional moments model."""

    def __init__(self, gen_layers, optimizer=None, learning_rate=1e-4,
                 num_par=None, history=None, meta=None, means=None,
                 stdevs=None, default_device=None, name=None, precision=None):
        super().__init__(gen_layers, None, loss='MeanSquaredError',
                         optimizer=optimizer, learning_rate=learning_rate,
                         history=history, meta=meta, means=means,
                         stdevs=stdevs, default_device=default_device,
                         name=name, precision=precision)
        self._num_par = num_par or 0

    # ---- checkpoint: one network
    def save(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        self.generator.save(os.path.join(out_dir, 'model_gen.pkl'))
        if self.history is not None:
            self.history.to_csv(os.path.join(out_dir, 'history.csv'))
        self.save_params(out_dir)

    @classmethod
    def load(cls, model_dir, verbose=True):
        params = cls.load_saved_params(model_dir, verbose=verbose)
        for unused in ('loss', 'optimizer_disc'):
            params.pop(unused, None)
        return cls(os.path.join(model_dir, 'model_gen.pkl'), **params)

    discriminator_weights = property(lambda self: [])
    weights = property(lambda self: self.generator_weights)

    @property
    def model_params(self):
        params = super().model_params
        for unused in ('loss', 'optimizer_disc', 'default_device'):
            params.pop(unused, None)
        if self.generator.built:
            self._num_par = int(sum(w.size for w in self.generator_weights))
        params['num_par'] = self._num_par
        return params

    def init_weights(self, lr_shape, hr_shape, device=None):
        if not self._gen.built:
            self._gen.build(tuple(lr_shape),
                            seed=getattr(type(self._gen), '_global_seed', None))

    # ---- loss
    def _masked(self, low_res, output_true, mask, **kw):
        _, details, _ = self._compute.loss_and_grads(
            low_res, output_true, self._loss_terms, train_gen=True,
            exo_names=self.hr_exo_features, mask=mask, **kw)
        if isinstance(details, LossFuture):
            return _Reported(details)
        return {k: details[k] for k in _REPORTED}

    def calc_loss(self, output_true, output_gen, mask):
        details = self._masked(None, output_true, mask, backward=False,
                               hi_res_gen=output_gen)
        return details['loss_gen'], details

    def calc_loss_cond_mom(self, output_true, output_gen, mask):
        loss, details = self.calc_loss(output_true, output_gen, mask)
        return loss, {'mean_squared_error': details['mean_squared_error']}

    def get_single_grad(self, low_res, hi_res_true, training_weights=None,
                        device_name=None, mask=None, defer=False, scal=None,
                        accumulate_wgrad=False, **unused):
        return 'gen', self._masked(low_res, hi_res_true, mask, backward=True,
                                   defer=defer, scal=scal,
                                   accumulate_wgrad=accumulate_wgrad)

    def calc_val_loss(self, batch_handler):
        n_val = len(batch_handler.val_data)
        for vb in batch_handler.val_data:
            self.update_loss_details(
                self._val_window,
                self._masked(vb.low_res, vb.output, vb.mask, backward=False),
                n_val)
        return self._val_window.means()

    # ---- training
    def _train_step(self, batch, multi_gpu=False):
        """one mini-batch (conditional.py:363-489's loop body): gradients of
        the masked loss + one optimizer step, returned as a ``LossFuture``.  A
        launch-bound step (``capture_steps``, as ``Sup3rGan._launch_batch``) is
        recorded once and replayed as one hipGraphLaunch (captured.py) — at
        BASELINE.md's shape (N, 4, 4, 4, 2) the ~250 launches of a step are all
        a few microseconds long."""
        def body(b):
            return [self.run_gradient_descent(
                b.low_res, b.output, None, optimizer=self.optimizer,
                multi_gpu=multi_gpu, mask=b.mask, defer=True)]
        rec = self._condmom_recorder(batch, multi_gpu)
        if rec is not None:
            return rec.run(batch, ('condmom',), [self.optimizer], body,
                           model=self)[0]
        return body(batch)[0]

    def _condmom_recorder(self, batch, multi_gpu):
        mode = self.capture_steps
        if not mode or self._replica_layout(multi_gpu)[1] != 1:
            return None
        from .compute import HipGanCompute
        own = (type(self).get_single_grad is Sup3rCondMom.get_single_grad
               and type(self).run_gradient_descent
               is Sup3rGan.run_gradient_descent
               and type(self._compute) is HipGanCompute)
        if not own:
            return None
        rec = getattr(self, '_recorder', None)
        if rec is None or rec.compute is not self._compute:
            from .captured import StepRecorder
            rec = self._recorder = StepRecorder(
                self._compute, fields=('low_res', 'output', 'mask'))
        return rec if rec.eligible(self, batch, mode, False) else None

    def _train_epoch(self, batch_handler, multi_gpu=False):
        """One pass over the handler; nothing is read back from the device
        until the last batch is enqueued."""
        n = len(batch_handler)
        pending = []
        for batch in batch_handler:
            self.init_weights(batch.low_res.shape, batch.output.shape)
            self._sync_replicas()
            pending.append(self._train_step(batch, multi_gpu))
        for step in pending:
            details = step.resolve() if isinstance(step, LossFuture) else step
            self.update_loss_details(self._train_window, details, n)
        return self._train_window.means()

    def train(self, batch_handler, input_resolution, n_epoch,
              checkpoint_int=None, out_dir='./condMom_{epoch}',
              early_stop_on=None, early_stop_threshold=0.005,
              early_stop_n_epoch=5, multi_gpu=False, tensorboard_log=False):
        if multi_gpu:
            self._join_replicas()
        self.set_norm_stats(batch_handler.means, batch_handler.stds)
        self.set_model_params(
            input_resolution=input_resolution,
            s_enhance=batch_handler.s_enhance,
            t_enhance=batch_handler.t_enhance,
            **{k: getattr(batch_handler, k) for k in (
                'smoothing', 'lr_features', 'hr_exo_features',
                'hr_out_features', 'smoothed_features')})
        epochs = self._ledger.next_epochs(n_epoch)
        t0 = time.time()
        for epoch in epochs:
            summary = self._train_epoch(batch_handler, multi_gpu=multi_gpu)
            summary.update(self.calc_val_loss(batch_handler))
            logger.info('epoch %d / %d: ' + ', '.join(
                f'{k} {summary[k]:.2e}' for k in ('train_loss_gen',
                                                  'val_loss_gen')
                if k in summary), epoch, epochs[-1])
            rate = self.get_optimizer_config(self.optimizer)['learning_rate']
            if self.finish_epoch(epoch, epochs, t0, summary, checkpoint_int,
                                 out_dir, early_stop_on, early_stop_threshold,
                                 early_stop_n_epoch,
                                 extras={'learning_rate_gen': rate}):
                break
        batch_handler.stop()
