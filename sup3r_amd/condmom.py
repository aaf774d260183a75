"""``Sup3rCondMom`` (conditional-moments model) on the MI355X engine:
generator only, masked-MSE loss, batches carry ``.output`` and ``.mask``.
Mirrors sup3r/models/conditional.py (ctor :23-99, calc_loss_cond_mom :221-241,
calc_loss :243-283, calc_val_loss :285-313, _train_epoch :315-361, train
:363-489); shares the conv stack, reverse pass and Adam with ``Sup3rGan``.
"""
import logging
import os
import time

import pandas as pd

from .gan import Sup3rGan

logger = logging.getLogger(__name__)


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
        self._num_par = num_par if num_par is not None else 0

    # -- persistence: generator only
    def save(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        self.generator.save(os.path.join(out_dir, 'model_gen.pkl'))
        if isinstance(self.history, pd.DataFrame):
            self.history.to_csv(os.path.join(out_dir, 'history.csv'))
        self.save_params(out_dir)
        logger.info('Saved model to disk in directory: {}'.format(out_dir))

    @classmethod
    def load(cls, model_dir, verbose=True):
        fp_gen = os.path.join(model_dir, 'model_gen.pkl')
        params = cls.load_saved_params(model_dir, verbose=verbose)
        for k in ('loss', 'optimizer_disc'):
            params.pop(k, None)
        return cls(fp_gen, **params)

    @property
    def discriminator_weights(self):
        return []

    @property
    def weights(self):
        return self.generator_weights

    @property
    def model_params(self):
        p = super().model_params
        p.pop('loss', None)
        p.pop('optimizer_disc', None)
        p.pop('default_device', None)
        if self.generator.built:
            self._num_par = int(sum(w.size for w in self.generator_weights))
        p['num_par'] = self._num_par
        return p

    def init_weights(self, lr_shape, hr_shape, device=None):
        if not self._gen.built:
            seed = getattr(type(self._gen), '_global_seed', None)
            self._gen.build(tuple(lr_shape), seed=seed)

    # -- loss
    def calc_loss_cond_mom(self, output_true, output_gen, mask):
        loss, details = self.calc_loss(output_true, output_gen, mask)
        return loss, {k: v for k, v in details.items() if k != 'loss_gen'}

    def calc_loss(self, output_true, output_gen, mask):
        loss, details, _ = self._compute.loss_and_grads(
            None, output_true, self._loss_terms, train_gen=True,
            exo_names=self.hr_exo_features, backward=False,
            hi_res_gen=output_gen, mask=mask)
        return loss, self._details(details)

    @staticmethod
    def _details(details):
        # conditional.py:280-282: {'mean_squared_error': ..., 'loss_gen': ...}
        return {'mean_squared_error': details['mean_squared_error'],
                'loss_gen': details['loss_gen']}

    def get_single_grad(self, low_res, hi_res_true, training_weights=None,
                        device_name=None, mask=None, **kwargs):
        _, details, _ = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms, train_gen=True,
            exo_names=self.hr_exo_features, backward=True, mask=mask)
        return 'gen', self._details(details)

    def calc_val_loss(self, batch_handler):
        logger.debug('Starting end-of-epoch validation loss calculation...')
        for val_batch in batch_handler.val_data:
            _, details, _ = self._compute.loss_and_grads(
                val_batch.low_res, val_batch.output, self._loss_terms,
                train_gen=True, exo_names=self.hr_exo_features,
                backward=False, mask=val_batch.mask)
            self._val_record = self.update_loss_details(
                self._val_record, self._details(details),
                len(batch_handler.val_data), prefix='val_')
        return self._val_record.mean(axis=0)

    # -- training
    def _train_epoch(self, batch_handler, multi_gpu=False):
        loss_details = {}
        for ib, batch in enumerate(batch_handler):
            self.init_weights(batch.low_res.shape, batch.output.shape)
            b_loss_details = self.run_gradient_descent(
                batch.low_res, batch.output, None, optimizer=self.optimizer,
                multi_gpu=multi_gpu, mask=batch.mask)
            self._train_record = self.update_loss_details(
                self._train_record, b_loss_details, len(batch_handler),
                prefix='train_')
            loss_details = self._train_record.mean().to_dict()
            logger.debug('Batch {} out of {} has epoch-average gen loss of: '
                         '{:.2e}. '.format(ib, len(batch_handler),
                                           loss_details['train_loss_gen']))
        return loss_details

    def train(self, batch_handler, input_resolution, n_epoch,
              checkpoint_int=None, out_dir='./condMom_{epoch}',
              early_stop_on=None, early_stop_threshold=0.005,
              early_stop_n_epoch=5, multi_gpu=False, tensorboard_log=False):
        self.set_norm_stats(batch_handler.means, batch_handler.stds)
        self.set_model_params(
            input_resolution=input_resolution,
            s_enhance=batch_handler.s_enhance,
            t_enhance=batch_handler.t_enhance,
            smoothing=batch_handler.smoothing,
            lr_features=batch_handler.lr_features,
            hr_exo_features=batch_handler.hr_exo_features,
            hr_out_features=batch_handler.hr_out_features,
            smoothed_features=batch_handler.smoothed_features)
        epochs = list(range(n_epoch))
        if self._history is None:
            self._history = pd.DataFrame(columns=['elapsed_time'])
            self._history.index.name = 'epoch'
        else:
            epochs = [e + int(self._history.index.values[-1]) + 1
                      for e in epochs]
        t0 = time.time()
        logger.info('Training model for {} epochs starting at epoch {}'.format(
            n_epoch, epochs[0]))
        for epoch in epochs:
            loss_details = self._train_epoch(batch_handler,
                                             multi_gpu=multi_gpu)
            loss_details.update(self.calc_val_loss(batch_handler))
            msg = 'Epoch {} of {} gen train loss: {:.2e} '.format(
                epoch, epochs[-1], loss_details['train_loss_gen'])
            if 'val_loss_gen' in loss_details:
                msg += 'gen val loss: {:.2e} '.format(
                    loss_details['val_loss_gen'])
            logger.info(msg)
            lr_g = self.get_optimizer_config(self.optimizer)['learning_rate']
            stop = self.finish_epoch(
                epoch, epochs, t0, loss_details, checkpoint_int, out_dir,
                early_stop_on, early_stop_threshold, early_stop_n_epoch,
                extras={'learning_rate_gen': lr_g})
            if stop:
                break
        batch_handler.stop()
