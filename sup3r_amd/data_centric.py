"""``Sup3rGanDC`` on the MI355X engine — the data-centric GAN whose validation
pass scores every (space bin, time bin) of the validation set and feeds the
normalised scores back to the batch handler as sampling weights.

API of sup3r/models/dc.py (``calc_val_loss_gen`` :16-62 returns the total and
content loss grids, ``calc_val_loss`` :64-116 updates the handler and returns
``mean_val_loss_gen`` / ``mean_val_loss_gen_content``); each bin is one
``Sup3rGan._get_hr_exo_and_loss`` on the device.
"""
import logging

import numpy as np

from .gan import Sup3rGan

logger = logging.getLogger(__name__)


def _unit_sum(v):
    """v / sum(v) as float32 (sampling weights)"""
    v = np.asarray(v, dtype=np.float32)
    return v / v.sum()


class Sup3rGanDC(Sup3rGan):
    """GAN that re-weights its training sampler by per-bin validation loss."""

    def _bin_scores(self, batch_handler, weight_gen_advers):
        """(2, n_space_bins, n_time_bins) float32: [0] total generator loss,
        [1] content loss; validation batch k is bin divmod(k, n_time_bins)."""
        n_s, n_t = batch_handler.n_space_bins, batch_handler.n_time_bins
        scores = np.zeros((2, n_s * n_t), dtype=np.float32)
        n_val = len(batch_handler.val_data)
        for k, batch in enumerate(batch_handler.val_data):
            logger.info('validation bin %d of %d', k, n_val)
            loss, details, *_ = self._get_hr_exo_and_loss(
                batch.low_res, batch.high_res,
                weight_gen_advers=weight_gen_advers)
            scores[0, k] = float(loss)
            scores[1, k] = float(details['loss_gen_content'])
        return scores.reshape(2, n_s, n_t)

    def calc_val_loss_gen(self, batch_handler, weight_gen_advers):
        total, content = self._bin_scores(batch_handler, weight_gen_advers)
        return total, content

    def calc_val_loss(self, batch_handler, weight_gen_advers):
        total, content = self.calc_val_loss_gen(batch_handler,
                                                weight_gen_advers)
        new = {'spatial_weights': _unit_sum(total.mean(axis=1)),
               'temporal_weights': _unit_sum(total.mean(axis=0))}
        for key, val in new.items():
            logger.debug('%s: %s -> %s', key,
                         getattr(batch_handler, key, None), val)
        batch_handler.update_weights(**new)
        return {'mean_val_loss_gen': round(float(total.mean()), 3),
                'mean_val_loss_gen_content': round(float(content.mean()), 3)}
