"""``Sup3rGanDC`` — data-centric GAN on the MI355X engine: the end-of-epoch
validation pass evaluates the generator loss per (space bin, time bin) and
hands the normalised bin losses back to the batch handler as sampling weights.
Mirrors sup3r/models/dc.py (calc_val_loss_gen :16-62, calc_val_loss :64-116);
the loss evaluation itself is ``Sup3rGan._get_hr_exo_and_loss`` on the device.
"""
import logging

import numpy as np

from .gan import Sup3rGan

logger = logging.getLogger(__name__)


class Sup3rGanDC(Sup3rGan):
    """Data-centric model using loss across time bins to select training
    observations"""

    def calc_val_loss_gen(self, batch_handler, weight_gen_advers):
        """Total and content generator losses of every validation bin, shape
        (n_space_bins, n_time_bins): batch i of ``val_data`` is bin
        (i // n_time_bins, i % n_time_bins)."""
        shape = (batch_handler.n_space_bins, batch_handler.n_time_bins)
        total_losses = np.zeros(shape, dtype=np.float32)
        content_losses = np.zeros(shape, dtype=np.float32)
        for i, batch in enumerate(batch_handler.val_data):
            logger.info(f'Calculating validation loss for batch {i} / '
                        f'{len(batch_handler.val_data)}...')
            loss, loss_details, _, _ = self._get_hr_exo_and_loss(
                low_res=batch.low_res, hi_res_true=batch.high_res,
                weight_gen_advers=weight_gen_advers)
            row, col = divmod(i, batch_handler.n_time_bins)
            total_losses[row, col] = float(loss)
            content_losses[row, col] = float(loss_details['loss_gen_content'])
        return total_losses, content_losses

    def calc_val_loss(self, batch_handler, weight_gen_advers):
        """Updates the batch handler's spatial / temporal sampling weights from
        the bin losses and returns the mean validation losses."""
        logger.debug('Starting end-of-epoch validation loss calculation...')
        loss_details = {}
        total_losses, content_losses = self.calc_val_loss_gen(
            batch_handler, weight_gen_advers)
        t_weights = total_losses.mean(axis=0)
        t_weights /= t_weights.sum()
        s_weights = total_losses.mean(axis=1)
        s_weights /= s_weights.sum()
        logger.debug(
            f'Previous spatial weights: {batch_handler.spatial_weights}')
        logger.debug(
            f'Previous temporal weights: {batch_handler.temporal_weights}')
        batch_handler.update_weights(spatial_weights=s_weights,
                                     temporal_weights=t_weights)
        logger.debug('New spatiotemporal weights (space, time):\n'
                     f'{total_losses / total_losses.sum()}')
        logger.debug(f'New spatial weights: {s_weights}')
        logger.debug(f'New temporal weights: {t_weights}')
        loss_details['mean_val_loss_gen'] = round(np.mean(total_losses), 3)
        loss_details['mean_val_loss_gen_content'] = round(
            np.mean(content_losses), 3)
        return loss_details
