"""``Sup3rGanWithObs`` on the MI355X engine: a GAN whose generator takes
sparse observation fields mid-network (``Sup3rConcatObs`` layers) and whose
training simulates them by sparsely sampling the true hi-res field.

Behaviour follows sup3r/models/with_obs.py: constructor :31-86, observation
masks ``_get_single_obs_mask`` / ``_get_obs_mask`` / ``_get_full_obs_mask``
:113-216 (same draws from ``np.random.default_rng(42)`` = the reference's
``RANDOM_GENERATOR``, sup3r/utilities/utilities.py:24), ``get_hr_exo_input``
:234-249, the extra loss terms of ``_get_hr_exo_and_loss`` :251-279,
``model_params`` :218-232.  The conv stack, reverse pass, Adam and the train
loop are ``Sup3rGan``'s; the observation loss runs on the device through the
masked content-loss kernel (``compute.loss_and_grads(obs=...)``).

An un-observed cell reaches the generator as 0 in normalised units (see
``spec.py``, ``Sup3rConcatObs``): phygnn's own NaN handling is not available
in this environment — UNVERIFIED against phygnn.
"""
import logging

import numpy as np

from .compute import CONTENT_KINDS
from .gan import Sup3rGan

logger = logging.getLogger(__name__)


class Sup3rGanWithObs(Sup3rGan):
    """Sup3r GAN with mid-network observation fusion."""

    def __init__(self, *args, onshore_obs_frac=None, offshore_obs_frac=None,
                 loss_obs_weight=0.0, loss_obs=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.onshore_obs_frac = {} if onshore_obs_frac is None \
            else onshore_obs_frac
        self.offshore_obs_frac = {} if offshore_obs_frac is None \
            else offshore_obs_frac
        loss_obs = self.loss_name if loss_obs is None else loss_obs
        self.loss_obs_name = loss_obs
        name = loss_obs if isinstance(loss_obs, str) else None
        if name is None and isinstance(loss_obs, dict):
            keys = [k for k in loss_obs if k != 'term_weights']
            name = keys[0] if len(keys) == 1 else None
        if name not in CONTENT_KINDS:
            # the obs loss compares boolean-masked 1-D vectors
            # (with_obs.py:92-97): only pointwise metrics are defined on them
            raise KeyError(f'loss_obs "{loss_obs}" has no MI355X kernel on '
                           f'masked cells (one of {sorted(CONTENT_KINDS)})')
        self._loss_obs_kind = CONTENT_KINDS[name]
        self.loss_obs_weight = loss_obs_weight
        self._obs_rng = np.random.default_rng(seed=42)

    # ---------------------------------------------------- observation masks
    @property
    def obs_training_inds(self):
        """channels of the true hi-res data that play the observations
        (with_obs.py:101-111)"""
        hr = [f.replace('_obs', '') for f in self.hr_features]
        return [hr.index(f.replace('_obs', '')) for f in self.obs_features]

    def _get_single_obs_mask(self, hi_res, spatial_frac, time_frac=1.0):
        """True = NOT observed, for one batch entry (with_obs.py:113-143)"""
        n_t = hi_res.shape[3] if self.is_5d else 1
        s_hit = self._obs_rng.uniform(size=hi_res.shape[1:3]) <= spatial_frac
        t_hit = self._obs_rng.uniform(size=n_t) <= time_frac
        seen = s_hit[:, :, None, None] & t_hit[None, None, :, None]
        mask = np.repeat(~seen, len(self.hr_out_features), axis=-1)
        return mask if self.is_5d else mask[:, :, 0]

    def _get_obs_mask(self, hi_res, spatial_frac, time_frac=1.0):
        """with_obs.py:145-194: one mask per batch entry, fractions drawn
        between the given bounds"""
        def bounds(v):
            return list(v) if isinstance(v, (list, tuple)) else [v, v]
        n = hi_res.shape[0]
        s_fracs = np.clip(self._obs_rng.uniform(*bounds(spatial_frac),
                                                size=n), 0, 1)
        t_fracs = np.clip(self._obs_rng.uniform(*bounds(time_frac), size=n),
                          0, 1)
        return np.stack([self._get_single_obs_mask(hi_res, s, t)
                         for s, t in zip(s_fracs, t_fracs)], axis=0)

    def _get_full_obs_mask(self, hi_res):
        """with_obs.py:196-216: onshore mask, replaced by the (sparser)
        offshore one where the topography is <= 0"""
        hi_res = np.asarray(hi_res)
        mask = self._get_obs_mask(hi_res, self.onshore_obs_frac['spatial'],
                                  self.onshore_obs_frac.get('time', 1.0))
        if 'topography' in self.hr_features and self.offshore_obs_frac:
            topo = hi_res[..., self.hr_features.index('topography')]
            off = self._get_obs_mask(hi_res,
                                     self.offshore_obs_frac['spatial'],
                                     self.offshore_obs_frac.get('time', 1.0))
            mask = np.where(topo[..., None] > 0, mask, off)
        return mask

    @property
    def model_params(self):
        params = super().model_params
        params['onshore_obs_frac'] = self.onshore_obs_frac
        params['offshore_obs_frac'] = self.offshore_obs_frac
        params['loss_obs_weight'] = self.loss_obs_weight
        params['loss_obs'] = self.loss_obs_name
        return params

    # ------------------------------------------------------------- training
    def get_hr_exo_input(self, hi_res_true):
        """with_obs.py:234-249: the sparse observation fields cut out of the
        true hi-res data (+ ``'mask'``, True = not observed)"""
        hi = np.asarray(hi_res_true.cpu().numpy()
                        if hasattr(hi_res_true, 'cpu') else hi_res_true)
        exo = {}
        if not self.obs_features:
            return exo
        mask = self._get_full_obs_mask(hi)
        for j, (name, idx) in enumerate(zip(self.obs_features,
                                            self.obs_training_inds)):
            col = hi[..., idx:idx + 1]
            exo[name] = np.where(mask[..., j:j + 1], np.float32(0),
                                 col).astype(np.float32)
        exo['mask'] = mask
        return exo

    def _obs_kwargs(self, hi_res_true, calc_loss_kwargs):
        kw = dict(calc_loss_kwargs)
        exo = self.get_hr_exo_input(hi_res_true)
        mask = exo.pop('mask', None)
        kw['extra_exo'] = exo
        if mask is not None and kw.get('train_gen', True):
            kw['obs'] = ((~mask).astype(np.float32), self._loss_obs_kind,
                         self.loss_obs_weight)
        return kw, mask

    def _get_hr_exo_and_loss(self, low_res, hi_res_true, **calc_loss_kwargs):
        """with_obs.py:251-279"""
        kw, mask = self._obs_kwargs(hi_res_true, calc_loss_kwargs)
        loss, details, hr_gen = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms,
            exo_names=self.hr_exo_features, backward=False, **kw)
        return loss, details, hr_gen, {'mask': mask}

    def get_single_grad(self, low_res, hi_res_true, training_weights=None,
                        device_name=None, **calc_loss_kwargs):
        kw, _ = self._obs_kwargs(hi_res_true, calc_loss_kwargs)
        _, details, _ = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms,
            exo_names=self.hr_exo_features, backward=True, **kw)
        which = 'gen' if calc_loss_kwargs.get('train_gen', True) else 'disc'
        return which, details

    def _post_batch(self, ib, b_loss_details, n_batches, previous_means):
        if 'obs_frac' in b_loss_details:
            logger.debug('Batch {} out of {} has obs_frac: {:.4e}'.format(
                ib + 1, n_batches, b_loss_details['obs_frac']))
        return super()._post_batch(ib, b_loss_details, n_batches,
                                   previous_means)
