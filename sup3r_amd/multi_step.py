"""``MultiStepGan`` — serial chain of single-step models.

Mirrors ``sup3r/models/multi_step.py:23-330`` of the reference: the same
constructor / ``load`` / ``generate`` surface, the 4-D <-> 5-D transposition
between spatial-only and spatiotemporal steps (:128-170), the feature matching
between steps (:172-196) and the per-step normalisation flags (:236-238).
Every step's conv stack runs on the MI355X through ``Sup3rGan.generate``.
"""
import json
import logging
import os

import numpy as np

from .utilities import ExoData

logger = logging.getLogger(__name__)


class MultiStepGan:
    """Ordered tuple of trained single-step models run back to back."""

    def __init__(self, models):
        self._models = tuple(models)

    def __len__(self):
        return len(self._models)

    @classmethod
    def load(cls, model_dirs, model_kwargs=None, verbose=True):
        """multi_step.py:42-84: one saved model directory per step; the class
        of each step is read from its ``model_params.json`` ``meta.class``."""
        import sup3r_amd
        if isinstance(model_dirs, str):
            model_dirs = [model_dirs]
        model_kwargs = model_kwargs or [{}] * len(model_dirs)
        if isinstance(model_kwargs, dict):
            model_kwargs = [model_kwargs]
        models = []
        for model_dir, kwargs in zip(model_dirs, model_kwargs):
            fp_params = os.path.join(model_dir, 'model_params.json')
            assert os.path.exists(fp_params), f'Could not find: {fp_params}'
            with open(fp_params) as f:
                params = json.load(f)
            meta = params.get('meta', {'class': 'Sup3rGan'})
            class_name = meta.get('class', 'Sup3rGan')
            Sup3rClass = getattr(sup3r_amd, class_name)
            models.append(Sup3rClass.load(model_dir, verbose=verbose, **kwargs))
        return cls(models)

    @property
    def models(self):
        return self._models

    @property
    def means(self):
        return tuple(model.means for model in self.models)

    @property
    def stdevs(self):
        return tuple(model.stdevs for model in self.models)

    @staticmethod
    def seed(s=0):
        from .gan import Sup3rGan
        Sup3rGan.seed(s=s)

    @staticmethod
    def _transpose_model_input(model, hi_res):
        """multi_step.py:128-170: a (n_obs, s1, s2, f) stack fed to a 5-D
        model becomes (1, s1, s2, n_obs-as-time, f) and vice versa."""
        if model.is_5d and len(hi_res.shape) == 4:
            hi_res = np.transpose(hi_res, axes=(1, 2, 0, 3))[np.newaxis]
        elif model.is_4d and len(hi_res.shape) == 5:
            msg = ('Recieved 5D input data with shape '
                   f'({hi_res.shape}) to a 4D model.')
            assert hi_res.shape[0] == 1, msg
            hi_res = np.transpose(hi_res[0], axes=(2, 0, 1, 3))
        else:
            msg = ('Recieved input data with shape '
                   f'{hi_res.shape} to a {model.input_dims}D model.')
            assert model.input_dims == len(hi_res.shape), msg
        return hi_res

    def _match_model_input(self, model_step, hi_res, exo_data):
        """multi_step.py:172-196: a step may use a subset of the previous
        step's output features."""
        if model_step > 0:
            current_model = self.models[model_step]
            previous_model = self.models[model_step - 1]
            output_feats = previous_model.hr_out_features
            input_feats = current_model.lr_features
            exo_data = exo_data or {}
            input_feats = [f for f in input_feats if f not in exo_data]
            if not set(input_feats).issubset(set(output_feats)):
                msg = ('Model step {} input features {} do not match '
                       'previous model step {} output features {}'.format(
                           model_step, input_feats, model_step - 1,
                           output_feats))
                logger.error(msg)
                raise ValueError(msg)
            lr_inds = [output_feats.index(fn) for fn in input_feats]
            hi_res = hi_res[..., lr_inds]
        return hi_res

    def generate(self, low_res, norm_in=True, un_norm_out=True,
                 exogenous_data=None):
        """multi_step.py:198-276."""
        if isinstance(exogenous_data, dict) and \
                not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        hi_res = np.array(low_res, copy=True)
        for i, model in enumerate(self.models):
            i_norm_in = not (i == 0 and not norm_in)
            i_un_norm_out = not (i + 1 == len(self.models) and not un_norm_out)
            i_exo_data = (None if exogenous_data is None
                          else exogenous_data.get_model_step_exo(i))
            try:
                hi_res = self._transpose_model_input(model, hi_res)
                hi_res = self._match_model_input(i, hi_res, i_exo_data)
                hi_res = model.generate(hi_res, norm_in=i_norm_in,
                                        un_norm_out=i_un_norm_out,
                                        exogenous_data=i_exo_data)
            except Exception as e:
                msg = ('Could not run model #{} of {} "{}" on tensor of '
                       'shape {}'.format(i + 1, len(self.models), model,
                                         hi_res.shape))
                logger.exception(msg)
                raise RuntimeError(msg) from e
        return hi_res

    # ---- aggregate views (multi_step.py:278-372)
    @property
    def meta(self):
        return tuple(model.meta for model in self.models)

    @property
    def lr_features(self):
        return self.models[0].lr_features

    @property
    def hr_out_features(self):
        return self.models[-1].hr_out_features

    @property
    def hr_exo_features(self):
        return [model.hr_exo_features for model in self.models]

    @property
    def model_params(self):
        return tuple(model.model_params for model in self.models)

    @property
    def s_enhancements(self):
        return [e for model in self.models for e in model.s_enhancements]

    @property
    def t_enhancements(self):
        return [e for model in self.models for e in model.t_enhancements]

    @property
    def s_enhance(self):
        return int(np.prod(self.s_enhancements))

    @property
    def t_enhance(self):
        return int(np.prod(self.t_enhancements))

    @property
    def input_dims(self):
        return self.models[0].input_dims

    @property
    def is_5d(self):
        return self.input_dims == 5

    @property
    def is_4d(self):
        return self.input_dims == 4
