"""``MultiStepGan`` — several trained single-step models run back to back
(what ``sup3r/models/multi_step.py:23-330`` provides: the
constructor / ``load`` / ``generate`` surface, the re-interpretation of the
array between spatial-only (4-D) and spatiotemporal (5-D) steps, the feature
selection between steps and the normalise-first / un-normalise-last rule).
Each step's conv stack runs on the MI355X through that model's ``generate``.
"""
import json
import logging
import os

import numpy as np

from .utilities import ExoData

logger = logging.getLogger(__name__)


def _as_model_rank(model, arr):
    """Present ``arr`` in the rank ``model`` consumes.  A stack of spatial
    fields ``(n, s1, s2, f)`` handed to a 5-D model is ONE sample whose time
    axis is the stack; a one-sample 5-D array handed to a 4-D model is a stack
    of its time steps."""
    if arr.ndim == model.input_dims:
        return arr
    if model.is_5d and arr.ndim == 4:
        return np.moveaxis(arr, 0, 2)[None]
    if model.is_4d and arr.ndim == 5:
        if arr.shape[0] != 1:
            raise AssertionError(
                f'a 4-D step can only take ONE 5-D sample, got {arr.shape}')
        return np.moveaxis(arr[0], 2, 0)
    raise AssertionError(f'{arr.shape} does not fit a {model.input_dims}-D '
                         'model')


class MultiStepGan:
    """Ordered tuple of trained single-step models."""

    def __init__(self, models):
        self._models = tuple(models)

    def __len__(self):
        return len(self._models)

    @classmethod
    def load(cls, model_dirs, model_kwargs=None, verbose=True):
        """One saved model directory per step; each step's class is the
        ``meta.class`` of its ``model_params.json`` (``Sup3rGan`` if absent)."""
        import sup3r_amd
        dirs = [model_dirs] if isinstance(model_dirs, str) else list(model_dirs)
        if model_kwargs is None:
            model_kwargs = [{}] * len(dirs)
        elif isinstance(model_kwargs, dict):
            model_kwargs = [model_kwargs]
        steps = []
        for d, kw in zip(dirs, model_kwargs):
            fp = os.path.join(d, 'model_params.json')
            if not os.path.exists(fp):
                raise AssertionError(f'{fp} does not exist')
            with open(fp) as f:
                meta = json.load(f).get('meta') or {}
            klass = getattr(sup3r_amd, meta.get('class', 'Sup3rGan'))
            steps.append(klass.load(d, verbose=verbose, **kw))
        return cls(steps)

    models = property(lambda self: self._models)
    means = property(lambda self: tuple(m.means for m in self._models))
    stdevs = property(lambda self: tuple(m.stdevs for m in self._models))
    meta = property(lambda self: tuple(m.meta for m in self._models))
    model_params = property(
        lambda self: tuple(m.model_params for m in self._models))
    lr_features = property(lambda self: self._models[0].lr_features)
    hr_out_features = property(lambda self: self._models[-1].hr_out_features)
    hr_exo_features = property(
        lambda self: [m.hr_exo_features for m in self._models])
    input_dims = property(lambda self: self._models[0].input_dims)
    is_5d = property(lambda self: self.input_dims == 5)
    is_4d = property(lambda self: self.input_dims == 4)

    @property
    def s_enhancements(self):
        return [e for m in self._models for e in m.s_enhancements]

    @property
    def t_enhancements(self):
        return [e for m in self._models for e in m.t_enhancements]

    s_enhance = property(lambda self: int(np.prod(self.s_enhancements)))
    t_enhance = property(lambda self: int(np.prod(self.t_enhancements)))

    @staticmethod
    def seed(s=0):
        from .gan import Sup3rGan
        Sup3rGan.seed(s=s)

    # kept under the reference's names for callers that reach for them
    @staticmethod
    def _transpose_model_input(model, hi_res):
        return _as_model_rank(model, hi_res)

    def _match_model_input(self, model_step, hi_res, exo_data):
        """Channels step ``model_step`` reads out of the previous step's
        output (its lo-res features minus what arrives as exogenous data), in
        the order it lists them."""
        if model_step == 0:
            return hi_res
        produced = self._models[model_step - 1].hr_out_features
        wanted = [f for f in self._models[model_step].lr_features
                  if f not in (exo_data or {})]
        missing = [f for f in wanted if f not in produced]
        if missing:
            raise ValueError(
                f'step {model_step} needs {missing}, step {model_step - 1} '
                f'only produces {produced}')
        return hi_res[..., [produced.index(f) for f in wanted]]

    def generate(self, low_res, norm_in=True, un_norm_out=True,
                 exogenous_data=None):
        """Run the chain.  Only the first step may skip normalising its input
        and only the last may skip un-normalising its output; every hand-over
        in between is in physical units."""
        if isinstance(exogenous_data, dict) and \
                not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        last = len(self._models) - 1
        arr = np.array(low_res, copy=True)
        for i, model in enumerate(self._models):
            step_exo = None if exogenous_data is None else \
                exogenous_data.get_model_step_exo(i)
            try:
                arr = self._match_model_input(
                    i, _as_model_rank(model, arr), step_exo)
                arr = model.generate(arr, norm_in=norm_in or i > 0,
                                     un_norm_out=un_norm_out or i < last,
                                     exogenous_data=step_exo)
            except Exception as e:
                raise RuntimeError(
                    f'step {i + 1} of {last + 1} ({type(model).__name__}) '
                    f'failed on an array of shape {arr.shape}') from e
        return arr
