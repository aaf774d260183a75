"""Small host utilities the model API relies on (counterparts of
sup3r/utilities/utilities.py ``Timer`` :261-335, ``camel_to_underscore``,
``safe_cast`` :140-152, and ``ExoData.get_combine_type_data``,
sup3r/preprocessing/data_handlers/exo.py:54-224)."""
import logging
import re
import time

import numpy as np

logger = logging.getLogger(__name__)


class Timer:
    """Wall-time decorator storing elapsed seconds per function name in
    ``.log`` (same keys the reference logs to tensorboard)."""

    def __init__(self):
        self.log = {}
        self._start = None
        self._stop = None

    def start(self):
        self._start = time.time()
        self._stop = None

    def stop(self):
        self._stop = time.time()

    @property
    def elapsed(self):
        end = time.time() if self._stop is None else self._stop
        return end - self._start

    @property
    def elapsed_str(self):
        return f'{round(self.elapsed, 5)} seconds'

    def __call__(self, func, call_id=None, log=False):
        def wrapper(*args, **kwargs):
            self.start()
            out = func(*args, **kwargs)
            self.stop()
            if call_id is not None:
                self.log.setdefault(call_id, {})[func.__name__] = self.elapsed
            else:
                self.log[func.__name__] = self.elapsed
            if log:
                logger.debug('Call to %s finished in %s', func.__name__,
                             self.elapsed_str)
            return out
        return wrapper


def camel_to_underscore(name):
    s1 = re.sub('(.)([A-Z][a-z]+)', r'\1_\2', name)
    return re.sub('([a-z0-9])([A-Z])', r'\1_\2', s1).lower()


def safe_cast(o):
    if hasattr(o, 'detach'):
        o = o.detach().cpu().numpy()
    if isinstance(o, (float, np.floating)):
        return float(o)
    if isinstance(o, (int, np.integer)):
        return int(o)
    if isinstance(o, (tuple, np.ndarray)):
        return list(o)
    if isinstance(o, (str, list)):
        return o
    return str(o)


def numpy_if_tensor(arr):
    """Duck-typed payload conversion (preprocessing/utilities.py:255-257),
    extended to device tensors."""
    if hasattr(arr, 'detach'):
        return arr.detach().cpu().numpy()
    return arr.numpy() if hasattr(arr, 'numpy') else arr


class LossValue(float):
    """A python float that also answers ``.numpy()`` like a 0-D tf.Tensor, so
    reference-style code (``loss.numpy() < other.numpy()``) keeps working."""

    def numpy(self):
        return float(self)


class ExoData(dict):
    """Feature -> {'steps': [{'model', 'combine_type', 'data'}, ...]} mapping
    with the lookup the generator needs (exo.py:54-224)."""

    def __init__(self, steps):
        super().__init__()
        if not isinstance(steps, dict):
            raise ValueError(
                'ExoData must be initialized with a dictionary of features.')
        for feat, entry in steps.items():
            assert 'steps' in entry, \
                f'ExoData entry for {feat} must have a "steps" key.'
            for i, step in enumerate(entry['steps']):
                assert 'data' in step and 'combine_type' in step, (
                    f'ExoData entry for {feat}, step #{i + 1}, must have a '
                    '"data" and "combine_type" key.')
        self.update(steps)

    def get_model_step_exo(self, model_step):
        """Entries of every feature whose 'model' index is ``model_step``
        (exo.py:108-130)."""
        out = {}
        for feature, entry in self.items():
            steps = [s for s in entry['steps'] if s.get('model') == model_step]
            if steps:
                out[feature] = {'steps': steps}
        return ExoData(out)

    def get_combine_type_data(self, feature, combine_type, model_step=None):
        steps = self[feature]['steps']
        if model_step is not None:
            steps = [s for s in steps if s.get('model') == model_step]
        types = [s['combine_type'] for s in steps]
        assert combine_type in types, (
            'Received exogenous_data without any combine_type '
            f'= "{combine_type}" steps.')
        return steps[types.index(combine_type)]['data']
