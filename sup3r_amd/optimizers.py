"""Optimizer config objects with the keras surface sup3r touches
(``get_config`` / ``from_config`` / ``.learning_rate`` / ``.iterations`` /
slot-variable names for the ``OptmGen/Adam/m/...`` history columns,
sup3r/models/abstract.py:321-350,543-587; ``get_optimizer_class``,
sup3r/models/utilities.py:150-158 resolves ANY ``tf.keras.optimizers`` name).
The update itself is one fused multi-tensor HIP kernel per step
(``s3_adam_step`` / ``s3_optimizer_step``): keras-2.15 ``update_step`` of
Adam, SGD (momentum / nesterov), RMSprop (not centered), Adagrad, Adamax and
AdamW.  Anything else (Nadam, Ftrl, Adadelta, Adafactor, Lion, amsgrad,
centered RMSprop, gradient clipping or EMA switched ON) raises ``KeyError``;
a verbatim keras ``get_config()`` dict (``ema_momentum=0.99`` with
``use_ema=False``, ``jit_compile=True``, ...) loads."""

# options of keras' base optimizer (every ``get_config()`` carries them,
# abstract.py:544-557 writes them into model_params.json).  Only the ones that
# change the update are refused, and only when they are switched on:
_CLIP_KW = ('clipnorm', 'clipvalue', 'global_clipnorm')      # on when not None
_EMA_KW = ('ema_momentum', 'ema_overwrite_frequency')        # read iff use_ema
_IGNORED_KW = ('jit_compile', 'is_legacy_optimizer')         # no maths in them
_BASE_KW = _CLIP_KW + _EMA_KW + _IGNORED_KW + ('use_ema',)


class _Optimizer:
    KIND = None
    DEFAULTS = {}

    def __init__(self, learning_rate=None, name=None, **kwargs):
        for k in _CLIP_KW:
            if kwargs.pop(k, None) is not None:
                raise KeyError(f'optimizer option "{k}" has no MI355X kernel '
                               'mapping')
        if kwargs.pop('use_ema', False):
            raise KeyError('optimizer option "use_ema" has no MI355X kernel '
                           'mapping')
        for k in _EMA_KW + _IGNORED_KW:
            kwargs.pop(k, None)
        if 'weight_decay' not in self.DEFAULTS and \
                kwargs.pop('weight_decay', None) not in (None, 0, 0.0):
            raise KeyError('weight_decay on this optimizer has no MI355X '
                           'kernel mapping (use AdamW)')
        conf = dict(self.DEFAULTS)
        if learning_rate is not None:
            conf['learning_rate'] = learning_rate
        unknown = set(kwargs) - set(conf)
        if unknown:
            raise KeyError(f'{type(self).__name__} got unknown / unsupported '
                           f'settings {sorted(unknown)}')
        conf.update(kwargs)
        self._check(conf)
        for k, v in conf.items():
            setattr(self, k, bool(v) if isinstance(self.DEFAULTS[k], bool)
                    else float(v))
        self.name = name or type(self).__name__
        self.iterations = 0

    def _check(self, conf):
        pass

    def get_config(self):
        conf = {'name': self.name}
        conf.update({k: getattr(self, k) for k in self.DEFAULTS})
        return conf

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def hyper(self):
        """hp[] of ``s3_optimizer_step`` (include/sup3r_hip.h)"""
        raise NotImplementedError


class Adam(_Optimizer):
    """keras-2.15 Adam (defaults: lr 1e-3 in keras; sup3r passes 1e-4)."""
    KIND = 0
    DEFAULTS = {'learning_rate': 1e-3, 'beta_1': 0.9, 'beta_2': 0.999,
                'epsilon': 1e-7, 'amsgrad': False}

    def _check(self, conf):
        if conf.get('amsgrad', False):
            raise KeyError('amsgrad=True has no MI355X kernel mapping')

    def hyper(self):
        return [self.learning_rate, self.beta_1, self.beta_2, self.epsilon]


class SGD(_Optimizer):
    KIND = 1
    DEFAULTS = {'learning_rate': 0.01, 'momentum': 0.0, 'nesterov': False}

    def hyper(self):
        return [self.learning_rate, self.momentum, float(self.nesterov)]


class RMSprop(_Optimizer):
    KIND = 2
    DEFAULTS = {'learning_rate': 1e-3, 'rho': 0.9, 'momentum': 0.0,
                'epsilon': 1e-7, 'centered': False}

    def _check(self, conf):
        if conf.get('centered', False):
            raise KeyError('centered RMSprop needs a third slot buffer: no '
                           'MI355X kernel mapping')

    def hyper(self):
        return [self.learning_rate, self.rho, self.momentum, self.epsilon]


class Adagrad(_Optimizer):
    KIND = 3
    DEFAULTS = {'learning_rate': 1e-3, 'initial_accumulator_value': 0.1,
                'epsilon': 1e-7}

    def hyper(self):
        return [self.learning_rate, self.epsilon,
                self.initial_accumulator_value]


class Adamax(_Optimizer):
    KIND = 4
    DEFAULTS = {'learning_rate': 1e-3, 'beta_1': 0.9, 'beta_2': 0.999,
                'epsilon': 1e-7}

    def hyper(self):
        return [self.learning_rate, self.beta_1, self.beta_2, self.epsilon]


class AdamW(_Optimizer):
    KIND = 5
    DEFAULTS = {'learning_rate': 1e-3, 'weight_decay': 0.004, 'beta_1': 0.9,
                'beta_2': 0.999, 'epsilon': 1e-7, 'amsgrad': False}
    _check = Adam._check

    def hyper(self):
        return [self.learning_rate, self.beta_1, self.beta_2, self.epsilon,
                self.weight_decay]


_CLASSES = (Adam, SGD, RMSprop, Adagrad, Adamax, AdamW)
OPTIMIZERS = {c.__name__: c for c in _CLASSES}
OPTIMIZERS.update({c.__name__.lower(): c for c in _CLASSES})


def get_optimizer_class(conf):
    """models/utilities.py:150-158."""
    name = conf['name']
    if name not in OPTIMIZERS:
        raise KeyError(f'optimizer "{name}" has no MI355X kernel mapping '
                       f'(available: {[c.__name__ for c in _CLASSES]})')
    return OPTIMIZERS[name]


def init_optimizer(optimizer, learning_rate):
    """abstract.py:321-350."""
    if isinstance(optimizer, dict):
        cls = get_optimizer_class(optimizer)
        conf = {k: v for k, v in optimizer.items()
                if k in cls.DEFAULTS or k in _BASE_KW
                or k == 'weight_decay'}
        if optimizer.get('name') not in (cls.__name__.lower(),):
            conf['name'] = optimizer['name']
        return cls.from_config(conf)
    if optimizer is None:
        return Adam(learning_rate=learning_rate)
    if isinstance(optimizer, str):
        return get_optimizer_class({'name': optimizer})(
            learning_rate=learning_rate)
    return optimizer
