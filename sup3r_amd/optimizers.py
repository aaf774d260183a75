"""Optimizer config objects with the keras surface sup3r touches
(``get_config`` / ``from_config`` / ``.learning_rate`` / ``.iterations`` /
slot-variable names for the ``OptmGen/Adam/m/...`` history columns,
sup3r/models/abstract.py:321-350,543-587).  The update itself is the fused
multi-tensor HIP kernel behind ``s3_adam_step``."""


class Adam:
    """keras-2.15 Adam hyper-parameters (defaults: lr 1e-3 in keras; sup3r
    passes 1e-4; beta_1 0.9, beta_2 0.999, epsilon 1e-7)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7, name='Adam', **kwargs):
        if kwargs.get('amsgrad', False):
            raise KeyError('amsgrad=True has no MI355X kernel mapping')
        self.learning_rate = float(learning_rate)
        self.beta_1 = float(beta_1)
        self.beta_2 = float(beta_2)
        self.epsilon = float(epsilon)
        self.name = name
        self.iterations = 0

    def get_config(self):
        return {'name': self.name, 'learning_rate': self.learning_rate,
                'beta_1': self.beta_1, 'beta_2': self.beta_2,
                'epsilon': self.epsilon, 'amsgrad': False}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


OPTIMIZERS = {'Adam': Adam, 'adam': Adam}


def get_optimizer_class(conf):
    """models/utilities.py:150-158."""
    name = conf['name']
    if name not in OPTIMIZERS:
        raise KeyError(f'optimizer "{name}" has no MI355X kernel mapping '
                       f'(available: {sorted(set(OPTIMIZERS))})')
    return OPTIMIZERS[name]


def init_optimizer(optimizer, learning_rate):
    """abstract.py:321-350."""
    if isinstance(optimizer, dict):
        cls = get_optimizer_class(optimizer)
        keys = ('learning_rate', 'beta_1', 'beta_2', 'epsilon', 'name',
                'amsgrad')
        return cls.from_config({k: v for k, v in optimizer.items()
                                if k in keys})
    if optimizer is None:
        return Adam(learning_rate=learning_rate)
    if isinstance(optimizer, str):
        return get_optimizer_class({'name': optimizer})(
            learning_rate=learning_rate)
    return optimizer
