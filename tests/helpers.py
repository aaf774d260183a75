"""Test fixtures: a synthetic batch handler implementing the BatchHandler
*protocol* sup3r's ``Sup3rGan.train`` consumes (attributes + iteration +
``DsetTuple``-like batches) without the reference's TF-backed
queue."""
import collections

import numpy as np

Batch = collections.namedtuple('Batch', ['low_res', 'high_res'])
MomBatch = collections.namedtuple(
    'MomBatch', ['low_res', 'high_res', 'output', 'mask'])


def coarsen(hr, s, t):
    """spatial mean-coarsening + temporal subsampling (what
    SingleBatchQueue.transform does on the host, batch_queues/base.py:32-87)."""
    if hr.ndim == 5:
        n, a, b, c, f = hr.shape
        lr = hr.reshape(n, a // s, s, b // s, s, c, f).mean(axis=(2, 4))
        return lr[:, :, :, ::t]
    n, a, b, f = hr.shape
    return hr.reshape(n, a // s, s, b // s, s, f).mean(axis=(2, 4))


def smooth_field(rng, shape):
    """Random field with large-scale structure (so super-resolution is
    learnable): sum of a few random plane waves + small noise."""
    grids = np.meshgrid(*[np.linspace(0, 1, n) for n in shape[1:-1]],
                        indexing='ij')
    out = np.zeros(shape, np.float32)
    for n in range(shape[0]):
        for f in range(shape[-1]):
            v = 0
            for _ in range(3):
                k = rng.uniform(-6, 6, size=len(grids))
                ph = rng.uniform(0, 2 * np.pi)
                v = v + np.sin(sum(ki * g for ki, g in zip(k, grids)) + ph)
            out[n, ..., f] = v / 2 + 0.05 * rng.standard_normal(shape[1:-1])
    return out


class ValData:
    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class SyntheticBatchHandler:
    def __init__(self, sample_shape, s_enhance, t_enhance, features,
                 batch_size=4, n_batches=3, n_val=1, seed=0, exo_features=()):
        rng = np.random.default_rng(seed)
        self.s_enhance, self.t_enhance = s_enhance, t_enhance
        self.lr_features = list(features)
        self.hr_out_features = list(features)
        self.hr_exo_features = list(exo_features)
        self.smoothing = None
        self.smoothed_features = []
        nf = len(features) + len(exo_features)
        is_5d = len(sample_shape) == 3 and sample_shape[2] > 1
        hr_sp = tuple(sample_shape) if is_5d else tuple(sample_shape[:2])
        self.hr_shape = hr_sp + (nf,)
        lr_sp = (hr_sp[0] // s_enhance, hr_sp[1] // s_enhance) + (
            (hr_sp[2] // t_enhance,) if is_5d else ())
        self.lr_shape = lr_sp + (len(features),)
        self.shapes = ((batch_size,) + self.lr_shape,
                       (batch_size,) + self.hr_shape)
        self.means = {f: 0.0 for f in list(features) + list(exo_features)}
        self.stds = {f: 1.0 for f in list(features) + list(exo_features)}

        def make():
            hr = smooth_field(rng, (batch_size,) + self.hr_shape)
            lr = coarsen(hr[..., :len(features)], s_enhance,
                         t_enhance if is_5d else 1)
            return Batch(lr.astype(np.float32), hr.astype(np.float32))
        self.batches = [make() for _ in range(n_batches)]
        self.val_data = ValData([make() for _ in range(n_val)])
        self.stopped = False

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)

    def stop(self):
        self.stopped = True


class SyntheticMomBatchHandler(SyntheticBatchHandler):
    """Conditional-moment batches (batch_queues/conditional.py:79-167): first
    moment target = the hi-res field itself, mask of ones with an optional
    zeroed border."""

    def __init__(self, *args, pad=0, **kwargs):
        super().__init__(*args, **kwargs)

        def to_mom(b):
            mask = np.ones_like(b.high_res)
            if pad:
                mask[:, :pad] = 0
                mask[:, -pad:] = 0
            return MomBatch(b.low_res, b.high_res, b.high_res.copy(), mask)
        self.batches = [to_mom(b) for b in self.batches]
        self.val_data = ValData([to_mom(b) for b in self.val_data.batches])
