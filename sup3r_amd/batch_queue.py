"""TF-free batch supply for training on the device.

What ``Sup3rGan.train`` consumes from a batch handler is small: iterate it for
``n_batches`` objects with ``.low_res`` / ``.high_res``, ``len()``, ``start()``
/ ``stop()``, ``shapes``, ``means`` / ``stds``, ``s_enhance`` / ``t_enhance``,
the feature lists and ``val_data`` (sup3r/models/base.py:624-828,1097-1191).
The reference supplies that with ``AbstractBatchQueue`` + ``SingleBatchQueue``
(sup3r/preprocessing/batch_queues/abstract.py:30-364, base.py:12-87) around a
``tf.queue.FIFOQueue`` and numpy / scipy coarsening on the host; this module
supplies the same surface — same constructor keywords, same attribute names —
with its own machinery:

* a :class:`_Feeder` owns the background thread: it draws raw hi-res batches
  from the samplers (weighted by their ``size``) into a bounded
  ``queue.Queue`` until told to stop — no TensorFlow anywhere;
* coarsening + smoothing (``SingleBatchQueue.transform``) run on the GPU
  (:class:`~sup3r_amd.batch_transform.DeviceBatchTransform`) when a batch is
  handed out, so what the training loop receives are device tensors.

Samplers are duck-typed the way the reference uses them: ``features``,
``sample_shape`` (hi-res s1, s2, t), ``batch_size``, optional ``size``
(relative sampling weight) and ``compute()``, ``next(sampler)`` ->
``(batch, s1, s2, t, features)``, optional ``lr_features`` / ``hr_features``
/ ``hr_features_ind`` / ``hr_out_features`` / ``hr_exo_features``.
"""
import logging
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .utilities import Timer

logger = logging.getLogger(__name__)


class DsetTuple(dict):
    """One batch: an ordered mapping member name -> array whose members are
    attributes as well and which indexes by position like a tuple
    (the role of sup3r/preprocessing/base.py:73-98)."""

    def __init__(self, **members):
        super().__init__(members)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return tuple(self.values())[key]
        return super().__getitem__(key)

    def __iter__(self):
        return iter(self.values())

    @property
    def dset_names(self):
        return list(self.keys())

    @property
    def dsets(self):
        return dict(self.items())


def _as_arrays(raw):
    """a sampler's return value (an array, or a tuple of them for dual
    samplers) as numpy"""
    if isinstance(raw, tuple):
        return tuple(np.asarray(a) for a in raw)
    return np.asarray(raw)


class _Feeder:
    """The background half of a queue: one daemon thread that keeps ``fifo``
    topped up with whatever ``draw()`` returns, ``workers`` draws at a time
    when a pool is given.  Restartable (a finished thread is replaced on the
    next ``start``)."""

    def __init__(self, draw, capacity, name, workers):
        self.draw, self.name = draw, name
        self.fifo = queue.Queue(maxsize=max(int(capacity), 0))
        self.capacity = int(capacity)
        self.workers = int(workers)
        self.pool = ThreadPoolExecutor(max_workers=self.workers)
        self.go = threading.Event()
        self.thread = self._new_thread()

    def _new_thread(self):
        return threading.Thread(target=self._loop, name=self.name,
                                daemon=True)

    def _loop(self):
        while self.go.is_set():
            want = self.capacity - self.fifo.qsize()
            if want <= 0:
                self.go.wait(0.001)              # full: look again shortly
                continue
            if self.workers > 1 and want > 1:
                drawn = list(self.pool.map(lambda _: self.draw(),
                                           range(want)))
            else:
                drawn = (self.draw() for _ in range(want))
            for item in drawn:
                while self.go.is_set():
                    try:
                        self.fifo.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        continue

    def start(self):
        self.go.set()
        if self.capacity <= 0 or self.thread.is_alive():
            return
        if self.thread.ident is not None:        # ran before: threads are
            self.thread = self._new_thread()     # single-use
        logger.info('%s queue: feeder thread started', self.name)
        self.thread.start()

    def stop(self):
        self.go.clear()
        if self.thread.is_alive():
            self.thread.join()
            logger.info('%s queue: feeder thread joined', self.name)

    def take(self):
        """next raw item, or None once the thread is gone and the FIFO empty"""
        while True:
            try:
                return self.fifo.get(timeout=0.05)
            except queue.Empty:
                if not self.thread.is_alive():
                    return None

    @property
    def backlog(self):
        return self.fifo.qsize()


class DeviceBatchQueue:
    """Hi-res sample batches from a list of samplers, coarsened / smoothed on
    the device when they are handed out."""

    BATCH_MEMBERS = ('low_res', 'high_res')

    def __init__(self, samplers, batch_size=16, n_batches=64, s_enhance=1,
                 t_enhance=1, queue_cap=None, transform_kwargs=None,
                 max_workers=1, thread_name='training', mode='lazy',
                 verbose=False, transform=None, seed=None):
        assert isinstance(samplers, list), (
            f'{type(self).__name__} requires a list of samplers. Received '
            f'type {type(samplers)}')
        self.containers = samplers
        self.batch_size, self.n_batches = int(batch_size), int(n_batches)
        self.s_enhance, self.t_enhance = int(s_enhance), int(t_enhance)
        self.mode, self.verbose = mode, verbose
        self.max_workers = int(max_workers)
        self.queue_cap = self.n_batches if queue_cap is None else queue_cap
        self.transform_kwargs = dict(transform_kwargs) if transform_kwargs \
            else {'smoothing_ignore': [], 'smoothing': None}
        self._transform = transform
        self._rng = np.random.default_rng(seed)
        self._rng_lock = threading.Lock()
        self._handed_out = 0
        self.container_index = 0
        self.timer = Timer()
        self._check_samplers()
        self._feeder = _Feeder(self.sample_batch,
                               0 if mode == 'eager' else self.queue_cap,
                               thread_name, self.max_workers)
        self._thread_name = thread_name

    # ------------------------------------------------- what the samplers say
    def _first(self, attr, default=None):
        return getattr(self.containers[0], attr, default)

    @property
    def features(self):
        return list(self._first('features'))

    @property
    def sample_shape(self):
        return tuple(self._first('sample_shape'))

    hr_sample_shape = sample_shape

    @property
    def lr_sample_shape(self):
        s1, s2, t = self.sample_shape
        return (s1 // self.s_enhance, s2 // self.s_enhance,
                t // self.t_enhance)

    @property
    def lr_features(self):
        return list(self._first('lr_features', self.features))

    @property
    def hr_features(self):
        return list(self._first('hr_features', self.features))

    @property
    def hr_out_features(self):
        return list(self._first('hr_out_features', self.hr_features))

    @property
    def hr_exo_features(self):
        return list(self._first('hr_exo_features', []))

    @property
    def hr_features_ind(self):
        ind = self._first('hr_features_ind')
        if ind is not None:
            return list(ind)
        feats = self.features
        return [feats.index(f) for f in self.hr_features]

    @property
    def container_weights(self):
        """sampling probability of each sampler: its share of the data"""
        w = np.fromiter((getattr(c, 'size', 1) for c in self.containers),
                        dtype=np.float64, count=len(self.containers))
        return (w / w.sum()).astype(np.float32)

    def _check_samplers(self):
        """the samplers must describe ONE kind of batch, and that batch must
        coarsen to whole cells; 'eager' mode loads their data up front"""
        feats, shape = self.features, self.sample_shape
        for c in self.containers[1:]:
            assert list(c.features) == feats, \
                'Received samplers with different sets of features.'
            assert tuple(c.sample_shape) == shape, (
                'Samplers have different values of "sample_shape": '
                f'{[tuple(k.sample_shape) for k in self.containers]}')
        seen = sorted({int(c.batch_size) for c in self.containers})
        assert seen == [self.batch_size], (
            f'Samplers have a different batch_size: {seen} than the '
            f'BatchQueue: {self.batch_size}')
        rem = (shape[0] % self.s_enhance, shape[1] % self.s_enhance,
               shape[2] % self.t_enhance)
        assert not any(rem), (
            f'The sample_shape {shape} is not consistent with the '
            f'enhancement factors {self.s_enhance, self.t_enhance}.')
        if self.mode == 'eager':
            logger.info('mode "eager": loading the sampler data now')
            for c in self.containers:
                if hasattr(c, 'compute'):
                    c.compute()

    # ----------------------------------------------------------- raw batches
    def get_random_container(self):
        with self._rng_lock:                     # (pool workers draw too)
            idx = self.container_index = int(self._rng.choice(
                len(self.containers), p=self.container_weights))
        return self.containers[idx]

    def sample_batch(self):
        """one raw batch from a sampler picked by weight"""
        return _as_arrays(next(self.get_random_container()))

    @property
    def queue_thread(self):
        return self._feeder.thread

    @property
    def queue_len(self):
        return self._feeder.backlog

    @property
    def running(self):
        return self._feeder.go.is_set()

    def start(self):
        """Keep the FIFO of raw batches full from now on (no-op in eager
        mode and for ``queue_cap`` 0: batches are then drawn on demand)."""
        self._feeder.start()

    def stop(self):
        """Stop drawing and join the feeder thread."""
        self._feeder.stop()

    def log_queue_info(self):
        return (f'{self._thread_name.title()} queue length: '
                f'{self.queue_len} / {self.queue_cap}')

    # -------------------------------------------------------- handing out
    def transform(self, samples, smoothing=None, smoothing_ignore=None,
                  temporal_coarsening_method='subsample'):
        """``SingleBatchQueue.transform`` (batch_queues/base.py:32-87) on the
        device: low_res = smooth(coarsen(samples)), high_res =
        samples[..., hr_features_ind]."""
        if self._transform is None:
            from .batch_transform import DeviceBatchTransform
            self._transform = DeviceBatchTransform(
                self.s_enhance, self.t_enhance, self.features,
                self.hr_features_ind).transform
        return self._transform(
            samples, smoothing=smoothing, smoothing_ignore=smoothing_ignore,
            temporal_coarsening_method=temporal_coarsening_method)

    def post_proc(self, samples):
        return DsetTuple(**dict(zip(
            self.BATCH_MEMBERS,
            self.transform(samples, **self.transform_kwargs))))

    def get_batch(self):
        """the next batch: a raw one from the feeder (drawn on the spot when
        no feeder runs), the length-1 time axis of spatial-only samples
        dropped, coarsened / smoothed"""
        raw = self._feeder.take() if self.queue_thread.is_alive() else None
        if raw is None:
            raw = self.sample_batch()
        if self.sample_shape[2] == 1:
            raw = tuple(a[..., 0, :] for a in raw) \
                if isinstance(raw, tuple) else raw[..., 0, :]
        return self.post_proc(raw)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        self._handed_out = 0
        self.start()
        return self

    def __next__(self):
        if self._handed_out >= self.n_batches:
            raise StopIteration
        self._handed_out += 1
        return self.timer(self.get_batch, log=self.verbose)()

    # --------------------------------------------------------------- shapes
    @property
    def lr_shape(self):
        return self.lr_sample_shape + (len(self.lr_features),)

    @property
    def hr_shape(self):
        return self.sample_shape + (len(self.hr_features),)

    @property
    def queue_shape(self):
        return [(self.batch_size,) + self.sample_shape
                + (len(self.features),)]

    @property
    def shapes(self):
        """(low_res, high_res) shapes of the batches ``__next__`` returns"""
        lr, hr = self.lr_shape, self.hr_shape
        if self.sample_shape[2] == 1:            # spatial-only: no time axis
            lr, hr = lr[:2] + lr[3:], hr[:2] + hr[3:]
        return (self.batch_size,) + lr, (self.batch_size,) + hr


class DeviceBatchHandler(DeviceBatchQueue):
    """Training queue + validation queue + normalisation stats: the object
    ``Sup3rGan.train`` consumes (``BatchHandlerFactory``,
    sup3r/preprocessing/batch_handlers/factory.py:33-310).  The reference
    builds its samplers from data containers; here ready samplers are handed
    in (duck-typed, see the module docstring) and ``means`` / ``stds`` are the
    per-feature dicts the model stores for ``norm_input`` / ``un_norm_output``
    (``StatsCollection`` is part of the data layer)."""

    def __init__(self, train_samplers, val_samplers=None, batch_size=16,
                 n_batches=64, s_enhance=1, t_enhance=1, means=None, stds=None,
                 queue_cap=None, transform_kwargs=None, max_workers=1,
                 mode='lazy', transform=None, seed=None):
        common = dict(batch_size=batch_size, n_batches=n_batches,
                      s_enhance=s_enhance, t_enhance=t_enhance,
                      queue_cap=queue_cap, transform_kwargs=transform_kwargs,
                      max_workers=max_workers, mode=mode, transform=transform,
                      seed=seed)
        super().__init__(samplers=train_samplers, **common)
        self.val_data = DeviceBatchQueue(
            samplers=val_samplers, thread_name='validation',
            **common) if val_samplers else []
        feats = self.features
        self.means = {f: np.float32(0.0) for f in feats} if means is None \
            else dict(means)
        self.stds = {f: np.float32(1.0) for f in feats} if stds is None \
            else dict(stds)

    @property
    def smoothing(self):
        return self.transform_kwargs.get('smoothing')

    @property
    def smoothed_features(self):
        if self.smoothing is None:
            return []
        skip = self.transform_kwargs.get('smoothing_ignore') or ()
        return [f for f in self.lr_features if f not in skip]

    def _both(self, what):
        if isinstance(self.val_data, DeviceBatchQueue):
            getattr(self.val_data, what)()
        getattr(DeviceBatchQueue, what)(self)

    def start(self):
        """start the validation queue along with the training queue"""
        self._both('start')

    def stop(self):
        """stop both queues"""
        self._both('stop')
