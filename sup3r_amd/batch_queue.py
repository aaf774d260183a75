"""TF-free batch queue feeding the device.

Mirrors ``AbstractBatchQueue`` / ``SingleBatchQueue``
(sup3r/preprocessing/batch_queues/abstract.py:30-364, base.py:12-87): a
dedicated thread keeps a FIFO of raw hi-res sample batches drawn from a list of
samplers; ``__next__`` dequeues one, squeezes the time axis of spatial-only
samples and runs ``transform`` — coarsening + smoothing, here on the GPU through
``DeviceBatchTransform`` — into a ``DsetTuple(low_res, high_res)``.  The
reference's FIFO is ``tf.queue.FIFOQueue`` (abstract.py:135-141) and therefore
needs TensorFlow; this one is a ``queue.Queue`` and hands out device tensors.

Samplers are duck-typed exactly as the reference uses them: ``features``,
``sample_shape`` (hi-res s1, s2, t), ``batch_size``, ``size`` (optional,
relative sampling weight), ``next(sampler)`` -> (batch, s1, s2, t, features),
and optionally ``lr_features`` / ``hr_features`` / ``hr_features_ind`` /
``hr_out_features`` / ``hr_exo_features``.
"""
import logging
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor, as_completed

import numpy as np

from .utilities import Timer

logger = logging.getLogger(__name__)


class DsetTuple:
    """namedtuple-like batch with dynamic attributes
    (sup3r/preprocessing/base.py:73-98)"""

    def __init__(self, **kwargs):
        self.dset_names = list(kwargs)
        self.__dict__.update(kwargs)

    @property
    def dsets(self):
        return {k: v for k, v in self.__dict__.items() if k in self.dset_names}

    def __iter__(self):
        return iter(self.dsets.values())

    def __getitem__(self, key):
        if isinstance(key, int):
            key = list(self.dsets)[key]
        return self.dsets[key]

    def __len__(self):
        return len(self.dsets)

    def __repr__(self):
        return f'DsetTuple({self.dsets})'


class DeviceBatchQueue:
    """Queue of hi-res sample batches, coarsened / smoothed on the device."""

    BATCH_MEMBERS = ('low_res', 'high_res')

    def __init__(self, samplers, batch_size=16, n_batches=64, s_enhance=1,
                 t_enhance=1, queue_cap=None, transform_kwargs=None,
                 max_workers=1, thread_name='training', mode='lazy',
                 verbose=False, transform=None, seed=None):
        msg = (f'{self.__class__.__name__} requires a list of samplers. '
               f'Received type {type(samplers)}')
        assert isinstance(samplers, list), msg
        self.containers = samplers
        self._batch_count = 0
        self._queue_thread = None
        self._training_flag = threading.Event()
        self._thread_name = thread_name
        self._thread_pool = ThreadPoolExecutor(max_workers=max_workers)
        self._rng = np.random.default_rng(seed)
        self.mode = mode
        self.s_enhance = s_enhance
        self.t_enhance = t_enhance
        self.batch_size = batch_size
        self.n_batches = n_batches
        self.queue_cap = n_batches if queue_cap is None else queue_cap
        self.max_workers = max_workers
        self.container_index = self.get_container_index()
        self.queue = queue.Queue(maxsize=max(self.queue_cap, 0))
        self.lr_sample_shape = (self.hr_sample_shape[0] // s_enhance,
                                self.hr_sample_shape[1] // s_enhance,
                                self.hr_sample_shape[2] // t_enhance)
        self.transform_kwargs = transform_kwargs or {'smoothing_ignore': [],
                                                     'smoothing': None}
        self.verbose = verbose
        self.timer = Timer()
        self._transform = transform
        self.preflight()

    # ------------------------------------------------------------ collection
    def check_shared_attr(self, attr):
        """the attribute every sampler must agree on (collections/base.py)"""
        vals = [getattr(c, attr) for c in self.containers]
        first = vals[0]
        msg = f'Samplers have different values of "{attr}": {vals}'
        assert all(np.array_equal(np.asarray(v, dtype=object),
                                  np.asarray(first, dtype=object))
                   for v in vals), msg
        return first

    @property
    def features(self):
        return list(self.containers[0].features)

    @property
    def sample_shape(self):
        return tuple(self.containers[0].sample_shape)

    hr_sample_shape = sample_shape

    @property
    def hr_features_ind(self):
        c = self.containers[0]
        if hasattr(c, 'hr_features_ind'):
            return list(c.hr_features_ind)
        return [self.features.index(f) for f in self.hr_features]

    @property
    def lr_features(self):
        return list(getattr(self.containers[0], 'lr_features', self.features))

    @property
    def hr_features(self):
        return list(getattr(self.containers[0], 'hr_features', self.features))

    @property
    def hr_out_features(self):
        return list(getattr(self.containers[0], 'hr_out_features',
                            self.hr_features))

    @property
    def hr_exo_features(self):
        return list(getattr(self.containers[0], 'hr_exo_features', []))

    @property
    def container_weights(self):
        sizes = np.array([getattr(c, 'size', 1) for c in self.containers],
                         dtype=np.float64)
        return (sizes / sizes.sum()).astype(np.float32)

    # --------------------------------------------------------------- checks
    def preflight(self):
        """Consistency of the samplers with each other and with the queue
        (feature lists, sample shape vs enhancement factors, batch size);
        eager mode materialises the sampler data first."""
        self.check_features()
        self.check_enhancement_factors()
        self.check_shared_attr('sample_shape')
        sizes = {int(c.batch_size) for c in self.containers}
        assert sizes == {int(self.batch_size)}, (
            f'Samplers have a different batch_size: {sorted(sizes)} than the '
            f'BatchQueue: {self.batch_size}')
        if self.mode == 'eager':
            logger.info('Received mode = "eager".')
            for c in self.containers:
                getattr(c, 'compute', lambda: None)()

    def check_features(self):
        ref = self.features
        for c in self.containers[1:]:
            assert list(c.features) == ref, \
                'Received samplers with different sets of features.'

    def check_enhancement_factors(self):
        s1, s2, t = self.sample_shape
        ok = not (s1 % self.s_enhance or s2 % self.s_enhance
                  or t % self.t_enhance)
        assert ok, (f'The sample_shape {self.sample_shape} is not consistent '
                    'with the enhancement factors '
                    f'{self.s_enhance, self.t_enhance}.')

    # ------------------------------------------------------------ transform
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
        tsamps = self.transform(samples, **self.transform_kwargs)
        return DsetTuple(**dict(zip(self.BATCH_MEMBERS, tsamps)))

    # ---------------------------------------------------------------- queue
    @property
    def queue_shape(self):
        return [(self.batch_size, *self.hr_sample_shape, len(self.features))]

    @property
    def queue_len(self):
        return self.queue.qsize() + self.queue_futures

    @property
    def queue_futures(self):
        return self._thread_pool._work_queue.qsize()

    @property
    def queue_thread(self):
        if self._queue_thread is None or not self._queue_thread.is_alive() \
                and self._queue_thread.ident is not None:
            self._queue_thread = threading.Thread(
                target=self.enqueue_batches, name=self._thread_name,
                daemon=True)
        return self._queue_thread

    def start(self):
        """Start thread to keep sample queue full for batches."""
        self._training_flag.set()
        if (not self.queue_thread.is_alive() and self.mode == 'lazy'
                and self.queue_cap > 0):
            logger.info(f'Starting {self._thread_name} queue.')
            self.queue_thread.start()

    def stop(self):
        """Stop loading batches."""
        self._training_flag.clear()
        thread = self._queue_thread
        if thread is not None and thread.is_alive():
            logger.info(f'Stopping {self._thread_name} queue.')
            thread.join()

    @property
    def running(self):
        return self._training_flag.is_set()

    def sample_batches(self, n_batches):
        """``n_batches`` raw batches: a list of arrays, or of futures when the
        thread pool has more than one worker"""
        if self.max_workers > 1 and n_batches > 1:
            return [self._thread_pool.submit(self.sample_batch)
                    for _ in range(n_batches)]
        return [self.sample_batch() for _ in range(n_batches)]

    def enqueue_batches(self):
        """Body of the queue thread: top the FIFO up to ``queue_cap`` until
        ``stop()`` clears the flag."""
        last_log = time.time()
        while self.running:
            room = self.queue_cap - self.queue.qsize()
            if room <= 0:
                time.sleep(0.001)
            else:
                for item in self.sample_batches(room):
                    done = item.result() if hasattr(item, 'result') else item
                    if not self._put(done):
                        break
            if time.time() - last_log > 60:
                logger.debug(self.log_queue_info())
                last_log = time.time()

    def _put(self, batch):
        """blocking put that gives up when the queue is stopped"""
        while self.running:
            try:
                self.queue.put(batch, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def get_container_index(self):
        indices = np.arange(0, len(self.containers))
        return int(self._rng.choice(indices, p=self.container_weights))

    def get_random_container(self):
        self.container_index = self.get_container_index()
        return self.containers[self.container_index]

    def sample_batch(self):
        """a batch of samples from a randomly chosen sampler, in memory"""
        out = next(self.get_random_container())
        if not isinstance(out, tuple):
            return np.asarray(out)
        return tuple(np.asarray(o) for o in out)

    # ------------------------------------------------------------- iteration
    def __len__(self):
        return self.n_batches

    def __iter__(self):
        self._batch_count = 0
        self.start()
        return self

    def get_batch(self):
        """next raw batch (from the FIFO while its thread is alive, otherwise
        sampled on the spot), time axis squeezed for spatial-only samples,
        then ``post_proc``"""
        use_queue = (self.mode != 'eager' and self.queue_cap > 0
                     and self.queue_thread.is_alive())
        samples = None
        while use_queue and samples is None:
            try:
                samples = self.queue.get(timeout=0.05)
            except queue.Empty:
                use_queue = self.queue_thread.is_alive()
        if samples is None:
            samples = self.sample_batch()
        if self.sample_shape[2] == 1:
            squeeze = lambda a: a[..., 0, :]          # noqa: E731
            samples = (tuple(squeeze(a) for a in samples)
                       if isinstance(samples, (list, tuple))
                       else squeeze(samples))
        return self.post_proc(samples)

    def __next__(self):
        if self._batch_count < self.n_batches:
            batch = self.timer(self.get_batch, log=self.verbose)()
            self._batch_count += 1
        else:
            raise StopIteration
        return batch

    def log_queue_info(self):
        return '{} queue length: {} / {}'.format(
            self._thread_name.title(), self.queue_len, self.queue_cap)

    @property
    def lr_shape(self):
        return (*self.lr_sample_shape, len(self.lr_features))

    @property
    def hr_shape(self):
        return (*self.hr_sample_shape, len(self.hr_features))

    @property
    def shapes(self):
        """Shapes of batches returned by ``__next__``"""
        lr_shape, hr_shape = self.lr_shape, self.hr_shape
        if self.sample_shape[2] == 1:
            lr_shape = (*lr_shape[:2], lr_shape[-1])
            hr_shape = (*hr_shape[:2], hr_shape[-1])
        return (self.batch_size, *lr_shape), (self.batch_size, *hr_shape)


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
        feats = list(train_samplers[0].features)
        self.means = dict(means) if means is not None else {
            f: np.float32(0.0) for f in feats}
        self.stds = dict(stds) if stds is not None else {
            f: np.float32(1.0) for f in feats}
        if not val_samplers:
            self.val_data = []
        else:
            self.val_data = DeviceBatchQueue(
                samplers=val_samplers, n_batches=n_batches,
                thread_name='validation', batch_size=batch_size,
                s_enhance=s_enhance, t_enhance=t_enhance, queue_cap=queue_cap,
                transform_kwargs=transform_kwargs, max_workers=max_workers,
                mode=mode, transform=transform, seed=seed)
        super().__init__(samplers=train_samplers, n_batches=n_batches,
                         batch_size=batch_size, s_enhance=s_enhance,
                         t_enhance=t_enhance, queue_cap=queue_cap,
                         transform_kwargs=transform_kwargs,
                         max_workers=max_workers, mode=mode,
                         transform=transform, seed=seed)

    @property
    def smoothing(self):
        return self.transform_kwargs.get('smoothing', None)

    @property
    def smoothed_features(self):
        ignore = self.transform_kwargs.get('smoothing_ignore', None) or []
        if self.smoothing is None:
            return []
        return [f for f in self.lr_features if f not in ignore]

    def start(self):
        """Start the val data batch queue in addition to the train batch
        queue."""
        if hasattr(self.val_data, 'start'):
            self.val_data.start()
        super().start()

    def stop(self):
        """Stop the val data batch queue in addition to the train batch
        queue."""
        if hasattr(self.val_data, 'stop'):
            self.val_data.stop()
        super().stop()
