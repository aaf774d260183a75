"""Per-chunk forward-pass executor on MI355X.

Counterpart of the hot-path half of ``sup3r.pipeline.forward_pass.ForwardPass``
(``run_generator`` :188-272, ``_reshape_data_chunk`` :274-337,
``pad_source_data`` :122-186, ``_output_check`` :384-425, ``run_chunk``
:582-673) and of the slice algebra of ``sup3r.pipeline.slicer``
(padded / unpadded lo-res slices, hi-res windows, crop slices, domain-edge
reflect padding).  File IO, bias correction, exo rasterisation and the job
launcher stay in sup3r (boundary / out of scope); here a chunk is
an in-memory array.

MI355X design points:
  * the model is loaded ONCE per process / GPU and its shape-specialised plan
    is reused for every chunk (the reference re-loads the model per chunk,
    forward_pass.py:638);
  * chunks are the data-parallel unit: rank r of N takes chunks r, r+N, ...
    (or a contiguous block) — embarrassingly parallel, no collective;
  * every chunk is padded to the SAME padded shape (interior overlap +
    reflect at domain edges) so one plan serves the whole domain.
"""
import collections
import json
import logging
import os

import numpy as np

from .strategy import (ArrayStrategy, ChunkSlicer,  # noqa: F401
                       ForwardPassChunk, chunk_slices)

logger = logging.getLogger(__name__)

_MODELS = {}


def get_model(model_class, kwargs):
    """sup3r.pipeline.utilities.get_model (utilities.py:11-24): the class by
    name from this package, ``Model.load(**kwargs)``.  The model is loaded
    ONCE per process and GPU (the reference re-loads it for every chunk,
    forward_pass.py:638): the loaded object, its device weights and its
    shape-specialised plans are cached on (class, kwargs)."""
    import sup3r_amd
    if isinstance(kwargs, str):
        kwargs = {'model_dir': kwargs}
    if _model_key(model_class, kwargs) in _MODELS:
        return _MODELS[_model_key(model_class, kwargs)]
    cls = getattr(sup3r_amd, str(model_class), None)
    if cls is None or not hasattr(cls, 'load'):
        msg = ('Could not load requested model class "{}" from '
               'sup3r_amd, Make sure you typed in the model class '
               'name correctly.'.format(model_class))
        logger.error(msg)
        raise KeyError(msg)
    key = _model_key(model_class, kwargs)
    if key not in _MODELS:
        _MODELS[key] = cls.load(**kwargs, verbose=True)
    return _MODELS[key]


def _model_key(model_class, kwargs):
    if isinstance(kwargs, str):
        kwargs = {'model_dir': kwargs}
    return (str(model_class), json.dumps(kwargs, sort_keys=True, default=str))


def register_model(model_class, kwargs, model):
    """Make an already loaded / freshly trained model the one ``get_model``
    returns for (model_class, kwargs) in this process."""
    _MODELS[_model_key(model_class, kwargs)] = model
    return model


def get_source_type(file_paths):
    """'h5' | 'nc' | 'npz' | None from a path (or the first of a list), the
    role of sup3r.preprocessing.utilities.get_source_type"""
    if file_paths is None:
        return None
    if isinstance(file_paths, (list, tuple)):
        file_paths = file_paths[0]
    ext = os.path.splitext(str(file_paths))[1].lower()
    if ext == '.h5':
        return 'h5'
    if ext == '.npz':
        return 'npz'
    return 'nc'


class _EnhancementMismatch(RuntimeError):
    """stated s_enhance / t_enhance vs the generator's output shape"""


class NpzOutputHandler:
    """Minimal chunk writer with the ``_write_output`` call of sup3r's
    ``OutputHandlerH5`` / ``OutputHandlerNC`` (sup3r/writers/base.py): the
    H5 / NetCDF writers themselves are out of scope (SURVEY.md §2 #18, h5py
    is not in this image).  ``data`` arrives already transformed by
    ``DeviceOutputTransform`` (u/v inversion, limits) on the device."""

    @classmethod
    def _write_output(cls, data, features, lat_lon, times, out_file,
                      meta_data=None, invert_uv=False, nn_fill=False,
                      max_workers=None, gids=None):
        tmp = out_file + '.tmp.npz'
        np.savez(tmp, data=np.asarray(data), features=np.array(features),
                 lat_lon=np.zeros(0) if lat_lon is None else
                 np.asarray(lat_lon),
                 times=np.zeros(0) if times is None else
                 np.asarray(times).astype('int64'),
                 gids=np.zeros(0) if gids is None else np.asarray(gids),
                 meta=json.dumps(meta_data or {}, default=str))
        os.replace(tmp, out_file)


class ResidentDomain:
    """A lo-res domain uploaded once (NaN-checked, reflect-padded by the
    slicer's halo, normalised with the model's statistics at upload time):
    the explicit handle ``ForwardPass.upload_domain`` returns and
    ``run_batched`` accepts in place of the array.  Nothing is cached behind
    the caller's back — a new or modified array needs a new upload."""

    def __init__(self, tensor, shape, stats_key):
        self.tensor, self.shape, self.stats_key = tensor, tuple(shape), \
            stats_key


class ChunkPathOptions(collections.namedtuple(
        'ChunkPathOptions', ['sdma_delivery', 'window_forward',
                             'device_chunks_4d', 'device_norm_4d',
                             'device_chains', 'input_prefetch'],
        defaults=[True, True, True, True, True, 8])):
    """How ``iter_chunks`` / ``run_chunk`` / ``run`` take a chunk batch to the
    device — an argument of those calls (``options=``) and of the constructor,
    NOT process state: two executors in one process may differ, and a thread
    cannot flip another's path.

    * ``sdma_delivery``: deliver by the SDMA engines through ROCr
      (s3_dma_d2h_begin); False: ``hipMemcpyAsync`` on a copy stream.  (When
      ROCr refuses a copy the process falls back for good, with one warning:
      that is a fact about the runtime, kept in ``ForwardPass._sdma_refused``.)
    * ``window_forward``: halo crop + un-normalisation inside the tail conv
      (s3_plan_forward_window) where the plan supports it; False: full output
      + s3_chunk_epilogue.
    * ``device_chunks_4d``: spatial (4-D) models on the device chunk path;
      False: chunk by chunk through ``model.generate``.
    * ``device_norm_4d``: ... with the transpose to time-major + norm_input on
      the device; False: host numpy, same bits.
    * ``device_chains``: ``MultiStepGan`` chains on the device chunk path;
      False: chunk by chunk through ``MultiStepGan.generate``, every
      hand-over through host numpy.
    * ``input_prefetch``: chunks read ahead from the caller's iterator on a
      helper thread (``get_input_chunk``: slicing + edge padding in numpy,
      ~0.35 ms per 75 x 75 x 48 chunk) while this thread waits for the device
      or launches a batch; 0: everything on the calling thread."""
    __slots__ = ()


DEFAULT_OPTIONS = ChunkPathOptions()


def _opts(options):
    if options is None:
        return DEFAULT_OPTIONS
    if isinstance(options, ChunkPathOptions):
        return options
    return DEFAULT_OPTIONS._replace(**dict(options))


def _read_ahead(chunks, depth):
    """``chunks`` pulled on a helper thread, at most ``depth`` ahead of the
    consumer; exceptions of the source surface at the consumer in order; the
    helper stops when the consumer goes away"""
    import queue
    import threading
    q, stop, end = queue.Queue(maxsize=max(1, int(depth))), threading.Event(), object()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for c in chunks:
                if not put((c, None)):
                    return
        except BaseException as e:      # noqa: BLE001 — re-raised by the consumer
            put((None, e))
            return
        put(end)
    th = threading.Thread(target=work, name='sup3r-amd-chunk-read-ahead',
                          daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is end:
                return
            c, err = item
            if err is not None:
                raise err
            yield c
    finally:
        stop.set()


class ForwardPass:
    """Per-chunk executor of the generator.

    Two faces.  The reference's (``sup3r.pipeline.forward_pass.ForwardPass``):
    ``ForwardPass(strategy, node_index)``, ``ForwardPass.run(strategy,
    node_index)``, ``fwp.get_input_chunk(chunk_index)`` and the classmethod
    ``run_chunk(chunk, model_kwargs, model_class, allowed_const, invert_uv,
    meta, nn_fill, output_workers)`` over ``ForwardPassChunk`` structures
    (exo data included) — ``strategy`` is duck-typed (``init_chunk``,
    ``node_chunks``, ``chunk_finished``, ``model_kwargs``, ...), so sup3r's
    ``ForwardPassStrategy`` or ``strategy.ArrayStrategy`` both drive it.  And
    the in-memory one: ``ForwardPass(model, slicer)`` with ``run_domain`` /
    ``run_batched`` over a lo-res array resident on the device."""

    OUTPUT_HANDLER_CLASS = {'npz': NpzOutputHandler}

    def __init__(self, model, slicer=0, rank=0, nranks=1, shard='interleave',
                 output_check=True, allowed_const=False, node_index=None,
                 options=None):
        """``ForwardPass(strategy, node_index=0)`` (forward_pass.py:44-64) or
        ``ForwardPass(model, slicer, rank, nranks, ...)``.

        ``allowed_const`` (ForwardPassStrategy.allowed_const,
        strategy.py:186-196): False = a constant output channel fails the
        chunk, True = any constant is fine, a value / list = only these
        constants are (0 for night-time clearsky ratio).  ``output_check=False``
        switches the whole check off."""
        self.strategy = None
        self.options = _opts(options)
        if hasattr(model, 'init_chunk'):
            strategy = model
            self.strategy = strategy
            self.node_index = int(slicer if node_index is None
                                  else node_index)
            model = get_model(strategy.model_class, strategy.model_kwargs)
            slicer = strategy.fwp_slicer
            allowed_const = getattr(strategy, 'allowed_const', allowed_const)
            rank, nranks = self.node_index, len(strategy.node_chunks)
            shard = 'block'
            output_type = get_source_type(
                getattr(strategy, 'out_pattern', None))
            assert output_type is None or \
                output_type in self.OUTPUT_HANDLER_CLASS or \
                output_type in ('h5', 'nc'), \
                f'Received bad output type {output_type}'
        self.model = model
        self.slicer = slicer
        self.rank, self.nranks = rank, nranks
        self.shard = shard
        self.output_check = output_check
        self.allowed_const = allowed_const
        if model.s_enhance != slicer.s_enhance or \
                model.t_enhance != slicer.t_enhance:
            raise RuntimeError(
                'model enhancement ({}, {}) does not match the slicer ({}, {})'
                .format(model.s_enhance, model.t_enhance, slicer.s_enhance,
                        slicer.t_enhance))

    @property
    def meta(self):
        """forward_pass.py:74-86: what goes into the output files' attrs"""
        import datetime
        return {'node_index': getattr(self, 'node_index', self.rank),
                'creation_date': datetime.datetime.now().strftime(
                    '%d/%m/%Y %H:%M:%S'),
                'model_meta': self.model.meta,
                'gan_params': self.model.model_params,
                'strategy_meta': getattr(self.strategy, 'meta', {})}

    def _get_step_enhance(self, step):
        """forward_pass.py:88-120: enhancement of an exo step's field
        relative to the lo-res input."""
        combine_type, model_step = step['combine_type'], step['model']
        assert combine_type.lower() in ('input', 'output', 'layer'), \
            f'Received weird combine_type {combine_type} for step: {step}'
        s_all = list(getattr(self.model, 's_enhancements',
                             [self.model.s_enhance]))
        t_all = list(getattr(self.model, 't_enhancements',
                             [self.model.t_enhance]))
        n = model_step if combine_type.lower() == 'input' else model_step + 1
        return (int(np.prod(s_all[:n], dtype=np.int64)),
                int(np.prod(t_all[:n], dtype=np.int64)))

    def get_input_chunk(self, chunk_index=0, mode='reflect'):
        """forward_pass.py:66-72: ``strategy.init_chunk`` + edge padding of
        the lo-res window and of its exo fields."""
        chunk = self.strategy.init_chunk(chunk_index)
        enh = None
        if chunk.exo_data is not None:
            enh = {f: [self._get_step_enhance(st)
                       for st in chunk.exo_data[f]['steps']]
                   for f in chunk.exo_data}
        chunk.input_data, chunk.exo_data = self.pad_source_data(
            chunk.input_data, chunk.pad_width, chunk.exo_data, mode=mode,
            enhancements=enh)
        return chunk

    # -- reference-compatible helpers ------------------------------------
    @staticmethod
    def pad_source_data(input_data, pad_width, exo_data=None, mode='reflect',
                        enhancements=None):
        """forward_pass.py:122-186: reflect-pad the chunk (and its exo data by
        the enhanced widths)."""
        out = np.pad(input_data, (*pad_width, (0, 0)), mode=mode)
        if exo_data is not None:
            for feature in exo_data:
                for i, step in enumerate(exo_data[feature]['steps']):
                    s_en, t_en = enhancements[feature][i] if enhancements \
                        else (step.get('s_enhance', 1),
                              step.get('t_enhance', 1))
                    ew = tuple((en * pw[0], en * pw[1]) for en, pw in
                               zip((s_en, s_en, t_en), pad_width)) + ((0, 0),)
                    new = step['data']
                    if new.ndim == 3 and mode in ('reflect', 'symmetric',
                                                  'edge', 'wrap'):
                        # a field without a time axis (topography): the
                        # reference repeats it along time (np.repeat) and pads
                        # the result; padding a time-constant field along
                        # time with these modes leaves it time-constant, so
                        # the SAME values come out of a spatial pad and a
                        # zero-stride (read-only) view along time — and the
                        # device executor uploads the field once per chunk
                        # instead of once per time step (108 MB per 75 x 75 x
                        # 48 chunk of a 10x model)
                        n_t = t_en * input_data.shape[2] + ew[2][0] + ew[2][1]
                        flat = np.pad(new, (ew[0], ew[1], (0, 0)), mode=mode)
                        exo_data[feature]['steps'][i]['data'] = \
                            np.broadcast_to(
                                flat[:, :, None, :],
                                flat.shape[:2] + (n_t, flat.shape[2]))
                        continue
                    if new.ndim == 3:
                        new = np.repeat(np.expand_dims(new, 2),
                                        t_en * input_data.shape[2], axis=2)
                    exo_data[feature]['steps'][i]['data'] = np.pad(
                        new, ew, mode=mode)
        return out, exo_data

    @staticmethod
    def _reshape_data_chunk(model, data_chunk, exo_data):
        """forward_pass.py:274-337."""
        if exo_data is not None:
            # every exo entry takes the layout of the model STEP that consumes
            # it (a multi-step model may run a spatial step, time on the batch
            # axis, in front of a spatio-temporal one)
            steps_of = getattr(model, 'models', [model])
            for feature in exo_data:
                for i, entry in enumerate(exo_data[feature]['steps']):
                    k = entry.get('model', 0)
                    assert k < len(steps_of), (
                        f'model index ({k}) for exo step {i} of "{feature}" '
                        'exceeds the number of model steps')
                    if steps_of[k].is_4d:
                        out = np.transpose(entry['data'], axes=(2, 0, 1, 3))
                    else:
                        out = np.expand_dims(entry['data'], axis=0)
                    exo_data[feature]['steps'][i]['data'] = np.asarray(out)
        if model.is_4d:
            return (np.asarray(np.transpose(data_chunk, axes=(2, 0, 1, 3))),
                    exo_data, 0, 1)
        return np.asarray(np.expand_dims(data_chunk, axis=0)), exo_data, 3, 1

    @classmethod
    def run_generator(cls, data_chunk, hr_crop_slices, model, s_enhance=None,
                      t_enhance=None, exo_data=None):
        """forward_pass.py:188-272."""
        data_chunk, exo_data, i_lr_t, i_lr_s = cls._reshape_data_chunk(
            model, data_chunk, exo_data)
        try:
            hi_res = model.generate(data_chunk, exogenous_data=exo_data)
        except Exception as e:
            msg = 'Forward pass failed on chunk with shape {}.'.format(
                data_chunk.shape)
            logger.exception(msg)
            raise RuntimeError(msg) from e
        if hi_res.ndim == 4:
            hi_res = np.expand_dims(np.transpose(hi_res, (1, 2, 0, 3)), axis=0)
        if s_enhance is not None and \
                hi_res.shape[1] != s_enhance * data_chunk.shape[i_lr_s]:
            msg = ('The stated spatial enhancement of {}x did not match the '
                   'low res / high res shapes of {} -> {}'.format(
                       s_enhance, data_chunk.shape, hi_res.shape))
            logger.error(msg)
            raise RuntimeError(msg)
        if t_enhance is not None and \
                hi_res.shape[3] != t_enhance * data_chunk.shape[i_lr_t]:
            msg = ('The stated temporal enhancement of {}x did not match the '
                   'low res / high res shapes of {} -> {}'.format(
                       t_enhance, data_chunk.shape, hi_res.shape))
            logger.error(msg)
            raise RuntimeError(msg)
        return hi_res[0][tuple(hr_crop_slices)]

    @staticmethod
    def _const_ok(allowed_const):
        """-> (skip the check entirely, tuple of permitted constants)"""
        if allowed_const is True:
            return True, ()
        if allowed_const is False or allowed_const is None:
            return False, ()
        if isinstance(allowed_const, (list, tuple)):
            return False, tuple(allowed_const)
        return False, (allowed_const,)

    @classmethod
    def _output_check(cls, out_data, allowed_const=False, features=None,
                      chunk_index=None):
        """forward_pass.py:384-425: NaNs, or an output channel that is one
        constant not listed in ``allowed_const``, mean the chunk failed."""
        skip, allowed = cls._const_ok(allowed_const)
        if skip:
            return False
        if np.isnan(out_data).any():
            logger.error('chunk %s: NaN in the generated output', chunk_index)
            return True
        for idf in range(out_data.shape[-1]):
            ch = out_data[..., idf]
            v0 = ch.flat[0]
            if ch.size > 1 and np.all(ch == v0) and v0 not in allowed:
                logger.error('chunk %s: output channel "%s" is constant (%s)',
                             chunk_index, features[idf] if features else idf,
                             v0)
                return True
        return False

    # -- execution ---------------------------------------------------------
    def chunk_input(self, domain, chunk_index):
        """Padded lo-res input of one chunk from the (s1, s2, t, f) domain."""
        c = self.slicer.chunks[chunk_index]
        data = domain[c['lr_pad_slice']]
        if any(p != (0, 0) for p in c['pad_width']):
            data, _ = self.pad_source_data(data, c['pad_width'])
        return data

    def run_domain_chunk(self, domain, chunk_index):
        """One chunk of an in-memory domain: pad -> NaN check -> generate ->
        enhancement check -> crop (-> output check)."""
        c = self.slicer.chunks[chunk_index]
        data = self.chunk_input(domain, chunk_index)
        if np.isnan(data).any():
            raise ValueError(
                f'Forward pass chunk {chunk_index} input data has NaN values')
        out = self.run_generator(data, c['hr_crop'], self.model,
                                 s_enhance=self.slicer.s_enhance,
                                 t_enhance=self.slicer.t_enhance)
        if self.output_check and self._output_check(
                out, self.allowed_const, self.model.hr_out_features,
                chunk_index):
            raise MemoryError(
                f'Forward pass output check failed on chunk {chunk_index}')
        return out

    def my_chunks(self):
        return self.slicer.rank_chunks(self.rank, self.nranks, self.shard)

    def _stats_key(self):
        m = self.model
        if getattr(m, 'means', None) is None:
            return None
        mu, sd = m._stats_for(m.lr_features)
        return (tuple(m.lr_features), mu.tobytes(), sd.tobytes())

    def upload_domain(self, domain):
        """NaN-check, reflect-pad by the slicer's halo, normalise and upload
        a lo-res ``(s1, s2, t, features)`` domain once; every chunk's padded
        input is then a plain window of the resident tensor (a chunk's edge
        padding mirrors the same cells the domain padding mirrors)."""
        sl, model = self.slicer, self.model
        ps, pt = sl.spatial_pad, sl.temporal_pad
        domain = np.asarray(domain)
        if np.isnan(domain).any():
            for idx in self.my_chunks():
                if np.isnan(domain[sl.chunks[idx]['lr_pad_slice']]).any():
                    raise ValueError(f'Forward pass chunk {idx} input '
                                     'data has NaN values')
        padded = np.pad(np.asarray(domain, dtype=np.float32),
                        ((ps, ps), (ps, ps), (pt, pt), (0, 0)),
                        mode='reflect')
        if model.means is not None:
            padded = np.asarray(model.norm_input(padded), dtype=np.float32)
        return ResidentDomain(model._gen.dev.to_device(padded), domain.shape,
                              self._stats_key())

    # -- MI355X-native executor ---------------------------------------------
    def run_batched(self, domain, out=None, writer=None, batch=8,
                    n_host_threads=16, direct_placement=False,
                    max_chunks=None):
        """Same result as :meth:`run`, organised for the device instead of
        chunk by chunk through host numpy (the reference's ``run_chunk`` loop,
        forward_pass.py:582-673, re-loads the model and round-trips every
        chunk through the host):

        * the lo-res domain is NaN-checked, reflect-padded, normalised and
          uploaded ONCE — every chunk's padded input is then a plain slice of
          the resident tensor (a chunk's edge padding mirrors the same cells
          the domain padding mirrors);
        * chunks of equal shape are stacked ``batch`` at a time into one
          generator launch sequence;
        * un-normalisation, the output check (NaN / constant channel,
          ``s3_chunk_stats``) and the halo crop happen on the device; only
          the cropped hi-res window crosses PCIe, into pinned double buffers
          on a copy stream, while the next batch computes;
        * placement into ``out`` runs on a host thread pool from the pinned
          buffers (numpy releases the GIL in the strided copies).  With
          ``direct_placement=True`` and a C-contiguous float32 ``out`` the
          array is registered with the HIP runtime instead and every cropped
          chunk is DMA'd straight into its window (``s3_d2h_window``, one
          pitched copy per chunk, no staging, no host memcpy) — measured
          slower on this platform (4.6 KB rows: 9 GB/s), kept as an option.

        ``max_chunks`` bounds the run to the first chunks of this rank's list
        (benchmarks).  ``domain`` is the lo-res array — checked, padded,
        normalised and uploaded by this call — or the ``ResidentDomain`` of an
        earlier :meth:`upload_domain` (repeated runs over one domain);
        residency is never implicit.

        Supports single-step 5-D models without exogenous inputs; anything
        else falls back to :meth:`run`."""
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor

        import torch

        from . import _lib
        model, sl = self.model, self.slicer
        gen = getattr(model, '_gen', None)
        # (models that override generate — SolarCC's temporal padding — or
        # that a user registered do not declare a device chunk path)
        simple = (gen is not None and getattr(model, 'is_5d', False)
                  and getattr(model, 'supports_device_chunks', False)
                  and not getattr(model, 'hr_exo_features', [])
                  and domain.shape[-1] == len(model.lr_features))
        if not simple:
            return self.run_chunks(domain, out=out, writer=writer)
        dev, L = gen.dev, _lib.lib()
        ids = self.my_chunks()
        if max_chunks is not None:
            ids = ids[:int(max_chunks)]
        if not ids:
            return 0
        ps, pt = sl.spatial_pad, sl.temporal_pad
        if isinstance(domain, ResidentDomain):
            if domain.stats_key != self._stats_key():
                raise RuntimeError(
                    'the resident domain was normalised with other statistics '
                    'than the model holds now; upload it again')
            dom_d = domain.tensor
        else:
            dom_d = self.upload_domain(domain).tensor
        n_in = int(dom_d.shape[-1])
        n_out = len(model.hr_out_features)
        if model.means is not None:
            mu, sd = model._stats_for(model.hr_out_features)
            scale = np.ascontiguousarray(sd, dtype=np.float32)
            shift = np.ascontiguousarray(mu, dtype=np.float32)
        pf = C.POINTER(C.c_float)

        # chunks grouped by padded input shape (edge chunks are ragged)
        groups = {}
        for idx in ids:
            c = sl.chunks[idx]
            shp = tuple(s_.stop - s_.start + lo + hi for s_, (lo, hi) in
                        zip(c['lr_pad_slice'], c['pad_width']))
            groups.setdefault(shp, []).append(idx)
        # direct placement: pitched DMA into the registered output array
        direct = (direct_placement and writer is None
                  and isinstance(out, np.ndarray)
                  and out.dtype == np.float32 and out.flags['C_CONTIGUOUS']
                  and out.ndim == 4)
        if direct:
            rc = L.s3_host_register(dev.ctx, C.c_void_p(out.ctypes.data),
                                    C.c_size_t(out.nbytes))
            direct = rc == 0
        copy_stream = torch.cuda.Stream(device=dev.torch_device)
        pool = ThreadPoolExecutor(max_workers=max(1, n_host_threads))
        pending = []          # (event, pinned buffer, chunk ids, stats)
        pinned = {}           # shape -> ring of pinned staging buffers
        toggle = {}
        busy = {}             # id(pinned buffer) -> placement futures in flight
        n_ring = 3

        def place(buf, k, idx):
            data = buf[k].numpy()
            hr_sl = sl.chunks[idx]['hr_slice']
            if writer is not None:
                writer(idx, hr_sl, np.array(data, copy=True))
            elif out is not None:
                out[hr_sl] = data

        def drain(keep):
            while len(pending) > keep:
                ev, buf, cids, stats = pending.pop(0)
                ev.synchronize()
                if direct:
                    buf = None        # (device block kept alive until here)
                st = stats.numpy().reshape(len(cids), 64, n_out, 3)
                mn, mx = st[..., 0].min(1), st[..., 1].max(1)
                nn = st[..., 2].sum(1)
                skip, allowed = self._const_ok(self.allowed_const)
                for k, idx in enumerate(cids):
                    const = [v for v, w in zip(mn[k], mx[k]) if v == w]
                    bad = not skip and (
                        nn[k].any() or any(v not in allowed for v in const))
                    if self.output_check and bad:
                        raise MemoryError('Forward pass output check failed '
                                          f'on chunk {idx}')
                if not direct:
                    # placement overlaps the next batches; the buffer is
                    # handed out again only once these are done
                    busy[id(buf)] = [pool.submit(place, buf, k, idx)
                                     for k, idx in enumerate(cids)]

        done = 0
        try:
            for shp, gids in groups.items():
                for b0 in range(0, len(gids), batch):
                    cids = gids[b0:b0 + batch]
                    # the chunks' padded windows, cut out of the resident
                    # domain into one (chunks, s1, s2, t, f) batch
                    x = dev.empty((len(cids),) + tuple(shp) + (n_in,))
                    dd1, dd2 = int(dom_d.shape[1]), int(dom_d.shape[2])
                    for k, idx in enumerate(cids):
                        c = sl.chunks[idx]
                        o = [s_.start - lo + p for s_, (lo, hi), p in zip(
                            c['lr_pad_slice'], c['pad_width'], (ps, ps, pt))]
                        src = dom_d.data_ptr() + 4 * n_in * (
                            (o[0] * dd1 + o[1]) * dd2 + o[2])
                        rc = L.s3_copy_block(
                            dev.ctx, C.c_void_p(src),
                            C.c_void_p(x[k].data_ptr()), shp[0], shp[1],
                            shp[2] * n_in, dd1 * dd2 * n_in, dd2 * n_in,
                            shp[1] * shp[2] * n_in, shp[2] * n_in)
                        _lib.check(rc, dev.ctx, 's3_copy_block')
                    ph = gen.plan(tuple(x.shape), training=False)
                    y = ph.forward(x)
                    if self.slicer.s_enhance * shp[0] != y.shape[1] or \
                            self.slicer.t_enhance * shp[2] != y.shape[3]:
                        raise RuntimeError(
                            'The stated enhancement of {}x / {}x did not match '
                            'the low res / high res shapes of {} -> {}'.format(
                                sl.s_enhance, sl.t_enhance, tuple(x.shape),
                                tuple(y.shape)))
                    if model.means is not None:
                        rc = L.s3_affine_channels(
                            dev.ctx, C.c_void_p(y.data_ptr()),
                            C.c_void_p(y.data_ptr()), n_out,
                            y.numel() // n_out, scale.ctypes.data_as(pf),
                            shift.ctypes.data_as(pf))
                        _lib.check(rc, dev.ctx, 's3_affine_channels')
                    # halo crop (the same for every chunk of a shape group)
                    crop = sl.chunks[cids[0]]['hr_crop']
                    y1, y2, y3 = (int(v) for v in y.shape[1:4])
                    cr = [(s_.start or 0, y_ if s_.stop is None else s_.stop)
                          for s_, y_ in zip(crop, (y1, y2, y3))]
                    cr = [(a, b if b >= 0 else y_ + b)
                          for (a, b), y_ in zip(cr, (y1, y2, y3))]
                    c1, c2, c3 = (b - a for a, b in cr)
                    yc = dev.empty((len(cids), c1, c2, c3, n_out))
                    for k in range(len(cids)):
                        src = y[k].data_ptr() + 4 * n_out * (
                            (cr[0][0] * y2 + cr[1][0]) * y3 + cr[2][0])
                        rc = L.s3_copy_block(
                            dev.ctx, C.c_void_p(src),
                            C.c_void_p(yc[k].data_ptr()), c1, c2, c3 * n_out,
                            y2 * y3 * n_out, y3 * n_out, c2 * c3 * n_out,
                            c3 * n_out)
                        _lib.check(rc, dev.ctx, 's3_copy_block')
                    stats_d = dev.empty((len(cids), 64, n_out, 3))
                    rc = L.s3_chunk_stats(
                        dev.ctx, C.c_void_p(yc.data_ptr()), len(cids),
                        yc.numel() // (len(cids) * n_out), n_out,
                        C.c_void_p(stats_d.data_ptr()))
                    _lib.check(rc, dev.ctx, 's3_chunk_stats')
                    key = tuple(yc.shape)
                    if direct:
                        drain(2)
                        stats_h = torch.empty(tuple(stats_d.shape),
                                              dtype=torch.float32,
                                              pin_memory=True)
                        ready = torch.cuda.Event()
                        ready.record()
                        with torch.cuda.stream(copy_stream):
                            copy_stream.wait_event(ready)
                            cs1, cs2 = int(yc.shape[1]), int(yc.shape[2])
                            row = int(yc.shape[3]) * int(yc.shape[4])
                            st0 = out.strides[0] // 4
                            st1 = out.strides[1] // 4
                            for k, idx in enumerate(cids):
                                hs = sl.chunks[idx]['hr_slice']
                                dst = out.ctypes.data + 4 * (
                                    hs[0].start * st0 + hs[1].start * st1
                                    + hs[2].start * n_out)
                                rc = L.s3_d2h_window(
                                    dev.ctx, C.c_void_p(yc[k].data_ptr()),
                                    C.c_void_p(dst), cs1, cs2, row, st0, st1,
                                    C.c_void_p(copy_stream.cuda_stream))
                                _lib.check(rc, dev.ctx, 's3_d2h_window')
                            stats_h.copy_(stats_d, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                        yc.record_stream(copy_stream)
                        stats_d.record_stream(copy_stream)
                        pending.append((ev, yc, cids, stats_h))
                        done += len(cids)
                        continue
                    if key not in pinned:
                        pinned[key] = [torch.empty(key, dtype=torch.float32,
                                                   pin_memory=True)
                                       for _ in range(n_ring)]
                        toggle[key] = 0
                    # at most n_ring - 1 batches between D2H and placement
                    drain(n_ring - 2)
                    buf = pinned[key][toggle[key]]
                    toggle[key] = (toggle[key] + 1) % n_ring
                    for f in busy.pop(id(buf), []):
                        f.result()        # (re-raises a writer's exception)
                    stats_h = torch.empty(tuple(stats_d.shape),
                                          dtype=torch.float32, pin_memory=True)
                    ready = torch.cuda.Event()
                    ready.record()
                    with torch.cuda.stream(copy_stream):
                        copy_stream.wait_event(ready)
                        buf.copy_(yc, non_blocking=True)
                        stats_h.copy_(stats_d, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    yc.record_stream(copy_stream)
                    stats_d.record_stream(copy_stream)
                    pending.append((ev, buf, cids, stats_h))
                    done += len(cids)
            drain(0)
            for futs in busy.values():
                for f in futs:
                    f.result()
        finally:
            pool.shutdown(wait=True)
            if direct:
                torch.cuda.synchronize()
                L.s3_host_unregister(dev.ctx, C.c_void_p(out.ctypes.data))
        return done

    # -- the reference's entry points over ForwardPassChunk structures -----
    @classmethod
    def _device_path(cls, model, chunk, options=None):
        """single-step generator on this package's engine, no exo field
        combined at the output: the chunk batches go through one plan on the
        device — 5-D models, and (round 5) 4-D (spatial) models, whose batch
        axis is the chunks' time axis (forward_pass.py:274-337); anything
        else (``MultiStepGan``, 'output' exo) takes ``run_generator`` ->
        ``model.generate`` chunk by chunk"""
        options = _opts(options)
        steps = getattr(model, 'models', None)
        if steps is not None:
            return cls._device_chain(model, chunk, options)
        if getattr(model, '_gen', None) is None or \
                not getattr(model, 'supports_device_chunks', False):
            return False
        if not getattr(model, 'is_5d', False):
            if not (getattr(model, 'is_4d', False) and
                    options.device_chunks_4d):
                return False
            # s3_chunk_time_first / s3_chunk_time_last carry at most 16
            # channels: a spatial model with more features goes chunk by
            # chunk through model.generate
            if len(getattr(model, 'lr_features', None) or ()) > 16 or \
                    len(getattr(model, 'hr_out_features', None) or ()) > 16:
                return False
        for entry in (chunk.exo_data or {}).values():
            if any(st['combine_type'].lower() == 'output'
                   for st in entry['steps']):
                return False
        return True

    @classmethod
    def _device_chain(cls, model, chunk, options=None):
        """a ``MultiStepGan`` whose steps all run on this package's engine
        with the base class's normalisation, spatial (4-D) steps first, then
        spatio-temporal (5-D) ones — the reference's production arrangements
        (examples/sup3rwind/run_configs/wind/config_fwp_spatial.json: two
        spatial steps; config_fwp_temporal.json: one 5-D step) — fp32
        statistics and 'input' exo fields, at most 16 channels at a
        hand-over, no 'output' exo: the hand-overs stay on the device
        (s3_step_handover); anything else through ``MultiStepGan.generate``"""
        from .gan import Sup3rGan as _BaseGan
        steps = list(model.models)
        options = _opts(options)
        if not options.device_chains or not steps:
            return False

        def base(m, name):
            f = getattr(type(m), name, None)
            return getattr(f, '__func__', f) is getattr(_BaseGan, name)
        seen_5d = False
        for i, m in enumerate(steps):
            if getattr(m, '_gen', None) is None or \
                    not getattr(m, 'supports_device_chunks', False) or \
                    hasattr(m, 'models'):
                return False
            # (hand_over re-implements the base class's input combination on
            # the device: a step that overrides any of it goes through
            # MultiStepGan.generate)
            if not all(base(m, name) for name in (
                    'norm_input', 'un_norm_output', 'generate',
                    '_combine_fwp_input')
                    if hasattr(_BaseGan, name)):
                return False
            if getattr(m, 'is_5d', False):
                seen_5d = True
            elif seen_5d or not (getattr(m, 'is_4d', False) and
                                 options.device_chunks_4d):
                return False
            if m._gen.dev is not steps[0]._gen.dev:
                return False
            if len(steps) > 1:
                if len(m.lr_features) > 16 or len(m.hr_out_features) > 16:
                    return False
                if m._means is not None:
                    for feats in (m.lr_features, m.hr_out_features):
                        mu, sd = m._stats_for(feats)
                        if mu.dtype != np.float32 or sd.dtype != np.float32:
                            return False
                elif i > 0 and steps[i - 1]._means is not None:
                    pass
        for entry in (chunk.exo_data or {}).values():
            for st in entry['steps']:
                if st['combine_type'].lower() == 'output':
                    return False
                if st.get('model', 0) > 0 and \
                        st['combine_type'].lower() == 'input' and \
                        np.asarray(st['data']).dtype != np.float32:
                    return False
        return True

    @classmethod
    def _crop_bounds(cls, crop, shape):
        """hr_crop_slice (None / negative stops, slicer.py:216-293) as
        (start, stop) per axis of a tensor of ``shape``"""
        out = []
        for s_, n in zip(crop, shape):
            a, b, _ = s_.indices(n)
            out.append((a, b))
        return out

    @classmethod
    def iter_chunks(cls, chunks, model, allowed_const=False, batch=8,
                    invert_uv=False, nn_fill=True, meta=None,
                    output_workers=None, return_data=True, write=True,
                    options=None):
        """Run ``ForwardPassChunk`` structures (already edge-padded:
        ``get_input_chunk``) through the generator, ``batch`` equal-shaped
        chunks per launch sequence, and yield ``(chunk, failed, output_data)``
        in input order.

        Per batch: the chunks' lo-res windows (+ 'input' exo channels) are
        normalised and stacked on the host (75 KB each), their 'layer' exo
        fields (topography ...) are normalised, stacked and uploaded next to
        them, one plan forward runs the batch; un-normalisation, the halo crop
        (``chunk.hr_crop_slice``), the output check (NaN / constant channel,
        forward_pass.py:384-425) run on the device; the cropped hi-res window
        crosses PCIe through pinned buffers on a copy stream while the next
        batch computes.  With ``write`` and a chunk's ``out_file`` set, the
        output epilogue (u/v inversion, limits: ``DeviceOutputTransform``)
        runs on the device and the registered ``OUTPUT_HANDLER_CLASS`` entry
        writes the result.  ``return_data=False`` skips the raw download when
        only the files are wanted (``run``).

        ``output_data`` of a device batch is a VIEW into a ring of pinned
        delivery buffers (``d2h_ring`` = 4 per output shape and per running
        generator, filled by the SDMA engines): it stays valid while the next
        ``d2h_ring - 2`` = two batches are yielded — consume it (write it,
        place it) or copy it before asking for more than that, and do not keep
        it past ``release_delivery_buffers``.  ``run`` / ``run_chunk`` hand
        out copies.  ``options``: a ``ChunkPathOptions`` (or a dict of its
        fields to change)."""
        options = _opts(options)
        pending = collections.deque()
        # the delivery rings belong to THIS generator: two interleaved
        # ``iter_chunks`` runs (threads, two models with one output shape)
        # get different lanes, consecutive runs reuse lane 0's pinned buffers
        lane = cls._take_lane()
        if options.input_prefetch and not isinstance(chunks, (list, tuple)):
            chunks = _read_ahead(chunks, options.input_prefetch)
        try:
            yield from cls._iter_chunks_lane(
                chunks, model, allowed_const, batch, invert_uv, nn_fill, meta,
                output_workers, return_data, write, pending, lane, options)
        finally:
            cls._lanes.discard(lane)

    @classmethod
    def _iter_chunks_lane(cls, chunks, model, allowed_const, batch, invert_uv,
                          nn_fill, meta, output_workers, return_data, write,
                          pending, lane, options):
        def flush(keep):
            # the newest batch is on the device's queue: the one before it may
            # start crossing PCIe (its forward is the one running or done)
            if len(pending) > 1:
                pending[-2].deliver()
            while len(pending) > keep:
                yield from pending.popleft()()

        group, shape = [], None
        for chunk in chunks:
            if not cls._device_path(model, chunk, options):
                yield from flush(0)
                if group:
                    pending.append(cls._launch_chunk_batch(
                        group, model, allowed_const, invert_uv, nn_fill, meta,
                        output_workers, return_data, write, lane, options))
                    group, shape = [], None
                    yield from flush(0)
                yield cls._run_chunk_host(chunk, model, allowed_const,
                                          invert_uv, nn_fill, meta,
                                          output_workers, write)
                continue
            key = (tuple(chunk.input_data.shape),
                   tuple((f, tuple(tuple(st['data'].shape)
                                   for st in e['steps']))
                         for f, e in sorted((chunk.exo_data or {}).items())),
                   tuple((s_.start, s_.stop) for s_ in chunk.hr_crop_slice))
            if group and (key != shape or len(group) >= batch):
                pending.append(cls._launch_chunk_batch(
                    group, model, allowed_const, invert_uv, nn_fill, meta,
                    output_workers, return_data, write, lane, options))
                group = []
                yield from flush(2)
            group.append(chunk)
            shape = key
        if group:
            pending.append(cls._launch_chunk_batch(
                group, model, allowed_const, invert_uv, nn_fill, meta,
                output_workers, return_data, write, lane, options))
        yield from flush(0)

    @classmethod
    def _run_chunk_host(cls, chunk, model, allowed_const, invert_uv, nn_fill,
                        meta, output_workers, write):
        """forward_pass.py:640-672 through ``model.generate``"""
        output_data = cls.run_generator(
            data_chunk=chunk.input_data, hr_crop_slices=chunk.hr_crop_slice,
            s_enhance=model.s_enhance, t_enhance=model.t_enhance,
            exo_data=chunk.exo_data, model=model)
        failed = cls._output_check(output_data, allowed_const=allowed_const)
        if write and chunk.out_file is not None and not failed:
            cls._write_chunk(chunk, model, output_data, invert_uv, nn_fill,
                             meta, output_workers)
        return chunk, failed, output_data

    @classmethod
    def _write_chunk(cls, chunk, model, data, invert_uv, nn_fill, meta,
                     output_workers):
        """the output epilogue on the device (writers/base.py:304-345), then
        the registered writer for the file type"""
        from .output_transform import DeviceOutputTransform
        logger.info(f'Saving forward pass output to {chunk.out_file}.')
        features = [f.lower() for f in model.hr_out_features]
        tr = DeviceOutputTransform(model._gen.dev if getattr(
            model, '_gen', None) is not None else None)
        x, features = tr.transform_output(
            data, features, chunk.hr_lat_lon, invert_uv=invert_uv,
            nn_fill=nn_fill)
        output_type = get_source_type(chunk.out_file)
        handler = cls.OUTPUT_HANDLER_CLASS.get(output_type)
        if handler is None:
            raise KeyError(
                f'no output handler registered for "{output_type}" files '
                f'({chunk.out_file}): the H5 / NetCDF writers live in sup3r '
                '(sup3r.writers.OutputHandlerH5 / OutputHandlerNC); register '
                'one in ForwardPass.OUTPUT_HANDLER_CLASS')
        # (already transformed: the handler must not invert / fill again)
        handler._write_output(
            data=x.cpu().numpy(), features=features,
            lat_lon=chunk.hr_lat_lon, times=chunk.hr_times,
            out_file=chunk.out_file, meta_data=meta, invert_uv=False,
            nn_fill=False, max_workers=output_workers, gids=chunk.gids)

    @classmethod
    def _launch_chunk_batch(cls, group, model, allowed_const, invert_uv,
                            nn_fill, meta, output_workers, return_data,
                            write, lane=0, options=None):
        """Enqueue one batch of equal-shaped chunks; returns the closure that
        waits for it and yields its ``(chunk, failed, output_data)``."""
        import ctypes as C

        import torch

        from . import _lib
        from .utilities import ExoData
        # a MultiStepGan chain runs step by step on the device; a single
        # model is a chain of one
        options = _opts(options)
        chain = hasattr(model, 'models')
        steps = list(model.models) if chain else [model]
        first, last = steps[0], steps[-1]
        gen = first._gen
        dev, L = gen.dev, _lib.lib()
        n = len(group)
        is_4d = not getattr(first, 'is_5d', False)
        xs, exos = [], []

        def step_exo(exo, i):
            """the exo entries model step ``i`` consumes (exo.py:108-130)"""
            if exo is None or not chain:
                return exo
            return exo.get_model_step_exo(i)
        for chunk in group:
            exo = chunk.exo_data
            if exo is not None and not isinstance(exo, ExoData):
                exo = ExoData(exo)
            exos.append(exo)
        # 4-D models: transpose + normalisation on the device
        # (s3_chunk_time_first, numpy's arithmetic) unless the model brings
        # its own norm_input
        from .gan import Sup3rGan as _BaseGan
        dev_norm = is_4d and options.device_norm_4d and \
            not any(step_exo(e, 0) for e in exos) and \
            getattr(type(first).norm_input, '__func__',
                    type(first).norm_input) is _BaseGan.norm_input
        for chunk, exo in zip(group, exos):
            # (one flat pass; the per-feature reduction over a (…, 2 .. 8)-wide
            # last axis — 0.7 ms per 75 x 75 x 48 chunk — only when it found one)
            if np.isnan(chunk.input_data).any():
                mask = np.isnan(chunk.input_data).any(axis=(0, 1, 2))
                feats = np.array(first.lr_features[:len(mask)])[mask]
                msg = f'Input data for {feats} contains NaN values!'
                logger.error(msg)
                raise RuntimeError(msg)
            if dev_norm:
                xs.append(np.asarray(chunk.input_data, dtype=np.float32))
                continue
            if is_4d:
                # (s1, s2, t, f) -> the t time steps as the batch of a 2-D
                # model (``_reshape_data_chunk``, forward_pass.py:274-337)
                # (contiguous: the normalisation below walks a transposed
                # view four times slower)
                x = first._combine_fwp_input(np.ascontiguousarray(np.transpose(
                    np.asarray(chunk.input_data), (2, 0, 1, 3))),
                    cls._batch_axis(step_exo(exo, 0), time_first=True))
            else:
                x = first._combine_fwp_input(
                    np.asarray(chunk.input_data)[None],
                    cls._batch_axis(step_exo(exo, 0)))
            xs.append(np.asarray(first.norm_input(x), dtype=np.float32))
        if dev_norm:
            raw = np.stack(xs, axis=0)                  # (n, s1, s2, t, f)
            x_shape = (n * raw.shape[3], raw.shape[1], raw.shape[2],
                       raw.shape[4])
            x = None
        else:
            x = np.concatenate(xs, axis=0) if n > 1 else xs[0]
            x_shape = tuple(x.shape)
        n_t = x_shape[0] // n          # 4-D: time steps per chunk
        lr_t = n_t if is_4d else x_shape[3]
        staged = []               # pinned upload buffers, alive until finish()
        pf = C.POINTER(C.c_float)
        i64x3 = C.c_int64 * 3

        def layer_exo_for(m, ph, i, rank4, nt):
            """the 'layer' exo fields step ``i``'s generator consumes
            mid-network, normalised, stacked over the batch, uploaded"""
            out = {}
            for name in ph.input_names:
                if name == 'x':
                    continue
                fields = []
                for exo in exos:
                    e = step_exo(exo, i)
                    assert e is not None and name in e, \
                        f'the generator needs exogenous feature "{name}"'
                    fields.append(np.asarray(
                        e.get_combine_type_data(name, 'layer')))
                out[name] = cls._exo_to_device(
                    dev, fields, rank4, staged,
                    lambda a, m=m, name=name: m._reshape_norm_exo(
                        tuple(a.shape), a, name))
                # (the plan of a 2-D model carries a time axis of one)
                want = tuple(int(v) for v in ph.in_shapes[name])
                if tuple(v for v in out[name].shape if v != 1) == \
                        tuple(v for v in want if v != 1):
                    out[name] = out[name].reshape(want)
                if tuple(out[name].shape) != want:
                    raise RuntimeError(
                        f'exogenous "{name}" of shape '
                        f'{tuple(out[name].shape)} cannot be laid over '
                        f'hi-res {want}')
            return out

        def hand_over(i, y, rank4, nt):
            """step i's normalised output -> step i + 1's normalised input
            (multi_step.py:233-259) without leaving the device"""
            m, nxt = steps[i], steps[i + 1]
            ysh = tuple(int(v) for v in y.shape)
            nxt4 = not getattr(nxt, 'is_5d', False)
            if rank4 and not nxt4:
                # (n t, H, W, C) -> n samples (H, W, t, C)
                # (_transpose_model_input, multi_step.py:107-146)
                yt = dev.empty((n, ysh[1], ysh[2], nt, ysh[3]))
                rc = L.s3_chunk_time_last(
                    dev.ctx, C.c_void_p(y.data_ptr()), n,
                    i64x3(nt, ysh[1], ysh[2]), i64x3(0, 0, 0),
                    i64x3(ysh[1], ysh[2], nt), ysh[3], None, None,
                    C.c_void_p(yt.data_ptr()))
                _lib.check(rc, dev.ctx, 's3_chunk_time_last')
                y, ysh = yt, tuple(int(v) for v in yt.shape)
            produced = list(m.hr_out_features)
            if len(produced) != ysh[-1]:
                raise RuntimeError(
                    f'step {i} produced {ysh[-1]} channels for '
                    f'{len(produced)} output features')
            nxt_exo = [step_exo(e, i + 1) for e in exos]
            wanted = [f for f in nxt.lr_features
                      if f not in (nxt_exo[0] or {})]
            missing = [f for f in wanted if f not in produced]
            if missing:
                raise ValueError(
                    f'step {i + 1} needs {missing}, step {i} only produces '
                    f'{produced}')
            extra = len(nxt.lr_features) - len(wanted)
            names = list(nxt.lr_features[-extra:]) if extra > 0 else []
            exo_t = None
            if names:
                fields = []
                for e in nxt_exo:
                    absent = [f for f in names if f not in (e or {})]
                    assert not absent, (f'exogenous_data lacks {absent} '
                                        '(combine_type "input")')
                    cols = [np.asarray(e.get_combine_type_data(f, 'input'))
                            for f in names]
                    fields.append(cols[0] if len(cols) == 1 else
                                  np.concatenate(cols, axis=-1))
                exo_t = cls._exo_to_device(dev, fields, nxt4, staged,
                                           lambda a: a)
                if tuple(exo_t.shape[:-1]) != ysh[:-1]:
                    raise RuntimeError(
                        f'"input" exo of step {i + 1} has shape '
                        f'{tuple(exo_t.shape)}, the data {ysh}')
            sc = sh_ = mu = sd = None
            if m._means is not None:
                mu0, sd0 = m._stats_for(m.hr_out_features)
                sc = np.ascontiguousarray(sd0, dtype=np.float32)
                sh_ = np.ascontiguousarray(mu0, dtype=np.float32)
            if nxt._means is not None:
                mu1, sd1 = nxt._stats_for(nxt.lr_features)
                if (sd1 == 0).any():
                    from warnings import warn
                    warn('a feature has zero standard deviation; dividing '
                         'by 1')
                    sd1 = np.where(sd1 == 0, 1, sd1)
                mu = np.ascontiguousarray(mu1, dtype=np.float32)
                sd = np.ascontiguousarray(sd1, dtype=np.float32)
            cmap = (C.c_int32 * len(wanted))(
                *[produced.index(f) for f in wanted])
            xn = dev.empty(ysh[:-1] + (len(wanted) + len(names),))
            n_pos = int(np.prod(ysh[:-1], dtype=np.int64))
            rc = L.s3_step_handover(
                dev.ctx, C.c_void_p(y.data_ptr()), n_pos, ysh[-1], cmap,
                len(wanted),
                sc.ctypes.data_as(pf) if sc is not None else None,
                sh_.ctypes.data_as(pf) if sh_ is not None else None,
                C.c_void_p(exo_t.data_ptr()) if exo_t is not None else None,
                len(names),
                mu.ctypes.data_as(pf) if mu is not None else None,
                sd.ctypes.data_as(pf) if sd is not None else None,
                C.c_void_p(xn.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_step_handover')
            return xn, nxt4
        try:
            if dev_norm:
                rawd = cls._upload_async(dev, raw, staged)
                xd = dev.empty(x_shape)
                mu = sd = None
                f32 = 1
                if first._means is not None:
                    mu, sd = first._stats_for(first.lr_features)
                    if len(mu) != x_shape[-1]:
                        raise RuntimeError(
                            f'{len(mu)} normalisation statistics for '
                            f'{x_shape[-1]} input features')
                    if (sd == 0).any():
                        from warnings import warn
                        warn('a feature has zero standard deviation; '
                             'dividing by 1')
                        sd = np.where(sd == 0, 1, sd)
                    f32 = int(mu.dtype == np.float32 and
                              sd.dtype == np.float32)
                    mu = np.ascontiguousarray(mu, dtype=np.float64)
                    sd = np.ascontiguousarray(sd, dtype=np.float64)
                pd = C.POINTER(C.c_double)
                rc = L.s3_chunk_time_first(
                    dev.ctx, C.c_void_p(rawd.data_ptr()), n,
                    (C.c_int64 * 3)(raw.shape[1], raw.shape[2], raw.shape[3]),
                    x_shape[-1],
                    mu.ctypes.data_as(pd) if mu is not None else None,
                    sd.ctypes.data_as(pd) if sd is not None else None, f32,
                    C.c_void_p(xd.data_ptr()))
                _lib.check(rc, dev.ctx, 's3_chunk_time_first')
            else:
                xd = cls._upload_async(dev, x, staged)
            # every step but the last: plan forward + hand-over on the device
            in_shape = x_shape
            for i in range(len(steps) - 1):
                ph_i = steps[i]._gen.plan(x_shape, training=False)
                y_i = ph_i.forward(xd, layer_exo_for(steps[i], ph_i, i, is_4d,
                                                     n_t))
                xd, is_4d = hand_over(i, y_i, is_4d, n_t)
                del y_i
                x_shape = tuple(int(v) for v in xd.shape)
            ph = last._gen.plan(x_shape, training=False)
            layer_exo = layer_exo_for(last, ph, len(steps) - 1, is_4d, n_t)
            # the enhancement checks of the reference (forward_pass.py:
            # _run_generator) on the plan's output shape
            yshape = tuple(int(v) for v in ph.out_shape)
            if model.s_enhance * in_shape[1] != yshape[1]:
                msg = ('The stated spatial enhancement of {}x did not match '
                       'the low res / high res shapes of {} -> {}'.format(
                           model.s_enhance, in_shape, yshape))
                logger.error(msg)
                raise _EnhancementMismatch(msg)
            if model.t_enhance * lr_t != (n_t if is_4d else yshape[3]):
                msg = ('The stated temporal enhancement of {}x did not match '
                       'the low res / high res shapes of {} -> {}'.format(
                           model.t_enhance, in_shape, yshape))
                logger.error(msg)
                raise _EnhancementMismatch(msg)
            n_out = yshape[-1]
            scale = shift = None
            if last.means is not None:
                mu, sd = last._stats_for(last.hr_out_features)
                scale = np.ascontiguousarray(sd, dtype=np.float32)
                shift = np.ascontiguousarray(mu, dtype=np.float32)
            # (4-D: the chunk's hi-res extents are (H, W, time steps))
            y1, y2, y3 = (yshape[1], yshape[2], n_t) if is_4d else yshape[1:4]
            cr = cls._crop_bounds(group[0].hr_crop_slice, (y1, y2, y3))
            c1, c2, c3 = (b - a for a, b in cr)
            yc = dev.empty((n, c1, c2, c3, n_out))
            stats_d = dev.empty((n, 64, n_out, 3))
            # halo crop + un-normalisation inside the tail conv (the window
            # forward: no full-size output, halo positions of the last conv
            # never computed) where the plan ends in the MFMA tail ...
            windowed = options.window_forward and ph.supports_window and \
                not is_4d
            if windowed:
                ph.forward_window(
                    xd, layer_exo, yc, [cr[0][0], cr[1][0], cr[2][0]],
                    [c1, c2, c3], cls._affine_tensor(dev, scale, shift))
                y = None
            else:
                y = ph.forward(xd, layer_exo)
        except (AssertionError, _EnhancementMismatch):
            raise
        except Exception as e:
            msg = 'Forward pass failed on chunk with shape {}.'.format(
                x_shape)
            logger.exception(msg)
            raise RuntimeError(msg) from e
        # ... else un-normalisation, halo crop and the output check's
        # statistics in one pass over the cropped window (s3_chunk_epilogue)
        # where the rows are 16-byte aligned, the three separate kernels
        # otherwise — same bits in every case
        fused = 1024 % n_out == 0 and n_out <= 16 and not any(
            (v * n_out) % 4 for v in (c3, cr[2][0], y3))
        if is_4d:
            # transpose to the chunk's (s1, s2, t) order + halo crop +
            # un-normalisation in one pass, then the output check's statistics
            i64x3 = C.c_int64 * 3
            rc = L.s3_chunk_time_last(
                dev.ctx, C.c_void_p(y.data_ptr()), n, i64x3(n_t, y1, y2),
                i64x3(cr[0][0], cr[1][0], cr[2][0]), i64x3(c1, c2, c3), n_out,
                scale.ctypes.data_as(pf) if scale is not None else None,
                shift.ctypes.data_as(pf) if shift is not None else None,
                C.c_void_p(yc.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_chunk_time_last')
            rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(yc.data_ptr()), n,
                                  yc.numel() // (n * n_out), n_out,
                                  C.c_void_p(stats_d.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_chunk_stats')
        elif windowed:
            rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(yc.data_ptr()), n,
                                  yc.numel() // (n * n_out), n_out,
                                  C.c_void_p(stats_d.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_chunk_stats')
        elif fused:
            i64x3 = C.c_int64 * 3
            rc = L.s3_chunk_epilogue(
                dev.ctx, C.c_void_p(y.data_ptr()), n, i64x3(y1, y2, y3),
                i64x3(cr[0][0], cr[1][0], cr[2][0]), i64x3(c1, c2, c3), n_out,
                scale.ctypes.data_as(pf) if scale is not None else None,
                shift.ctypes.data_as(pf) if shift is not None else None,
                C.c_void_p(yc.data_ptr()), C.c_void_p(stats_d.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_chunk_epilogue')
        else:
            if scale is not None:
                rc = L.s3_affine_channels(
                    dev.ctx, C.c_void_p(y.data_ptr()),
                    C.c_void_p(y.data_ptr()), n_out, y.numel() // n_out,
                    scale.ctypes.data_as(pf), shift.ctypes.data_as(pf))
                _lib.check(rc, dev.ctx, 's3_affine_channels')
            for k in range(n):
                src = y[k].data_ptr() + 4 * n_out * (
                    (cr[0][0] * y2 + cr[1][0]) * y3 + cr[2][0])
                rc = L.s3_copy_block(
                    dev.ctx, C.c_void_p(src), C.c_void_p(yc[k].data_ptr()),
                    c1, c2, c3 * n_out, y2 * y3 * n_out, y3 * n_out,
                    c2 * c3 * n_out, c3 * n_out)
                _lib.check(rc, dev.ctx, 's3_copy_block')
            rc = L.s3_chunk_stats(dev.ctx, C.c_void_p(yc.data_ptr()), n,
                                  yc.numel() // (n * n_out), n_out,
                                  C.c_void_p(stats_d.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_chunk_stats')
        del y
        copy_stream = cls._copy_stream(dev)
        stats_h = torch.empty(tuple(stats_d.shape), dtype=torch.float32,
                              pin_memory=True)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            stats_h.copy_(stats_d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        stats_d.record_stream(copy_stream)
        state = {'ticket': None, 'host': None}

        def deliver():
            """start the batch's device -> host DMA (idempotent).  The SDMA
            engine is driven through ROCr (s3_dma_d2h_begin), which takes no
            HIP event as a dependency: the host waits for the batch's forward
            first — by the time ``iter_chunks`` calls this the NEXT batch is
            already enqueued behind it, so the device does not idle."""
            if not return_data or state['host'] is not None:
                return
            ready.synchronize()
            host_ptr, host_arr = cls._delivery_buffer(
                dev, tuple(yc.shape), lane)
            state['host_ptr'] = host_ptr
            ticket = C.c_uint64()
            rc = -1
            if options.sdma_delivery and not cls._sdma_refused:
                rc = L.s3_dma_d2h_begin(
                    dev.ctx, C.c_void_p(yc.data_ptr()), C.c_void_p(host_ptr),
                    host_arr.size * 4, C.byref(ticket))
                if rc != 0:
                    # ROCr refused (no SDMA queue in this container, a foreign
                    # allocator ...): say so once and copy through HIP instead
                    logger.warning(
                        'SDMA delivery unavailable (%s): falling back to '
                        'hipMemcpyAsync for the chunk batches',
                        _lib.last_error(dev.ctx))
                    ForwardPass._sdma_refused = True
            if rc != 0:
                rc = L.s3_d2h_async(
                    dev.ctx, C.c_void_p(yc.data_ptr()), C.c_void_p(host_ptr),
                    host_arr.size * 4, C.c_void_p(copy_stream.cuda_stream))
                _lib.check(rc, dev.ctx, 's3_d2h_async')
                state['copy_ev'] = torch.cuda.Event()
                state['copy_ev'].record(copy_stream)
                ticket = None
            state['ticket'], state['host'] = ticket, host_arr

        def finish():
            deliver()
            ev.synchronize()
            staged.clear()
            st = stats_h.numpy().reshape(n, 64, n_out, 3)
            mn, mx = st[..., 0].min(1), st[..., 1].max(1)
            nn = st[..., 2].sum(1)
            skip, allowed = cls._const_ok(allowed_const)
            fails = []
            for k in range(n):
                failed = False
                if not skip:
                    if nn[k].any():
                        logger.error('Forward pass output contains NaN '
                                     'values!')
                        failed = True
                    else:
                        for i in range(n_out):
                            if mn[k, i] == mx[k, i] and \
                                    mn[k, i] not in allowed:
                                logger.error('All values are the same for '
                                             f'feature channel {i}!')
                                failed = True
                                break
                fails.append(failed)
            for k, chunk in enumerate(group):
                if write and chunk.out_file is not None and not fails[k]:
                    cls._write_chunk(chunk, model, yc[k], invert_uv, nn_fill,
                                     meta, output_workers)
            host_arr = state['host']
            if state['ticket'] is not None:
                rc = L.s3_dma_wait(dev.ctx, state['ticket'],
                                   int(dev.comm_timeout_s * 1000))
                state['ticket'] = None
                if rc != 0:
                    # the engine may still write into that buffer: it leaves
                    # the ring for good (never reused, never freed)
                    cls._retire_delivery_buffer(dev, tuple(yc.shape), lane,
                                                state['host_ptr'])
                _lib.check(rc, dev.ctx, 's3_dma_wait')
            elif state.get('copy_ev') is not None:
                state['copy_ev'].synchronize()
            for k, chunk in enumerate(group):
                # (a view into this generator's ring of delivery buffers:
                # valid while the next ``d2h_ring - 2`` batches are yielded —
                # batch k + d2h_ring claims the buffer of batch k when batch
                # k + d2h_ring + 1 is enqueued, just before k + d2h_ring - 1
                # is handed out)
                yield (chunk, fails[k],
                       host_arr[k] if host_arr is not None else None)
        finish.deliver = deliver
        return finish

    @classmethod
    def _exo_to_device(cls, dev, fields, rank4, keep, prep):
        """the exo fields of a batch's chunks -> one device tensor in the
        model's layout: ``fields`` = per chunk ``(s1, s2, t, c)``; a 2-D model
        (``rank4``) takes ``(n t, s1, s2, c)`` (the time steps on the batch
        axis, forward_pass.py:303-337), a 3-D one ``(n, s1, s2, t, c)``;
        ``prep`` (normalisation) runs on the host in that layout.  Fields
        that are constant in time — the zero-stride views ``pad_source_data``
        builds from 3-D exo data — cross PCIe once and are laid over the
        time steps on the device (s3_broadcast_axis)"""
        import ctypes as C

        from . import _lib
        n = len(fields)
        const = all(f.ndim == 4 and f.shape[2] > 1 and f.strides[2] == 0
                    for f in fields)
        lay = (lambda f: np.transpose(f, (2, 0, 1, 3))) if rank4 else \
            (lambda f: f[None])
        if const:
            nt = int(fields[0].shape[2])
            parts = [np.asarray(prep(lay(f[:, :, :1])), dtype=np.float32)
                     for f in fields]
            small = cls._upload_async(
                dev, np.concatenate(parts, axis=0) if n > 1 else parts[0],
                keep)
            h, w, c = (int(v) for v in (fields[0].shape[0],
                                        fields[0].shape[1],
                                        fields[0].shape[3]))
            out = dev.empty((n * nt, h, w, c) if rank4 else (n, h, w, nt, c))
            rc = _lib.lib().s3_broadcast_axis(
                dev.ctx, C.c_void_p(small.data_ptr()), n,
                1 if rank4 else h * w, h * w * c if rank4 else c, nt,
                C.c_void_p(out.data_ptr()))
            _lib.check(rc, dev.ctx, 's3_broadcast_axis')
            return out
        parts = [np.asarray(prep(lay(f)), dtype=np.float32) for f in fields]
        return cls._upload_async(
            dev, np.concatenate(parts, axis=0) if n > 1 else parts[0], keep)

    _upload_streams = {}

    @classmethod
    def _upload_async(cls, dev, arr, keep):
        """host array -> device tensor without blocking the host OR the compute
        stream: a pageable ``.to(device)`` on the compute stream returns only
        when the copy has run, i.e. after every kernel enqueued before it — the
        host could not prepare and enqueue the next batch under the current
        one.  The array goes through a pinned staging tensor (kept alive in
        ``keep`` until the batch is finished) and a non-blocking copy on an
        UPLOAD stream of its own (round 6: on the compute stream the copy of
        batch k + 1 — 157 us per 4 x 75 x 75 x 48 chunk batch — ran between the
        last kernel of batch k and the first of batch k + 1 with the device
        idle; now it crosses PCIe under batch k's kernels and the compute
        stream only waits for its event)."""
        import torch
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        stage = torch.empty(arr.shape, dtype=torch.float32, pin_memory=True)
        stage.numpy()[...] = arr
        keep.append(stage)
        with cls._lane_lock:
            up = cls._upload_streams.get(dev.index)
            if up is None:
                up = cls._upload_streams[dev.index] = torch.cuda.Stream(
                    device=dev.torch_device)
        compute = torch.cuda.current_stream()
        with torch.cuda.stream(up):
            t = stage.to(dev.torch_device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(up)
        compute.wait_event(done)
        # (allocated under the upload stream, used on the compute stream)
        t.record_stream(compute)
        return t

    @staticmethod
    def _batch_axis(exo, time_first=False):
        """the exo structure of ONE chunk with a leading batch axis on every
        field — or, for a 4-D model, the field's time axis moved there
        (``_reshape_data_chunk``, forward_pass.py:303-337), without touching
        the caller's arrays"""
        if exo is None:
            return None
        from .utilities import ExoData
        if time_first:
            return ExoData({f: {'steps': [dict(st, data=np.transpose(
                np.asarray(st['data']), (2, 0, 1, 3))) for st in e['steps']]}
                for f, e in exo.items()})
        return ExoData({f: {'steps': [dict(st, data=np.asarray(st['data'])[
            None]) for st in e['steps']]} for f, e in exo.items()})

    #: a batch's hi-res chunks land in a ring of pinned host buffers
    #: (s3_host_alloc) per output shape, allocated once: pinning 368 MB per
    #: batch cost 8.9 ms of host time each (torch.empty(pin_memory=True))
    d2h_ring = 4
    #: ROCr refused an SDMA copy in this process (no SDMA queue in the
    #: container, a foreign allocator ...): every later batch copies through
    #: HIP.  A latch about the runtime, set once — the steering flags are
    #: ``ChunkPathOptions``
    _sdma_refused = False
    _aff_cache = {}
    _delivery = {}
    _lanes = set()
    _lane_lock = __import__('threading').Lock()

    @classmethod
    def _take_lane(cls):
        with cls._lane_lock:
            lane = 0
            while lane in cls._lanes:
                lane += 1
            cls._lanes.add(lane)
        return lane

    @classmethod
    def _retire_delivery_buffer(cls, dev, shape, lane, ptr):
        ring = cls._delivery.get((dev.index, shape, lane))
        if ring:
            ring['bufs'] = [b for b in ring['bufs'] if b[0] != ptr]
            ring['next'] = 0

    @classmethod
    def _affine_tensor(cls, dev, scale, shift):
        """scale[C] then shift[C] on the device (None without statistics);
        one upload per distinct set of statistics"""
        if scale is None:
            return None
        key = (dev.index, scale.tobytes(), shift.tobytes())
        with cls._lane_lock:
            t = cls._aff_cache.get(key)
            if t is None:
                if len(cls._aff_cache) > 64:
                    cls._aff_cache.clear()
                t = cls._aff_cache[key] = dev.to_device(
                    np.concatenate([scale, shift]).astype(np.float32))
        return t

    @classmethod
    def release_delivery_buffers(cls):
        """free the pinned delivery rings (up to ``d2h_ring`` x the batch's
        cropped output per output shape); views handed out earlier become
        invalid"""
        from . import _lib
        from .engine import Device
        import ctypes as C
        for (index, _, _), ring in list(cls._delivery.items()):
            dev = Device.get(index)
            for ptr, _arr in ring['bufs']:
                _lib.lib().s3_host_free(dev.ctx, C.c_void_p(ptr))
        cls._delivery.clear()

    @classmethod
    def _delivery_buffer(cls, dev, shape, lane=0):
        import ctypes as C

        from . import _lib
        key = (dev.index, shape, lane)
        ring = cls._delivery.setdefault(key, {'bufs': [], 'next': 0})
        if len(ring['bufs']) < cls.d2h_ring:
            n = int(np.prod(shape))
            ptr = C.c_void_p()
            rc = _lib.lib().s3_host_alloc(dev.ctx, n * 4, 0, C.byref(ptr))
            _lib.check(rc, dev.ctx, 's3_host_alloc')
            arr = np.ctypeslib.as_array(
                (C.c_float * n).from_address(ptr.value)).reshape(shape)
            ring['bufs'].append((ptr.value, arr))
            ring['next'] = len(ring['bufs']) % cls.d2h_ring
            return ring['bufs'][-1]
        buf = ring['bufs'][ring['next'] % cls.d2h_ring]
        ring['next'] += 1
        return buf

    _copy_streams = {}

    @classmethod
    def _copy_stream(cls, dev):
        import torch
        with cls._lane_lock:
            if dev.index not in cls._copy_streams:
                cls._copy_streams[dev.index] = torch.cuda.Stream(
                    device=dev.torch_device)
            return cls._copy_streams[dev.index]

    @classmethod
    def run_chunk(cls, chunk, model_kwargs, model_class, allowed_const,
                  invert_uv=False, meta=None, nn_fill=True,
                  output_workers=None, options=None):
        """Run a forward pass on a single spatiotemporal chunk
        (forward_pass.py:582-673): same arguments, same return value
        ``(failed, output_data)`` — ``output_data`` the cropped, un-normalised
        hi-res array ``(s1, s2, t, features)``; when ``chunk.out_file`` is set
        and the chunk did not fail the (device-)transformed output is written
        through ``OUTPUT_HANDLER_CLASS``.  The model is loaded once per
        process (``get_model``), not per chunk."""
        logger.info(f'Running forward pass for chunk_index={chunk.index}.')
        model = get_model(model_class, model_kwargs)
        (_, failed, output_data), = cls.iter_chunks(
            [chunk], model, allowed_const=allowed_const, batch=1,
            invert_uv=invert_uv, nn_fill=nn_fill, meta=meta,
            output_workers=output_workers, options=options)
        # (the caller owns what it gets: not a view of the delivery ring)
        return failed, np.array(output_data)

    @classmethod
    def run(cls, strategy, node_index, batch=8, return_data=False,
            options=None):
        """Forward passes on all chunks of one node (forward_pass.py:427-449,
        ``_run_serial`` :451-500) — one node = one process = one GPU; the
        node's chunks are the rank's share of the data-parallel work, there
        is no collective.  Chunks are stacked ``batch`` at a time on the
        device (``iter_chunks``).  Raises ``MemoryError`` on a failed chunk
        like the reference.  Returns the number of chunks run (and, with
        ``return_data``, the list of ``(chunk_index, output_data)``)."""
        if strategy.node_finished(node_index):
            return (0, []) if return_data else 0
        fwp = cls(strategy, node_index=node_index, options=options)
        todo = [int(i) for i in strategy.node_chunks[node_index]
                if not strategy.chunk_finished(int(i))]

        def chunks():
            for i in todo:
                yield fwp.get_input_chunk(chunk_index=i)
        done, kept = 0, []
        meta = fwp.meta
        for chunk, failed, data in cls.iter_chunks(
                chunks(), fwp.model, allowed_const=strategy.allowed_const,
                batch=batch, invert_uv=getattr(strategy, 'invert_uv', False),
                nn_fill=getattr(strategy, 'nn_fill', True), meta=meta,
                output_workers=getattr(strategy, 'output_workers', None),
                return_data=return_data, options=fwp.options):
            if failed:
                raise MemoryError(
                    f'Forward pass for chunk_index {chunk.index} failed '
                    'with constant output or NaNs.')
            if hasattr(strategy, 'mark_finished'):
                strategy.mark_finished(chunk.index)
            if return_data:
                kept.append((chunk.index, np.array(data)))
            done += 1
        logger.info('Finished forward passes on %d chunks', done)
        return (done, kept) if return_data else done

    _run_serial = run

    def run_chunks(self, domain, out=None, writer=None):
        """The chunk-by-chunk loop over an in-memory domain: one
        ``model.generate`` per chunk through host numpy.  Works for every model (exogenous inputs,
        4-D and multi-step models); host-bound at ~10 chunks/s."""
        done = 0
        for idx in self.my_chunks():
            hr = self.run_domain_chunk(domain, idx)
            sl = self.slicer.chunks[idx]['hr_slice']
            if writer is not None:
                writer(idx, sl, hr)
            elif out is not None:
                out[sl] = hr
            done += 1
        return done

    def run_domain(self, domain, out=None, writer=None, batch=8):
        """Process this rank's chunks of ``domain`` (s1, s2, t, features) —
        what ``ForwardPass.run`` (forward_pass.py:427-449) is to a strategy.

        ``out``: optional pre-allocated hi-res array (s1*s, s2*s, t*te, f_out)
        each cropped chunk is placed into (ranks write disjoint windows).
        ``writer(chunk_index, hr_slice, data)`` is called per chunk instead when
        given (file output lives in sup3r's writers).  Returns the number of
        chunks run.

        Single-step 5-D models without exogenous inputs go through the
        device-resident executor (:meth:`run_batched`, ~300 chunks/s, the same
        bits — ``tests/test_forward_pass_gpu.py``); everything else through
        the chunk-by-chunk loop (:meth:`run_chunks`)."""
        model = self.model
        if getattr(model, '_gen', None) is not None and \
                getattr(model, 'is_5d', False) and \
                getattr(model, 'supports_device_chunks', False) and \
                not getattr(model, 'hr_exo_features', []) and \
                domain.shape[-1] == len(getattr(model, 'lr_features', [])):
            return self.run_batched(domain, out=out, writer=writer,
                                    batch=batch)
        return self.run_chunks(domain, out=out, writer=writer)
