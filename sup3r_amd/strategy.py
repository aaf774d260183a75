"""Chunk algebra and the chunk structure of the forward pass.

``ChunkSlicer`` restates the slice algebra of ``sup3r.pipeline.slicer.
ForwardPassSlicer`` (padded / unpadded lo-res slices, hi-res windows, crop
slices, domain-edge reflect padding; chunk ordering of ``chunk_lookup``,
slicer.py:473-483,668-716).  ``ForwardPassChunk`` has the fields of the
reference's structure of that name (sup3r/pipeline/strategy.py:37-54) — the
executor (``forward_pass.ForwardPass.run_chunk`` / ``run``) only duck-types
it, so the reference's own objects go through unchanged.  ``ArrayStrategy``
is the in-memory counterpart of ``ForwardPassStrategy`` for the attributes
the executor reads (``init_chunk`` strategy.py:520-581, ``node_chunks``
:363-372, ``chunk_finished``, ``model_kwargs`` / ``model_class`` /
``allowed_const`` / ``invert_uv`` / ``nn_fill`` / ``output_workers``): file
IO, bias correction and exo rasterisation stay in sup3r (out of
scope), here the lo-res domain and the exo fields are arrays.
"""
import copy
import os
from dataclasses import dataclass, field

import numpy as np


def chunk_slices(size, chunk):
    """[slice(0, chunk), slice(chunk, 2*chunk), ...] covering ``size``
    (sup3r.pipeline.utilities.get_chunk_slices semantics, step 1)."""
    out, start = [], 0
    while start < size:
        stop = min(start + chunk, size)
        out.append(slice(start, stop))
        start = stop
    return out


@dataclass
class ChunkSlicer:
    """Index algebra for tiling a lo-res domain (s1, s2, t) into generator
    chunks with overlap, and placing the cropped hi-res output.

    For chunk ``i``: ``lr_pad_slice`` is the in-domain padded window read from
    the source, ``pad_width`` the extra reflect padding applied at domain edges
    so every chunk sees ``spatial_pad`` / ``temporal_pad`` cells of context on
    all sides, ``hr_crop`` removes the enhanced padding from the generator
    output and ``hr_slice`` is where the result lands in the hi-res domain."""

    coarse_shape: tuple
    time_steps: int
    s_enhance: int
    t_enhance: int
    chunk_shape: tuple
    spatial_pad: int = 0
    temporal_pad: int = 0
    chunks: list = field(default_factory=list, init=False)

    def __post_init__(self):
        s1 = chunk_slices(self.coarse_shape[0], self.chunk_shape[0])
        s2 = chunk_slices(self.coarse_shape[1], self.chunk_shape[1])
        tt = chunk_slices(self.time_steps, self.chunk_shape[2])
        self.n_spatial_chunks = len(s1) * len(s2)
        self.n_time_chunks = len(tt)
        dims = (self.coarse_shape[0], self.coarse_shape[1], self.time_steps)
        pads = (self.spatial_pad, self.spatial_pad, self.temporal_pad)
        enh = (self.s_enhance, self.s_enhance, self.t_enhance)
        # chunk index = t_idx * n_spatial + (s1_idx * n_s2 + s2_idx), the
        # ordering of ForwardPassSlicer.chunk_lookup (slicer.py:473-483)
        for t_sl in tt:
            for a in s1:
                for b in s2:
                    lr = (a, b, t_sl)
                    lr_pad, pad_width, hr_crop, hr_slice = [], [], [], []
                    for sl, n, p, e in zip(lr, dims, pads, enh):
                        start, stop = max(0, sl.start - p), min(n, sl.stop + p)
                        lr_pad.append(slice(start, stop))
                        lo = max(0, p - sl.start)
                        hi = max(0, sl.stop + p - n)
                        pad_width.append((lo, hi))
                        hr_crop.append(slice(
                            p * e, p * e + (sl.stop - sl.start) * e))
                        hr_slice.append(slice(sl.start * e, sl.stop * e))
                    self.chunks.append(dict(
                        lr_slice=lr, lr_pad_slice=tuple(lr_pad),
                        pad_width=tuple(pad_width), hr_crop=tuple(hr_crop),
                        hr_slice=tuple(hr_slice)))

    @property
    def n_chunks(self):
        return len(self.chunks)

    @property
    def hr_shape(self):
        return (self.coarse_shape[0] * self.s_enhance,
                self.coarse_shape[1] * self.s_enhance,
                self.time_steps * self.t_enhance)

    @property
    def extra_padding(self):
        """per chunk reflect padding at the domain edges (slicer.py:675-716)"""
        return [c['pad_width'] for c in self.chunks]

    def get_chunk_indices(self, chunk_index):
        return (chunk_index % self.n_spatial_chunks,
                chunk_index // self.n_spatial_chunks)

    def rank_chunks(self, rank, nranks, mode='interleave'):
        """Chunk ids of one rank.  'interleave' balances ragged edge chunks;
        'block' mirrors ForwardPassStrategy.node_chunks (np.array_split,
        strategy.py:363-372)."""
        ids = np.arange(self.n_chunks)
        if mode == 'block':
            return list(np.array_split(ids, nranks)[rank])
        return list(ids[rank::nranks])


@dataclass
class ForwardPassChunk:
    """One chunk going through the generator: the fields of
    ``sup3r.pipeline.strategy.ForwardPassChunk`` (strategy.py:37-54)."""

    input_data: np.ndarray
    exo_data: dict
    hr_crop_slice: tuple
    lr_pad_slice: tuple
    hr_lat_lon: np.ndarray
    hr_times: object
    gids: np.ndarray
    out_file: str
    pad_width: tuple
    index: int

    def __post_init__(self):
        self.shape = self.input_data.shape


class ArrayStrategy:
    """What ``ForwardPass`` reads from a ``ForwardPassStrategy``, over an
    in-memory lo-res domain ``(s1, s2, t, features)``.

    ``exo_data``: feature -> {'steps': [{'model', 'combine_type', 'data',
    's_enhance', 't_enhance'}]} over the WHOLE domain at each step's
    resolution (``ExoData``, preprocessing/data_handlers/exo.py:54-273); 3-D
    ``(s1, s2, 1)`` fields are constant in time.  ``max_nodes``: number of
    nodes (= ranks, one per GPU) the chunk list is split over with
    ``np.array_split`` like strategy.py:363-372."""

    def __init__(self, domain, model_kwargs, fwp_chunk_shape, spatial_pad=1,
                 temporal_pad=1, model_class='Sup3rGan', exo_data=None,
                 out_pattern=None, allowed_const=False, invert_uv=False,
                 nn_fill=True, output_workers=None, max_nodes=1,
                 lat_lon=None, time_index=None, meta=None, s_enhance=None,
                 t_enhance=None, model=None):
        from .forward_pass import get_model
        self.domain = domain
        self.model_kwargs = model_kwargs
        self.model_class = model_class
        self.spatial_pad, self.temporal_pad = spatial_pad, temporal_pad
        self.exo_data = exo_data
        self.out_pattern = out_pattern
        self.allowed_const = allowed_const
        self.invert_uv, self.nn_fill = invert_uv, nn_fill
        self.output_workers = output_workers
        self.max_nodes = max_nodes
        self.pass_workers = 1
        self.meta = dict(meta or {})
        self.lat_lon, self.time_index = lat_lon, time_index
        if s_enhance is None or t_enhance is None:
            model = model or get_model(model_class, model_kwargs)
            s_enhance, t_enhance = model.s_enhance, model.t_enhance
        self.s_enhance, self.t_enhance = int(s_enhance), int(t_enhance)
        shape = tuple(int(v) for v in domain.shape[:3])
        self.fwp_chunk_shape = tuple(
            int(c or n) for c, n in zip(fwp_chunk_shape, shape))
        self.fwp_slicer = ChunkSlicer(
            shape[:2], shape[2], self.s_enhance, self.t_enhance,
            self.fwp_chunk_shape, spatial_pad=spatial_pad,
            temporal_pad=temporal_pad)
        self._finished = set()
        self._out_files = None
        self._gids = None

    # -- what the executor reads -----------------------------------------
    @property
    def n_chunks(self):
        return self.fwp_slicer.n_chunks

    @property
    def out_files(self):
        """strategy.py:455-472 (a cached property there too)"""
        if self._out_files is None:
            self._out_files = self._make_out_files()
        return self._out_files

    def _make_out_files(self):
        sl = self.fwp_slicer
        if self.out_pattern is None:
            return [None] * sl.n_chunks
        assert '{file_id}' in self.out_pattern, \
            'out_pattern must include a {file_id} format key'
        os.makedirs(os.path.dirname(self.out_pattern) or '.', exist_ok=True)
        return [self.out_pattern.format(
            file_id=f'{str(i).zfill(6)}_{str(j).zfill(6)}')
            for i in range(sl.n_time_chunks)
            for j in range(sl.n_spatial_chunks)]

    def chunk_finished(self, chunk_index, log=True):
        out_file = self.out_files[chunk_index]
        return chunk_index in self._finished or (
            out_file is not None and os.path.exists(out_file))

    @property
    def node_chunks(self):
        chunks = [c for c in range(self.n_chunks)]
        n = max(1, min(self.max_nodes or len(chunks), len(chunks)))
        return np.array_split(chunks, n)

    def node_finished(self, node_index):
        return all(self.chunk_finished(i)
                   for i in self.node_chunks[node_index])

    def _exo_chunk(self, lr_slices):
        """ExoData.get_chunk (exo.py:205-273): every step's field cut to the
        chunk's lo-res extent times the step's enhancement."""
        if self.exo_data is None:
            return None
        out = {}
        for feature, entry in self.exo_data.items():
            steps = []
            for step in entry['steps']:
                s_en = int(step.get('s_enhance', 1))
                t_en = int(step.get('t_enhance', 1))
                new = {k: v for k, v in step.items() if k != 'data'}
                new.setdefault('s_enhance', s_en)
                new.setdefault('t_enhance', t_en)
                ens = (s_en, s_en, t_en)
                sl = tuple(slice(s_.start * e, s_.stop * e)
                           for s_, e in zip(lr_slices, ens))
                data = step['data']
                new['data'] = np.array(data[sl[:data.ndim - 1]])
                steps.append(new)
            out[feature] = {'steps': steps}
        return out

    def init_chunk(self, chunk_index=0):
        """strategy.py:520-581: the chunk's un-padded-at-the-edges lo-res
        window, its exo data, crop slices and edge padding."""
        assert chunk_index < self.n_chunks, (chunk_index, self.n_chunks)
        sl = self.fwp_slicer
        c = sl.chunks[chunk_index]
        data = np.array(self.domain[c['lr_pad_slice']])
        hr = c['hr_slice']
        lat_lon = None if self.lat_lon is None else \
            np.asarray(self.lat_lon)[hr[0], hr[1]]
        times = None if self.time_index is None else \
            self.time_index[hr[2]]
        if self._gids is None:
            n1, n2 = sl.hr_shape[:2]
            self._gids = np.arange(n1 * n2).reshape(n1, n2)
        gids = self._gids[hr[0], hr[1]]
        return ForwardPassChunk(
            input_data=data, exo_data=self._exo_chunk(c['lr_pad_slice']),
            lr_pad_slice=c['lr_pad_slice'], hr_crop_slice=c['hr_crop'],
            hr_lat_lon=lat_lon, hr_times=times, gids=gids,
            out_file=self.out_files[chunk_index], pad_width=c['pad_width'],
            index=chunk_index)

    def mark_finished(self, chunk_index):
        self._finished.add(int(chunk_index))

    def copy(self):
        return copy.copy(self)
