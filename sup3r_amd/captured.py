"""Recorded training steps: one ``hipGraphLaunch`` per mini-batch.

``Sup3rGan._train_batch`` (sup3r/models/base.py:944-1031) on the reference's
own CPU-runnable case (tests/training/test_train_gan.py:45-114: batch 15 of
5 x 5 -> 10 x 10) is ~650 kernel launches of a few microseconds each: the step
is bound by launch latency, not by anything a kernel could do better.  The
launch sequence of a mini-batch is the same every time the same networks
train on the same shapes, so it is recorded once (``s3_capture_begin`` ..
``s3_capture_end``, include/sup3r_hip.h) and replayed.

What makes a replay equal to the eager step:

* inputs: the batch is copied INTO the recorded input buffers;
* buffers: every tensor handed out while recording (``Device._retain``) lives
  as long as the graph; plan arenas are static (a ``clear_plans`` or a changed
  option starts a new record);
* scalars that change per step: the optimizer's ``alpha(t)`` etc. are staged on
  the device before every replay (``s3_optimizer_stage``), the recorded update
  launch reads them there; ``optimizer.iterations`` is counted by the replay;
* weight-version logic on the host (re-pack of filter images, the shared
  D(hi_res_true)): both stores are touched before recording so that every
  pack is part of the graph, and after every replay so that eager code
  re-packs;
* loss scalars: each replay snapshots the recorded scalar buffers into the
  ``LossFuture`` it returns (an epoch may resolve them all at its end).

Not recorded (the step then runs eagerly, as before): multi-GPU steps (RCCL),
structured loss terms (seeded projections), subclasses with their own
gradient routine, anything that uploads inside the step.  A capture that
fails falls back to eager for that key with one warning.
"""
import collections
import ctypes as C
import logging
from warnings import warn

from . import _lib
from .compute import LossFuture
from .engine import _torch

logger = logging.getLogger(__name__)


class _Recorded:
    def __init__(self):
        self.graph = None
        self.inputs = {}          # field name -> recorded input buffer
        self.retained = []        # every buffer of the step
        self.opt_steps = []       # (net, optimizer) in launch order
        self.futures = []         # (scal, recipe, scale) per network step
        self.nodes = 0

    def __del__(self):
        try:
            if self.graph:
                _lib.lib().s3_graph_destroy(self.graph)
                self.graph = None
        except Exception:
            pass


class StepRecorder:
    """Per-model cache of recorded ``_launch_batch`` sequences."""

    WARM = 2          # eager runs of a key before it is recorded
    MAX_ELEMS = 1 << 20   # 'auto': hi-res batch elements up to which a step is launch-bound
    # recorded graphs kept alive at once: each holds its hipGraph and every
    # buffer of its step.  The key contains values that move during a long
    # training (the adversarial weight of ``update_adversarial_weights``, a
    # replaced optimizer, a new plan epoch): the least recently used records
    # beyond this bound are destroyed, and records that can never be replayed
    # again (stale plan epoch / options / optimizer objects) go at once.
    MAX_RECORDS = 6

    def __init__(self, compute, fields=('low_res', 'high_res')):
        """``fields``: the attributes of a batch the step reads (``Sup3rGan``:
        low_res / high_res; ``Sup3rCondMom``: low_res / output / mask) — each
        gets a static device buffer the batch is copied into before a replay"""
        self.compute = compute
        self.fields = tuple(fields)
        self.dev = compute.dev
        self._entries = collections.OrderedDict()
        self.replays = 0
        self.evicted = 0

    def _state_digest(self, model, nets):
        """what is baked into a recorded graph beyond shapes: the content-loss
        terms (kinds, weights and kwargs are kernel arguments and part of the
        LossFuture recipe), each network's precision (selects the plan) and
        whether D(hi_res_true) may be shared between the two steps"""
        def term(t):
            name, kind, weight, kw = t
            return (str(name), repr(kind), float(weight),
                    tuple(sorted((str(k), repr(v))
                                 for k, v in (kw or {}).items())))
        return (tuple(term(t) for t in model._loss_terms),
                tuple(str(getattr(n, 'precision', None)) for n in nets),
                bool(getattr(self.compute, 'share_dtrue_allowed', True)))

    def _prune(self, live):
        """drop the records no key can reach any more, then the least recently
        used ones beyond MAX_RECORDS (``live`` = (options_key, plan epochs,
        optimizer ids) of the step being run)"""
        # key = (input shapes, options_key, plan epochs, optimizers, digest, ...)
        stale = [k for k in self._entries if (k[1], k[2]) != live[:2]
                 or not set(i for i, _ in k[3]) <= live[2]]
        for k in stale:
            self._drop(k)
        recorded = [k for k, e in self._entries.items() if e['rec'] is not None]
        for k in recorded[:max(0, len(recorded) - self.MAX_RECORDS)]:
            self._drop(k)
        # un-recorded bookkeeping entries are tiny but unbounded too
        while len(self._entries) > 8 * self.MAX_RECORDS:
            self._drop(next(iter(self._entries)))

    def _drop(self, key):
        ent = self._entries.pop(key, None)
        if ent and ent['rec'] is not None:
            rec, ent['rec'] = ent['rec'], None
            # replays are asynchronous (hipGraphLaunch on the context's
            # stream): the record being dropped may be the one launched on the
            # previous step — wait for the device before its exec graph and
            # the buffers it reads go away (eviction is rare)
            if rec.graph or rec.retained:
                self.dev.sync()
            if rec.graph:
                _lib.lib().s3_graph_destroy(rec.graph)
                rec.graph = None
            rec.retained = []
            self.evicted += 1

    # ------------------------------------------------------------------ use
    def eligible(self, model, batch, mode, multi_gpu):
        if not mode or multi_gpu or self.dev.nranks > 1:
            return False
        if any(isinstance(kind, str) for _, kind, _, _ in model._loss_terms):
            return False
        if any(getattr(batch, f, None) is None for f in self.fields):
            return False
        if mode == 'auto':
            worst = 0
            for f in self.fields:
                n = 1
                for v in getattr(batch, f).shape:
                    n *= int(v)
                worst = max(worst, n)
            return worst <= self.MAX_ELEMS
        return True

    def run(self, batch, key, optimizers, body, model=None):
        """``body(resident batch) -> [LossFuture, ...]`` is the eager step;
        returns its futures — from a replay once ``key`` has been seen
        ``WARM`` times."""
        nets = [n for n in (self.compute.gen, self.compute.disc)
                if n is not None]
        epochs = tuple(getattr(n, 'plan_epoch', 0) for n in nets)
        key = (tuple(tuple(getattr(batch, f).shape) for f in self.fields),
               self.dev.options_key, epochs,
               tuple((id(o), o.KIND) for o in optimizers),
               self._state_digest(model, nets) if model is not None else ()
               ) + tuple(key)
        live_opts = set(id(o) for o in optimizers)
        if model is not None:
            live_opts |= {id(getattr(model, a, None))
                          for a in ('optimizer', 'optimizer_disc')}
        live = (self.dev.options_key, epochs, live_opts)
        if key not in self._entries:
            self._prune(live)
        ent = self._entries.setdefault(key, {'seen': 0, 'rec': None,
                                             'bad': False})
        self._entries.move_to_end(key)
        if ent['rec'] is None:
            ready = all(o.iterations >= 1 for o in optimizers)
            if ent['bad'] or ent['seen'] < self.WARM or not ready:
                ent['seen'] += 1
                return body(self._resident(batch))
            try:
                ent['rec'] = self._record(batch, body, nets)
            except Exception as e:       # noqa: BLE001 — eager still works
                ent['bad'] = True
                warn(f'training step not recorded ({e}); running it eagerly',
                     RuntimeWarning)
                return body(self._resident(batch))
            logger.debug('recorded a training step: %d graph nodes',
                         ent['rec'].nodes)
            self._prune(live)     # the new record is the most recent one
        return self._replay(ent['rec'], batch, nets)

    # ------------------------------------------------------------- internals
    def _resident(self, batch):
        import types
        return types.SimpleNamespace(**{
            f: self.dev.to_device(getattr(batch, f)) for f in self.fields})

    def _load(self, rec, batch):
        torch = _torch()
        for f in self.fields:
            dst, src = rec.inputs[f], getattr(batch, f)
            if not isinstance(src, torch.Tensor):
                src = self.dev.to_device(src)
            dst.copy_(src.reshape(dst.shape))

    def _record(self, batch, body, nets):
        L = _lib.lib()
        dev, compute = self.dev, self.compute
        rec = _Recorded()
        import types
        for f in self.fields:
            rec.inputs[f] = dev.empty(tuple(getattr(batch, f).shape))
        Static = types.SimpleNamespace(**rec.inputs)
        for n in nets:            # every filter pack becomes part of the graph
            n.touch()
        _lib.check(L.s3_capture_begin(dev.ctx), dev.ctx, 's3_capture_begin')
        dev._retain = rec.retained
        compute._recording = rec.opt_steps
        try:
            futures = body(Static)
            if not all(isinstance(f, LossFuture) for f in futures):
                raise RuntimeError('a step that reads back cannot be replayed')
            h = C.c_void_p()
            _lib.check(L.s3_capture_end(dev.ctx, C.byref(h)), dev.ctx,
                       's3_capture_end')
        except BaseException:
            L.s3_capture_abort(dev.ctx)
            for n in nets:        # packs were recorded, not run: not current
                n.touch()
            raise
        finally:
            dev._retain = None
            compute._recording = None
        rec.graph = h
        rec.nodes = int(L.s3_graph_nodes(h))
        for f in futures:
            rec.futures.append((f._scal, f._recipe, f._scale))
        return rec

    def _replay(self, rec, batch, nets):
        L = _lib.lib()
        dev = self.dev
        self._load(rec, batch)
        for net, opt in rec.opt_steps:
            opt.iterations += 1
            net.optimizer_stage(opt.KIND, opt.hyper(), opt.iterations)
        _lib.check(L.s3_graph_launch(rec.graph), dev.ctx, 's3_graph_launch')
        for n in nets:
            n.touch()
        self.replays += 1
        return [LossFuture(scal.clone(), recipe, scale, dev=dev)
                for scal, recipe, scale in rec.futures]
