"""``Sup3rGan`` on MI355X: the reference's Python surface over the HIP engine.

Drop-in for ``sup3r.models.Sup3rGan`` on the hot path: the
constructor, ``generate / discriminate / calc_loss / train / save / load``,
the history schema, the ``meta`` keys, ``model_params.json`` and the checkpoint
file names are the reference's, so ``ForwardPassStrategy`` / ``ForwardPass``
and a ``BatchHandler`` drive it unchanged.  What each method must DO is taken
from sup3r/models/base.py (Sup3rGan), abstract.py (AbstractSingleModel) and
interface.py (AbstractInterface) — cited per method; how it is done is this
package's own:

* all arithmetic is ``compute.HipGanCompute`` (libsup3r_hip.so); nothing here
  touches a tensor element;
* loss scalars stay on the device as ``LossFuture``s while the host enqueues
  the all-reduce, the Adam step and the other network's step: one device ->
  host read per mini-batch when the train / skip gating needs the running
  discriminator loss, one per epoch otherwise;
* the per-batch record is ``ledger.LossWindow`` (numpy rings), pandas is used
  once per epoch for the history row;
* data-parallel training is one process per GPU: ``train(multi_gpu=True)``
  joins the RCCL communicator, takes this rank's slice of every mini-batch,
  SUMs the gradients (abstract.py:785-805) and steps identically everywhere.
"""
import copy
import json
import logging
import os
import pprint
import re
import time
from warnings import warn

import numpy as np

from . import __version__
from . import _lib
from .compute import HipGanCompute, LossFuture, parse_loss_spec
from .ledger import History, LossWindow
from .optimizers import get_optimizer_class, init_optimizer
from .spec import EXO_CLASSES, OBS_CLASSES, load_hidden_layers
from .utilities import ExoData, Timer, numpy_if_tensor, safe_cast

logger = logging.getLogger(__name__)

VERSION_RECORD = {'sup3r_amd': __version__, 'backend': 'libsup3r_hip (gfx950)'}

_META_KEYS = ('input_resolution', 'lr_features', 'hr_exo_features',
              'hr_out_features', 'smoothed_features', 's_enhance',
              't_enhance', 'smoothing')
_HANDLER_KEYS = ('smoothing', 'lr_features', 'hr_exo_features',
                 'hr_out_features', 'smoothed_features')


def _leading_int(text):
    return int(re.search(r'\d+', text).group(0))



def _per_channel(x, mu, sigma, fn, _period=2048):
    """``fn(x, mu, sigma)`` broadcast over the last (feature) axis — the same
    element-wise arithmetic, hence the same bits, as the plain numpy
    expression, but with the statistics tiled to rows of ``_period`` cells:
    numpy walks a broadcast over a 2 … 8-wide last axis with an inner loop of
    that length (2.4 ms for the (48, 75, 75, 2) batch of a spatial chunk,
    0.3 ms this way)."""
    x = np.asarray(x)
    c = x.shape[-1] if x.ndim else 0
    if not (x.ndim >= 2 and x.flags.c_contiguous and mu.shape == (c,)
            and sigma.shape == (c,) and x.size >= 4 * c * _period):
        return fn(x, mu, sigma)
    flat = x.reshape(-1)
    row = c * _period
    main = (flat.size // row) * row
    mu_t, sg_t = np.tile(mu, _period), np.tile(sigma, _period)
    out = np.empty(flat.shape, np.result_type(x.dtype, mu.dtype, sigma.dtype))
    out[:main].reshape(-1, row)[...] = fn(flat[:main].reshape(-1, row), mu_t,
                                          sg_t)
    if main < flat.size:
        out[main:].reshape(-1, c)[...] = fn(flat[main:].reshape(-1, c), mu,
                                            sigma)
    return out.reshape(x.shape)

class Sup3rGan:
    """Basic sup3r GAN model on MI355X."""

    # tests may install another factory with the same interface; the product
    # default is the HIP engine and nothing else
    _compute_factory = HipGanCompute
    # ``generate`` is the plain normalise -> plan -> un-normalise composition:
    # ``ForwardPass.iter_chunks`` may stack chunks and run the plan itself
    # (subclasses that change ``generate`` switch this off)
    supports_device_chunks = True
    # data-parallel training: size of the gradient buckets handed to RCCL
    # while the backward pass is still running (0 / None: one all-reduce of
    # the whole buffer after it).  xGMI is point-to-point: few, large
    # collectives — 16 MB keeps every link busy and still leaves 9 buckets of
    # the 148 MB discriminator to hide under its backward pass.
    allreduce_bucket_bytes = 16 << 20

    def __init__(self, gen_layers, disc_layers, loss='MeanSquaredError',
                 optimizer=None, learning_rate=1e-4, optimizer_disc=None,
                 learning_rate_disc=None, history=None, meta=None, means=None,
                 stdevs=None, default_device=None, name=None, precision=None):
        self.timer = Timer()
        self.default_device = default_device or 'gpu:0'
        self.name = name if name is not None else self.__class__.__name__
        self._meta = meta if meta is not None else {}
        self.loss_name = loss
        self._loss_terms = parse_loss_spec(loss)
        self._ledger = History(history)
        self._train_window = LossWindow('train_')
        self._val_window = LossWindow('val_')
        last = self._ledger.last_row()
        self._train_window.seed(last)
        self._val_window.seed(last)
        self._optimizer = init_optimizer(optimizer, learning_rate)
        self._optimizer_disc = init_optimizer(
            optimizer_disc or copy.deepcopy(optimizer),
            learning_rate_disc or learning_rate)
        gen_spec, gen_w = self._read_network(gen_layers, 'generator')
        disc_spec, disc_w = self._read_network(disc_layers, 'discriminator')
        self._compute = self._compute_factory(gen_spec, disc_spec,
                                              precision=precision)
        self._gen = self._compute.gen
        self._disc = self._compute.disc
        for net, w in ((self._gen, gen_w), (self._disc, disc_w)):
            if net is not None and w is not None:
                net.set_weights(w)
                net._from_file = True
        self._means = means
        self._stdevs = stdevs
        self.total_batches = 0
        self.virtual_gpus = None     # tests: split every batch like N GPUs would
        self._replicas_synced = False

    # ------------------------------------------------------------- loading
    def _read_network(self, model, role):
        """What ``load_network`` accepts (abstract.py:57-111): a list / dict of
        hidden layers, a ``.json`` config (recorded under
        ``meta['config_<role>']``) or a saved network file.  Returns
        (hidden_layers, weights | None)."""
        if model is None:
            return None, None
        if isinstance(model, (list, dict)):
            return load_hidden_layers(model), None
        if isinstance(model, str) and model.endswith('.json'):
            with open(model) as f:
                cfg = json.load(f)
            self._meta[f'config_{role}'] = cfg
            nested = cfg.get('meta', {}).get(f'config_{role}', {})
            for holder in (cfg, nested):
                if 'hidden_layers' in holder:
                    return holder['hidden_layers'], None
            raise KeyError(
                f'{model}: no "hidden_layers" (nor "meta/config_{role}/'
                f'hidden_layers") among {sorted(cfg)}')
        if isinstance(model, str) and model.endswith('.pkl'):
            from .engine import read_network_file
            hidden, weights, _ = read_network_file(model)
            return hidden, weights
        raise TypeError(f'cannot build the {role} from a '
                        f'{type(model).__name__}')

    @staticmethod
    def seed(s=0):
        """Reproducible weight initialisation (interface.py:59-69)."""
        from .engine import Network
        Network._global_seed = s
        np.random.seed(s)

    # ---------------------------------------------------------- properties
    means = property(lambda self: self._means)
    stdevs = property(lambda self: self._stdevs)
    optimizer = property(lambda self: self._optimizer)
    optimizer_disc = property(lambda self: self._optimizer_disc)
    generator = property(lambda self: self._gen)
    discriminator = property(lambda self: self._disc)
    generator_weights = property(lambda self: self._gen.weights)
    discriminator_weights = property(lambda self: self._disc.weights)
    version_record = property(lambda self: VERSION_RECORD)

    @property
    def history(self):
        return self._ledger.frame

    @property
    def _history(self):            # subclasses written against the old name
        return self._ledger.frame

    @property
    def weights(self):
        return self.generator_weights + self.discriminator_weights

    @property
    def meta(self):
        self._meta.setdefault('class', self.__class__.__name__)
        return self._meta

    @property
    def input_dims(self):
        return self._gen.layers[0].rank or 5

    is_5d = property(lambda self: self.input_dims == 5)
    is_4d = property(lambda self: self.input_dims == 4)

    def _layer_product(self, attr):
        return int(np.prod([getattr(layer, attr, 1)
                            for layer in self._gen.layers]))

    def get_s_enhance_from_layers(self):
        return self._layer_product('_spatial_mult')

    def get_t_enhance_from_layers(self):
        return self._layer_product('_temporal_mult')

    def _enhance(self, key, from_layers):
        if self.meta.get(key) is None:
            self.meta[key] = from_layers()
        return self.meta[key]

    @property
    def s_enhance(self):
        return self._enhance('s_enhance', self.get_s_enhance_from_layers)

    @property
    def t_enhance(self):
        return self._enhance('t_enhance', self.get_t_enhance_from_layers)

    s_enhancements = property(lambda self: [self.s_enhance])
    t_enhancements = property(lambda self: [self.t_enhance])

    @property
    def input_resolution(self):
        res = self.meta.get('input_resolution')
        assert res is not None, 'meta["input_resolution"] has not been set'
        return res

    def _resolution_numbers(self):
        """({'spatial': km, 'temporal': min} of the input, of the output)"""
        fine = {k: _leading_int(v) for k, v in self.input_resolution.items()}
        factor = {'spatial': self.s_enhance, 'temporal': self.t_enhance}
        return fine, {k: v // factor[k] for k, v in fine.items()}

    @property
    def output_resolution(self):
        if self.meta.get('output_resolution') is None and \
                self.meta.get('input_resolution') is not None:
            fine, coarse = self._resolution_numbers()
            self.meta['output_resolution'] = {
                k: v.replace(str(fine[k]), str(coarse[k]))
                for k, v in self.input_resolution.items()}
        return self.meta.get('output_resolution')

    lr_features = property(lambda self: self.meta.get('lr_features', []))
    hr_out_features = property(
        lambda self: self.meta.get('hr_out_features', []))
    smoothing = property(lambda self: self.meta.get('smoothing'))
    smoothed_features = property(
        lambda self: self.meta.get('smoothed_features', []))

    @property
    def obs_features(self):
        seen = []
        for layer in self._gen.layers:
            if layer.cls in OBS_CLASSES:
                seen += [f for f in layer.kwargs.get('features', [layer.name])
                         if f not in seen]
        return seen

    @property
    def hr_exo_features(self):
        names = [layer.name for layer in self._gen.layers
                 if layer.cls in EXO_CLASSES]
        for f in self.obs_features:
            base = f.replace('_obs', '')
            if base not in self.hr_out_features:
                names.append(base)
        return names

    @property
    def hr_features(self):
        return self.hr_out_features + self.hr_exo_features

    @property
    def model_params(self):
        def plain(stats):
            return None if stats is None else \
                {k: float(v) for k, v in stats.items()}
        both = self._means is not None and self._stdevs is not None
        return {'name': self.name, 'loss': self.loss_name,
                'version_record': self.version_record,
                'optimizer': self.get_optimizer_config(self.optimizer),
                'optimizer_disc': self.get_optimizer_config(
                    self.optimizer_disc),
                'means': plain(self._means) if both else self._means,
                'stdevs': plain(self._stdevs) if both else self._stdevs,
                'meta': self.meta, 'default_device': self.default_device}

    # ------------------------------------------------------- normalisation
    def set_norm_stats(self, new_means, new_stdevs):
        """abstract.py:133-195: per-feature float32 statistics."""
        if new_means is None or new_stdevs is None:
            return
        for label, stats in (('means', new_means), ('stdevs', new_stdevs)):
            if not isinstance(stats, dict):
                raise TypeError(f'{label} must map feature name -> value, '
                                f'got {type(stats).__name__}')
        self._means = {k: np.float32(v) for k, v in new_means.items()}
        self._stdevs = {k: np.float32(v) for k, v in new_stdevs.items()}

    def _stats_for(self, features):
        absent = [f for f in features if f not in self._means]
        if absent:
            raise KeyError(f'no normalisation statistics for {absent}; have '
                           f'{sorted(self._means)}')
        return (np.array([self._means[f] for f in features]),
                np.array([self._stdevs[f] for f in features]))

    def norm_input(self, low_res):
        """abstract.py:197-238: (x - mean) / std per lo-res feature."""
        if self._means is None:
            return low_res
        low_res = numpy_if_tensor(low_res)
        mu, sigma = self._stats_for(self.lr_features)
        if (sigma == 0).any():
            warn('a feature has zero standard deviation; dividing by 1')
            sigma = np.where(sigma == 0, 1, sigma)
        return _per_channel(low_res, mu, sigma, lambda x, m, s_: (x - m) / s_)

    def un_norm_output(self, output):
        """abstract.py:240-275: x * std + mean per hi-res output feature."""
        if self._means is None:
            return output
        mu, sigma = self._stats_for(self.hr_out_features)
        return _per_channel(numpy_if_tensor(output), mu, sigma,
                            lambda x, m, s_: x * s_ + m)

    # ------------------------------------------------------------- forward
    def _exo_channels(self, data, wanted, exogenous_data, combine_type):
        """Append the ``combine_type`` exo features that ``wanted`` lists
        beyond the channels ``data`` already has (interface.py:259-356)."""
        if exogenous_data is None:
            return data
        if not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        extra = len(wanted) - data.shape[-1]
        names = wanted[-extra:] if extra > 0 else []
        absent = [f for f in names if f not in exogenous_data]
        assert not absent, (f'exogenous_data lacks {absent} '
                            f'(combine_type "{combine_type}")')
        parts = [data]
        for f in names:
            arr = exogenous_data.get_combine_type_data(f, combine_type)
            if arr is not None:
                parts.append(arr)
        return np.concatenate(parts, axis=-1) if len(parts) > 1 else data

    def _combine_fwp_input(self, low_res, exogenous_data=None):
        return self._exo_channels(low_res, self.lr_features, exogenous_data,
                                  'input')

    def _combine_fwp_output(self, hi_res, exogenous_data=None):
        return self._exo_channels(hi_res, self.hr_out_features,
                                  exogenous_data, 'output')

    def _reshape_norm_exo(self, hr_shape, hi_res_exo, exo_name, norm_in=True):
        """abstract.py:916-979 with the target hi-res shape known up front
        (the plan's exo input shape) instead of the running tensor."""
        if hi_res_exo is None:
            return None
        exo = np.asarray(numpy_if_tensor(hi_res_exo))
        if norm_in and self._means is not None:
            key = exo_name if exo_name in self._means else \
                exo_name.replace('_obs', '')
            exo = (exo.copy() - self._means[key]) / self._stdevs[key]
        if exo_name in self.obs_features:
            # sparse observations: NaN = not observed (with_obs.py:24-29);
            # carried as 0 in normalised units (spec.py, Sup3rConcatObs)
            exo = np.nan_to_num(exo, nan=0.0)
        if exo.ndim == 3:                       # (s1, s2, 1): one per sample
            exo = np.broadcast_to(exo[None], (hr_shape[0],) + exo.shape)
        if exo.ndim == 4 and len(hr_shape) == 5:    # constant in time
            exo = np.broadcast_to(exo[:, :, :, None],
                                  exo.shape[:3] + (hr_shape[3], exo.shape[3]))
        if exo.ndim != len(hr_shape):
            raise RuntimeError(f'exogenous "{exo_name}" of shape {exo.shape} '
                               f'cannot be laid over hi-res {hr_shape}')
        return np.ascontiguousarray(exo)

    def generate(self, low_res, norm_in=True, un_norm_out=True,
                 exogenous_data=None):
        """Public generate (abstract.py:1037-1105): numpy in, numpy out."""
        if exogenous_data is not None and \
                not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        x = self._combine_fwp_input(np.asarray(numpy_if_tensor(low_res)),
                                    exogenous_data)
        if norm_in:
            x = self.norm_input(x)
        x = np.asarray(x, dtype=np.float32)
        try:
            ph = self._gen.plan(x.shape, training=False)
            dev = self._gen.dev
            layer_exo = {}
            for name in ph.input_names:
                if name == 'x':
                    continue
                if name in self.obs_features and (
                        exogenous_data is None or name not in exogenous_data):
                    # the reference runs the obs layer without the field and
                    # the next layer fails on the channel count
                    # (abstract.py:1003-1013 -> :1093-1098)
                    raise RuntimeError(
                        f'exogenous_data is missing observation feature '
                        f'"{name}": the generator cannot run without it')
                assert exogenous_data is not None and \
                    name in exogenous_data, \
                    f'the generator needs exogenous feature "{name}"'
                sh = ph.in_shapes[name]
                view = tuple(sh) if self.is_5d else (sh[0], sh[1], sh[2],
                                                     sh[4])
                arr = self._reshape_norm_exo(
                    view, exogenous_data.get_combine_type_data(name, 'layer'),
                    name, norm_in=norm_in)
                layer_exo[name] = dev.to_device(arr.astype(np.float32))
            hi_res = ph.forward(dev.to_device(x), layer_exo)
        except (AssertionError, RuntimeError):
            raise
        except Exception as e:
            raise RuntimeError(f'generator failed on input of shape '
                               f'{x.shape}: {e}') from e
        hi_res = hi_res.cpu().numpy()
        if un_norm_out:
            hi_res = self.un_norm_output(hi_res)
        return self._combine_fwp_output(hi_res, exogenous_data)

    def _tf_generate(self, low_res, hi_res_exo=None):
        """Normalised low-res in, device tensor out (abstract.py:1131-1173)."""
        try:
            return self._compute.tf_generate(low_res, hi_res_exo)
        except (KeyError, AssertionError):
            raise
        except Exception as e:
            raise RuntimeError('generator failed on input of shape '
                               f'{tuple(np.shape(low_res))}') from e

    def discriminate(self, hi_res, norm_in=False):
        """base.py:237-281: numpy in, numpy logits out."""
        hi_res = np.asarray(numpy_if_tensor(hi_res))
        if norm_in and self._means is not None:
            mu, sigma = self._stats_for(self.hr_out_features)
            hi_res = (hi_res.copy() - mu.astype(np.float32)) \
                / sigma.astype(np.float32)
        return self._tf_discriminate(hi_res).cpu().numpy()

    def _tf_discriminate(self, hi_res):
        """base.py:283-313."""
        try:
            return self._compute.tf_discriminate(hi_res)
        except Exception as e:
            raise RuntimeError('discriminator failed on input of shape '
                               f'{tuple(np.shape(hi_res))}') from e

    # ---------------------------------------------------------------- loss
    def calc_loss(self, hi_res_true, hi_res_gen, weight_gen_advers=0.001,
                  train_gen=True, train_disc=False, compute_disc=False):
        """base.py:830-911 (forward value only; gradients come from
        ``get_single_grad``)."""
        loss, details, _ = self._compute.loss_and_grads(
            None, hi_res_true, self._loss_terms,
            weight_gen_advers=weight_gen_advers, train_gen=train_gen,
            train_disc=train_disc, compute_disc=compute_disc,
            exo_names=self.hr_exo_features, backward=False,
            hi_res_gen=hi_res_gen)
        return loss, details

    def _get_hr_exo_and_loss(self, low_res, hi_res_true, **calc_loss_kwargs):
        """abstract.py:1175-1188."""
        loss, details, hr_gen = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms,
            exo_names=self.hr_exo_features, backward=False,
            **calc_loss_kwargs)
        return loss, details, hr_gen, None

    def get_single_grad(self, low_res, hi_res_true, training_weights=None,
                        device_name=None, **calc_loss_kwargs):
        """abstract.py:1190-1238.  The gradients stay on the device (flat
        gradient buffer of the trained network); returned is the handle name
        ('gen' | 'disc') plus the loss details (or their ``LossFuture`` when
        called with ``defer=True``)."""
        _, details, _ = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms,
            exo_names=self.hr_exo_features, backward=True,
            **calc_loss_kwargs)
        which = 'gen' if calc_loss_kwargs.get('train_gen', True) else 'disc'
        return which, details

    # ------------------------------------------------------- data parallel
    def _replica_layout(self, multi_gpu):
        """(shards this process computes, shards in total).  One process per
        GPU: a rank computes its own shard; ``virtual_gpus`` makes one process
        walk all shards the way the reference's single process drives its GPU
        threads (abstract.py:807-841)."""
        if not multi_gpu:
            return [0], 1
        if self.virtual_gpus and self.virtual_gpus > 1:
            return list(range(self.virtual_gpus)), int(self.virtual_gpus)
        dev = self._gen.dev
        if dev.nranks > 1:
            return [dev.rank], dev.nranks
        return [0], 1

    def _join_replicas(self):
        """``train(multi_gpu=True)`` under torchrun: create this rank's RCCL
        communicator once (WORLD_SIZE > 1)."""
        from . import distributed
        if self.virtual_gpus:
            return
        if self._gen.dev.nranks == 1 and distributed.env_rank()[2] > 1:
            distributed.init_data_parallel()

    def _sync_replicas(self):
        """Replicas must hold identical weights before the first summed step:
        rank 0's are broadcast (weights + Adam slots) once both networks
        exist."""
        if self._replicas_synced or self._gen.dev.nranks == 1 or \
                self.virtual_gpus:
            return
        self._compute.broadcast_state(root=0)
        self._replicas_synced = True

    def run_gradient_descent(self, low_res, hi_res_true, training_weights=None,
                             optimizer=None, multi_gpu=False, defer=False,
                             **calc_loss_kwargs):
        """abstract.py:843-914: gradients of one mini-batch + one optimizer
        step.  ``multi_gpu``: the batch is split on axis 0 into equal shards
        (``tf.split``, :819-825), one per GPU; per-shard gradients are SUMMED
        (:785-805) — across processes by one RCCL all-reduce of the flat
        gradient buffer — and every replica applies the identical step.  The
        reported loss details are the MEAN over shards (the reference reports
        the last GPU's, :792), identical on every rank so that the train /
        skip decisions taken from them cannot diverge.

        ``defer=True`` returns a ``LossFuture`` (no host sync here)."""
        optimizer = optimizer or self.optimizer
        mine, n_shards = self._replica_layout(multi_gpu)
        can_defer = getattr(self._compute, 'supports_defer', False)
        if n_shards == 1:
            if can_defer:
                calc_loss_kwargs['defer'] = True
            which, details = self.get_single_grad(low_res, hi_res_true,
                                                  **calc_loss_kwargs)
        else:
            if not can_defer:
                raise RuntimeError(
                    f'{type(self).__name__} has no sharded gradient path')
            from .distributed import shard_batch
            scal = None
            extra = {k: calc_loss_kwargs.pop(k) for k in ('mask',)
                     if k in calc_loss_kwargs}
            for j, r in enumerate(mine):
                kw = dict(calc_loss_kwargs)
                for k, v in extra.items():
                    kw[k] = None if v is None else shard_batch(v, r, n_shards)
                # (the loss kernels WRITE their scalar slots: one buffer per
                # shard, summed on the device)
                part = self._compute.new_scalars()
                if len(mine) < n_shards and j == len(mine) - 1 and \
                        self.allreduce_bucket_bytes:
                    # the other shards live on other GPUs: the RCCL SUM runs
                    # bucket by bucket under this (last) backward pass
                    kw['overlap_bucket'] = int(self.allreduce_bucket_bytes)
                which, details = self.get_single_grad(
                    shard_batch(low_res, r, n_shards),
                    shard_batch(hi_res_true, r, n_shards), defer=True,
                    scal=part, accumulate_wgrad=j > 0, **kw)
                if scal is None:
                    scal = part
                else:
                    self._compute.add_scalars(scal, part)
                    details._scal = scal
            if len(mine) < n_shards:       # the other shards live elsewhere
                self._compute.allreduce_grads(which)
                self._compute.allreduce_scalars(scal)
            details._scale = 1.0 / n_shards
        self._compute.apply(which, optimizer)
        if isinstance(details, LossFuture) and not defer:
            details = details.resolve()
        return details

    # ------------------------------------------------------------ optimizer
    @staticmethod
    def get_optimizer_config(optimizer):
        return {k: (float(v) if isinstance(v, np.floating) else
                    int(v) if isinstance(v, np.integer) else v)
                for k, v in optimizer.get_config().items()}

    def get_optimizer_state(self, optimizer, net):
        """abstract.py:566-587: learning rate, step count and the mean |slot|
        of every variable (history columns ``Optm*/Adam/{m,v}/...``)."""
        state = {'learning_rate':
                 self.get_optimizer_config(optimizer)['learning_rate']}
        if net is None or not net.built:
            return state
        state['iteration'] = float(optimizer.iterations)
        order = {}
        for i, p in enumerate(net.param_table):
            li = order.setdefault(p['layer'], len(order))
            tag = f'layer{li}/{p["kind"]}'
            state[f'{optimizer.name}/m/{tag}'] = net.mean_abs(_lib.BUF_M, i)
            state[f'{optimizer.name}/v/{tag}'] = net.mean_abs(_lib.BUF_V, i)
        return state

    def update_optimizer(self, option='generator', **kwargs):
        """base.py:326-348: new optimizer settings, step count kept."""
        option = option.lower()
        for key, attr in (('gen', '_optimizer'), ('disc', '_optimizer_disc')):
            if key in option or 'all' in option:
                old = getattr(self, attr)
                conf = dict(self.get_optimizer_config(old), **kwargs)
                new = get_optimizer_class(conf).from_config(conf)
                new.iterations = old.iterations
                setattr(self, attr, new)

    # ------------------------------------------------------------- weights
    def init_weights(self, lr_shape, hr_shape, device=None):
        """base.py:394-437: build both networks for these shapes (glorot
        uniform kernels, zero biases), no-op once built."""
        seed = getattr(type(self._gen), '_global_seed', None)
        if not self._gen.built:
            logger.info('building the generator for lo-res %s', lr_shape)
            self._gen.build(tuple(lr_shape), seed=seed)
        n_out = self._gen.plan(tuple(lr_shape)).out_shape[-1]
        if self.hr_out_features:
            assert n_out == len(self.hr_out_features), (
                f'the generator writes {n_out} features, hr_out_features '
                f'lists {len(self.hr_out_features)}')
        if self._disc is not None and not self._disc.built:
            self._disc.build(tuple(hr_shape),
                             seed=None if seed is None else seed + 1)

    # ----------------------------------------------------------- save/load
    def save_params(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'model_params.json'), 'w') as f:
            json.dump(self.model_params, f, sort_keys=True, indent=2,
                      default=safe_cast)

    def save(self, out_dir):
        """base.py:133-157: model_gen.pkl, model_disc.pkl, history.csv,
        model_params.json."""
        os.makedirs(out_dir, exist_ok=True)
        self.generator.save(os.path.join(out_dir, 'model_gen.pkl'))
        self.discriminator.save(os.path.join(out_dir, 'model_disc.pkl'))
        if self.history is not None:
            self.history.to_csv(os.path.join(out_dir, 'history.csv'))
        self.save_params(out_dir)
        logger.info('checkpoint written to %s', out_dir)

    @staticmethod
    def load_saved_params(out_dir, verbose=True):
        """abstract.py:352-402: constructor kwargs from model_params.json."""
        with open(os.path.join(out_dir, 'model_params.json')) as f:
            params = json.load(f)
        fp = os.path.join(out_dir, 'history.csv')
        params['history'] = fp if os.path.exists(fp) else None
        versions = params.pop('version_record', None)
        if versions is not None and verbose:
            logger.info('model written by:\n%s',
                        pprint.pformat(versions, indent=2))
        if params.get('means') is not None and \
                params.get('stdevs') is not None:
            for key in ('means', 'stdevs'):
                params[key] = {k: np.float32(v)
                               for k, v in params[key].items()}
        return params

    @classmethod
    def _load(cls, model_dir, verbose=True):
        if verbose:
            logger.info('loading %s from %s', cls.__name__, model_dir)
        return (os.path.join(model_dir, 'model_gen.pkl'),
                os.path.join(model_dir, 'model_disc.pkl'),
                cls.load_saved_params(model_dir, verbose=verbose))

    @classmethod
    def load(cls, model_dir, verbose=True):
        fp_gen, fp_disc, params = cls._load(model_dir, verbose=verbose)
        return cls(fp_gen, fp_disc, **params)

    # ---------------------------------------------------- model parameters
    def set_model_params(self, **kwargs):
        """interface.py:453-499: record what the batch handler says about the
        data; a value that contradicts an earlier training run only warns,
        inconsistent enhancement factors / resolutions raise."""
        if 'hr_out_features' in kwargs:
            self.meta['hr_out_features'] = kwargs['hr_out_features']
        given_exo = list(kwargs.get('hr_exo_features') or [])
        assert list(self.hr_exo_features) == given_exo, (
            f'the generator layers imply hi-res exo features '
            f'{self.hr_exo_features}, the data handler provides {given_exo}')
        for key in _META_KEYS:
            if key not in kwargs:
                continue
            have = self.meta.get(key)
            if have is None:
                self.meta[key] = kwargs[key]
            elif have != kwargs[key]:
                warn(f'{key}: trained before with {have}, now given '
                     f'{kwargs[key]}')
        self._ensure_valid_enhancement_factors()
        self._ensure_valid_input_resolution()

    def _ensure_valid_input_resolution(self):
        """interface.py:420-436."""
        if self.meta.get('input_resolution') is None:
            return
        fine, coarse = self._resolution_numbers()
        s, t = self.meta['s_enhance'], self.meta['t_enhance']
        ok = (coarse['spatial'] > 0 and coarse['temporal'] > 0
              and coarse['spatial'] * s == fine['spatial']
              and coarse['temporal'] * t == fine['temporal'])
        if not ok:
            raise RuntimeError(
                f'input resolution {self.input_resolution} is not a multiple '
                f'of the enhancement factors (spatial {s}, temporal {t})')

    def _ensure_valid_enhancement_factors(self):
        """interface.py:438-451."""
        s, t = self.meta.get('s_enhance'), self.meta.get('t_enhance')
        if s is None or t is None:
            return
        ls, lt = (self.get_s_enhance_from_layers(),
                  self.get_t_enhance_from_layers())
        if ls != s and lt != t:
            raise RuntimeError(
                f'the generator layers enhance by (spatial {ls}, temporal '
                f'{lt}) but (spatial {s}, temporal {t}) was requested')

    @staticmethod
    def check_batch_handler_attrs(batch_handler):
        return {k: getattr(batch_handler, k) for k in _HANDLER_KEYS
                if hasattr(batch_handler, k)}

    # -------------------------------------------------------- bookkeeping
    @staticmethod
    def update_loss_details(record, new_data, max_batches, prefix=None):
        """abstract.py:589-622 on a ``LossWindow`` (``record``)."""
        record.resize(max_batches)
        record.push({k: numpy_if_tensor(v) for k, v in new_data.items()})
        return record

    @staticmethod
    def log_loss_details(loss_details, level='INFO'):
        emit = logger.info if level.upper() == 'INFO' else logger.debug
        for k in sorted(loss_details):
            v = loss_details[k]
            emit('\t%s: %s', k, v if isinstance(v, str) else f'{v:.2e}')

    @staticmethod
    def early_stop(history, column, threshold=0.005, n_epoch=5):
        """abstract.py:643-685 on a DataFrame or a ``ledger.History``."""
        table = history if isinstance(history, History) else History(history)
        stop, tail = table.plateaued(column, threshold, n_epoch)
        if stop:
            logger.info('early stop: "%s" moved by %s (< %s) over the last '
                        '%d epochs', column, tail, threshold, n_epoch)
        return stop

    def finish_epoch(self, epoch, epochs, t0, loss_details, checkpoint_int,
                     out_dir, early_stop_on, early_stop_threshold,
                     early_stop_n_epoch, extras=None):
        """abstract.py:698-783: history row, periodic / final checkpoint,
        early-stop test.  Returns True when training should stop."""
        self.log_loss_details(loss_details)
        row = {'elapsed_time': time.time() - t0}
        row.update({k: float(v) for k, v in loss_details.items()})
        row.update({k: safe_cast(v) for k, v in (extras or {}).items()})
        self._ledger.write(epoch, row)
        due = checkpoint_int is not None and epoch % checkpoint_int == 0
        stop = early_stop_on is not None and self.early_stop(
            self._ledger, early_stop_on, threshold=early_stop_threshold,
            n_epoch=early_stop_n_epoch)
        if due or stop or epoch == epochs[-1]:
            assert '{epoch}' in out_dir, \
                f'out_dir needs an "{{epoch}}" field for checkpoints: {out_dir}'
            self.save(out_dir.format(epoch=epoch))
        return stop

    @staticmethod
    def get_weight_update_fraction(history, comparison_key,
                                   update_bounds=(0.5, 0.95), update_frac=0.0):
        """base.py:439-476: > 1 to grow, < 1 to shrink, 1 inside the bounds."""
        val = history[comparison_key]
        if isinstance(val, (list, tuple, np.ndarray)):
            val = val[-1]
        lo, hi = update_bounds
        if val < lo:
            return 1 + update_frac
        return 1 / (1 + update_frac) if val > hi else 1

    def update_adversarial_weights(self, history, adaptive_update_fraction,
                                   adaptive_update_bounds, weight_gen_advers,
                                   train_disc):
        """base.py:551-606: the adversarial weight follows the fraction of
        batches the discriminator had to train."""
        if adaptive_update_fraction <= 0 or not train_disc:
            return weight_gen_advers
        factor = self.get_weight_update_fraction(
            history, 'disc_train_frac', update_frac=adaptive_update_fraction,
            update_bounds=adaptive_update_bounds)
        if factor != 1:
            logger.debug('adversarial weight %.4e -> %.4e', weight_gen_advers,
                         weight_gen_advers * factor)
        return weight_gen_advers * factor

    # --------------------------------------------------------------- train
    def calc_val_loss(self, batch_handler, weight_gen_advers):
        """base.py:913-942: running validation means after the epoch."""
        n_val = len(batch_handler.val_data)
        for batch in batch_handler.val_data:
            _, details, _, _ = self._get_hr_exo_and_loss(
                batch.low_res, batch.high_res,
                weight_gen_advers=weight_gen_advers)
            self.update_loss_details(self._val_window, details, n_val)
        return self._val_window.means()

    def _launch_batch(self, batch, train_gen, only_gen, gen_too_good,
                      train_disc, only_disc, disc_too_good, weight_gen_advers,
                      multi_gpu=False):
        """Enqueue the generator and / or discriminator step of one
        mini-batch (base.py:944-1031) without reading anything back.
        Returns ([LossFuture | dict, ...], trained_gen, trained_disc)."""
        do_gen = only_gen or (train_gen and not gen_too_good)
        do_disc = only_disc or (train_disc and not disc_too_good)
        scope = getattr(self._compute, 'batch_scope', None)

        def body(batch):
            steps = []
            if scope is not None:
                scope(True)
            try:
                if do_gen:
                    steps.append(self.run_gradient_descent(
                        batch.low_res, batch.high_res, None,
                        optimizer=self.optimizer,
                        weight_gen_advers=weight_gen_advers, train_gen=True,
                        train_disc=False, compute_disc=train_disc,
                        multi_gpu=multi_gpu, defer=True))
                if do_disc:
                    steps.append(self.run_gradient_descent(
                        batch.low_res, batch.high_res, None,
                        optimizer=self.optimizer_disc,
                        weight_gen_advers=weight_gen_advers, train_gen=False,
                        train_disc=True, multi_gpu=multi_gpu, defer=True))
            finally:
                if scope is not None:
                    scope(False)
            return steps
        rec = self._step_recorder(batch, multi_gpu) if (do_gen or do_disc) \
            else None
        if rec is not None:
            # a launch-bound step is recorded once and replayed as one graph
            # launch (captured.py)
            opts = ([self.optimizer] if do_gen else []) + \
                ([self.optimizer_disc] if do_disc else [])
            steps = rec.run(batch, (do_gen, do_disc, bool(train_disc),
                                    float(weight_gen_advers)), opts, body,
                            model=self)
        else:
            # one upload per mini-batch: both steps read the same device
            # tensors (and the discriminator's pass over the true field is
            # shared)
            steps = body(self._resident(batch))
        return steps, do_gen, do_disc

    # 'auto': record the launch sequence of small (launch-bound) mini-batches
    # and replay it as one hipGraphLaunch; True: whenever possible; False: never
    capture_steps = 'auto'

    def _step_recorder(self, batch, multi_gpu):
        mode = self.capture_steps
        # (multi_gpu on a single rank without virtual shards is the plain step)
        if not mode or self._replica_layout(multi_gpu)[1] != 1:
            return None
        from .compute import HipGanCompute
        own = (type(self).get_single_grad is Sup3rGan.get_single_grad
               and type(self).run_gradient_descent
               is Sup3rGan.run_gradient_descent
               and type(self._compute) is HipGanCompute)
        if not own:
            return None
        rec = getattr(self, '_recorder', None)
        if rec is None or rec.compute is not self._compute:
            from .captured import StepRecorder
            rec = self._recorder = StepRecorder(self._compute)
        return rec if rec.eligible(self, batch, mode, False) else None

    def _resident(self, batch):
        dev = getattr(self._gen, 'dev', None)
        if dev is None or not hasattr(dev, 'to_device'):
            return batch

        class Resident:
            low_res = dev.to_device(batch.low_res)
            high_res = dev.to_device(batch.high_res)
        return Resident

    @staticmethod
    def _settle(steps, trained_gen, trained_disc):
        details = {}
        for s in steps:
            details.update(s.resolve() if isinstance(s, LossFuture) else s)
        details = {k: float(v) for k, v in details.items()}
        details['gen_train_frac'] = float(trained_gen)
        details['disc_train_frac'] = float(trained_disc)
        return details

    def _train_batch(self, batch, train_gen, only_gen, gen_too_good,
                     train_disc, only_disc, disc_too_good, weight_gen_advers,
                     multi_gpu=False):
        """base.py:944-1031: one mini-batch, loss details as floats."""
        return self._settle(*self._launch_batch(
            batch, train_gen, only_gen, gen_too_good, train_disc, only_disc,
            disc_too_good, weight_gen_advers, multi_gpu))

    def _post_batch(self, ib, b_loss_details, n_batches, previous_means):
        """base.py:1033-1095: fold one mini-batch into the running means."""
        self._train_window.resize(n_batches)
        self._train_window.push(b_loss_details, carry=previous_means)
        if not b_loss_details['gen_train_frac'] and \
                not b_loss_details['disc_train_frac']:
            warn(f'neither network trained on batch {ib} of {n_batches}')
        return self._train_window.means()

    def _train_epoch(self, batch_handler, weight_gen_advers, train_gen,
                     train_disc, disc_loss_bounds, multi_gpu=False):
        """base.py:1097-1191.  The discriminator sits a batch out while its
        running loss is at or below the lower bound, the generator while it is
        above the upper one.  Only that gating needs a per-batch read of the
        loss scalars; when one network trains alone the futures of the whole
        epoch are read back together at its end."""
        lr_shape, hr_shape = batch_handler.shapes
        self.init_weights(lr_shape, hr_shape)
        self._sync_replicas()
        lo, hi = float(np.min(disc_loss_bounds)), float(np.max(disc_loss_bounds))
        n_batches = len(batch_handler)
        means = self._train_window.means()
        only_gen = train_gen and not train_disc
        only_disc = train_disc and not train_gen
        gated = train_gen and train_disc
        backlog = []
        for ib, batch in enumerate(batch_handler):
            t_batch = time.time()
            loss_disc = means.get('train_loss_disc', 0)
            disc_too_good = loss_disc <= lo
            gen_too_good = train_disc and loss_disc > hi
            launched = self._launch_batch(
                batch, train_gen, only_gen, gen_too_good, train_disc,
                only_disc, disc_too_good, weight_gen_advers, multi_gpu)
            if gated:
                means = self._post_batch(ib, self._settle(*launched),
                                         n_batches, means)
            else:
                backlog.append(launched)
            logger.debug('batch %d / %d enqueued in %.4f s', ib + 1,
                         n_batches, time.time() - t_batch)
        for ib, launched in enumerate(backlog):
            means = self._post_batch(ib, self._settle(*launched), n_batches,
                                     means)
        self.total_batches += n_batches
        out = self._train_window.means()
        out['total_batches'] = int(self.total_batches)
        return out

    def train(self, batch_handler, input_resolution, n_epoch,
              weight_gen_advers=0.001, train_gen=True, train_disc=True,
              disc_loss_bounds=(0.45, 0.6), checkpoint_int=None,
              out_dir='./gan_{epoch}', early_stop_on=None,
              early_stop_threshold=0.005, early_stop_n_epoch=5,
              adaptive_update_bounds=(0.9, 0.99), adaptive_update_fraction=0.0,
              multi_gpu=False, tensorboard_log=False,
              tensorboard_profile=False):
        """base.py:624-828 (the tensorboard options are accepted and ignored:
        profiling on MI355X goes through rocprofv3, see tools/)."""
        if multi_gpu:
            self._join_replicas()
        self.set_norm_stats(batch_handler.means, batch_handler.stds)
        self.set_model_params(
            input_resolution=input_resolution,
            s_enhance=batch_handler.s_enhance,
            t_enhance=batch_handler.t_enhance,
            **self.check_batch_handler_attrs(batch_handler))
        epochs = self._ledger.next_epochs(n_epoch)
        t0 = time.time()
        logger.info('training %d epochs from epoch %d, adversarial weight %s',
                    n_epoch, epochs[0], weight_gen_advers)
        for epoch in epochs:
            t_epoch = time.time()
            summary = self._train_epoch(
                batch_handler, weight_gen_advers, train_gen, train_disc,
                disc_loss_bounds, multi_gpu=multi_gpu)
            summary.update(self.calc_val_loss(batch_handler,
                                              weight_gen_advers))
            logger.info('epoch %d / %d: ' + ', '.join(
                f'{k} {summary[k]:.2e}' for k in (
                    'train_loss_gen', 'train_loss_disc', 'val_loss_gen',
                    'val_loss_disc') if k in summary), epoch, epochs[-1])
            extras = {'weight_gen_advers': weight_gen_advers,
                      'disc_loss_bound_0': disc_loss_bounds[0],
                      'disc_loss_bound_1': disc_loss_bounds[1]}
            for tag, opt, net in (('OptmGen', self.optimizer, self._gen),
                                  ('OptmDisc', self.optimizer_disc,
                                   self._disc)):
                extras.update({f'{tag}/{k}': v for k, v in
                               self.get_optimizer_state(opt, net).items()})
            weight_gen_advers = self.update_adversarial_weights(
                summary, adaptive_update_fraction, adaptive_update_bounds,
                weight_gen_advers, train_disc)
            stop = self.finish_epoch(
                epoch, epochs, t0, summary, checkpoint_int, out_dir,
                early_stop_on, early_stop_threshold, early_stop_n_epoch,
                extras=extras)
            logger.info('epoch %d took %.2f s', epoch, time.time() - t_epoch)
            if stop:
                break
        logger.info('%d epochs in %.2f s', n_epoch, time.time() - t0)
        batch_handler.stop()
