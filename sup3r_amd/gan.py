"""``Sup3rGan`` with the reference's Python surface, computing on MI355X.

Drop-in for ``sup3r.models.Sup3rGan`` on the hot path: same constructor, the
same ``generate / discriminate / calc_loss / train / save / load`` methods and
bookkeeping (history schema, meta keys, model_params.json, checkpoint layout),
so ``ForwardPassStrategy`` / ``ForwardPass`` and the ``BatchHandler`` protocol
drive it unchanged.  Reference being mirrored (behaviour, not code):
sup3r/models/base.py (Sup3rGan), sup3r/models/abstract.py
(AbstractSingleModel), sup3r/models/interface.py (AbstractInterface).

All arithmetic is delegated to ``compute.HipGanCompute`` (libsup3r_hip.so);
this module is host control logic only.
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
import pandas as pd

from . import __version__
from . import _lib
from .compute import HipGanCompute, parse_loss_spec
from .optimizers import get_optimizer_class, init_optimizer
from .spec import EXO_CLASSES, OBS_CLASSES, load_hidden_layers
from .utilities import (ExoData, LossValue, Timer, numpy_if_tensor, safe_cast)

logger = logging.getLogger(__name__)

VERSION_RECORD = {'sup3r_amd': __version__, 'backend': 'libsup3r_hip (gfx950)'}


class Sup3rGan:
    """Basic sup3r GAN model on MI355X."""

    # tests may install another factory with the same interface; the product
    # default is the HIP engine and nothing else
    _compute_factory = HipGanCompute

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
        self._history = history
        if isinstance(self._history, str):
            self._history = pd.read_csv(self._history, index_col=0)
        self._train_record = pd.DataFrame()
        self._val_record = pd.DataFrame()
        self._init_records()
        optimizer_disc = optimizer_disc or copy.deepcopy(optimizer)
        learning_rate_disc = learning_rate_disc or learning_rate
        self._optimizer = init_optimizer(optimizer, learning_rate)
        self._optimizer_disc = init_optimizer(optimizer_disc,
                                              learning_rate_disc)
        gen_spec, gen_w = self._load_network_spec(gen_layers, 'generator')
        disc_spec, disc_w = self._load_network_spec(disc_layers,
                                                    'discriminator')
        self._compute = self._compute_factory(gen_spec, disc_spec,
                                              precision=precision)
        self._gen = self._compute.gen
        self._disc = self._compute.disc
        if gen_w is not None:
            self._gen.set_weights(gen_w)
        if disc_w is not None and self._disc is not None:
            self._disc.set_weights(disc_w)
        self._means = means
        self._stdevs = stdevs
        self.total_batches = 0
        self._tb_writer = None

    # ------------------------------------------------------------- loading
    def _load_network_spec(self, model, name):
        """load_network (abstract.py:57-111): hidden-layer list, .json config
        or saved network file -> (hidden_layers, weights | None)."""
        if model is None:
            return None, None
        if isinstance(model, str) and model.endswith('.json'):
            with open(model) as f:
                cfg = json.load(f)
            self._meta[f'config_{name}'] = cfg
            if 'hidden_layers' in cfg:
                return cfg['hidden_layers'], None
            if 'meta' in cfg and f'config_{name}' in cfg['meta'] and \
                    'hidden_layers' in cfg['meta'][f'config_{name}']:
                return cfg['meta'][f'config_{name}']['hidden_layers'], None
            msg = ('Could not load model from json config, need '
                   '"hidden_layers" key or "meta/config_{}/hidden_layers" '
                   ' at top level but only found: {}'.format(name,
                                                             cfg.keys()))
            logger.error(msg)
            raise KeyError(msg)
        if isinstance(model, str) and model.endswith('.pkl'):
            import pickle
            with open(model, 'rb') as f:
                d = pickle.load(f)
            if not isinstance(d, dict) or \
                    d.get('format') != 'sup3r_amd.network.v1':
                raise TypeError(
                    'Something went wrong. Tried to load a custom network '
                    f'but "{model}" is not a sup3r_amd network file')
            return d['hidden_layers'], (d['weights'] or None)
        if isinstance(model, (list, dict)):
            return load_hidden_layers(model), None
        msg = ('Something went wrong. Tried to load a custom network but '
               'ended up with a model of type "{}"'.format(type(model)))
        logger.error(msg)
        raise TypeError(msg)

    @staticmethod
    def seed(s=0):
        """Reproducible weight initialisation (interface.py:59-69)."""
        from .engine import Network
        Network._global_seed = s
        np.random.seed(s)

    # ---------------------------------------------------------- properties
    @property
    def means(self):
        return self._means

    @property
    def stdevs(self):
        return self._stdevs

    @property
    def optimizer(self):
        return self._optimizer

    @property
    def optimizer_disc(self):
        return self._optimizer_disc

    @property
    def history(self):
        return self._history

    @property
    def generator(self):
        return self._gen

    @property
    def discriminator(self):
        return self._disc

    @property
    def generator_weights(self):
        return self._gen.weights

    @property
    def discriminator_weights(self):
        return self._disc.weights

    @property
    def weights(self):
        return self.generator_weights + self.discriminator_weights

    @property
    def meta(self):
        if 'class' not in self._meta:
            self._meta['class'] = self.__class__.__name__
        return self._meta

    @property
    def version_record(self):
        return VERSION_RECORD

    @property
    def input_dims(self):
        return self._gen.layers[0].rank or 5

    @property
    def is_5d(self):
        return self.input_dims == 5

    @property
    def is_4d(self):
        return self.input_dims == 4

    def get_s_enhance_from_layers(self):
        return int(np.prod([getattr(layer, '_spatial_mult', 1)
                            for layer in self._gen.layers]))

    def get_t_enhance_from_layers(self):
        return int(np.prod([getattr(layer, '_temporal_mult', 1)
                            for layer in self._gen.layers]))

    @property
    def s_enhance(self):
        s = self.meta.get('s_enhance', None)
        if s is None:
            s = self.get_s_enhance_from_layers()
        self.meta['s_enhance'] = s
        return s

    @property
    def t_enhance(self):
        t = self.meta.get('t_enhance', None)
        if t is None:
            t = self.get_t_enhance_from_layers()
        self.meta['t_enhance'] = t
        return t

    @property
    def s_enhancements(self):
        return [self.s_enhance]

    @property
    def t_enhancements(self):
        return [self.t_enhance]

    @property
    def input_resolution(self):
        res = self.meta.get('input_resolution', None)
        assert res is not None, \
            'model.input_resolution is None. This needs to be set.'
        return res

    def _get_numerical_resolutions(self):
        ires = {k: int(re.search(r'\d+', v).group(0))
                for k, v in self.input_resolution.items()}
        enh = {'spatial': self.s_enhance, 'temporal': self.t_enhance}
        return ires, {k: v // enh[k] for k, v in ires.items()}

    @property
    def output_resolution(self):
        out = self.meta.get('output_resolution', None)
        if self.meta.get('input_resolution') is not None and out is None:
            ires, ores = self._get_numerical_resolutions()
            out = {k: v.replace(str(ires[k]), str(ores[k]))
                   for k, v in self.input_resolution.items()}
            self.meta['output_resolution'] = out
        return out

    @property
    def lr_features(self):
        return self.meta.get('lr_features', [])

    @property
    def hr_out_features(self):
        return self.meta.get('hr_out_features', [])

    @property
    def obs_features(self):
        feats = []
        for layer in self._gen.layers:
            if layer.cls in OBS_CLASSES:
                for f in layer.kwargs.get('features', [layer.name]):
                    if f not in feats:
                        feats.append(f)
        return feats

    @property
    def hr_exo_features(self):
        feats = [layer.name for layer in self._gen.layers
                 if layer.cls in EXO_CLASSES]
        feats += [f.replace('_obs', '') for f in self.obs_features
                  if f.replace('_obs', '') not in self.hr_out_features]
        return feats

    @property
    def hr_features(self):
        return self.hr_out_features + self.hr_exo_features

    @property
    def smoothing(self):
        return self.meta.get('smoothing', None)

    @property
    def smoothed_features(self):
        return self.meta.get('smoothed_features', [])

    @property
    def model_params(self):
        means, stdevs = self._means, self._stdevs
        if means is not None and stdevs is not None:
            means = {k: float(v) for k, v in means.items()}
            stdevs = {k: float(v) for k, v in stdevs.items()}
        return {'name': self.name, 'loss': self.loss_name,
                'version_record': self.version_record,
                'optimizer': self.get_optimizer_config(self.optimizer),
                'optimizer_disc': self.get_optimizer_config(
                    self.optimizer_disc),
                'means': means, 'stdevs': stdevs, 'meta': self.meta,
                'default_device': self.default_device}

    # ------------------------------------------------------- normalisation
    def set_norm_stats(self, new_means, new_stdevs):
        if new_means is not None and new_stdevs is not None:
            logger.info('Setting new normalization statistics...')
            if not isinstance(new_means, dict) or \
                    not isinstance(new_stdevs, dict):
                msg = ('Means and stdevs need to be dictionaries with keys as '
                       'feature names but received means of type '
                       f'{type(new_means)} and stdevs of type '
                       f'{type(new_stdevs)}')
                logger.error(msg)
                raise TypeError(msg)
            self._means = {k: np.float32(v) for k, v in new_means.items()}
            self._stdevs = {k: np.float32(v) for k, v in new_stdevs.items()}

    def _stats_for(self, features):
        missing = [f for f in features if f not in self._means]
        if any(missing):
            msg = (f'Could not find features {missing} in means/stdevs: '
                   f'{self._means}/{self._stdevs}')
            logger.error(msg)
            raise KeyError(msg)
        means = np.array([self._means[f] for f in features])
        stdevs = np.array([self._stdevs[f] for f in features])
        return means, stdevs

    def norm_input(self, low_res):
        if self._means is not None:
            low_res = numpy_if_tensor(low_res)
            means, stdevs = self._stats_for(self.lr_features)
            if any(stdevs == 0):
                stdevs = np.where(stdevs == 0, 1, stdevs)
                msg = 'Some standard deviations are zero.'
                logger.warning(msg)
                warn(msg)
            low_res = (low_res.copy() - means) / stdevs
        return low_res

    def un_norm_output(self, output):
        if self._means is not None:
            output = numpy_if_tensor(output)
            means, stdevs = self._stats_for(self.hr_out_features)
            output = (output * stdevs) + means
        return output

    # ------------------------------------------------------------- forward
    def _combine_fwp_input(self, low_res, exogenous_data=None):
        if exogenous_data is None:
            return low_res
        if not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        fnum_diff = len(self.lr_features) - low_res.shape[-1]
        exo_feats = [] if fnum_diff <= 0 else self.lr_features[-fnum_diff:]
        assert all(f in exogenous_data for f in exo_feats), (
            f'Provided exogenous_data: {exogenous_data} is missing some '
            f'required features ({exo_feats})')
        for feature in exo_feats:
            exo_input = exogenous_data.get_combine_type_data(feature, 'input')
            if exo_input is not None:
                low_res = np.concatenate((low_res, exo_input), axis=-1)
        return low_res

    def _combine_fwp_output(self, hi_res, exogenous_data=None):
        if exogenous_data is None:
            return hi_res
        if not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        fnum_diff = len(self.hr_out_features) - hi_res.shape[-1]
        exo_feats = [] if fnum_diff <= 0 else self.hr_out_features[-fnum_diff:]
        assert all(f in exogenous_data for f in exo_feats), (
            f'Provided exogenous_data is missing some required features '
            f'({exo_feats})')
        for feature in exo_feats:
            exo_output = exogenous_data.get_combine_type_data(feature,
                                                              'output')
            if exo_output is not None:
                hi_res = np.concatenate((hi_res, exo_output), axis=-1)
        return hi_res

    def _reshape_norm_exo(self, hr_shape, hi_res_exo, exo_name, norm_in=True):
        """abstract.py:916-979 with the target hi-res shape known up front
        (the plan's exo input shape) instead of the running tensor."""
        if hi_res_exo is None:
            return hi_res_exo
        hi_res_exo = np.asarray(numpy_if_tensor(hi_res_exo))
        if norm_in and self._means is not None:
            key = exo_name if exo_name in self._means else \
                exo_name.replace('_obs', '')
            hi_res_exo = (hi_res_exo.copy() - self._means[key]) \
                / self._stdevs[key]
        if hi_res_exo.ndim == 3:
            hi_res_exo = np.repeat(hi_res_exo[None], hr_shape[0], axis=0)
        if hi_res_exo.ndim == 4 and len(hr_shape) == 5:
            hi_res_exo = np.repeat(np.expand_dims(hi_res_exo, 3),
                                   hr_shape[3], axis=3)
        if hi_res_exo.ndim != len(hr_shape):
            msg = ('hi_res and hi_res_exo arrays are not of the same rank: '
                   '{} and {}'.format(hr_shape, hi_res_exo.shape))
            logger.error(msg)
            raise RuntimeError(msg)
        return hi_res_exo

    def generate(self, low_res, norm_in=True, un_norm_out=True,
                 exogenous_data=None):
        """Public generate (abstract.py:1037-1105): numpy in, numpy out."""
        if exogenous_data is not None and \
                not isinstance(exogenous_data, ExoData):
            exogenous_data = ExoData(exogenous_data)
        low_res = self._combine_fwp_input(np.asarray(
            numpy_if_tensor(low_res)), exogenous_data)
        if norm_in and self._means is not None:
            low_res = self.norm_input(low_res)
        low_res = np.asarray(low_res, dtype=np.float32)
        try:
            ph = self._gen.plan(low_res.shape, training=False)
            hr_exo = {}
            for name in ph.input_names:
                if name == 'x':
                    continue
                msg = f'exogenous_data is missing required feature "{name}"'
                assert exogenous_data is not None and \
                    name in exogenous_data, msg
                exo = exogenous_data.get_combine_type_data(name, 'layer')
                sh = ph.in_shapes[name]
                hr_shape = tuple(sh) if self.is_5d else (
                    sh[0], sh[1], sh[2], sh[4])
                hr_exo[name] = self._reshape_norm_exo(
                    hr_shape, exo, name, norm_in=norm_in).astype(np.float32)
            dev = self._gen.dev
            hi_res = ph.forward(dev.to_device(low_res),
                                {k: dev.to_device(v)
                                 for k, v in hr_exo.items()})
        except AssertionError:
            raise
        except Exception as e:
            msg = ('Could not run the generator on tensor of shape {}: {}'
                   .format(low_res.shape, e))
            logger.error(msg)
            raise RuntimeError(msg) from e
        hi_res = hi_res.cpu().numpy()
        if un_norm_out and self._means is not None:
            hi_res = self.un_norm_output(hi_res)
        return self._combine_fwp_output(hi_res, exogenous_data)

    def _tf_generate(self, low_res, hi_res_exo=None):
        """Normalised low-res in, device tensor out (abstract.py:1131-1173)."""
        try:
            return self._compute.tf_generate(low_res, hi_res_exo)
        except (KeyError, AssertionError):
            raise
        except Exception as e:
            msg = 'Could not run the generator on tensor of shape {}'.format(
                tuple(np.shape(low_res)))
            logger.error(msg)
            raise RuntimeError(msg) from e

    def discriminate(self, hi_res, norm_in=False):
        """base.py:237-281: numpy in, numpy logits out."""
        hi_res = np.asarray(numpy_if_tensor(hi_res))
        if norm_in and self._means is not None:
            mean, std = self._stats_for(self.hr_out_features)
            hi_res = (hi_res.copy() - mean.astype(np.float32)) \
                / std.astype(np.float32)
        return self._tf_discriminate(hi_res).cpu().numpy()

    def _tf_discriminate(self, hi_res):
        try:
            return self._compute.tf_discriminate(hi_res)
        except Exception as e:
            msg = ('Could not run the discriminator on tensor of shape {}'
                   .format(tuple(np.shape(hi_res))))
            logger.error(msg)
            raise RuntimeError(msg) from e

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
        loss, details, hr_gen = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms,
            exo_names=self.hr_exo_features, backward=False,
            **calc_loss_kwargs)
        return loss, details, hr_gen, None

    def get_single_grad(self, low_res, hi_res_true, training_weights=None,
                        device_name=None, **calc_loss_kwargs):
        """abstract.py:1190-1238.  The gradients stay on the device (flat
        gradient buffer of the trained network); returned is the handle name
        ('gen' | 'disc') plus the loss details."""
        _, details, _ = self._compute.loss_and_grads(
            low_res, hi_res_true, self._loss_terms,
            exo_names=self.hr_exo_features, backward=True,
            **calc_loss_kwargs)
        which = 'gen' if calc_loss_kwargs.get('train_gen', True) else 'disc'
        return which, details

    def run_gradient_descent(self, low_res, hi_res_true, training_weights=None,
                             optimizer=None, multi_gpu=False,
                             **calc_loss_kwargs):
        """abstract.py:843-914.  ``multi_gpu``: this process holds 1/N of the
        mini-batch; per-rank gradients are SUMMED by one RCCL all-reduce
        (the reference sums per-GPU gradient lists on the host,
        abstract.py:785-805) before the identical Adam step on every rank."""
        if optimizer is None:
            optimizer = self.optimizer
        start = time.time()
        which, details = self.get_single_grad(low_res, hi_res_true,
                                              **calc_loss_kwargs)
        if multi_gpu and self._gen.dev.nranks > 1:
            self._compute.allreduce_grads(which)
        self._compute.apply(which, optimizer)
        logger.debug('Finished single gradient descent step in %.4f seconds',
                     time.time() - start)
        return details

    # ------------------------------------------------------------ optimizer
    @staticmethod
    def get_optimizer_config(optimizer):
        conf = optimizer.get_config()
        for k, v in conf.items():
            if isinstance(v, np.floating):
                conf[k] = float(v)
            elif isinstance(v, np.integer):
                conf[k] = int(v)
        return conf

    def get_optimizer_state(self, optimizer, net):
        """abstract.py:566-587: learning rate + mean |slot| per variable."""
        state = {'learning_rate':
                 self.get_optimizer_config(optimizer)['learning_rate']}
        if net is None or not net.built:
            return state
        state['iteration'] = float(optimizer.iterations)
        layer_ids = {}
        for i, p in enumerate(net.param_table):
            li = layer_ids.setdefault(p['layer'], len(layer_ids))
            vname = f'layer{li}/{p["kind"]}'
            state[f'{optimizer.name}/m/{vname}'] = net.mean_abs(_lib.BUF_M, i)
            state[f'{optimizer.name}/v/{vname}'] = net.mean_abs(_lib.BUF_V, i)
        return state

    def update_optimizer(self, option='generator', **kwargs):
        if 'gen' in option.lower() or 'all' in option.lower():
            conf = self.get_optimizer_config(self.optimizer)
            conf.update(**kwargs)
            it = self._optimizer.iterations
            self._optimizer = get_optimizer_class(conf).from_config(conf)
            self._optimizer.iterations = it
        if 'disc' in option.lower() or 'all' in option.lower():
            conf = self.get_optimizer_config(self.optimizer_disc)
            conf.update(**kwargs)
            it = self._optimizer_disc.iterations
            self._optimizer_disc = get_optimizer_class(conf).from_config(conf)
            self._optimizer_disc.iterations = it

    # ------------------------------------------------------------- weights
    def init_weights(self, lr_shape, hr_shape, device=None):
        """base.py:394-437: build both networks for these shapes (glorot
        uniform kernels, zero biases), no-op once built."""
        if not self._gen.built:
            logger.info('Initializing model weights on the MI355X')
            seed = getattr(type(self._gen), '_global_seed', None)
            self._gen.build(tuple(lr_shape), seed=seed)
        out_shape = self._gen.plan(tuple(lr_shape)).out_shape
        msg = (f'Number of model outputs {out_shape[-1]} does not match the '
               'number of computed hr_out_features '
               f'{len(self.hr_out_features)}')
        if self.hr_out_features:
            assert out_shape[-1] == len(self.hr_out_features), msg
        if self._disc is not None and not self._disc.built:
            seed = getattr(type(self._disc), '_global_seed', None)
            self._disc.build(tuple(hr_shape),
                             seed=None if seed is None else seed + 1)

    # ----------------------------------------------------------- save/load
    def save_params(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'model_params.json'), 'w') as f:
            json.dump(self.model_params, f, sort_keys=True, indent=2,
                      default=safe_cast)

    def save(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        self.generator.save(os.path.join(out_dir, 'model_gen.pkl'))
        self.discriminator.save(os.path.join(out_dir, 'model_disc.pkl'))
        if isinstance(self.history, pd.DataFrame):
            self.history.to_csv(os.path.join(out_dir, 'history.csv'))
        self.save_params(out_dir)
        logger.info('Saved GAN to disk in directory: {}'.format(out_dir))

    @staticmethod
    def load_saved_params(out_dir, verbose=True):
        with open(os.path.join(out_dir, 'model_params.json')) as f:
            params = json.load(f)
        fp_history = os.path.join(out_dir, 'history.csv')
        params['history'] = fp_history if os.path.exists(fp_history) else None
        if 'version_record' in params:
            version_record = params.pop('version_record')
            if verbose:
                logger.info('Loading model from disk that was created with '
                            'the following package versions: \n{}'.format(
                                pprint.pformat(version_record, indent=2)))
        means, stdevs = params.get('means'), params.get('stdevs')
        if means is not None and stdevs is not None:
            params['means'] = {k: np.float32(v) for k, v in means.items()}
            params['stdevs'] = {k: np.float32(v) for k, v in stdevs.items()}
        return params

    @classmethod
    def _load(cls, model_dir, verbose=True):
        if verbose:
            logger.info('Loading GAN from disk in directory: {}'.format(
                model_dir))
        fp_gen = os.path.join(model_dir, 'model_gen.pkl')
        fp_disc = os.path.join(model_dir, 'model_disc.pkl')
        return fp_gen, fp_disc, cls.load_saved_params(model_dir,
                                                      verbose=verbose)

    @classmethod
    def load(cls, model_dir, verbose=True):
        fp_gen, fp_disc, params = cls._load(model_dir, verbose=verbose)
        return cls(fp_gen, fp_disc, **params)

    # ---------------------------------------------------- model parameters
    def set_model_params(self, **kwargs):
        """interface.py:453-499."""
        keys = ('input_resolution', 'lr_features', 'hr_exo_features',
                'hr_out_features', 'smoothed_features', 's_enhance',
                't_enhance', 'smoothing')
        keys = [k for k in keys if k in kwargs]
        if 'hr_out_features' in kwargs:
            self.meta['hr_out_features'] = kwargs['hr_out_features']
        hr_exo_feat = kwargs.get('hr_exo_features', []) or []
        msg = (f'Expected high-res exo features {self.hr_exo_features} based '
               'on model architecture but received "hr_exo_features" from '
               f'data handler: {hr_exo_feat}')
        assert list(self.hr_exo_features) == list(hr_exo_feat), msg
        for var in keys:
            val = self.meta.get(var, None)
            if val is None:
                self.meta[var] = kwargs[var]
            elif val != kwargs[var]:
                msg = ('Model was previously trained with {var}={} but '
                       'received new {var}={}'.format(val, kwargs[var],
                                                      var=var))
                logger.warning(msg)
                warn(msg)
        self._ensure_valid_enhancement_factors()
        self._ensure_valid_input_resolution()

    def _ensure_valid_input_resolution(self):
        if self.meta.get('input_resolution') is None:
            return
        ires, ores = self._get_numerical_resolutions()
        s_enhance, t_enhance = self.meta['s_enhance'], self.meta['t_enhance']
        check = (ores['temporal'] > 0 and ores['spatial'] > 0
                 and ires['temporal'] / ores['temporal'] == t_enhance
                 and ires['spatial'] / ores['spatial'] == s_enhance)
        if not check:
            msg = (f'Enhancement factors (s_enhance={s_enhance}, '
                   f't_enhance={t_enhance}) do not evenly divide input '
                   f'resolution ({self.input_resolution})')
            logger.error(msg)
            raise RuntimeError(msg)

    def _ensure_valid_enhancement_factors(self):
        t_enhance = self.meta.get('t_enhance', None)
        s_enhance = self.meta.get('s_enhance', None)
        if s_enhance is None or t_enhance is None:
            return
        layer_se = self.get_s_enhance_from_layers()
        layer_te = self.get_t_enhance_from_layers()
        if not (layer_se == s_enhance or layer_te == t_enhance):
            msg = ('Enhancement factors computed from layer attributes '
                   f'(s_enhance={layer_se}, t_enhance={layer_te}) conflict '
                   f'with user provided values (s_enhance={s_enhance}, '
                   f't_enhance={t_enhance})')
            logger.error(msg)
            raise RuntimeError(msg)

    @staticmethod
    def check_batch_handler_attrs(batch_handler):
        return {k: getattr(batch_handler, k, None)
                for k in ['smoothing', 'lr_features', 'hr_exo_features',
                          'hr_out_features', 'smoothed_features']
                if hasattr(batch_handler, k)}

    # -------------------------------------------------------- bookkeeping
    def _init_records(self):
        if self._history is not None:
            train_cols = [c for c in self._history.columns if 'train_' in c]
            val_cols = [c for c in self._history.columns if 'val_' in c]
            self._train_record = self._history[train_cols].iloc[-1:]
            self._train_record = self._train_record.reset_index(drop=True)
            self._val_record = self._history[val_cols].iloc[-1:]
            self._val_record = self._val_record.reset_index(drop=True)

    @staticmethod
    def update_loss_details(record, new_data, max_batches, prefix=None):
        new_index = 0 if len(record) == 0 else record.index[-1] + 1
        for k, v in new_data.items():
            key = k if prefix is None or prefix in k else prefix + k
            record.loc[new_index, key] = float(numpy_if_tensor(v))
        return record.iloc[-max_batches:]

    @staticmethod
    def log_loss_details(loss_details, level='INFO'):
        for k, v in sorted(loss_details.items()):
            fmt = '\t{}: {}' if isinstance(v, str) else '\t{}: {:.2e}'
            (logger.info if level.lower() == 'info' else logger.debug)(
                fmt.format(k, v))

    @staticmethod
    def early_stop(history, column, threshold=0.005, n_epoch=5):
        stop = False
        if history is not None and len(history) > n_epoch + 1:
            diffs = np.abs(np.diff(history[column]))
            if all(diffs[-n_epoch:] < threshold):
                stop = True
                logger.info('Found early stop condition, loss values "{}" '
                            'have absolute relative differences less than '
                            'threshold {}: {}'.format(column, threshold,
                                                      diffs[-n_epoch:]))
        return stop

    def finish_epoch(self, epoch, epochs, t0, loss_details, checkpoint_int,
                     out_dir, early_stop_on, early_stop_threshold,
                     early_stop_n_epoch, extras=None):
        self.log_loss_details(loss_details)
        self._history.at[epoch, 'elapsed_time'] = time.time() - t0
        for k, v in loss_details.items():
            self._history.at[epoch, k] = float(v)
        last_epoch = epoch == epochs[-1]
        chp = checkpoint_int is not None and (epoch % checkpoint_int) == 0
        if last_epoch or chp:
            msg = ('Model output dir for checkpoint models should have '
                   f'{"{epoch}"} but did not: {out_dir}')
            assert '{epoch}' in out_dir, msg
            self.save(out_dir.format(epoch=epoch))
        stop = False
        if early_stop_on is not None and early_stop_on in self._history:
            stop = self.early_stop(self._history, early_stop_on,
                                   threshold=early_stop_threshold,
                                   n_epoch=early_stop_n_epoch)
            if stop:
                self.save(out_dir.format(epoch=epoch))
        if extras is not None:
            for k, v in extras.items():
                self._history.at[epoch, k] = safe_cast(v)
        return stop

    @staticmethod
    def get_weight_update_fraction(history, comparison_key,
                                   update_bounds=(0.5, 0.95), update_frac=0.0):
        val = history[comparison_key]
        if isinstance(val, (list, tuple, np.ndarray)):
            val = val[-1]
        if val < update_bounds[0]:
            return 1 + update_frac
        if val > update_bounds[1]:
            return 1 / (1 + update_frac)
        return 1

    def update_adversarial_weights(self, history, adaptive_update_fraction,
                                   adaptive_update_bounds, weight_gen_advers,
                                   train_disc):
        if adaptive_update_fraction > 0:
            update_frac = 1
            if train_disc:
                update_frac = self.get_weight_update_fraction(
                    history, 'disc_train_frac',
                    update_frac=adaptive_update_fraction,
                    update_bounds=adaptive_update_bounds)
                weight_gen_advers *= update_frac
            if update_frac != 1:
                logger.debug(
                    f'New discriminator weight: {weight_gen_advers:.4e}')
        return weight_gen_advers

    # --------------------------------------------------------------- train
    def calc_val_loss(self, batch_handler, weight_gen_advers):
        logger.debug('Starting end-of-epoch validation loss calculation...')
        for batch in batch_handler.val_data:
            _, v_loss_details, _, _ = self._get_hr_exo_and_loss(
                batch.low_res, batch.high_res,
                weight_gen_advers=weight_gen_advers)
            self._val_record = self.update_loss_details(
                self._val_record, v_loss_details,
                len(batch_handler.val_data), prefix='val_')
        return self._val_record.mean(axis=0)

    def _train_batch(self, batch, train_gen, only_gen, gen_too_good,
                     train_disc, only_disc, disc_too_good, weight_gen_advers,
                     multi_gpu=False):
        trained_gen = trained_disc = False
        loss_details = {}
        if only_gen or (train_gen and not gen_too_good):
            trained_gen = True
            loss_details.update(self.timer(self.run_gradient_descent)(
                batch.low_res, batch.high_res, None,
                weight_gen_advers=weight_gen_advers, optimizer=self.optimizer,
                train_gen=True, train_disc=False, compute_disc=train_disc,
                multi_gpu=multi_gpu))
        if only_disc or (train_disc and not disc_too_good):
            trained_disc = True
            loss_details.update(self.timer(self.run_gradient_descent)(
                batch.low_res, batch.high_res, None,
                weight_gen_advers=weight_gen_advers,
                optimizer=self.optimizer_disc, train_gen=False,
                train_disc=True, multi_gpu=multi_gpu))
        loss_details = {k: float(v) for k, v in loss_details.items()}
        loss_details['gen_train_frac'] = float(trained_gen)
        loss_details['disc_train_frac'] = float(trained_disc)
        return loss_details

    def _post_batch(self, ib, b_loss_details, n_batches, previous_means):
        for key, val in previous_means.items():
            if key.startswith('train_'):
                b_loss_details.setdefault(key.replace('train_', ''), val)
        self._train_record = self.update_loss_details(
            self._train_record, b_loss_details, n_batches, prefix='train_')
        trained_gen = bool(self._train_record['gen_train_frac'].values[-1])
        trained_disc = bool(self._train_record['disc_train_frac'].values[-1])
        if not trained_gen and not trained_disc:
            msg = ('For some reason none of the GAN networks trained during '
                   'batch {} out of {}!'.format(ib, n_batches))
            logger.warning(msg)
            warn(msg)
        return self._train_record.mean(axis=0).to_dict()

    def _train_epoch(self, batch_handler, weight_gen_advers, train_gen,
                     train_disc, disc_loss_bounds, multi_gpu=False):
        lr_shape, hr_shape = batch_handler.shapes
        self.init_weights(lr_shape, hr_shape)
        disc_th_low = np.min(disc_loss_bounds)
        disc_th_high = np.max(disc_loss_bounds)
        loss_means = self._train_record.mean().to_dict()
        loss_means.setdefault('train_loss_disc', 0)
        loss_means.setdefault('train_loss_gen', 0)
        only_gen = train_gen and not train_disc
        only_disc = train_disc and not train_gen
        for ib, batch in enumerate(batch_handler):
            start = time.time()
            loss_disc = loss_means['train_loss_disc']
            disc_too_good = loss_disc <= disc_th_low
            disc_too_bad = (loss_disc > disc_th_high) and train_disc
            gen_too_good = disc_too_bad
            b_loss_details = self.timer(self._train_batch, log=True)(
                batch, train_gen, only_gen, gen_too_good, train_disc,
                only_disc, disc_too_good, weight_gen_advers, multi_gpu)
            loss_means = self.timer(self._post_batch, log=True)(
                ib, b_loss_details, len(batch_handler), loss_means)
            logger.info(f'Finished batch step {ib + 1} / '
                        f'{len(batch_handler)} in '
                        f'{time.time() - start:.4f} seconds')
        self.total_batches += len(batch_handler)
        loss_details = self._train_record.mean().to_dict()
        loss_details['total_batches'] = int(self.total_batches)
        return loss_details

    def train(self, batch_handler, input_resolution, n_epoch,
              weight_gen_advers=0.001, train_gen=True, train_disc=True,
              disc_loss_bounds=(0.45, 0.6), checkpoint_int=None,
              out_dir='./gan_{epoch}', early_stop_on=None,
              early_stop_threshold=0.005, early_stop_n_epoch=5,
              adaptive_update_bounds=(0.9, 0.99), adaptive_update_fraction=0.0,
              multi_gpu=False, tensorboard_log=False,
              tensorboard_profile=False):
        """base.py:624-828 (tensorboard options are accepted and ignored:
        profiling on MI355X goes through rocprofv3, see tools/)."""
        self.set_norm_stats(batch_handler.means, batch_handler.stds)
        params = self.check_batch_handler_attrs(batch_handler)
        self.set_model_params(input_resolution=input_resolution,
                              s_enhance=batch_handler.s_enhance,
                              t_enhance=batch_handler.t_enhance, **params)
        epochs = list(range(n_epoch))
        if self._history is None:
            self._history = pd.DataFrame(columns=['elapsed_time'])
            self._history.index.name = 'epoch'
        else:
            epochs = [e + int(self._history.index.values[-1]) + 1
                      for e in epochs]
        t0 = time.time()
        logger.info('Training model with adversarial weight: {} for {} epochs '
                    'starting at epoch {}'.format(weight_gen_advers, n_epoch,
                                                  epochs[0]))
        for epoch in epochs:
            t_epoch = time.time()
            loss_details = self._train_epoch(
                batch_handler, weight_gen_advers, train_gen, train_disc,
                disc_loss_bounds, multi_gpu=multi_gpu)
            loss_details.update(
                self.calc_val_loss(batch_handler, weight_gen_advers))
            msg = f'Epoch {epoch} of {epochs[-1]} '
            msg += 'gen/disc train loss: {:.2e}/{:.2e} '.format(
                loss_details['train_loss_gen'],
                loss_details['train_loss_disc'])
            if 'val_loss_gen' in loss_details and \
                    'val_loss_disc' in loss_details:
                msg += 'gen/disc val loss: {:.2e}/{:.2e} '.format(
                    loss_details['val_loss_gen'],
                    loss_details['val_loss_disc'])
            logger.info(msg)
            extras = {'weight_gen_advers': weight_gen_advers,
                      'disc_loss_bound_0': disc_loss_bounds[0],
                      'disc_loss_bound_1': disc_loss_bounds[1]}
            opt_g = self.get_optimizer_state(self.optimizer, self._gen)
            opt_d = self.get_optimizer_state(self.optimizer_disc, self._disc)
            extras.update({f'OptmGen/{k}': v for k, v in opt_g.items()})
            extras.update({f'OptmDisc/{k}': v for k, v in opt_d.items()})
            weight_gen_advers = self.update_adversarial_weights(
                loss_details, adaptive_update_fraction,
                adaptive_update_bounds, weight_gen_advers, train_disc)
            stop = self.finish_epoch(
                epoch, epochs, t0, loss_details, checkpoint_int, out_dir,
                early_stop_on, early_stop_threshold, early_stop_n_epoch,
                extras=extras)
            logger.info('Finished training epoch in {:.4f} seconds'.format(
                time.time() - t_epoch))
            if stop:
                break
        logger.info('Finished training {} epochs in {:.4f} seconds'.format(
            n_epoch, time.time() - t0))
        batch_handler.stop()
