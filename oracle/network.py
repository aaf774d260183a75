"""phygnn.CustomNetwork restated (TEST INFRASTRUCTURE): ``hidden_layers`` JSON
-> ordered layer list -> eager forward / reverse-mode backward.

Follows ``sup3r/models/abstract.py:57-111`` (``load_network``) and the parse
rules of phygnn 0.0.33 ``HiddenLayers`` (not vendored; restated from its
published behaviour): ``{"n": N, "repeat": [...]}`` blocks are expanded
in order, ``class`` names resolve to keras / phygnn layers with the remaining
keys as kwargs, same-``name`` ``SkipConnection`` entries share ONE instance,
class-less dicts with ``units`` are Dense (+ optional Activation).
"""
import copy

import numpy as np

from . import layers as L

EXO_LAYERS = (L.Sup3rConcat, L.Sup3rAdder)


def expand_repeats(hidden_layers):
    out = []
    for layer in hidden_layers:
        if 'repeat' in layer and 'n' in layer:
            for _ in range(int(layer['n'])):
                out += copy.deepcopy(layer['repeat'])
        elif 'repeat' in layer:
            raise KeyError('Keyword "repeat" was found in layer but "n" was '
                           'not: {}'.format(layer))
        else:
            out.append(copy.deepcopy(layer))
    return out


def _make_layer(cls, kw, skips):
    if cls == 'FlexiblePadding':
        return L.FlexiblePadding(**kw)
    if cls == 'Conv2D':
        return L.ConvND(2, **kw)
    if cls == 'Conv3D':
        return L.ConvND(3, **kw)
    if cls == 'Conv2DTranspose':
        return L.ConvTransposeND(2, **kw)
    if cls == 'Conv3DTranspose':
        return L.ConvTransposeND(3, **kw)
    if cls == 'Cropping2D':
        return L.Cropping(kw.get('cropping', ((0, 0), (0, 0))), 2)
    if cls == 'Cropping3D':
        return L.Cropping(kw.get('cropping', ((1, 1), (1, 1), (1, 1))), 3)
    if cls == 'LeakyReLU':
        return L.LeakyReLU(**kw)
    if cls == 'Activation':
        return L.Activation(**kw)
    if cls == 'ReLU':
        return L.Activation('relu')
    if cls == 'SpatialExpansion':
        return L.SpatialExpansion(**kw)
    if cls == 'SpatioTemporalExpansion':
        return L.SpatioTemporalExpansion(**kw)
    if cls == 'SkipConnection':
        name = kw['name']
        if name not in skips:
            skips[name] = L.SkipConnection(name)
        return skips[name]
    if cls == 'Flatten':
        return L.Flatten()
    if cls == 'Dense':
        return L.Dense(**kw)
    if cls == 'Sup3rConcat':
        return L.Sup3rConcat(**kw)
    if cls == 'Sup3rConcatObs':
        # sup3r_amd's stated semantics (spec.py): the obs field arrives with
        # un-observed cells already 0 and joins as one more channel
        return L.Sup3rConcat(**kw)
    if cls == 'Sup3rAdder':
        return L.Sup3rAdder(**kw)
    raise KeyError(f'layer class {cls!r} is not restated in the oracle')


class Network:
    """Ordered layer list with eager forward/backward (oracle of
    phygnn.CustomNetwork as used by Sup3rGan)."""

    def __init__(self, hidden_layers, name=None):
        if isinstance(hidden_layers, dict):
            hidden_layers = hidden_layers['hidden_layers']
        self.name = name
        self.layers = []
        # emulation of bf16 STORAGE in the HIP plan (tests): indices of layers
        # whose output is rounded to bfloat16 before the next layer sees it
        self.emu_store_round = set()
        skips = {}
        for spec in expand_repeats(hidden_layers):
            spec = dict(spec)
            if 'class' in spec:
                cls = spec.pop('class')
                self.layers.append(_make_layer(cls, spec, skips))
            else:
                act = spec.pop('activation', None)
                spec.pop('dropout', None)
                if spec.pop('batch_normalization', None) is not None:
                    raise KeyError('batch_normalization is not restated')
                if 'units' in spec:
                    self.layers.append(L.Dense(**spec))
                if act is not None:
                    self.layers.append(L.Activation(act))

    # -- weights in keras order: kernel, bias per layer in layer order
    @property
    def weight_layers(self):
        seen, out = set(), []
        for layer in self.layers:
            if id(layer) in seen:
                continue
            seen.add(id(layer))
            if hasattr(layer, 'kernel'):
                out.append(layer)
        return out

    @property
    def weights(self):
        out = []
        for layer in self.weight_layers:
            out += list(layer.weights)
        return out

    @property
    def grads(self):
        out = []
        for layer in self.weight_layers:
            out += list(layer.grads)
        return out

    def set_weights(self, arrays):
        arrays = list(arrays)
        i = 0
        for layer in self.weight_layers:
            layer.kernel = np.array(arrays[i])
            i += 1
            if layer.use_bias:
                layer.bias = np.array(arrays[i])
                i += 1
        assert i == len(arrays)

    def init_weights(self, x, exo=None, seed=0, bias_scale=0.0):
        """Run one forward to build shapes (keras lazy build), then draw
        glorot-uniform kernels from ``default_rng(seed)`` in keras weight
        order; biases zero (keras default) or N(0, bias_scale)."""
        self.forward(x, exo)
        rng = np.random.default_rng(seed)
        for layer in self.weight_layers:
            layer.kernel = L.glorot_uniform(layer.kernel.shape, rng,
                                            layer.kernel.dtype)
            if layer.use_bias:
                b = rng.standard_normal(layer.bias.shape) * bias_scale
                layer.bias = b.astype(layer.bias.dtype)

    def forward(self, x, exo=None):
        """``exo``: dict name -> hi-res exo array (as ``_tf_generate``'s
        ``hi_res_exo``, abstract.py:1107-1129)."""
        for layer in self.layers:
            if isinstance(layer, L.SkipConnection):
                layer._cache = None
                layer._dcache = None
                layer._fwd_roles = []
        for i, layer in enumerate(self.layers):
            try:
                if isinstance(layer, EXO_LAYERS):
                    x = layer.forward(x, None if exo is None
                                      else exo.get(layer.name))
                else:
                    x = layer.forward(x)
                if i in self.emu_store_round:
                    x = L.round_bf16(x)
            except Exception as e:
                raise RuntimeError(
                    'Could not run layer #{} "{}" on tensor of shape {}'
                    .format(i, type(layer).__name__, x.shape)) from e
        return x

    def backward(self, dy):
        for layer in reversed(self.layers):
            dy = layer.backward(dy)
        return dy

    def cast(self, dtype):
        for layer in self.weight_layers:
            layer.kernel = layer.kernel.astype(dtype)
            if layer.use_bias:
                layer.bias = layer.bias.astype(dtype)
        return self
