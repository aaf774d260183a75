"""``hidden_layers`` JSON (the sup3r / phygnn layer-spec surface) -> logical
layer list -> fused device plan.

Host-side only: no arithmetic happens here.  Mirrors what
``AbstractSingleModel.load_network`` (sup3r/models/abstract.py:57-111) gets
from ``phygnn.CustomNetwork(hidden_layers=...)``: repeat blocks expanded in
order, same-name ``SkipConnection`` shared, keras defaults (``padding='valid'``,
``use_bias=True``, fused ``activation`` kwarg), layer attributes
``_spatial_mult`` / ``_temporal_mult`` / ``rank`` that
``AbstractInterface`` reads (sup3r/models/interface.py:71-123).

The *plan* is what crosses the C-ABI (include/sup3r_hip.h): a list of tensors
(always 5-D ``(N, s1, s2, t, C)``; 4-D nets use t = 1) and fused ops.  The key
fusion is exact index algebra, not an approximation:

    REFLECT pad P -> conv(k, valid, stride 1) -> crop C
      == conv over a *virtually* reflect-padded input with low offset P - C
    REFLECT pad P -> Conv2DTranspose | Conv3DTranspose(k, stride 1, valid)
      -> crop C   (C >= k-1)
      == the same with the kernel flipped and (C_in, C_out) swapped,
         low offset (k - 1) + P - C

plus bias / LeakyReLU / ReLU / residual-add epilogues and the depth-to-space
store permutation (DCR order) of Spatial(Temporal)Expansion.
"""
import copy
import json

import numpy as np

# op kinds (must match include/sup3r_hip.h)
OP_CONV = 1
OP_REPEAT_T = 2
OP_D2S = 3
OP_ACT = 4
OP_ADD = 5
OP_CONCAT = 6
OP_DENSE = 7
OP_PAD = 8
OP_CROP = 9
OP_VIEW = 10
OP_ROLL_T = 11
OP_DILATE = 12

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1
WL_CONV, WL_CONVT = 0, 1

EXO_CLASSES = ('Sup3rConcat', 'Sup3rAdder')
OBS_CLASSES = ('Sup3rConcatObs', 'Sup3rObsModel')


def load_hidden_layers(model):
    """abstract.py:75-94 json handling (str path | dict | list)."""
    if isinstance(model, str):
        with open(model) as f:
            model = json.load(f)
    if isinstance(model, dict):
        if 'hidden_layers' in model:
            return model['hidden_layers']
        raise KeyError('Could not load model from json config, need '
                       '"hidden_layers" key at top level but only found: '
                       '{}'.format(list(model.keys())))
    return model


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


def _tup(v, n):
    if isinstance(v, (int, np.integer)):
        return (int(v),) * n
    v = tuple(int(i) for i in v)
    if len(v) != n:
        raise ValueError(f'expected {n} values, got {v}')
    return v


class LayerSpec:
    """One logical (keras / phygnn) layer.  ``instance`` identifies shared
    objects (same-name SkipConnection)."""

    def __init__(self, cls, kwargs, instance):
        self.cls = cls
        self.kwargs = kwargs
        self.instance = instance
        self.name = kwargs.get('name', None)
        self._spatial_mult = int(kwargs.get('spatial_mult', 1)) \
            if cls in ('SpatialExpansion', 'SpatioTemporalExpansion') else 1
        self._temporal_mult = int(kwargs.get('temporal_mult', 1)) \
            if cls == 'SpatioTemporalExpansion' else 1
        self.rank = None
        if cls == 'FlexiblePadding':
            self.rank = len(kwargs['paddings'])
        elif cls in ('Conv2D', 'Conv2DTranspose', 'Cropping2D',
                     'SpatialExpansion'):
            self.rank = 4
        elif cls in ('Conv3D', 'Conv3DTranspose', 'Cropping3D',
                     'SpatioTemporalExpansion'):
            self.rank = 5

    def __repr__(self):
        return f'<{self.cls} {self.kwargs}>'


SUPPORTED = ('FlexiblePadding', 'Conv2D', 'Conv3D', 'Conv2DTranspose',
             'Conv3DTranspose',
             'Cropping2D', 'Cropping3D', 'LeakyReLU', 'Activation', 'ReLU',
             'SkipConnection', 'SpatialExpansion', 'SpatioTemporalExpansion',
             'Flatten', 'Dense', 'Sup3rConcat', 'Sup3rAdder', 'Sup3rConcatObs')


# keras kwargs a layer may carry: the ones the lowering reads, and the ones
# that do not change the forward arithmetic.  Anything else (dilation_rate,
# groups, a non-default data_format ...) would silently compute something
# different from keras, so it raises like any other unsupported feature.
_BENIGN = {'name', 'dtype', 'trainable', 'kernel_initializer',
           'bias_initializer', 'kernel_regularizer', 'bias_regularizer',
           'activity_regularizer', 'kernel_constraint', 'bias_constraint',
           'input_shape', 'batch_input_shape'}
_NEUTRAL = {'data_format': ('channels_last', None), 'dilation_rate': (1,),
            'groups': (1,), 'output_padding': (None,)}
_CONV_KW = {'filters', 'kernel_size', 'strides', 'padding', 'activation',
            'use_bias'}
_READ_KW = {
    'Conv2D': _CONV_KW, 'Conv3D': _CONV_KW, 'Conv2DTranspose': _CONV_KW,
    'Conv3DTranspose': _CONV_KW,
    'Dense': {'units', 'activation', 'use_bias'},
    'LeakyReLU': {'alpha', 'negative_slope'},
    'Activation': {'activation'}, 'ReLU': set(),
    'Cropping2D': {'cropping'}, 'Cropping3D': {'cropping'},
    'FlexiblePadding': {'paddings', 'mode'},
}


def _check_kwargs(cls, kw, index):
    known = _READ_KW.get(cls)
    if known is None:
        return
    for k, v in kw.items():
        if k in known or k in _BENIGN:
            continue
        if k in _NEUTRAL:
            vals = v if isinstance(v, (list, tuple)) else [v]
            if all(x in _NEUTRAL[k] for x in vals):
                continue
        raise KeyError(f'{cls} (hidden layer #{index}): keyword "{k}"={v!r} '
                       'changes the arithmetic and has no MI355X kernel '
                       'mapping')
    if cls == 'ReLU' and any(kw.get(k) not in (None, 0, 0.0) for k in (
            'max_value', 'negative_slope', 'threshold')):
        raise KeyError(f'ReLU (hidden layer #{index}) with {kw} has no '
                       'MI355X kernel mapping')


def parse_layers(hidden_layers):
    """hidden_layers -> list[LayerSpec] with phygnn HiddenLayers semantics."""
    layers, skips = [], {}
    for i, spec in enumerate(expand_repeats(load_hidden_layers(hidden_layers))):
        spec = dict(spec)
        if 'class' in spec:
            cls = spec.pop('class')
            if cls not in SUPPORTED:
                raise KeyError(
                    f'Layer class "{cls}" (hidden layer #{i}) has no '
                    'MI355X kernel mapping in sup3r_amd')
            _check_kwargs(cls, spec, i)
            if cls == 'SkipConnection':
                nm = spec['name']
                if nm not in skips:
                    skips[nm] = LayerSpec(cls, spec, ('skip', nm))
                layers.append(skips[nm])
            else:
                layers.append(LayerSpec(cls, spec, ('layer', len(layers))))
        else:
            act = spec.pop('activation', None)
            if spec.pop('dropout', None) is not None:
                raise KeyError('dropout layers are not supported')
            if spec.pop('batch_normalization', None) is not None:
                raise KeyError('batch_normalization is not supported')
            if 'units' in spec:
                layers.append(LayerSpec('Dense', spec, ('layer', len(layers))))
            if act is not None:
                layers.append(LayerSpec('Activation', {'activation': act},
                                        ('layer', len(layers))))
    return layers


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _act_code(name, alpha=0.0):
    if name is None or name == 'linear':
        return ACT_NONE, 0.0
    if name == 'relu':
        return ACT_RELU, 0.0
    if name == 'leaky_relu':
        return ACT_LEAKY, float(alpha)
    raise KeyError(f'activation "{name}" has no MI355X kernel mapping')


class Plan:
    """Fused device plan for ONE concrete input shape (batch included)."""

    def __init__(self):
        self.tensors = []      # list of [n, d1, d2, d3, c]
        self.ops = []          # list of dict
        self.params = []       # list of dict(shape=keras shape, kind, layer)
        self.inputs = {}       # name -> tensor id ('x' + exo names)
        self.output = None
        self.out_rank = None   # 2 (dense logits), 4 or 5
        self.layer_out_shapes = []   # keras-view shape after each layer
        # (tensor id, op index | -1) holding the result after each layer; the
        # layers of one fused group all point at the group's op
        self.layer_out = []

    def new_tensor(self, shape5):
        self.tensors.append([int(v) for v in shape5])
        return len(self.tensors) - 1

    def n_params(self):
        return sum(int(np.prod(p['shape'])) for p in self.params)


def _reflect_ok(n, lo, hi):
    return lo <= n - 1 and hi <= n - 1


def build_plan(layers, in_shape, param_table=None, fuse=True):
    """Shape inference + fusion.

    Parameters
    ----------
    layers : list[LayerSpec]
    in_shape : tuple   keras-view input shape (N, s1, s2[, t], C)
    param_table : list | None
        Existing parameter table of this network (from a previous plan at
        another input shape); verified to match, so plans share one store.
    fuse : bool
        False lowers every layer to its own op (used by tests to cross-check
        the fused plan).
    """
    in_shape = tuple(int(v) for v in in_shape)
    rank = len(in_shape)
    if rank not in (2, 4, 5):
        raise ValueError(f'unsupported input rank {rank}')
    plan = Plan()
    if rank == 5:
        cur_shape = list(in_shape)
    elif rank == 4:
        cur_shape = [in_shape[0], in_shape[1], in_shape[2], 1, in_shape[3]]
    else:
        cur_shape = [in_shape[0], 1, 1, 1, in_shape[1]]
    cur = plan.new_tensor(cur_shape)
    plan.inputs['x'] = cur
    nd = 3 if rank == 5 else 2      # spatial dims seen by keras
    flat = rank == 2
    skip_cache = {}
    pend = None      # pending virtual padding: dict(lo=[3], hi=[3], mode)
    layer_param = {}   # instance -> (w_id, b_id)

    def cur_dims():
        return plan.tensors[cur]

    def flush_pad():
        """Materialise a pending FlexiblePadding that no conv consumed."""
        nonlocal cur, pend
        if pend is None:
            return
        sh = cur_dims()
        out = [sh[0]] + [sh[1 + d] + pend['lo'][d] + pend['hi'][d]
                         for d in range(3)] + [sh[4]]
        t = plan.new_tensor(out)
        plan.ops.append(dict(kind=OP_PAD, in0=cur, out=t, lo=list(pend['lo']),
                             hi=list(pend['hi']), pad_mode=pend['mode']))
        cur, pend = t, None

    def get_param(layer, wshape, bshape, wl):
        key = layer.instance
        if key in layer_param:
            return layer_param[key]
        wid = len(plan.params)
        plan.params.append(dict(shape=tuple(wshape), kind='kernel',
                                layout=wl, layer=key))
        bid = -1
        if bshape is not None:
            bid = len(plan.params)
            plan.params.append(dict(shape=tuple(bshape), kind='bias',
                                    layout=WL_CONV, layer=key))
        layer_param[key] = (wid, bid)
        return wid, bid

    def keras_view(sh, flat_now):
        if flat_now:
            return (sh[0], sh[4])
        if nd == 3:
            return tuple(sh)
        return (sh[0], sh[1], sh[2], sh[4])

    i = 0
    n_layers = len(layers)
    while i < n_layers:
        L = layers[i]
        cls, kw = L.cls, L.kwargs
        consumed = 1
        n_ops_before = len(plan.ops)
        if cls == 'FlexiblePadding':
            pads = [tuple(int(v) for v in p) for p in kw['paddings']]
            if len(pads) != nd + 2:
                raise RuntimeError(
                    f'FlexiblePadding rank {len(pads)} does not match tensor '
                    f'rank {nd + 2}')
            if pads[0] != (0, 0) or pads[-1] != (0, 0):
                raise KeyError('padding of batch/channel axes is unsupported')
            mode = kw.get('mode', 'REFLECT').upper()
            if mode not in ('REFLECT', 'CONSTANT'):
                raise KeyError(f'pad mode {mode} has no kernel mapping')
            flush_pad()
            lo = [pads[1 + d][0] if d < nd else 0 for d in range(3)]
            hi = [pads[1 + d][1] if d < nd else 0 for d in range(3)]
            sh = cur_dims()
            if mode == 'REFLECT':
                for d in range(3):
                    if not _reflect_ok(sh[1 + d], lo[d], hi[d]):
                        raise RuntimeError(
                            'REFLECT padding {} exceeds dim {} - 1'.format(
                                (lo[d], hi[d]), sh[1 + d]))
            pend = dict(lo=lo, hi=hi,
                        mode=PAD_REFLECT if mode == 'REFLECT' else PAD_ZERO)
            if not fuse:
                flush_pad()
        elif cls in ('Conv2D', 'Conv3D', 'Conv2DTranspose', 'Conv3DTranspose'):
            cnd = 3 if cls in ('Conv3D', 'Conv3DTranspose') else 2
            if cnd != nd:
                raise RuntimeError(f'{cls} applied to rank-{nd + 2} tensor')
            is_t = cls in ('Conv2DTranspose', 'Conv3DTranspose')
            filters = int(kw['filters'])
            k = list(_tup(kw['kernel_size'], cnd)) + [1] * (3 - cnd)
            s = list(_tup(kw.get('strides', 1), cnd)) + [1] * (3 - cnd)
            padding = kw.get('padding', 'valid').lower()
            act, alpha = _act_code(kw.get('activation', None))
            use_bias = kw.get('use_bias', True)
            sh = cur_dims()
            cin = sh[4]
            # virtual padded extent seen by the conv
            plo = pend['lo'] if pend else [0, 0, 0]
            phi = pend['hi'] if pend else [0, 0, 0]
            pmode = pend['mode'] if pend else PAD_ZERO
            ext = [sh[1 + d] + plo[d] + phi[d] for d in range(3)]
            if is_t:
                if padding != 'valid':
                    raise KeyError(f'{cls} with padding != valid has no '
                                   'kernel mapping')
                if any(v != 1 for v in s):
                    if any(k[d] < s[d] for d in range(3)):
                        # keras 'valid' gives in * s + max(k - s, 0) cells,
                        # the zero-insertion lowering (in - 1) * s + k: they
                        # agree only for k >= s
                        raise KeyError(
                            f'{cls} with kernel_size {k[:cnd]} < strides '
                            f'{s[:cnd]} has no kernel mapping')
                    # strided transpose: y[i s + k] += x[i] w[k] is the
                    # stride-1 transpose of x with s - 1 zeros inserted
                    # between its cells (the padding before it is real data
                    # for the transpose, so it is materialised first)
                    flush_pad()
                    sh = cur_dims()
                    dil = [sh[0]] + [(sh[1 + d] - 1) * s[d] + 1
                                     for d in range(3)] + [sh[4]]
                    t_d = plan.new_tensor(dil)
                    plan.ops.append(dict(kind=OP_DILATE, in0=cur, out=t_d,
                                         stride=list(s)))
                    cur, sh = t_d, dil
                    plo, phi, pmode = [0] * 3, [0] * 3, PAD_ZERO
                    ext = [sh[1 + d] for d in range(3)]
                    s = [1, 1, 1]
                full = [ext[d] + k[d] - 1 for d in range(3)]
                # look ahead for the crop that removes the zero-tail region
                crop_lo, crop_hi = [0, 0, 0], [0, 0, 0]
                if i + 1 < n_layers and layers[i + 1].cls == f'Cropping{cnd}D':
                    c = _crop_list(layers[i + 1].kwargs.get('cropping', 0), cnd)
                    for d in range(cnd):
                        crop_lo[d], crop_hi[d] = c[d]
                    consumed = 2
                need = [k[d] - 1 for d in range(3)]
                if pmode == PAD_REFLECT and any(
                        crop_lo[d] < need[d] or crop_hi[d] < need[d]
                        for d in range(3)):
                    # (behind a fused REFLECT pad the zero tails of the
                    # transpose would mix with mirrored data)
                    raise KeyError(
                        f'{cls} whose zero tails are not cropped '
                        '(cropping < kernel_size - 1) has no kernel mapping')
                out_sp = [full[d] - crop_lo[d] - crop_hi[d] for d in range(3)]
                lo = [(k[d] - 1) + plo[d] - crop_lo[d] for d in range(3)]
                wshape = tuple(k[:cnd]) + (filters, cin)
                wl = WL_CONVT
            else:
                if padding == 'same':
                    if pend is not None:
                        flush_pad()
                        sh = cur_dims()
                        plo, phi, pmode = [0] * 3, [0] * 3, PAD_ZERO
                        ext = [sh[1 + d] for d in range(3)]
                    sp = [_same_pad(ext[d], k[d], s[d]) for d in range(3)]
                    plo = [p[0] for p in sp]
                    phi = [p[1] for p in sp]
                    pmode = PAD_ZERO
                    ext = [sh[1 + d] + plo[d] + phi[d] for d in range(3)]
                elif padding != 'valid':
                    raise KeyError(f'padding "{padding}" is unsupported')
                conv_out = [(ext[d] - k[d]) // s[d] + 1 for d in range(3)]
                crop_lo, crop_hi = [0, 0, 0], [0, 0, 0]
                nxt = layers[i + 1].cls if i + 1 < n_layers else None
                if fuse and nxt in ('Cropping2D', 'Cropping3D') and \
                        all(v == 1 for v in s):
                    c = _crop_list(layers[i + 1].kwargs.get('cropping', 0),
                                   nd)
                    for d in range(nd):
                        crop_lo[d], crop_hi[d] = c[d]
                    consumed = 2
                out_sp = [conv_out[d] - crop_lo[d] - crop_hi[d]
                          for d in range(3)]
                lo = [plo[d] - crop_lo[d] * s[d] for d in range(3)]
                wshape = tuple(k[:cnd]) + (cin, filters)
                wl = WL_CONV
            if any(v <= 0 for v in out_sp):
                raise RuntimeError(
                    f'{cls} output shape {out_sp} is not positive for input '
                    f'{sh}')
            if pmode == PAD_REFLECT:
                # every tap index must stay inside the reflectable range
                for d in range(3):
                    mx = (out_sp[d] - 1) * s[d] + k[d] - 1 - lo[d]
                    if lo[d] > sh[1 + d] - 1 or \
                            mx - (sh[1 + d] - 1) > sh[1 + d] - 1:
                        raise RuntimeError('REFLECT padding exceeds dim')
            wid, bid = get_param(L, wshape, (filters,) if use_bias else None,
                                 wl)
            op = dict(kind=OP_CONV, in0=cur, w=wid, b=bid, k=k, stride=s,
                      lo=lo, pad_mode=pmode, wlayout=wl, act=act, alpha=alpha,
                      res=-1, res_before_act=0, d2s=1, cin=cin, cout=filters)
            pend = None
            out_shape = [sh[0]] + out_sp + [filters]
            # ---- epilogue fusion: [d2s] [act] [skip-end add]
            j = i + consumed
            if fuse:
                if j < n_layers and layers[j].cls in (
                        'SpatialExpansion', 'SpatioTemporalExpansion') and \
                        layers[j]._temporal_mult == 1 and \
                        layers[j]._spatial_mult > 1:
                    b = layers[j]._spatial_mult
                    if filters % (b * b) != 0:
                        raise RuntimeError(
                            'Spatial expansion of factor {} is being '
                            'attempted on input tensor of shape {}, but the '
                            'last dimension of the input tensor ({}) must be '
                            'divisible by the spatial factor squared ({}).'
                            .format(b, keras_view(out_shape, False), filters,
                                    b * b))
                    op['d2s'] = b
                    out_shape = [sh[0], out_sp[0] * b, out_sp[1] * b,
                                 out_sp[2], filters // (b * b)]
                    j += 1
                if op['act'] == ACT_NONE and j < n_layers and \
                        layers[j].cls in ('LeakyReLU', 'Activation', 'ReLU'):
                    a = _layer_act(layers[j])
                    if a is not None:
                        op['act'], op['alpha'] = a
                        j += 1
                if j < n_layers and layers[j].cls == 'SkipConnection' and \
                        layers[j].name in skip_cache and \
                        op['act'] == ACT_NONE and \
                        plan.tensors[skip_cache[layers[j].name]] == out_shape:
                    op['res'] = skip_cache.pop(layers[j].name)
                    j += 1
            t = plan.new_tensor(out_shape)
            op['out'] = t
            plan.ops.append(op)
            cur = t
            consumed = j - i
        elif cls in ('Cropping2D', 'Cropping3D'):
            flush_pad()
            c = _crop_list(kw.get('cropping', 0), nd) + [(0, 0)] * (3 - nd)
            sh = cur_dims()
            out = [sh[0]] + [sh[1 + d] - c[d][0] - c[d][1]
                             for d in range(3)] + [sh[4]]
            t = plan.new_tensor(out)
            plan.ops.append(dict(kind=OP_CROP, in0=cur, out=t,
                                 lo=[c[d][0] for d in range(3)]))
            cur = t
        elif cls in ('LeakyReLU', 'Activation', 'ReLU'):
            flush_pad()
            a = _layer_act(L)
            if a is None:
                raise KeyError(f'activation {kw} has no kernel mapping')
            t = plan.new_tensor(cur_dims())
            plan.ops.append(dict(kind=OP_ACT, in0=cur, out=t, act=a[0],
                                 alpha=a[1]))
            cur = t
        elif cls == 'SkipConnection':
            flush_pad()
            if L.name in skip_cache:
                other = skip_cache.pop(L.name)
                if plan.tensors[other] != cur_dims():
                    raise RuntimeError('SkipConnection shape mismatch')
                t = plan.new_tensor(cur_dims())
                plan.ops.append(dict(kind=OP_ADD, in0=cur, in1=other, out=t))
                cur = t
            else:
                skip_cache[L.name] = cur
        elif cls in ('SpatialExpansion', 'SpatioTemporalExpansion'):
            flush_pad()
            if cls == 'SpatialExpansion' and nd != 2:
                raise RuntimeError('SpatialExpansion needs a 4D tensor')
            if cls == 'SpatioTemporalExpansion' and nd != 3:
                raise RuntimeError('SpatioTemporalExpansion needs a 5D tensor')
            m, b = L._temporal_mult, L._spatial_mult
            if m > 1:
                meth = kw.get('temporal_method', 'nearest')
                sh = cur_dims()
                if meth == 'nearest':
                    t = plan.new_tensor(
                        [sh[0], sh[1], sh[2], sh[3] * m, sh[4]])
                    plan.ops.append(dict(kind=OP_REPEAT_T, in0=cur, out=t,
                                         rep=m))
                    cur = t
                elif meth == 'depth_to_time':
                    # (.., t, c) -> (.., t*m, c/m) is a pure view of
                    # channels-last memory; then tf.roll along t
                    if sh[4] % m != 0:
                        raise RuntimeError(
                            'Temporal expansion of factor {} is being '
                            'attempted on input tensor of shape {}, but the '
                            'last dimension ({}) must be divisible by the '
                            'temporal factor.'.format(
                                m, keras_view(sh, False), sh[4]))
                    t = plan.new_tensor(
                        [sh[0], sh[1], sh[2], sh[3] * m, sh[4] // m])
                    plan.ops.append(dict(kind=OP_VIEW, in0=cur, out=t))
                    cur = t
                    roll = int(kw.get('t_roll', 0))
                    if roll % (sh[3] * m) != 0:
                        t2 = plan.new_tensor(plan.tensors[t])
                        plan.ops.append(dict(kind=OP_ROLL_T, in0=cur, out=t2,
                                             rep=roll))
                        cur = t2
                else:
                    raise KeyError(
                        f'temporal_method "{meth}" has no kernel mapping '
                        '(only "nearest" and "depth_to_time")')
            if b > 1:
                sh = cur_dims()
                if sh[4] % (b * b) != 0:
                    raise RuntimeError(
                        'Spatial expansion of factor {} is being attempted '
                        'on input tensor of shape {}, but the last dimension '
                        'of the input tensor ({}) must be divisible by the '
                        'spatial factor squared ({}).'.format(
                            b, keras_view(sh, False), sh[4], b * b))
                t = plan.new_tensor([sh[0], sh[1] * b, sh[2] * b, sh[3],
                                     sh[4] // (b * b)])
                plan.ops.append(dict(kind=OP_D2S, in0=cur, out=t, d2s=b))
                cur = t
        elif cls in EXO_CLASSES or cls == 'Sup3rConcatObs':
            # Sup3rConcatObs (phygnn, not vendored): a sparse observation
            # field joins the tensor as one more channel.  Its NaN handling
            # cannot be restated from anything under /root/reference; here an
            # un-observed cell carries 0 in normalised units (the feature
            # mean) — the fill is applied where the field is prepared
            # (Sup3rGan._reshape_norm_exo / Sup3rGanWithObs), the device op is
            # the plain concat.  UNVERIFIED against phygnn.
            if cls == 'Sup3rConcatObs':
                # one observation channel per layer is all that is restated
                # here: a layer with several `features` or with
                # `exo_features` (abstract.py:1001-1035 stacks them into two
                # extra call arguments) would be lowered wrongly, not slowly
                if len(kw.get('features', [L.name])) > 1 or \
                        kw.get('exo_features'):
                    raise KeyError(
                        f'Sup3rConcatObs layer "{L.name}" with several '
                        '`features` or with `exo_features` has no kernel '
                        'mapping (one observation channel per layer only)')
                import warnings
                warnings.warn(
                    f'Sup3rConcatObs layer "{L.name}": phygnn is not '
                    'available to this build, the layer is lowered as a '
                    'channel concat with un-observed (NaN) cells at the '
                    'feature mean — UNVERIFIED against phygnn; a generator '
                    'trained with the reference\'s layer may produce '
                    'different fields', RuntimeWarning, stacklevel=2)
            flush_pad()
            sh = cur_dims()
            if cls in ('Sup3rConcat', 'Sup3rConcatObs'):
                e = plan.new_tensor(sh[:4] + [1])
                plan.inputs[L.name] = e
                t = plan.new_tensor(sh[:4] + [sh[4] + 1])
                plan.ops.append(dict(kind=OP_CONCAT, in0=cur, in1=e, out=t))
            else:
                e = plan.new_tensor(sh[:4] + [1])
                plan.inputs[L.name] = e
                t = plan.new_tensor(sh)
                plan.ops.append(dict(kind=OP_ADD, in0=cur, in1=e, out=t,
                                     bcast_c=1))
            cur = t
        elif cls == 'Flatten':
            flush_pad()
            sh = cur_dims()
            # row-major (s1, s2, t, C) flatten is a pure view of NDHWC memory
            t = plan.new_tensor([sh[0], 1, 1, 1, sh[1] * sh[2] * sh[3] * sh[4]])
            plan.ops.append(dict(kind=OP_VIEW, in0=cur, out=t))
            cur = t
            flat = True
        elif cls == 'Dense':
            flush_pad()
            sh = cur_dims()
            units = int(kw['units'])
            act, alpha = _act_code(kw.get('activation', None))
            use_bias = kw.get('use_bias', True)
            wid, bid = get_param(L, (sh[4], units),
                                 (units,) if use_bias else None, WL_CONV)
            op = dict(kind=OP_DENSE, in0=cur, w=wid, b=bid, act=act,
                      alpha=alpha, cin=sh[4], cout=units)
            j = i + 1
            if fuse and act == ACT_NONE and j < n_layers and \
                    layers[j].cls in ('LeakyReLU', 'Activation', 'ReLU'):
                a = _layer_act(layers[j])
                if a is not None:
                    op['act'], op['alpha'] = a
                    j += 1
            t = plan.new_tensor(sh[:4] + [units])
            op['out'] = t
            plan.ops.append(op)
            cur = t
            consumed = j - i
        else:
            raise KeyError(f'Layer class "{cls}" has no kernel mapping')
        consumed_op = len(plan.ops) > n_ops_before
        for _ in range(consumed):
            plan.layer_out_shapes.append(keras_view(cur_dims(), flat))
            plan.layer_out.append((cur, len(plan.ops) - 1 if consumed_op else -1))
        i += consumed
    flush_pad()
    if skip_cache:
        # an un-closed SkipConnection is legal in phygnn (identity)
        pass
    plan.output = cur
    plan.out_rank = 2 if flat else nd + 2
    plan.out_shape = keras_view(cur_dims(), flat)
    if param_table is not None:
        a = [(p['shape'], p['kind']) for p in param_table]
        b = [(p['shape'], p['kind']) for p in plan.params]
        if a != b:
            raise RuntimeError(
                'network weights were built for a different input shape: '
                f'{a} vs {b}')
    return plan


def _crop_list(cropping, nd):
    if isinstance(cropping, (int, np.integer)):
        return [(int(cropping), int(cropping))] * nd
    out = []
    for c in cropping:
        if isinstance(c, (int, np.integer)):
            out.append((int(c), int(c)))
        else:
            out.append((int(c[0]), int(c[1])))
    if len(out) != nd:
        raise ValueError('bad cropping spec')
    return out


def _layer_act(L):
    if L.cls == 'LeakyReLU':
        # keras-2.15 LeakyReLU default alpha = 0.3
        return ACT_LEAKY, float(L.kwargs.get(
            'alpha', L.kwargs.get('negative_slope', 0.3)))
    if L.cls == 'ReLU':
        return ACT_RELU, 0.0
    if L.cls == 'Activation':
        name = L.kwargs.get('activation')
        if name == 'relu':
            return ACT_RELU, 0.0
        if name in (None, 'linear'):
            return ACT_NONE, 0.0
        return None
    return None


# ------------------------------------------------------------------ weights
def keras_to_canonical(arr, layout):
    """keras weight -> device-canonical ``[taps..., C_in, C_out]`` fp32.
    Conv kernels are already canonical; Conv2DTranspose ``(kh, kw, Co, Ci)``
    is flipped in (h, w) and (Co, Ci) swapped."""
    arr = np.asarray(arr, dtype=np.float32)
    if layout == WL_CONVT:
        nd = arr.ndim - 2
        arr = np.flip(arr, axis=tuple(range(nd)))
        arr = np.swapaxes(arr, -1, -2)
    return np.ascontiguousarray(arr)


def canonical_to_keras(arr, layout):
    arr = np.asarray(arr, dtype=np.float32)
    if layout == WL_CONVT:
        nd = arr.ndim - 2
        arr = np.swapaxes(arr, -1, -2)
        arr = np.flip(arr, axis=tuple(range(nd)))
    return np.ascontiguousarray(arr)


def canonical_shape(shape, layout):
    shape = tuple(shape)
    if layout == WL_CONVT:
        return shape[:-2] + (shape[-1], shape[-2])
    return shape


def glorot_uniform(shape, rng):
    """keras glorot_uniform on the KERAS-layout shape."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)
