"""Independent torch-CPU (float64, autograd) implementation of the layer
semantics, used ONLY to cross-check the numpy oracle (second opinion: oneDNN /
ATen kernels + autograd vs hand-written numpy forward/backward).

Adapters: tensors are kept channels-last like keras; each op permutes to
torch's channels-first, runs ``torch.nn.functional``, and permutes back.
"""
import copy

import numpy as np
import torch
import torch.nn.functional as F


def _expand(hidden_layers):
    out = []
    for layer in hidden_layers:
        if 'repeat' in layer:
            for _ in range(layer['n']):
                out += copy.deepcopy(layer['repeat'])
        else:
            out.append(copy.deepcopy(layer))
    return out


def _cf(x):   # channels-last -> channels-first
    nd = x.dim()
    return x.permute(0, nd - 1, *range(1, nd - 1))


def _cl(x):   # channels-first -> channels-last
    nd = x.dim()
    return x.permute(0, *range(2, nd), 1)


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class TorchNet:
    """Runs a hidden_layers spec with torch ops; weights given in keras
    layout/order (list of numpy arrays) and converted on the fly."""

    def __init__(self, hidden_layers, weights, dtype=torch.float64):
        if isinstance(hidden_layers, dict):
            hidden_layers = hidden_layers['hidden_layers']
        self.spec = _expand(hidden_layers)
        self.dtype = dtype
        self.weights = [torch.tensor(np.asarray(w), dtype=dtype,
                                     requires_grad=True) for w in weights]

    def forward(self, x, exo=None):
        wi = 0
        skips = {}
        for spec in self.spec:
            spec = dict(spec)
            cls = spec.pop('class', None)
            if cls is None:
                if 'units' in spec:
                    w, b = self.weights[wi], self.weights[wi + 1]
                    wi += 2
                    x = x @ w + b
                if spec.get('activation') == 'relu':
                    x = F.relu(x)
                continue
            if cls == 'FlexiblePadding':
                pads = spec['paddings'][1:-1]
                flat = []
                for lo, hi in reversed(pads):
                    flat += [lo, hi]
                mode = {'REFLECT': 'reflect', 'CONSTANT': 'constant'}[
                    spec.get('mode', 'REFLECT').upper()]
                x = _cl(F.pad(_cf(x), flat, mode=mode))
            elif cls in ('Conv2D', 'Conv3D'):
                nd = 2 if cls == 'Conv2D' else 3
                w, b = self.weights[wi], self.weights[wi + 1]
                wi += 2
                k = w.shape[:nd]
                s = spec.get('strides', 1)
                s = (s,) * nd if isinstance(s, int) else tuple(s)
                xc = _cf(x)
                if spec.get('padding', 'valid').lower() == 'same':
                    flat = []
                    for d in reversed(range(nd)):
                        lo, hi = _same_pad(x.shape[1 + d], k[d], s[d])
                        flat += [lo, hi]
                    xc = F.pad(xc, flat)
                wt = w.permute(nd + 1, nd, *range(nd))   # (Co, Ci, k...)
                conv = F.conv2d if nd == 2 else F.conv3d
                x = _cl(conv(xc, wt, b, stride=s))
                act = spec.get('activation')
                if act == 'relu':
                    x = F.relu(x)
            elif cls in ('Conv2DTranspose', 'Conv3DTranspose'):
                w, b = self.weights[wi], self.weights[wi + 1]
                wi += 2
                s = spec.get('strides', 1)
                # keras (k.., Co, Ci) -> torch conv_transpose (Ci, Co, k..)
                if cls == 'Conv2DTranspose':
                    wt = w.permute(3, 2, 0, 1)
                    x = _cl(F.conv_transpose2d(_cf(x), wt, b, stride=s))
                else:
                    wt = w.permute(4, 3, 0, 1, 2)
                    x = _cl(F.conv_transpose3d(_cf(x), wt, b, stride=s))
                if spec.get('activation') == 'relu':
                    x = F.relu(x)
            elif cls in ('Cropping2D', 'Cropping3D'):
                c = spec['cropping']
                nd = 2 if cls == 'Cropping2D' else 3
                c = [(c, c)] * nd if isinstance(c, int) else [
                    (i, i) if isinstance(i, int) else tuple(i) for i in c]
                sl = [slice(None)] + [slice(lo, x.shape[1 + d] - hi)
                                      for d, (lo, hi) in enumerate(c)]
                x = x[tuple(sl)]
            elif cls == 'LeakyReLU':
                x = F.leaky_relu(x, spec.get('alpha', 0.3))
            elif cls == 'Activation':
                assert spec['activation'] == 'relu'
                x = F.relu(x)
            elif cls == 'SkipConnection':
                nm = spec['name']
                if nm in skips:
                    x = x + skips.pop(nm)
                else:
                    skips[nm] = x
            elif cls == 'SpatialExpansion':
                x = self._d2s(x, spec.get('spatial_mult', 1))
            elif cls == 'SpatioTemporalExpansion':
                m = spec.get('temporal_mult', 1)
                b = spec.get('spatial_mult', 1)
                if m > 1:
                    assert spec.get('temporal_method', 'nearest') == 'nearest'
                    x = torch.repeat_interleave(x, m, dim=3)
                if b > 1:
                    n, s1, s2, t, c = x.shape
                    xt = x.permute(0, 3, 1, 2, 4).reshape(n * t, s1, s2, c)
                    yt = self._d2s(xt, b)
                    x = yt.reshape(n, t, s1 * b, s2 * b, -1).permute(
                        0, 2, 3, 1, 4)
            elif cls == 'Flatten':
                x = x.reshape(x.shape[0], -1)
            elif cls == 'Dense':
                w, b = self.weights[wi], self.weights[wi + 1]
                wi += 2
                x = x @ w + b
            elif cls == 'Sup3rConcat':
                x = torch.cat((x, exo[spec['name']]), dim=-1)
            elif cls == 'Sup3rAdder':
                x = x + exo[spec['name']]
            else:
                raise KeyError(cls)
        assert wi == len(self.weights)
        return x

    @staticmethod
    def _d2s(x, b):
        """DCR depth_to_space via torch.pixel_shuffle (CRD) + explicit
        channel permutation: torch wants channel index c*b*b + i*b + j, TF's
        is (i*b + j)*Co + c."""
        n, h, w, c = x.shape
        co = c // (b * b)
        xc = x.reshape(n, h, w, b * b, co).permute(0, 4, 3, 1, 2)
        xc = xc.reshape(n, co * b * b, h, w)
        return F.pixel_shuffle(xc, b).permute(0, 2, 3, 1)
