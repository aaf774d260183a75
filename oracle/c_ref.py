"""The generator / discriminator forward AS TF EXECUTES IT, with the
convolutions on the C + OpenMP restatement (oracle/conv_ref.c) and the cheap
layers in numpy (TEST / BASELINE INFRASTRUCTURE).  Takes a built
``oracle.network.Network`` so that it computes the same function with the same
weights as the numpy oracle."""
import ctypes as C
import time

import numpy as np

from . import layers as L
from .build_ref import load

_PF = C.POINTER(C.c_float)


def conv_valid(lib, x, kernel, bias, strides):
    """keras Conv2D / Conv3D, padding valid, channels-last fp32"""
    nd = x.ndim - 2
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(kernel, np.float32)
    if nd == 2:
        x5 = x.reshape(x.shape[:3] + (1, x.shape[3]))
        k = tuple(w.shape[:2]) + (1,)
        s = tuple(strides) + (1,)
    else:
        x5, k, s = x, tuple(w.shape[:3]), tuple(strides)
    n, d0, d1, d2, cin = x5.shape
    cout = w.shape[-1]
    o = [(d - kk) // ss + 1 for d, kk, ss in zip((d0, d1, d2), k, s)]
    y = np.empty((n, o[0], o[1], o[2], cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = lib.s3ref_conv_valid(
        x5.ctypes.data_as(_PF), n, d0, d1, d2, cin, w.ctypes.data_as(_PF),
        b.ctypes.data_as(_PF) if b is not None else None, k[0], k[1], k[2],
        s[0], s[1], s[2], cout, y.ctypes.data_as(_PF))
    if rc != 0:
        raise RuntimeError('s3ref_conv_valid rejected the geometry')
    return y if nd == 3 else y.reshape(y.shape[:3] + (cout,))


def forward(net, x, exo=None, lib=None, native=False):
    """-> (y, seconds, build description).  ``net``: built oracle network."""
    if lib is None:
        lib, how = load(native=native)
    else:
        how = ''
    for layer in net.layers:
        if isinstance(layer, L.SkipConnection):
            layer._cache = None
            layer._dcache = None
            layer._fwd_roles = []
    t0 = time.time()
    x = np.asarray(x, np.float32)
    for layer in net.layers:
        if isinstance(layer, L.ConvND):
            pads = layer._pad_amounts(x.shape[1:1 + layer.nd])
            if any(p != (0, 0) for p in pads):
                x = np.pad(x, [(0, 0)] + list(pads) + [(0, 0)])
            x = conv_valid(lib, x, layer.kernel,
                           layer.bias if layer.use_bias else None,
                           layer.strides)
            x = L._act_forward(layer.activation, x)
        elif isinstance(layer, (L.Sup3rConcat, L.Sup3rAdder)):
            x = layer.forward(x, None if exo is None else exo.get(layer.name))
        else:
            x = layer.forward(x)
    return x, time.time() - t0, how
