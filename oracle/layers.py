"""Layer-by-layer numpy restatement of the keras / phygnn layers that the
sup3r generator + discriminator configs instantiate (TEST INFRASTRUCTURE).

Reference call sites: the eager layer loop of
``sup3r/models/abstract.py:1131-1173`` (``_tf_generate``) and
``sup3r/models/base.py:283-313`` (``_tf_discriminate``); layer classes named
in ``sup3r/configs/**.json`` and ``tests/data/config_disc_*_test.json``.
The arithmetic itself lives in tensorflow 2.15.1 / keras 2.15.0 /
nrel_phygnn 0.0.33 (pixi.lock:314,320,352) which are NOT vendored; their
published semantics are restated here.

Every layer has ``forward(x, *extra)`` and ``backward(dy)`` (returns dx and
fills ``self.grads`` in keras weight order).  Arrays are channels-last,
``(N, s1, s2, C)`` or ``(N, s1, s2, t, C)``; dtype follows the input so the
same code runs fp32 (parity) and fp64 (finite-difference checks).

Emulation switches (test infrastructure for the bf16 throughput mode of the
HIP path; all off by default, i.e. the reference's fp32 arithmetic):
``emu_fwd_round`` / ``emu_wgrad_round`` / ``emu_dgrad_round`` on a Conv layer round its matrix
operands to bfloat16 (round-to-nearest-even, as ``v_cvt_pk_bf16_f32``) before
the contraction, which then accumulates in the working dtype (run the network
in float64 to get the exact sum of the rounded products); ``emu_mask`` on an
activation (or a conv with a fused one) replaces the sign pattern used by the
backward pass — the tests pass the masks the DEVICE used, so that a
pre-activation within round-off of zero cannot make the two backward passes
differ by a whole flipped unit.
"""
import numpy as np


def round_bf16(x):
    """fp32 -> bfloat16 (round to nearest even) -> back, any float dtype."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    # in 32-bit arithmetic: the rounding increment cannot wrap for finite
    # values (biased exponent < 0xFF), and Inf / NaN keep their class
    u = a.view(np.uint32)
    r = (u >> np.uint32(16)) & np.uint32(1)
    r += np.uint32(0x7FFF)
    r += u
    r &= np.uint32(0xFFFF0000)
    special = (u & np.uint32(0x7F800000)) == np.uint32(0x7F800000)
    if special.any():
        r[special] = u[special] & np.uint32(0xFFFF0000)
        nan = special & ((u & np.uint32(0x007FFFFF)) != 0)
        r[nan] |= np.uint32(0x00400000)
    out = r.view(np.float32).reshape(a.shape)
    return out.astype(np.asarray(x).dtype, copy=False)


def _limit_blas_threads():
    """The oracle's GEMMs are tall and skinny (1e5 positions x 64 channels):
    OpenBLAS at its default of 64 threads on the 256-CPU GPU host ran them
    4 x SLOWER than at 8 (tools/dbg/blas_threads.py: 0.51 s vs 0.13 s per
    full-size C2 conv), which was most of the GPU suite's wall time."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=8, user_api='blas')
    except Exception:        # no threadpoolctl: slower, not wrong
        pass


_limit_blas_threads()


def _tuple(v, n):
    if isinstance(v, (int, np.integer)):
        return (int(v),) * n
    v = tuple(int(i) for i in v)
    assert len(v) == n
    return v


class Layer:
    """Base: no weights, identity."""

    weights = ()
    grads = ()
    name = None

    def forward(self, x, *extra):
        return x

    def backward(self, dy):
        return dy

    def __call__(self, x, *extra):
        return self.forward(x, *extra)


class FlexiblePadding(Layer):
    """phygnn FlexiblePadding -> ``tf.pad(x, paddings, mode)`` (K1).
    REFLECT mirrors without repeating the edge sample (== np.pad 'reflect')."""

    def __init__(self, paddings, mode='REFLECT', **_):
        self.paddings = [tuple(int(i) for i in p) for p in paddings]
        self.mode = mode.upper()
        self.rank = len(self.paddings)

    def _np_mode(self):
        return {'REFLECT': 'reflect', 'CONSTANT': 'constant',
                'SYMMETRIC': 'symmetric'}[self.mode]

    def forward(self, x, *extra):
        self._in_shape = x.shape
        return np.pad(x, self.paddings, mode=self._np_mode())

    def backward(self, dy):
        # adjoint of the gather x_pad = x[idx]: scatter-add through the same
        # index map, one axis at a time
        if getattr(self, 'emu_grad_round', False):
            # (the device keeps the padded-frame data gradient of the conv
            # behind this pad as bf16 between its kernel and the fold)
            dy = round_bf16(dy)
        dx = dy
        for ax, (lo, hi) in enumerate(self.paddings):
            if lo == 0 and hi == 0:
                continue
            n = self._in_shape[ax]
            if self.mode == 'CONSTANT':
                sl = [slice(None)] * dx.ndim
                sl[ax] = slice(lo, lo + n)
                dx = dx[tuple(sl)]
                continue
            idx = np.pad(np.arange(n), (lo, hi), mode=self._np_mode())
            dxm = np.moveaxis(dx, ax, 0)
            # the un-padded middle maps one to one (a bulk copy); the lo + hi
            # border planes are added one by one in index order — what
            # np.add.at(out, idx, dx) does, without its per-element dispatch
            # (6 s of a full-size C2 backward pass)
            outm = np.array(dxm[lo:lo + n])
            for j in list(range(lo)) + list(range(lo + n, lo + n + hi)):
                outm[idx[j]] += dxm[j]
            dx = np.moveaxis(outm, 0, ax)
        return dx


class Cropping(Layer):
    """keras Cropping2D / Cropping3D: symmetric int or per-side pairs."""

    def __init__(self, cropping, ndim_spatial, **_):
        if isinstance(cropping, (int, np.integer)):
            crop = [(int(cropping), int(cropping))] * ndim_spatial
        else:
            crop = []
            for c in cropping:
                if isinstance(c, (int, np.integer)):
                    crop.append((int(c), int(c)))
                else:
                    crop.append((int(c[0]), int(c[1])))
        assert len(crop) == ndim_spatial
        self.cropping = crop

    def forward(self, x, *extra):
        self._in_shape = x.shape
        sl = [slice(None)]
        for ax, (lo, hi) in enumerate(self.cropping):
            sl.append(slice(lo, x.shape[ax + 1] - hi))
        sl.append(slice(None))
        self._sl = tuple(sl)
        return x[self._sl]

    def backward(self, dy):
        dx = np.zeros(self._in_shape, dtype=dy.dtype)
        dx[self._sl] = dy
        return dx


def same_padding(n, k, s):
    """TF 'SAME' padding: out = ceil(n/s); extra pad goes on the END."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    lo = total // 2
    return lo, total - lo


def _act_forward(name, x, alpha=None):
    if name is None or name == 'linear':
        return x
    if name == 'relu':
        return np.where(x > 0, x, 0).astype(x.dtype)
    if name == 'leaky_relu':
        return np.where(x > 0, x, alpha * x).astype(x.dtype)
    if name == 'sigmoid':
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)
    if name == 'tanh':
        return np.tanh(x)
    raise KeyError(f'activation {name!r} not restated in the oracle')


def _act_backward(name, y_pre, y, dy, alpha=None, mask=None):
    if name is None or name == 'linear':
        return dy
    pos = (y_pre > 0) if mask is None else np.asarray(mask, bool)
    if name == 'relu':
        return dy * pos
    if name == 'leaky_relu':
        return dy * np.where(pos, 1.0, alpha).astype(dy.dtype)
    if name == 'sigmoid':
        return dy * y * (1 - y)
    if name == 'tanh':
        return dy * (1 - y * y)
    raise KeyError(name)


class ConvND(Layer):
    """keras Conv2D / Conv3D (K2, K3, K5): cross-correlation, kernel layout
    ``(k_1, ..., k_d, C_in, C_out)``, ``padding`` valid|same, bias add, optional
    fused ``activation`` kwarg."""

    def __init__(self, nd, filters, kernel_size, strides=1, padding='valid',
                 activation=None, use_bias=True, **_):
        self.nd = nd
        self.filters = int(filters)
        self.kernel_size = _tuple(kernel_size, nd)
        self.strides = _tuple(strides, nd)
        self.padding = padding.lower()
        self.activation = activation
        self.use_bias = use_bias
        self.rank = nd + 2
        self.kernel = None
        self.bias = None

    def build(self, cin, rng=None, dtype=np.float32):
        shape = self.kernel_size + (cin, self.filters)
        self.kernel = glorot_uniform(shape, rng, dtype)
        self.bias = np.zeros((self.filters,), dtype=dtype)

    @property
    def weights(self):
        return [self.kernel, self.bias] if self.use_bias else [self.kernel]

    def _pad_amounts(self, in_sp):
        if self.padding == 'valid':
            return [(0, 0)] * self.nd
        return [same_padding(n, k, s) for n, k, s in
                zip(in_sp, self.kernel_size, self.strides)]

    def forward(self, x, *extra):
        if self.kernel is None:
            self.build(x.shape[-1], dtype=x.dtype)
        nd = self.nd
        in_sp = x.shape[1:1 + nd]
        pads = self._pad_amounts(in_sp)
        self._pads = pads
        self._in_shape = x.shape
        if any(p != (0, 0) for p in pads):
            x = np.pad(x, [(0, 0)] + pads + [(0, 0)], mode='constant')
        self._xp = x
        out_sp = tuple((x.shape[1 + d] - self.kernel_size[d]) // self.strides[d]
                       + 1 for d in range(nd))
        self._out_sp = out_sp
        n = x.shape[0]
        cin = x.shape[-1]
        w = self.kernel.astype(x.dtype, copy=False)
        if getattr(self, 'emu_fwd_round', False):
            x, w = round_bf16(x), round_bf16(w)
        if all(s == 1 for s in self.strides):
            # unit strides: in the FLAT index of the padded array a tap is a
            # constant offset, so x[p + tap] for every output position p is
            # the contiguous slice x2[off:] — one GEMM per tap on views, no
            # gather copy (the per-tap window copies of the general path were
            # most of the oracle's time at full C2 size).  Rows of the result
            # that are not output positions (p beyond an extent) are cropped.
            x2 = np.ascontiguousarray(x).reshape(-1, cin)
            ypad = np.zeros((x2.shape[0], self.filters), dtype=x.dtype)
            for tap in np.ndindex(*self.kernel_size):
                off = self._flat_offset(tap, x.shape)
                m = x2.shape[0] - off
                ypad[:m] += x2[off:] @ w[tap]
            crop = (slice(None),) + tuple(slice(0, o) for o in out_sp)
            y = np.ascontiguousarray(
                ypad.reshape(x.shape[:-1] + (self.filters,))[crop])
        else:
            y = np.zeros((n * int(np.prod(out_sp)), self.filters),
                         dtype=x.dtype)
            for tap in np.ndindex(*self.kernel_size):
                xs = x[self._tap_slices(tap)]
                y += xs.reshape(-1, cin) @ w[tap]
            y = y.reshape((n,) + out_sp + (self.filters,))
        if self.use_bias:
            y = y + self.bias.astype(x.dtype, copy=False)
        self._pre = y
        self._y = _act_forward(self.activation, y)
        return self._y

    def _flat_offset(self, tap, xshape):
        """offset of ``tap`` in the flat position index of the padded array"""
        off, stride = 0, 1
        for d in range(self.nd - 1, -1, -1):
            off += tap[d] * stride
            stride *= xshape[1 + d]
        return off

    def _tap_slices(self, tap):
        sl = [slice(None)]
        for d in range(self.nd):
            s = self.strides[d]
            sl.append(slice(tap[d], tap[d] + (self._out_sp[d] - 1) * s + 1, s))
        sl.append(slice(None))
        return tuple(sl)

    def backward(self, dy):
        dy = _act_backward(self.activation, self._pre, self._y, dy,
                           mask=getattr(self, 'emu_mask', None))
        x = self._xp
        cin = x.shape[-1]
        w = self.kernel.astype(dy.dtype, copy=False)
        dy2 = dy.reshape(-1, self.filters)
        # (float64 accumulation: numpy's float32 reduction along axis 0 is a
        # sequential running sum — 1.8 M equal-magnitude terms of an MAE
        # gradient bias it by ~1e-3)
        db = dy2.sum(axis=0, dtype=np.float64).astype(dy2.dtype)
        # (device kernels: the weight gradient rounds x and dPre, the data
        # gradient dPre and w)
        dyw = dyd = dy2
        if getattr(self, 'emu_wgrad_round', False):
            x, dyw = round_bf16(x), round_bf16(dy2)
        if getattr(self, 'emu_dgrad_round', False):
            w, dyd = round_bf16(w), round_bf16(dy2)
        dw = np.zeros(self.kernel.shape, dtype=dy.dtype)
        dxp = np.zeros(x.shape, dtype=dy.dtype)
        if all(s_ == 1 for s_ in self.strides):
            # (the flat-offset form of forward(): dPre embedded in a zero
            # array of the padded extent, contiguous slices on both sides)
            emb = (slice(None),) + tuple(slice(0, o) for o in self._out_sp)

            def embed(a2):
                e = np.zeros(x.shape[:-1] + (self.filters,), dtype=a2.dtype)
                e[emb] = a2.reshape(dy.shape)
                return e.reshape(-1, self.filters)
            ew = embed(dyw)
            ed = ew if dyd is dyw else embed(dyd)
            x2 = np.ascontiguousarray(x).reshape(-1, cin)
            dx2 = dxp.reshape(-1, cin)
            for tap in np.ndindex(*self.kernel_size):
                off = self._flat_offset(tap, x.shape)
                m = x2.shape[0] - off
                dw[tap] = x2[off:].T @ ew[:m]
                dx2[off:] += ed[:m] @ w[tap].T
        else:
            for tap in np.ndindex(*self.kernel_size):
                sl = self._tap_slices(tap)
                xs = x[sl]
                dw[tap] = xs.reshape(-1, cin).T @ dyw
                dxp[sl] += (dyd @ w[tap].T).reshape(xs.shape)
        self.grads = [dw, db] if self.use_bias else [dw]
        sl = [slice(None)]
        for d, (lo, hi) in enumerate(self._pads):
            sl.append(slice(lo, dxp.shape[1 + d] - hi))
        sl.append(slice(None))
        return dxp[tuple(sl)]


class ConvTransposeND(Layer):
    """keras Conv2DTranspose / Conv3DTranspose (K4), padding valid, no output
    padding: ``y[n, i*s + k, co] += x[n, i, ci] * w[k, co, ci]``; kernel layout
    ``(k_1, ..., k_d, C_out, C_in)``; out = (in - 1) * s + k."""

    def __init__(self, nd, filters, kernel_size, strides=1, padding='valid',
                 activation=None, use_bias=True, **_):
        assert padding.lower() == 'valid', 'only valid padding is restated'
        self.nd = nd
        self.filters = int(filters)
        self.kernel_size = _tuple(kernel_size, nd)
        self.strides = _tuple(strides, nd)
        self.activation = activation
        self.use_bias = use_bias
        self.rank = nd + 2
        self.kernel = None
        self.bias = None

    def build(self, cin, rng=None, dtype=np.float32):
        shape = self.kernel_size + (self.filters, cin)
        self.kernel = glorot_uniform(shape, rng, dtype)
        self.bias = np.zeros((self.filters,), dtype=dtype)

    @property
    def weights(self):
        return [self.kernel, self.bias] if self.use_bias else [self.kernel]

    def _tap_slices(self, tap, in_sp):
        sl = [slice(None)]
        for d in range(self.nd):
            s = self.strides[d]
            sl.append(slice(tap[d], tap[d] + (in_sp[d] - 1) * s + 1, s))
        sl.append(slice(None))
        return tuple(sl)

    def forward(self, x, *extra):
        if self.kernel is None:
            self.build(x.shape[-1], dtype=x.dtype)
        nd = self.nd
        in_sp = x.shape[1:1 + nd]
        out_sp = tuple((in_sp[d] - 1) * self.strides[d] + self.kernel_size[d]
                       for d in range(nd))
        self._x = x
        w = self.kernel.astype(x.dtype, copy=False)
        if getattr(self, 'emu_fwd_round', False):
            x, w = round_bf16(x), round_bf16(w)
        y = np.zeros((x.shape[0],) + out_sp + (self.filters,), dtype=x.dtype)
        x2 = x.reshape(-1, x.shape[-1])
        for tap in np.ndindex(*self.kernel_size):
            sl = self._tap_slices(tap, in_sp)
            y[sl] += (x2 @ w[tap].T).reshape(x.shape[:-1] + (self.filters,))
        if self.use_bias:
            y = y + self.bias.astype(x.dtype, copy=False)
        self._pre = y
        self._y = _act_forward(self.activation, y)
        return self._y

    def backward(self, dy):
        dy = _act_backward(self.activation, self._pre, self._y, dy,
                           mask=getattr(self, 'emu_mask', None))
        x = self._x
        in_sp = x.shape[1:1 + self.nd]
        w = self.kernel.astype(dy.dtype, copy=False)
        x2 = x.reshape(-1, x.shape[-1])
        db = dy.reshape(-1, self.filters).sum(axis=0, dtype=np.float64).astype(
            dy.dtype)
        dyw = dyd = dy
        if getattr(self, 'emu_wgrad_round', False):
            x2, dyw = round_bf16(x2), round_bf16(dy)
        if getattr(self, 'emu_dgrad_round', False):
            w, dyd = round_bf16(w), round_bf16(dy)
        dw = np.zeros(self.kernel.shape, dtype=dy.dtype)
        dx = np.zeros(x2.shape, dtype=dy.dtype)
        for tap in np.ndindex(*self.kernel_size):
            sl = self._tap_slices(tap, in_sp)
            dw[tap] = dyw[sl].reshape(-1, self.filters).T @ x2
            dx += dyd[sl].reshape(-1, self.filters) @ w[tap]
        self.grads = [dw, db] if self.use_bias else [dw]
        return dx.reshape(x.shape)


class LeakyReLU(Layer):
    """keras LeakyReLU(alpha) (K6); keras-2.15 default alpha is 0.3."""

    def __init__(self, alpha=0.3, **_):
        self.alpha = float(alpha)

    def forward(self, x, *extra):
        self._x = x
        return np.where(x > 0, x, self.alpha * x).astype(x.dtype)

    def backward(self, dy):
        m = getattr(self, 'emu_mask', None)
        pos = (self._x > 0) if m is None else np.asarray(m, bool)
        return dy * np.where(pos, 1.0, self.alpha).astype(dy.dtype)


class Activation(Layer):
    """keras Activation(name)."""

    def __init__(self, activation, **_):
        self.activation = activation

    def forward(self, x, *extra):
        self._x = x
        self._y = _act_forward(self.activation, x)
        return self._y

    def backward(self, dy):
        return _act_backward(self.activation, self._x, self._y, dy,
                             mask=getattr(self, 'emu_mask', None))


def depth_to_space(x, b):
    """tf.nn.depth_to_space, NHWC, DCR channel order (K8):
    out[n, h*b+i, w*b+j, c] = in[n, h, w, (i*b + j)*C_out + c]."""
    n, h, w, c = x.shape
    co = c // (b * b)
    y = x.reshape(n, h, w, b, b, co)
    y = y.transpose(0, 1, 3, 2, 4, 5)
    return y.reshape(n, h * b, w * b, co)


def space_to_depth(y, b):
    """Adjoint / inverse of :func:`depth_to_space`."""
    n, hb, wb, co = y.shape
    h, w = hb // b, wb // b
    x = y.reshape(n, h, b, w, b, co)
    x = x.transpose(0, 1, 3, 2, 4, 5)
    return x.reshape(n, h, w, b * b * co)


class SpatialExpansion(Layer):
    """phygnn SpatialExpansion (4-D): depth_to_space(x, spatial_mult)."""

    def __init__(self, spatial_mult=1, **_):
        self._spatial_mult = int(spatial_mult)

    def forward(self, x, *extra):
        if x.shape[-1] % self._spatial_mult ** 2 != 0:
            raise RuntimeError(
                'Spatial expansion of factor {} is being attempted on input '
                'tensor of shape {}, but the last dimension of the input '
                'tensor ({}) must be divisible by the spatial factor squared '
                '({}).'.format(self._spatial_mult, x.shape, x.shape[-1],
                               self._spatial_mult ** 2))
        return depth_to_space(x, self._spatial_mult)

    def backward(self, dy):
        return space_to_depth(dy, self._spatial_mult)


class SpatioTemporalExpansion(Layer):
    """phygnn SpatioTemporalExpansion (5-D), K7 + K8: temporal first
    (``tf.image.resize(method='nearest')`` per s1-slice on (N, s2, t, C) ->
    integer factor m: out[..., j, :] = in[..., j // m, :]), then per-time-step
    depth_to_space(spatial_mult)."""

    def __init__(self, spatial_mult=1, temporal_mult=1,
                 temporal_method='nearest', t_roll=0, **_):
        self._spatial_mult = int(spatial_mult)
        self._temporal_mult = int(temporal_mult)
        self._temporal_meth = temporal_method
        self._t_roll = int(t_roll)
        if self._temporal_mult > 1 and temporal_method not in (
                'nearest', 'depth_to_time'):
            raise KeyError(f'temporal_method {temporal_method!r} is not '
                           'restated in the oracle')

    def forward(self, x, *extra):
        assert x.ndim == 5
        m, b = self._temporal_mult, self._spatial_mult
        self._in_shape = x.shape
        if m > 1:
            if self._temporal_meth == 'depth_to_time':
                n, s1, s2, t, c = x.shape
                x = x.reshape(n, s1, s2, t * m, c // m)
                x = np.roll(x, self._t_roll, axis=3)
            else:
                x = np.repeat(x, m, axis=3)
        if b > 1:
            n, s1, s2, t, c = x.shape
            if c % (b * b) != 0:
                raise RuntimeError('channels not divisible by spatial_mult^2')
            xt = np.moveaxis(x, 3, 1).reshape(n * t, s1, s2, c)
            yt = depth_to_space(xt, b)
            x = np.moveaxis(yt.reshape(n, t, s1 * b, s2 * b, c // (b * b)),
                            1, 3)
        return np.ascontiguousarray(x)

    def backward(self, dy):
        m, b = self._temporal_mult, self._spatial_mult
        if b > 1:
            n, s1b, s2b, t, co = dy.shape
            dt = np.moveaxis(dy, 3, 1).reshape(n * t, s1b, s2b, co)
            dx = space_to_depth(dt, b)
            dy = np.moveaxis(dx.reshape(n, t, s1b // b, s2b // b, co * b * b),
                             1, 3)
        if m > 1:
            if self._temporal_meth == 'depth_to_time':
                dy = np.roll(dy, -self._t_roll, axis=3)
                dy = dy.reshape(self._in_shape)
            else:
                n, s1, s2, tm, c = dy.shape
                dy = dy.reshape(n, s1, s2, tm // m, m, c).sum(axis=4)
        return np.ascontiguousarray(dy)


class SkipConnection(Layer):
    """phygnn SkipConnection (K9): first call caches its input and returns it
    unchanged; the second call (same instance, shared by ``name``) returns
    ``x + cache`` and clears the cache."""

    def __init__(self, name, **_):
        self.name = name
        self._cache = None
        self._dcache = None

    def forward(self, x, *extra):
        if self._cache is None:
            self._cache = x
            self._fwd_roles = getattr(self, '_fwd_roles', [])
            self._fwd_roles.append('start')
            return x
        out = x + self._cache
        self._cache = None
        self._fwd_roles.append('end')
        return out

    def backward(self, dy):
        # called in reverse order: 'end' first (stash the branch gradient),
        # then 'start' (join it)
        role = self._fwd_roles.pop()
        if role == 'end':
            self._dcache = dy
            return dy
        out = dy + self._dcache
        self._dcache = None
        return out


class Flatten(Layer):
    """keras Flatten: row-major over (s1, s2[, t], C)."""

    def forward(self, x, *extra):
        self._in_shape = x.shape
        return x.reshape(x.shape[0], -1)

    def backward(self, dy):
        return dy.reshape(self._in_shape)


class Dense(Layer):
    """keras Dense (K10): y = x @ W + b, W:(in, units)."""

    def __init__(self, units, activation=None, use_bias=True, **_):
        self.units = int(units)
        self.activation = activation
        self.use_bias = use_bias
        self.kernel = None
        self.bias = None

    def build(self, cin, rng=None, dtype=np.float32):
        self.kernel = glorot_uniform((cin, self.units), rng, dtype)
        self.bias = np.zeros((self.units,), dtype=dtype)

    @property
    def weights(self):
        return [self.kernel, self.bias] if self.use_bias else [self.kernel]

    def forward(self, x, *extra):
        if self.kernel is None:
            self.build(x.shape[-1], dtype=x.dtype)
        self._x = x
        w = self.kernel.astype(x.dtype, copy=False)
        if getattr(self, 'emu_fwd_round', False):
            x, w = round_bf16(x), round_bf16(w)
        y = x @ w
        if self.use_bias:
            y = y + self.bias.astype(x.dtype, copy=False)
        self._pre = y
        self._y = _act_forward(self.activation, y)
        return self._y

    def backward(self, dy):
        dy = _act_backward(self.activation, self._pre, self._y, dy,
                           mask=getattr(self, 'emu_mask', None))
        x2 = self._x.reshape(-1, self._x.shape[-1])
        dy2 = dy.reshape(-1, self.units)
        dw = x2.T @ dy2
        self.grads = [dw, dy2.sum(axis=0, dtype=np.float64).astype(
            dy2.dtype)] if self.use_bias else [dw]
        return (dy2 @ self.kernel.astype(dy.dtype, copy=False).T
                ).reshape(self._x.shape)


class Sup3rConcat(Layer):
    """phygnn Sup3rConcat (K11): concat([x, hi_res_exo], axis=-1)."""

    def __init__(self, name, **_):
        self.name = name

    def forward(self, x, hi_res_adder=None, *extra):
        self._nx = x.shape[-1]
        if hi_res_adder is None:
            return x
        return np.concatenate((x, hi_res_adder.astype(x.dtype)), axis=-1)

    def backward(self, dy):
        return dy[..., :self._nx]


class Sup3rAdder(Layer):
    """phygnn Sup3rAdder (K11): x + hi_res_exo (broadcast on channels)."""

    def __init__(self, name, **_):
        self.name = name

    def forward(self, x, hi_res_adder=None, *extra):
        if hi_res_adder is None:
            return x
        return x + hi_res_adder.astype(x.dtype)

    def backward(self, dy):
        return dy


def glorot_uniform(shape, rng=None, dtype=np.float32):
    """keras glorot_uniform: limit = sqrt(6 / (fan_in + fan_out)) with
    fan_in = shape[-2] * receptive, fan_out = shape[-1] * receptive."""
    rng = rng if rng is not None else np.random.default_rng(0)
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)
