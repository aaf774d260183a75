"""numpy interpreter of the fused device plan (sup3r_amd.spec.Plan) — test
infrastructure that lets the fusion algebra (virtual reflect padding, flipped
ConvTranspose kernels, epilogue / depth-to-space fusion) be checked against the
layer-by-layer oracle on CPU, with no GPU and no HIP library involved.  It
restates, in the simplest possible form, what each C-ABI op must compute."""
import numpy as np

from sup3r_amd import spec as S


def _src_index(n_out, n_in, k, stride, lo, mode):
    """(n_out, k) source indices and validity mask for one spatial dim."""
    o = np.arange(n_out)[:, None]
    t = np.arange(k)[None, :]
    idx = o * stride + t - lo
    if mode == S.PAD_REFLECT:
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx > n_in - 1, 2 * (n_in - 1) - idx, idx)
        valid = np.ones_like(idx, dtype=bool)
        assert idx.min() >= 0 and idx.max() <= n_in - 1
    else:
        valid = (idx >= 0) & (idx <= n_in - 1)
        idx = np.clip(idx, 0, n_in - 1)
    return idx, valid


def _act(y, act, alpha):
    if act == S.ACT_RELU:
        return np.where(y > 0, y, 0).astype(y.dtype)
    if act == S.ACT_LEAKY:
        return np.where(y > 0, y, alpha * y).astype(y.dtype)
    return y


def d2s5(y, b):
    n, h, w, t, c = y.shape
    co = c // (b * b)
    y = y.reshape(n, h, w, t, b, b, co).transpose(0, 1, 4, 2, 5, 3, 6)
    return y.reshape(n, h * b, w * b, t, co)


def run_plan(plan, params, inputs, dtype=np.float64):
    """params: canonical-layout arrays in plan.params order; inputs: dict name
    -> keras-view array.  Returns keras-view output."""
    T = [None] * len(plan.tensors)
    for name, tid in plan.inputs.items():
        T[tid] = np.asarray(inputs[name], dtype=dtype).reshape(
            plan.tensors[tid])
    for op in plan.ops:
        kind = op['kind']
        x = T[op['in0']]
        osh = plan.tensors[op['out']]
        if kind == S.OP_CONV:
            w = np.asarray(params[op['w']], dtype=dtype)
            k = op['k']
            w = w.reshape(k[0], k[1], k[2], op['cin'], op['cout'])
            b = op['d2s']
            sp = [osh[1] // b, osh[2] // b, osh[3]]
            y = np.zeros((x.shape[0], sp[0], sp[1], sp[2], op['cout']), dtype)
            idx = [_src_index(sp[d], x.shape[1 + d], k[d], op['stride'][d],
                              op['lo'][d], op['pad_mode']) for d in range(3)]
            for a in range(k[0]):
                for bb in range(k[1]):
                    for c in range(k[2]):
                        i0, v0 = idx[0][0][:, a], idx[0][1][:, a]
                        i1, v1 = idx[1][0][:, bb], idx[1][1][:, bb]
                        i2, v2 = idx[2][0][:, c], idx[2][1][:, c]
                        xs = x[:, i0][:, :, i1][:, :, :, i2]
                        m = (v0[:, None, None] & v1[None, :, None]
                             & v2[None, None, :])
                        xs = xs * m[None, :, :, :, None]
                        y += xs @ w[a, bb, c]
            if op['b'] >= 0:
                y = y + np.asarray(params[op['b']], dtype=dtype)
            if b > 1:
                y = d2s5(y, b)
            y = _act(y, op['act'], op['alpha'])
            if op['res'] >= 0:
                y = y + T[op['res']]
            T[op['out']] = y
        elif kind == S.OP_REPEAT_T:
            T[op['out']] = np.repeat(x, op['rep'], axis=3)
        elif kind == S.OP_D2S:
            T[op['out']] = d2s5(x, op['d2s'])
        elif kind == S.OP_ACT:
            T[op['out']] = _act(x, op['act'], op['alpha'])
        elif kind == S.OP_ADD:
            T[op['out']] = x + T[op['in1']]
        elif kind == S.OP_CONCAT:
            T[op['out']] = np.concatenate((x, T[op['in1']]), axis=-1)
        elif kind == S.OP_DENSE:
            w = np.asarray(params[op['w']], dtype=dtype)
            y = x @ w
            if op['b'] >= 0:
                y = y + np.asarray(params[op['b']], dtype=dtype)
            T[op['out']] = _act(y, op['act'], op['alpha'])
        elif kind == S.OP_PAD:
            pw = [(0, 0)] + [(op['lo'][d], op['hi'][d]) for d in range(3)] \
                + [(0, 0)]
            mode = 'reflect' if op['pad_mode'] == S.PAD_REFLECT else 'constant'
            T[op['out']] = np.pad(x, pw, mode=mode)
        elif kind == S.OP_CROP:
            lo = op['lo']
            T[op['out']] = x[:, lo[0]:lo[0] + osh[1], lo[1]:lo[1] + osh[2],
                             lo[2]:lo[2] + osh[3]]
        elif kind == S.OP_DILATE:
            st = op['stride']
            y = np.zeros(osh, dtype=x.dtype)
            y[:, ::st[0], ::st[1], ::st[2]] = x
            T[op['out']] = y
        elif kind == S.OP_VIEW:
            T[op['out']] = x.reshape(osh)
        elif kind == S.OP_ROLL_T:
            T[op['out']] = np.roll(x, op['rep'], axis=3)
        else:
            raise KeyError(kind)
        assert list(T[op['out']].shape) == list(osh), (op, T[op['out']].shape)
    return T[plan.output].reshape(plan.out_shape)
