"""Test fixtures: a synthetic batch handler implementing the BatchHandler
*protocol* sup3r's ``Sup3rGan.train`` consumes (attributes + iteration +
``DsetTuple``-like batches) without the reference's TF-backed
queue."""
import collections

import numpy as np

def switch(name, value=1):
    """Set (``None``: remove) a kernel-selection option of the GPU context —
    the default of every plan created afterwards (``include/sup3r_hip.h``
    "options"; a plan keeps the options it was created with).  The autouse
    fixture of ``tests/conftest.py`` removes whatever a test set."""
    from sup3r_amd.engine import Device
    Device.get().set_option(name, value)


Batch = collections.namedtuple('Batch', ['low_res', 'high_res'])
MomBatch = collections.namedtuple(
    'MomBatch', ['low_res', 'high_res', 'output', 'mask'])


def coarsen(hr, s, t):
    """spatial mean-coarsening + temporal subsampling (what
    SingleBatchQueue.transform does on the host, batch_queues/base.py:32-87)."""
    if hr.ndim == 5:
        n, a, b, c, f = hr.shape
        lr = hr.reshape(n, a // s, s, b // s, s, c, f).mean(axis=(2, 4))
        return lr[:, :, :, ::t]
    n, a, b, f = hr.shape
    return hr.reshape(n, a // s, s, b // s, s, f).mean(axis=(2, 4))


def smooth_field(rng, shape):
    """Random field with large-scale structure (so super-resolution is
    learnable): sum of a few random plane waves + small noise."""
    grids = np.meshgrid(*[np.linspace(0, 1, n) for n in shape[1:-1]],
                        indexing='ij')
    out = np.zeros(shape, np.float32)
    for n in range(shape[0]):
        for f in range(shape[-1]):
            v = 0
            for _ in range(3):
                k = rng.uniform(-6, 6, size=len(grids))
                ph = rng.uniform(0, 2 * np.pi)
                v = v + np.sin(sum(ki * g for ki, g in zip(k, grids)) + ph)
            out[n, ..., f] = v / 2 + 0.05 * rng.standard_normal(shape[1:-1])
    return out


class ValData:
    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class SyntheticBatchHandler:
    def __init__(self, sample_shape, s_enhance, t_enhance, features,
                 batch_size=4, n_batches=3, n_val=1, seed=0, exo_features=()):
        rng = np.random.default_rng(seed)
        self.s_enhance, self.t_enhance = s_enhance, t_enhance
        self.lr_features = list(features)
        self.hr_out_features = list(features)
        self.hr_exo_features = list(exo_features)
        self.smoothing = None
        self.smoothed_features = []
        nf = len(features) + len(exo_features)
        is_5d = len(sample_shape) == 3 and sample_shape[2] > 1
        hr_sp = tuple(sample_shape) if is_5d else tuple(sample_shape[:2])
        self.hr_shape = hr_sp + (nf,)
        lr_sp = (hr_sp[0] // s_enhance, hr_sp[1] // s_enhance) + (
            (hr_sp[2] // t_enhance,) if is_5d else ())
        self.lr_shape = lr_sp + (len(features),)
        self.shapes = ((batch_size,) + self.lr_shape,
                       (batch_size,) + self.hr_shape)
        self.means = {f: 0.0 for f in list(features) + list(exo_features)}
        self.stds = {f: 1.0 for f in list(features) + list(exo_features)}

        def make():
            hr = smooth_field(rng, (batch_size,) + self.hr_shape)
            lr = coarsen(hr[..., :len(features)], s_enhance,
                         t_enhance if is_5d else 1)
            return Batch(lr.astype(np.float32), hr.astype(np.float32))
        self.batches = [make() for _ in range(n_batches)]
        self.val_data = ValData([make() for _ in range(n_val)])
        self.stopped = False

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)

    def stop(self):
        self.stopped = True


class SyntheticMomBatchHandler(SyntheticBatchHandler):
    """Conditional-moment batches (batch_queues/conditional.py:79-167): first
    moment target = the hi-res field itself, mask of ones with an optional
    zeroed border."""

    def __init__(self, *args, pad=0, **kwargs):
        super().__init__(*args, **kwargs)

        def to_mom(b):
            mask = np.ones_like(b.high_res)
            if pad:
                mask[:, :pad] = 0
                mask[:, -pad:] = 0
            return MomBatch(b.low_res, b.high_res, b.high_res.copy(), mask)
        self.batches = [to_mom(b) for b in self.batches]
        self.val_data = ValData([to_mom(b) for b in self.val_data.batches])


# --------------------------------------------------------------------------
# bf16 emulation: make the numpy oracle do the roundings a HIP plan does
# --------------------------------------------------------------------------
_BF16_WGRAD = ('bf16_trunk', 'bf16_gen', 'bf16_2d', 'c2', 'tail')
_BF16_DGRAD = ('mfma_frame', 'mfma_valid', 'mfma_chunked', 'fewch_frame', 's2',
               'c2', 'gconv')


def emulate_plan(ref, ph, masks=False, rounding=True, sample=slice(None)):
    """Configure the oracle network ``ref`` (oracle.network.Network over the
    same ``hidden_layers``) to reproduce the numerics of the HIP plan handle
    ``ph`` (sup3r_amd.engine.PlanHandle):

    * ``rounding``: a conv whose device kernel rounds its operands to bf16
      (``s3_plan_op_info``: forward, weight gradient, data gradient) does the
      same in the oracle, and a tensor the plan STORES as bf16 is rounded
      after the last layer of its fused group;
    * ``masks``: the sign pattern of every activation in the backward pass is
      the one the device used (read from the training plan's saved
      activations), so a pre-activation within round-off of zero cannot flip
      a whole unit between the two backward passes.  Call after the device
      forward; ``sample`` selects the batch entries the oracle ran.

    Returns the number of (convs with bf16 operands, tensors stored as bf16,
    masks installed)."""
    from sup3r_amd import spec as S
    plan = ph.plan
    assert len(plan.layer_out) == len(ref.layers), \
        (len(plan.layer_out), len(ref.layers))
    groups = {}
    for li, (_, oi) in enumerate(plan.layer_out):
        if oi >= 0:
            groups.setdefault(oi, []).append(li)
    n_ops = n_store = n_mask = 0
    is_bf16_plan = ph.precision == 1          # S3_PREC_BF16
    for oi, lis in groups.items():
        op = plan.ops[oi]
        last = lis[-1]
        if rounding and ph.tensor_is_bf16(op['out']):
            ref.emu_store_round.add(last)
            n_store += 1
        wl = [ref.layers[li] for li in lis
              if hasattr(ref.layers[li], 'kernel')]
        if op['kind'] == S.OP_CONV and rounding:
            info = ph.op_info(oi)
            assert len(wl) == 1, (oi, lis)
            wl[0].emu_fwd_round = bool(info['fwd_bf16_ops'])
            wl[0].emu_wgrad_round = is_bf16_plan and info['wgrad'] in _BF16_WGRAD
            wl[0].emu_dgrad_round = is_bf16_plan and info['dgrad'] in _BF16_DGRAD
            n_ops += int(info['fwd_bf16_ops'])
            if is_bf16_plan and info.get('dgrad_frame16'):
                # the conv's padded-frame data gradient is stored as bf16: the
                # pad layer of the group folds rounded values
                # (a pad fused into the conv produces no op of its own: it
                # sits right before the group, with op index -1)
                li = min(lj for lj in lis if hasattr(ref.layers[lj], 'kernel'))
                pads = []
                while li > 0 and not pads:
                    li -= 1
                    if type(ref.layers[li]).__name__ == 'FlexiblePadding':
                        pads.append(ref.layers[li])
                    elif li not in lis and plan.layer_out[li][1] >= 0:
                        break
                assert len(pads) == 1, (oi, lis)
                pads[0].emu_grad_round = True
        if masks and op.get('act', 0):
            y = ph.tensor(op['out'])[sample]
            acts = [ref.layers[li] for li in lis
                    if type(ref.layers[li]).__name__ in ('LeakyReLU',
                                                         'Activation')]
            tgt = acts[0] if acts else (wl[0] if wl else None)
            assert tgt is not None, (oi, lis)
            want = tgt._pre.shape if hasattr(tgt, 'kernel') else tgt._x.shape
            m = y > 0
            b = int(op.get('d2s', 1) or 1)
            if b > 1 and hasattr(tgt, 'kernel'):
                # the activation of a conv that also carries the
                # depth-to-space store: the device tensor is the permuted one,
                # out[n, h b + i, w b + j, t, c] = pre[n, h, w, t, (i b + j) Co
                # + c] (DCR) — undo the permutation for the conv's own mask
                n_, s1, s2, t_, co = m.shape
                m = m.reshape(n_, s1 // b, b, s2 // b, b, t_, co)
                m = m.transpose(0, 1, 3, 5, 2, 4, 6).reshape(
                    n_, s1 // b, s2 // b, t_, b * b * co)
            if m.size != int(np.prod(want)):
                # a conv with a fused activation kwarg followed by a crop
                # (Conv2DTranspose(activation=relu) + Cropping2D): the oracle
                # masks BEFORE the crop, the device stores after it
                crops = [ref.layers[li] for li in lis
                         if type(ref.layers[li]).__name__ == 'Cropping']
                assert len(crops) == 1, (oi, lis)
                m = embed_mask(want, m, crops[0].cropping)
            tgt.emu_mask = m.reshape(want)
            n_mask += 1
    return n_ops, n_store, n_mask


def rel_linf(a, b):
    """max |a - b| / max(1, max |b|)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def rel_max(a, b):
    """max |a - b| / max |b|"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean())
                 / max(1e-30, np.sqrt((b ** 2).mean())))


def teacher_forced_check(ref, ph, x, exo=None, sample=slice(None),
                         allow_missing=False):
    """Per-op parity of a whole network, free of error propagation.

    A deep stack in bf16 is chaotic at the rounding level: two correct
    implementations that accumulate in a different order flip a few roundings
    per layer, every flip perturbs ~1700 downstream sums by ~1e-4 relative,
    which flips ~2.5 % of THEIR roundings, and after a handful of layers the
    two results are decorrelated at the full bf16 error (measured: the
    bf16-emulating oracle with fp32 vs fp64 accumulation differs from itself
    by 1.8e-2 on the 37-conv generator).  An end-to-end bound can therefore
    never separate "kernel wrong" from "mode imprecise".  This check can: the
    oracle (configured by ``emulate_plan`` to round where the device rounds)
    walks the layer list, and at the end of every fused group its result is
    compared with the tensor the DEVICE stored for that op — then REPLACED by
    it, so the next group starts from exactly the device's input.

    Needs a training plan (keeps every activation) after its forward — or an
    inference plan created with the option ``KEEP_ACTIVATIONS``; with
    ``allow_missing`` an op whose output the device never materialises (a
    concat / skip add fused into a conv: ``s3_plan_tensor_read`` says "no
    buffer") is stepped over — the oracle carries its own value to the next
    op, whose output then covers both.
    Returns one dict per op: ``frac`` of elements that differ from the
    device value (after the same storage rounding; fp32-stored tensors: by
    more than the accumulation noise), ``excess`` = the largest difference
    BEYOND one bf16 spacing of the value (bf16-stored) or the largest
    difference (fp32-stored), both relative to the tensor's scale — to be
    compared with ``noise`` (2e-5, fp32 accumulation order) —, ``bf16``.
    Afterwards the oracle's cached layer inputs ARE the device's activations:
    ``ref.backward`` is then a backward pass over identical operands."""
    from oracle import layers as L
    from sup3r_amd import spec as S
    plan = ph.plan
    groups = {}
    for li, (_, oi) in enumerate(plan.layer_out):
        if oi >= 0:
            groups.setdefault(oi, []).append(li)
    last_to_op = {lis[-1]: oi for oi, lis in groups.items()}
    for layer in ref.layers:
        if isinstance(layer, L.SkipConnection):
            layer._cache = None
            layer._dcache = None
            layer._fwd_roles = []
    stats = []
    carry = None
    h = np.asarray(x, np.float32)[sample]
    for i, layer in enumerate(ref.layers):
        if isinstance(layer, (L.Sup3rConcat, L.Sup3rAdder)):
            e = None if exo is None else exo.get(layer.name)
            h = layer.forward(h, None if e is None else e[sample])
        else:
            h = layer.forward(h)
        if i not in last_to_op:
            continue
        oi = last_to_op[i]
        op = plan.ops[oi]
        try:
            dev = ph.tensor(op['out'])
        except RuntimeError as e:
            if allow_missing and 'no buffer' in str(e):
                stats.append(dict(op=oi, kind=op['kind'], skipped=True))
                # (a value the device rounds in registers on its way into
                # the fused consumer: emulate_plan of the unfused variant of
                # the plan put the layer into emu_store_round)
                if i in ref.emu_store_round:
                    h = L.round_bf16(h)
                    # that rounding may flip too: one spacing of THIS value is
                    # carried into the bound of the op that covers it
                    carry = np.exp2(np.floor(np.log2(np.maximum(
                        np.abs(h).astype(np.float64), 1e-30))) - 7)
                continue
            raise
        dev = dev.reshape((-1,) + h.shape[1:])[sample]
        is16 = ph.tensor_is_bf16(op['out'])
        mine = L.round_bf16(h) if is16 else h
        diff = np.abs(mine.astype(np.float64) - dev)
        scale = max(1.0, float(np.abs(dev).max()))
        # fp32 accumulation in a different order moves a sum by ~1e-6 of the
        # magnitude of its terms (the tensor's scale), whatever the sum is
        noise = 2e-5 * scale
        if is16:
            # spacing of bf16 numbers around |v|: 2^(floor(log2|v|) - 7); a
            # sum that lands on the other side of a rounding boundary differs
            # by exactly one spacing
            mag = np.maximum(np.maximum(np.abs(dev), np.abs(mine)).astype(
                np.float64), 1e-30)
            ulp = np.exp2(np.floor(np.log2(mag)) - 7)
            if carry is not None:
                ulp = ulp + (carry if carry.shape == ulp.shape
                             else float(carry.max()))
            bad = diff > 0
            excess = float(((diff - ulp) / scale).max())
        else:
            bad = diff > noise
            excess = float((diff / scale).max())
        stats.append(dict(op=oi, kind=op['kind'], bf16=bool(is16),
                          frac=float(bad.mean()), excess=excess,
                          noise=noise / scale, shape=tuple(h.shape)))
        h = dev.astype(np.float32)
        carry = None
    return stats


def embed_mask(tgt_shape, dev_mask, crop):
    """device mask of a CROPPED tensor placed into the un-cropped shape the
    oracle layer works on (positions outside the crop get no gradient)"""
    full = np.zeros(tgt_shape, bool)
    sl = [slice(None)] + [slice(lo, n - hi) for (lo, hi), n in
                          zip(crop, tgt_shape[1:-1])] + [slice(None)]
    full[tuple(sl)] = dev_mask.reshape(full[tuple(sl)].shape)
    return full
