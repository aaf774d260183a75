#!/usr/bin/env python
"""Regenerates ``tests/golden/tf_*.npz`` / ``tf_slicer_C3.json`` FROM THE
REFERENCE ITSELF — the only route from "parity unpinned" to pinned.

Run on a machine that has the reference's environment (``pip install
NREL-sup3r`` = tensorflow 2.15.1, keras 2.15.0, nrel-phygnn 0.0.33; NOT
available in the build container nor on the GPU box):

    python tools/make_tf_goldens.py [--out tests/golden]

It imports ``sup3r.models.Sup3rGan`` / ``Sup3rCondMom`` and
``sup3r.pipeline.slicer.ForwardPassSlicer``, builds the models from THIS
repo's JSON configs (identical ``hidden_layers`` spec language), runs the
reference's own code paths on seeded synthetic inputs and stores inputs, every
weight array in keras order, and the outputs:

  tf_gen_3x_4x_2f_fwd.npz   x (1,5,5,4,2)    -> y, via _tf_generate
                            (sup3r/models/abstract.py:1131-1173)
  tf_gen_2x_2f_fwd.npz      x (3,10,10,2)    -> y
  tf_gen_5x_12x_2f_fwd.npz  x (1,6,6,6,4)    -> y  (the C2 generator, small)
  tf_disc_st_same.npz       x (2,12,12,16,2) -> logits, via _tf_discriminate
                            (sup3r/models/base.py:283-313)
  tf_disc_st_valid.npz      x (1,64,64,112,2) -> logits (production disc)
  tf_gan_step.npz           calc_loss + get_single_grad of the generator and
                            discriminator steps (base.py:830-911,
                            abstract.py:1190-1238) on the 2x/4x test GAN:
                            loss details + every gradient
  tf_adam_3steps.npz        keras Adam.apply_gradients x 3 (abstract.py:899)
  tf_slicer_C3.json         ForwardPassSlicer(coarse_shape=(400,400),
                            time_steps=720, chunk_shape=(20,20,48), s_enhance
                            =5, t_enhance=12, spatial_pad=1, temporal_pad=2):
                            slice lists + pad widths (pipeline/slicer.py)

``tests/test_tf_goldens.py`` picks these files up when present: the numpy
oracle (CPU) and the HIP path (GPU) are then checked against TensorFlow's own
numbers instead of only against each other.  This script never travels to the
GPU box as anything but text and nothing in the product imports it.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')


def _np(v):
    return v.numpy() if hasattr(v, 'numpy') else np.asarray(v)


def _weights(net):
    return {f'w{i:03d}': _np(w) for i, w in enumerate(net.weights)}


def _slices(sl):
    return [[s.start, s.stop] for s in sl]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    args = ap.parse_args()
    try:
        import tensorflow as tf
        from sup3r.models import Sup3rGan
        from sup3r.pipeline.slicer import ForwardPassSlicer
    except Exception as e:                      # pragma: no cover
        sys.exit('this script needs the reference environment (tensorflow '
                 f'2.15 + phygnn + sup3r): {e!r}')
    os.makedirs(args.out, exist_ok=True)
    rng = np.random.default_rng(42)
    versions = {'tensorflow': tf.__version__}
    try:
        import phygnn
        import sup3r
        versions.update(phygnn=phygnn.__version__, sup3r=sup3r.__version__)
    except Exception:
        pass

    def gan(gen, disc, **kw):
        Sup3rGan.seed(0)
        return Sup3rGan(os.path.join(CFG, gen), os.path.join(CFG, disc),
                        learning_rate=1e-4, **kw)

    def randomise(model):
        """non-zero biases so that a bias bug cannot hide"""
        for w in model.weights:
            if len(w.shape) == 1:
                w.assign(rng.normal(0, 0.1, size=w.shape).astype(np.float32))

    # ---- generators
    for name, gen, disc, xs, hs in (
            ('gen_3x_4x_2f', 'gen_3x_4x_2f.json', 'disc_st_same.json',
             (1, 5, 5, 4, 2), (1, 15, 15, 16, 2)),
            ('gen_2x_2f', 'gen_2x_2f.json', 'disc_s_same.json',
             (3, 10, 10, 2), (3, 20, 20, 2)),
            ('gen_5x_12x_2f', 'gen_5x_12x_2f.json', 'disc_st_same.json',
             (1, 6, 6, 6, 4), (1, 30, 30, 72, 2))):
        m = gan(gen, disc)
        m.init_weights(xs, hs)
        randomise(m)
        x = rng.standard_normal(xs).astype(np.float32)
        y = _np(m._tf_generate(x))
        assert y.shape == hs, (name, y.shape)
        np.savez_compressed(os.path.join(args.out, f'tf_{name}_fwd.npz'),
                            x=x, y=y, config=gen, **_weights(m.generator))
    # ---- discriminators
    for name, disc, hs in (('disc_st_same', 'disc_st_same.json',
                            (2, 12, 12, 16, 2)),
                           ('disc_st_valid', 'disc_st.json',
                            (1, 64, 64, 112, 2))):
        m = gan('test_gen_st_2x_4x_2f.json', disc)
        m.init_weights((hs[0], 4, 4, 4, 2), hs)
        randomise(m)
        x = rng.standard_normal(hs).astype(np.float32)
        y = _np(m._tf_discriminate(x))
        np.savez_compressed(os.path.join(args.out, f'tf_{name}.npz'), x=x,
                            y=y, config=disc, **_weights(m.discriminator))
    # ---- one GAN step of each kind
    m = gan('test_gen_st_2x_4x_2f.json', 'test_disc_st_same.json',
            loss='MeanAbsoluteError')
    lr = rng.standard_normal((3, 4, 4, 4, 2)).astype(np.float32)
    hr = rng.standard_normal((3, 8, 8, 16, 2)).astype(np.float32)
    m.init_weights(lr.shape, hr.shape)
    randomise(m)
    out = {'low_res': lr, 'high_res': hr, 'weight_gen_advers': 1e-2}
    out.update({'gen_' + k: v for k, v in _weights(m.generator).items()})
    out.update({'disc_' + k: v for k, v in _weights(m.discriminator).items()})
    for tag, tw, kw in (
            ('genstep', m.generator_weights,
             dict(train_gen=True, train_disc=False, compute_disc=True)),
            ('discstep', m.discriminator_weights,
             dict(train_gen=False, train_disc=True))):
        grad, details = m.get_single_grad(lr, hr, tw, weight_gen_advers=1e-2,
                                          **kw)
        for i, g in enumerate(grad):
            out[f'{tag}_g{i:03d}'] = _np(g)
        for k, v in details.items():
            out[f'{tag}_{k}'] = np.float64(_np(v))
    np.savez_compressed(os.path.join(args.out, 'tf_gan_step.npz'), **out)
    # ---- keras Adam, three steps
    w = [tf.Variable(rng.standard_normal(s).astype(np.float32))
         for s in ((3, 3, 2, 5), (5,), (20, 3), (3,))]
    opt = tf.keras.optimizers.Adam(learning_rate=1e-2)
    rec = {f'w0_{i}': _np(v) for i, v in enumerate(w)}
    for t in range(3):
        gs = [rng.standard_normal(v.shape).astype(np.float32) for v in w]
        opt.apply_gradients(zip([tf.constant(g) for g in gs], w))
        for i, (g, v) in enumerate(zip(gs, w)):
            rec[f'g{t}_{i}'] = g
            rec[f'w{t + 1}_{i}'] = _np(v)
    np.savez_compressed(os.path.join(args.out, 'tf_adam_3steps.npz'), **rec)
    # ---- the C3 slicer
    s = ForwardPassSlicer(coarse_shape=(400, 400), time_steps=720, s_enhance=5,
                          t_enhance=12, time_slice=slice(None),
                          temporal_pad=2, spatial_pad=1,
                          chunk_shape=(20, 20, 48))
    rec = {'n_chunks': int(s.n_chunks),
           'n_spatial_chunks': int(s.n_spatial_chunks),
           'n_time_chunks': int(s.n_time_chunks),
           's1_lr_slices': _slices(s.s1_lr_slices),
           's2_lr_slices': _slices(s.s2_lr_slices),
           's1_lr_pad_slices': _slices(s.s1_lr_pad_slices),
           's2_lr_pad_slices': _slices(s.s2_lr_pad_slices),
           't_lr_slices': _slices(s.t_lr_slices),
           't_lr_pad_slices': _slices(s.t_lr_pad_slices),
           's1_hr_crop_slices': _slices(s.s1_hr_crop_slices),
           's2_hr_crop_slices': _slices(s.s2_hr_crop_slices),
           't_hr_crop_slices': _slices(s.t_hr_crop_slices),
           'chunk_indices': [list(map(int, s.get_chunk_indices(i)))
                             for i in (0, 1, 399, 400, 5999)],
           'pad_width': {str(i): [list(map(int, p)) for p in
                                  s.get_pad_width(i)]
                         for i in (0, 19, 399, 400, 5999)},
           'versions': versions}
    with open(os.path.join(args.out, 'tf_slicer_C3.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    print('wrote', sorted(p for p in os.listdir(args.out)
                          if p.startswith('tf_')), versions)


if __name__ == '__main__':
    main()
