#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ with the numpy oracle
(oracle/), which is itself pinned against torch-CPU autograd and exact-integer
permutation cases (tests/test_oracle_vs_torch.py).  The reference (TF/phygnn)
cannot be imported in this container, so these are restatement goldens
("parity unpinned" at the TF level — SURVEY.md §8c); they freeze the oracle's
behaviour and give the GPU tests box-independent expected values.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import gan as G  # noqa: E402
from oracle import layers as L  # noqa: E402
from oracle.network import Network  # noqa: E402

CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')


def load(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


def net_case(fname, cfg, shape, exo_shape=None, seed=1):
    rng = np.random.default_rng(42)
    x = rng.standard_normal(shape).astype(np.float32)
    exo = None
    if exo_shape is not None:
        exo = {'topography': rng.standard_normal(exo_shape).astype(np.float32)}
    net = Network(load(cfg))
    net.init_weights(x, exo, seed=seed, bias_scale=0.1)
    y = net.forward(x, exo)
    dy = np.random.default_rng(2).standard_normal(y.shape).astype(np.float32)
    dx = net.backward(dy)
    out = {'x': x, 'y': y, 'dy': dy, 'dx': dx}
    if exo:
        out['exo'] = exo['topography']
    for i, (w, g) in enumerate(zip(net.weights, net.grads)):
        out[f'w{i}'] = w
        out[f'g{i}'] = g.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, fname), **out)


def main():
    net_case('gen_st_2x_4x_2f.npz', 'test_gen_st_2x_4x_2f.json',
             (1, 5, 5, 4, 2))
    net_case('gen_st_3x_4x_2f_topo.npz', 'test_gen_st_3x_4x_2f_topo.json',
             (1, 4, 4, 4, 2), exo_shape=(1, 12, 12, 16, 1))
    net_case('gen_s_2x_2f.npz', 'test_gen_s_2x_2f.json', (3, 10, 10, 2))
    net_case('disc_st_same.npz', 'test_disc_st_same.json', (2, 12, 12, 16, 2))
    net_case('disc_st_valid.npz', 'test_disc_st_valid.json',
             (1, 14, 13, 15, 2))
    # exact-integer permutation ops
    perm = {}
    for b in (2, 3, 5):
        x = np.arange(2 * 3 * 2 * b * b * 2, dtype=np.float32).reshape(
            2, 3, 2, b * b * 2)
        perm[f'd2s_x_b{b}'] = x
        perm[f'd2s_y_b{b}'] = L.depth_to_space(x, b)
    for m in (2, 3):
        x = np.arange(1 * 2 * 2 * 4 * 3, dtype=np.float32).reshape(
            1, 2, 2, 4, 3)
        perm[f'trepeat_x_m{m}'] = x
        perm[f'trepeat_y_m{m}'] = L.SpatioTemporalExpansion(
            temporal_mult=m).forward(x)
    np.savez_compressed(os.path.join(HERE, 'permutation_ops.npz'), **perm)
    # GAN loss algebra + Adam
    rng = np.random.default_rng(7)
    dt = (rng.standard_normal((15, 1)) * 2).astype(np.float32)
    dg = (rng.standard_normal((15, 1)) * 2).astype(np.float32)
    a = rng.standard_normal((3, 6, 6, 4, 2)).astype(np.float32)
    b = rng.standard_normal((3, 6, 6, 4, 2)).astype(np.float32)
    ld, g_t, g_g = G.rel_bce(dt.astype(np.float64), dg.astype(np.float64))
    la, ga_t, _ = G.rel_bce(dg.astype(np.float64), dt.astype(np.float64))
    mae, gmae, _ = G.mae(a.astype(np.float64), b.astype(np.float64))
    mse, gmse, _ = G.mse(a.astype(np.float64), b.astype(np.float64))
    w = rng.standard_normal((5, 4)).astype(np.float32)
    out = dict(d_true=dt, d_gen=dg, loss_disc=ld, g_true=g_t, g_gen=g_g,
               loss_gen_advers=la, g_advers=ga_t, hr_gen=a, hr_true=b,
               mae=mae, g_mae=gmae, mse=mse, g_mse=gmse, adam_w0=w)
    opt = G.Adam(learning_rate=1e-3)
    wk = w.copy()
    for t in range(3):
        g = rng.standard_normal(w.shape).astype(np.float32)
        opt.apply_gradients([g], [wk])
        out[f'adam_g{t + 1}'] = g
        out[f'adam_w{t + 1}'] = wk.copy()
        out[f'adam_m{t + 1}'] = opt.m[0].copy()
        out[f'adam_v{t + 1}'] = opt.v[0].copy()
    np.savez_compressed(os.path.join(HERE, 'gan_loss_adam.npz'), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
