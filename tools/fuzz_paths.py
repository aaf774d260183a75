#!/usr/bin/env python
"""Shape fuzz of the specialised bf16 training kernels against their general
fallbacks (same bf16 operands): random extents / batch sizes, each special path
forced on with its MIN_TILES override and compared with the path disabled.
python tools/fuzz_paths.py [--n 12] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def conv(f, s=1, pad='valid'):
    return [{'class': 'Conv3D', 'filters': f, 'kernel_size': 3, 'strides': s,
             'padding': pad}, {'alpha': 0.2, 'class': 'LeakyReLU'}]


def run(spec, shape, x, dy, env):
    import torch
    from sup3r_amd.engine import Network
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        net = Network(spec, precision='bf16')
        net.build(shape, seed=1)
        ph = net.plan(shape, training=True)
        y = ph.forward(net.dev.to_device(x))
        if dy is None:
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(y.device)
        dx = ph.backward(dy, need_dx=True)
        return y.cpu().numpy(), dx.cpu().numpy(), [np.array(g) for g in net.grads], dy
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def rel(a, b):
    d = float(np.sqrt(((a - b) ** 2).mean()))
    n = float(np.sqrt((b ** 2).mean()))
    return d / max(n, 1e-30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=12)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--poison', action='store_true',
                    help='plan buffers start as NaN patterns in the specialised run')
    args = ap.parse_args()
    from sup3r_amd.configs.author_configs import pcc
    rng = np.random.default_rng(args.seed)
    force = {'SUP3R_AMD_HALO32_MIN_TILES': '1', 'SUP3R_AMD_FEWCH_HALO_MIN_TILES': '1',
             'SUP3R_AMD_DGRAD_S2_MIN_TILES': '1', 'SUP3R_AMD_PERSIST_DGRAD_MIN_TILES': '1',
             'SUP3R_AMD_HALO_S2_MIN_TILES': '1'}
    if args.poison:
        force['SUP3R_AMD_POISON_ALLOC'] = '1'
    off = dict(force, SUP3R_AMD_NO_HALO32='1', SUP3R_AMD_NO_FEWCH_HALO='1',
               SUP3R_AMD_NO_DGRAD_S2='1', SUP3R_AMD_NO_DGRAD_C2='1',
               SUP3R_AMD_NO_WGRAD_TAIL='1', SUP3R_AMD_NO_WGRAD_C2='1',
               SUP3R_AMD_NO_DGRAD_CHUNKED='1', SUP3R_AMD_NO_DGRAD_FEWCH='1',
               SUP3R_AMD_NO_MASK_FUSE='1', SUP3R_AMD_BF16_TRAIN_ACT='0',
               SUP3R_AMD_NO_DISC_BF16='1', SUP3R_AMD_NO_BIAS_FUSE='1',
               SUP3R_AMD_NO_PERSIST_DGRAD='1', SUP3R_AMD_NO_HALO_S2='1', SUP3R_AMD_NO_WGRAD_WS='1',
               SUP3R_AMD_NO_DPRE16='1', SUP3R_AMD_NO_GCONV_SPLITK='1', SUP3R_AMD_NO_TILE_NF2='1',
               SUP3R_AMD_NO_FOLD16='1', SUP3R_AMD_NO_WGRAD_GEN_PF='1', SUP3R_AMD_NO_BATCHED_PACK='1')
    off.pop('SUP3R_AMD_POISON_ALLOC', None)
    worst = 0.0
    for it in range(args.n):
        kind = it % 4
        n = int(rng.integers(1, 9))
        if kind == 0:      # discriminator-style stack
            dims = (int(rng.integers(13, 30)), int(rng.integers(13, 30)), int(rng.integers(21, 70)))
            spec = conv(32) + conv(32, 2) + conv(64) + [{'class': 'Flatten'}, {'class': 'Dense', 'units': 1}]
            shape = (n, *dims, 2)
        elif kind == 1:    # generator tail: 64 -> 200 + d2s -> 8 -> 2
            dims = (int(rng.integers(4, 9)), int(rng.integers(4, 9)), int(rng.integers(16, 50)))
            spec = pcc(3, 64) + pcc(3, 64) + pcc(3, 200, act=False) + \
                [{'class': 'SpatioTemporalExpansion', 'spatial_mult': 5},
                 {'alpha': 0.2, 'class': 'LeakyReLU'}] + pcc(3, 2, act=False)
            shape = (n, *dims, 4)
        elif kind == 3:    # the production discriminator's conv stack, small extents
            dims = (int(rng.integers(36, 48)), int(rng.integers(36, 48)), int(rng.integers(60, 90)))
            n = min(n, 3)
            spec = conv(32) + conv(32, 2) + conv(64) + conv(64, 2) + conv(128) + conv(128, 2) + \
                [{'class': 'Flatten'}, {'class': 'Dense', 'units': 16}, {'alpha': 0.2, 'class': 'LeakyReLU'},
                 {'class': 'Dense', 'units': 1}]
            shape = (n, *dims, 2)
        else:              # residual trunk (bf16 saved activations, wgrad bf16)
            dims = (int(rng.integers(6, 14)), int(rng.integers(6, 14)), int(rng.integers(17, 60)))
            blk = [{'class': 'SkipConnection', 'name': 'b'}] + pcc(3, 64) + pcc(3, 64, act=False) + \
                [{'class': 'SkipConnection', 'name': 'b'}]
            spec = pcc(3, 64) + blk + [{'class': 'SkipConnection', 'name': 'c'}] + pcc(3, 64) + \
                pcc(3, 64, act=False) + [{'class': 'SkipConnection', 'name': 'c'}] + pcc(3, 2, act=False)
            shape = (n, *dims, 4)
        x = rng.standard_normal(shape).astype(np.float32)
        y1, dx1, g1, dy = run(spec, shape, x, None, force)
        y2, dx2, g2, _ = run(spec, shape, x, dy, off)
        errs = [rel(y1, y2), rel(dx1, dx2)] + [rel(a, b) for a, b in zip(g1, g2)]
        ok = all(np.isfinite(e) for e in errs) and max(errs) < (3e-2 if kind >= 2 else 1e-2 if kind == 1 else 5e-3)
        worst = max(worst, max(errs))
        print(f'case {it} kind {kind} shape {shape}: max rel rms {max(errs):.2e} {"ok" if ok else "FAIL"}')
        if not ok:
            print('   per-tensor:', ['%.1e' % e for e in errs])
            sys.exit(1)
    print(f'all {args.n} cases ok, worst rel rms {worst:.2e}')


if __name__ == '__main__':
    main()
