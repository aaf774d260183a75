#!/usr/bin/env python
"""Race screen: the same K training steps of the C2 GAN twice from the same
seed — every kernel reduces in a fixed order, so the weights after K steps
must be bit-identical.  A data race (an operand consumed before its wait, an
LDS hand-over without its barrier) shows up as a difference.
python tools/determinism_soak.py [--config c2] [--batch 8] [--steps 12] [--runs 3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(config, batch, steps):
    import torch
    import bench
    from sup3r_amd import Sup3rGan
    from sup3r_amd.engine import Device
    Sup3rGan.seed(7)
    model, lr_s, hr_s, what, _ = bench.train_models(config)
    dev = Device.get()
    rng = np.random.default_rng(3)
    losses = []
    model.init_weights((batch,) + lr_s, (batch,) + hr_s)
    for k in range(steps):
        class Batch:
            low_res = dev.to_device(rng.standard_normal((batch,) + lr_s).astype(np.float32))
            high_res = dev.to_device(rng.standard_normal((batch,) + hr_s).astype(np.float32))
        d = model._train_batch(Batch, True, False, False, True, False, False, 1e-3)
        losses.append((d['loss_gen'], d['loss_disc']))
    torch.cuda.synchronize()
    w = [np.array(a) for a in model.generator.weights] + \
        [np.array(a) for a in model.discriminator.weights]
    del model
    torch.cuda.empty_cache()
    return w, losses, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--runs', type=int, default=3)
    a = ap.parse_args()
    ref, l0, what = run(a.config, a.batch, a.steps)
    assert all(np.isfinite(v).all() for v in ref), 'non-finite weights'
    print(f'{what}, batch {a.batch}, {a.steps} steps: losses (gen, disc) first {l0[0]} last {l0[-1]}')
    bad = 0
    for r in range(1, a.runs):
        w, l, _ = run(a.config, a.batch, a.steps)
        diff = [i for i, (x, y) in enumerate(zip(ref, w)) if not np.array_equal(x, y)]
        print(f'run {r}: {len(diff)} of {len(w)} weight tensors differ'
              + (f' (first: #{diff[0]}, max |d| {np.abs(ref[diff[0]] - w[diff[0]]).max():.3e})' if diff else '')
              + f'; losses identical: {l == l0}')
        bad += len(diff)
    print('DETERMINISTIC' if not bad else 'NOT DETERMINISTIC')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
