#!/usr/bin/env python
"""cProfile of the host side of Sup3rGan._train_batch on a small config:
python tools/train_hostprof.py [--gen ...] [--disc ...] [--lr-shape ...]"""
import argparse
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gen', default='gen_2x_2f.json')
    ap.add_argument('--disc', default='disc_s_same.json')
    ap.add_argument('--lr-shape', default='15,5,5,2')
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    import torch
    from sup3r_amd import Sup3rGan
    lr_shape = tuple(int(v) for v in args.lr_shape.split(','))
    model = Sup3rGan(os.path.join(CFG, args.gen), os.path.join(CFG, args.disc),
                     loss='MeanAbsoluteError')
    s, t = model.s_enhance, model.t_enhance
    hr_shape = (lr_shape[0], lr_shape[1] * s, lr_shape[2] * s) + (
        (lr_shape[3] * t, 2) if len(lr_shape) == 5 else (2,))
    rng = np.random.default_rng(0)

    class B:
        low_res = rng.standard_normal(lr_shape).astype(np.float32)
        high_res = rng.standard_normal(hr_shape).astype(np.float32)
    model.init_weights(lr_shape, hr_shape)
    step = lambda: model._train_batch(B, True, False, False, True, False, False, 1e-3)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.iters):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(35)


if __name__ == '__main__':
    main()
