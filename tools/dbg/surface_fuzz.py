#!/usr/bin/env python
"""Shape fuzz over the shipped generator specs: random batch / extents, bf16 inference with the default
kernel selection against the same plan on the general kernels (NO_PERSIST, NO_CONV2D_WS, NO_TAIL_SLIDE,
NO_CONV2D_OUT, NO_FUSED2D: same bf16 operands, other summation orders), and one training step (gradients
finite).  python tools/dbg/surface_fuzz.py [--n 2] [--seed 0] [--nmax 24] [--train-div 3]"""
import argparse
import glob
import json
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd.engine import Network  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=2)
ap.add_argument('--seed', type=int, default=0)
ap.add_argument('--nmax', type=int, default=24)
ap.add_argument('--train-div', type=int, default=3)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
SURF = os.path.join(ROOT, 'sup3r_amd', 'configs', 'sup3r')
GENERAL = {'NO_PERSIST': 1, 'NO_CONV2D_WS': 1, 'NO_TAIL_SLIDE': 1, 'NO_CONV2D_OUT': 1, 'NO_FUSED2D': 1}
worst, bad = 0.0, 0
for path in sorted(glob.glob(SURF + '/*/gen_*.json')):
    rel = os.path.relpath(path, SURF)
    spec = json.load(open(path))
    feats = int([p for p in os.path.basename(rel)[:-5].split('_') if p.endswith('f')][0][:-1])
    st = 'spatial/' not in rel and '_5x_1x_' not in rel
    for _ in range(a.n):
        n = int(rng.integers(1, a.nmax + 1))
        h, w = int(rng.integers(8, 49)), int(rng.integers(8, 49))
        t = int(rng.integers(4, 21))
        cin = feats + (1 if 'wind_5x_1x' in rel or 'wind_3x_4x' in rel else 0) + (2 if 'solar_1x_8x' in rel else 0) + \
            (2 if 'trh_1x_24x' in rel else 0) + (2 if 'solar_5x_1x' in rel else 0)
        shape = (n, h, w, t, cin) if st else (n, h, w, cin)
        tag = f'{rel} {shape}'
        try:
            net = Network(spec, precision='bf16')
            net.build(shape, seed=int(rng.integers(1 << 20)))
        except Exception as e:                      # (a spec / shape the plan refuses: say so, go on)
            print(f'{tag}: not built: {str(e)[:90]}', flush=True)
            continue
        dev = net.dev
        x = rng.standard_normal(shape).astype(np.float32)
        ph = net.plan(shape, training=False)
        exo = {k: dev.to_device(rng.standard_normal(tuple(sh)).astype(np.float32)) for k, sh in ph.in_shapes.items() if k != 'x'}
        y = ph.forward(dev.to_device(x), exo).cpu().numpy()
        pg = net.plan(shape, training=False, options=GENERAL)
        yg = pg.forward(dev.to_device(x), exo).cpu().numpy()
        err = float(np.abs(y - yg).max() / max(np.abs(yg).max(), 1e-30))
        fin = bool(np.isfinite(y).all())
        del ph, pg
        net.clear_plans()
        nt = max(1, n // a.train_div)
        tshape = (nt,) + shape[1:]
        pt = net.plan(tshape, training=True)
        exot = {k: dev.to_device(rng.standard_normal(tuple(sh)).astype(np.float32)) for k, sh in pt.in_shapes.items() if k != 'x'}
        yt = pt.forward(dev.to_device(x[:nt]), exot)
        pt.backward(dev.to_device(rng.standard_normal(tuple(yt.shape)).astype(np.float32)), need_dx=False)
        gfin = all(np.isfinite(np.asarray(g)).all() for g in net.grads)
        del pt
        net.clear_plans()
        ok = fin and gfin and err < 3e-2
        worst = max(worst, err)
        bad += 0 if ok else 1
        print(f'{tag}: default vs general kernels {err:.2e}, finite {fin}, training grads finite {gfin}' + ('' if ok else '  <-- CHECK'),
              flush=True)
print(f'worst {worst:.2e}, {bad} cases to check')
