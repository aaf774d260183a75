#!/usr/bin/env python
"""Is the trunk kernel limited by the board's power cap or by its schedule?

Runs the C2 generator forward (32 chunks per step, bf16) twice — on random
operands and on all-zero operands (zero weights, zero input: the same
instruction stream, no bit toggling in the MFMA datapath) — while polling
rocm-smi, and prints ms per step, the mean shader clock and socket power of
each.  A schedule-bound kernel takes the same time on both.

    python tools/power_probe.py [--steps 300]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def poll(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(['rocm-smi', '--showclocks', '--showpower'],
                               capture_output=True, text=True, timeout=5).stdout
            m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t)
            p = re.search(r'Power \(W\): ([\d.]+)', t)
            if m and p:
                out.append((int(m.group(1)), float(p.group(1))))
        except Exception:
            pass
        time.sleep(0.1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    import torch
    from sup3r_amd.engine import Device, Network
    cfg = os.path.join(ROOT, 'sup3r_amd', 'configs', 'gen_5x_12x_2f.json')
    with open(cfg) as f:
        spec = json.load(f)
    dev = Device.get(0)
    shape = (args.batch, 16, 16, 24, 4)
    res = {}
    for what in ('random', 'zeros'):
        net = Network(spec, name='generator', device=dev, precision='bf16')
        net.build(shape, seed=0)
        x = np.random.default_rng(42).standard_normal(shape).astype(np.float32)
        if what == 'zeros':
            net.set_weights([np.zeros_like(w) for w in net.weights])
            x[:] = 0
        ph = net.plan(shape, training=False)
        xd = dev.to_device(x)
        out = dev.empty((args.batch, 80, 80, 288, 2))
        for _ in range(5):
            ph.forward(xd, out=out)
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=poll, args=(stop, samples))
        th.start()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ph.forward(xd, out=out)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        stop.set()
        th.join()
        busy = [s for s in samples if s[1] > 600] or samples or [(0, 0.0)]
        res[what] = {'ms_per_step': 1e3 * el / args.steps,
                     'samples_per_s': args.batch * args.steps / el,
                     'sclk_MHz': float(np.mean([s[0] for s in busy])),
                     'power_W': float(np.mean([s[1] for s in busy])),
                     'n_smi_samples': len(busy)}
        del ph
        net.clear_plans()
    res['zeros_over_random'] = (res['zeros']['samples_per_s'] /
                                res['random']['samples_per_s'])
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
