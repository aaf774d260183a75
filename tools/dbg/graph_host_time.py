"""host time of Sup3rGan._launch_batch (no sync) eager vs replayed graph, C1 shape"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from sup3r_amd import Sup3rGan
from sup3r_amd.engine import Device
CFG = os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs')
LR, HR = (15, 5, 5, 2), (15, 10, 10, 2)
for mode in (False, True):
    Sup3rGan.seed(1)
    m = Sup3rGan(os.path.join(CFG, 'gen_2x_2f.json'), os.path.join(CFG, 'disc_s_same.json'),
                 loss='MeanAbsoluteError', precision='bf16')
    m.capture_steps = mode
    m.init_weights(LR, HR)
    dev = Device.get()
    rng = np.random.default_rng(0)
    class B:
        low_res = dev.to_device(rng.standard_normal(LR).astype(np.float32))
        high_res = dev.to_device(rng.standard_normal(HR).astype(np.float32))
    args = (B, True, False, False, True, False, False, 1e-3)
    for _ in range(5):
        m._train_batch(*args)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(200):
        h0 = time.perf_counter()
        launched = m._launch_batch(*args)
        host += time.perf_counter() - h0
        m._settle(*launched)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f'capture_steps={mode}: {el / 200 * 1e3:.3f} ms per _train_batch, host enqueue {host / 200 * 1e3:.3f} ms')
