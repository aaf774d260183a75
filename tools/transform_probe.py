"""Throughput of the device batch transform against the
numpy/scipy host path it replaces.  Usage: python tools/transform_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch  # noqa: E402
from oracle.transform import transform as host_transform  # noqa: E402
from sup3r_amd.batch_transform import DeviceBatchTransform  # noqa: E402

shape, s, t = (32, 96, 96, 96, 2), 3, 4          # C4-like: 3x / 4x, batch 32
feats = ['u', 'v']
rng = np.random.default_rng(0)
x = rng.standard_normal(shape).astype(np.float32)
tr = DeviceBatchTransform(s, t, feats)
xd = tr.dev.to_device(x)
for smoothing in (None, 0.8):
    for _ in range(3):
        tr.transform(xd, smoothing=smoothing, temporal_coarsening_method='average')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        tr.transform(xd, smoothing=smoothing, temporal_coarsening_method='average')
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    lr_bytes = x.nbytes / (s * s * t)
    moved = x.nbytes + lr_bytes + (4 * lr_bytes if smoothing else 0)
    t1 = time.perf_counter()
    host_transform(x, s, t, feats, [0, 1], smoothing, None, 'average')
    host_ms = (time.perf_counter() - t1) * 1e3
    print(f'smoothing={smoothing}: device {ms:.3f} ms ({moved / ms / 1e6:.0f} GB/s '
          f'of algorithmic traffic), host numpy/scipy {host_ms:.1f} ms')
