"""How far is the bf16 throughput mode from the exact-fp32 parity mode on the
full C2 generator (random glorot weights, O(1) inputs)?  The fp32 mode is
itself within 1e-3 of the oracle (tests/test_hip_parity.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sup3r_amd.engine import Network  # noqa: E402

spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs', 'gen_5x_12x_2f.json')))
shape = (8, 16, 16, 24, 4)
x = np.random.default_rng(42).standard_normal(shape).astype(np.float32)
n32 = Network(spec, precision='f32')
n32.build(shape, seed=0)
y32 = n32(x).cpu().numpy()
n16 = Network(spec, precision='bf16')
n16.set_weights(n32.weights)
y16 = n16(x).cpu().numpy()
d = np.abs(y16 - y32)
print(f'C2 batch 8: |y| max {np.abs(y32).max():.4f} rms {np.sqrt((y32 ** 2).mean()):.4f}; '
      f'bf16 vs fp32 mode: L-inf {d.max():.3e}, rms {np.sqrt((d ** 2).mean()):.3e}, '
      f'rel L-inf {d.max() / np.abs(y32).max():.3e}')
