"""How much of a C2 training step is the host's turnaround after the per-batch
loss read-back?  K x _train_batch (settled every batch, what a gated epoch
does) against K x _launch_batch settled together at the end."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import bench
from sup3r_amd.engine import Device

model, lr_s, hr_s, what, gflop = bench.train_models('c2')
n = 8
dev = Device.get()
rng = np.random.default_rng(0)


class Batch:
    low_res = dev.to_device(rng.standard_normal((n,) + lr_s).astype(np.float32))
    high_res = dev.to_device(rng.standard_normal((n,) + hr_s).astype(np.float32))


model.init_weights((n,) + lr_s, (n,) + hr_s)
args = (Batch, True, False, False, True, False, False, 1e-3)
for _ in range(3):
    model._train_batch(*args)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    model._train_batch(*args)
torch.cuda.synchronize()
a = (time.perf_counter() - t0) / K
t0 = time.perf_counter()
held = [model._launch_batch(*args) for _ in range(K)]
t1 = time.perf_counter()
for h in held:
    model._settle(*h)
torch.cuda.synchronize()
b = (time.perf_counter() - t0) / K
print('settled every batch %.3f ms | settled at the end %.3f ms | host enqueue %.3f ms per batch'
      % (a * 1e3, b * 1e3, (t1 - t0) / K * 1e3))
