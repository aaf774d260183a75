import time, numpy as np, os
from threadpoolctl import threadpool_limits
rng = np.random.default_rng(0)
M = 22 * 22 * 294
x2 = rng.standard_normal((M, 64)).astype(np.float32)
ew = rng.standard_normal((M, 64)).astype(np.float32)
w = rng.standard_normal((27, 64, 64)).astype(np.float32)
def fwd():
    y = np.zeros((M, 64), np.float32)
    for t in range(27):
        off = 7 * t
        y[:M - off] += x2[off:] @ w[t]
    return y
def bwd():
    dx = np.zeros((M, 64), np.float32)
    for t in range(27):
        off = 7 * t
        dw = x2[off:].T @ ew[:M - off]
        dx[off:] += ew[:M - off] @ w[t].T
    return dx
for n in (1, 2, 4, 8, 16, 32, 64):
    with threadpool_limits(limits=n, user_api='blas'):
        fwd(); t = time.time(); fwd(); tf = time.time() - t
        bwd(); t = time.time(); bwd(); tb = time.time() - t
    print(f'openblas threads {n:3d}: fwd {tf:.3f} s  bwd {tb:.3f} s', flush=True)
import torch
xt, et, wt = torch.from_numpy(x2), torch.from_numpy(ew), torch.from_numpy(w)
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    def tfwd():
        y = torch.zeros((M, 64))
        for t in range(27):
            off = 7 * t
            y[:M - off] += xt[off:] @ wt[t]
        return y
    def tbwd():
        dx = torch.zeros((M, 64))
        for t in range(27):
            off = 7 * t
            dw = xt[off:].T @ et[:M - off]
            dx[off:] += et[:M - off] @ wt[t].T
        return dx
    tfwd(); t = time.time(); tfwd(); tf = time.time() - t
    tbwd(); t = time.time(); tbwd(); tb = time.time() - t
    print(f'torch threads {n:3d}: fwd {tf:.3f} s  bwd {tb:.3f} s', flush=True)
