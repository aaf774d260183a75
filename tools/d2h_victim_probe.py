"""What does a device -> host delivery running beside the C3 forward cost it,
by mechanism?  The batch-8 plan runs in a loop (per-op HIP events) while a
thread delivers 368 MB buffers back to back.  python tools/d2h_victim_probe.py"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from sup3r_amd import Sup3rGan, _lib  # noqa: E402

mechs = sys.argv[1].split(',') if len(sys.argv) > 1 else \
    ['alone', 'torch', 'memcpy', 'k16', 'sdma']
batch = 8
feats = ['u_100m', 'v_100m', 'temperature_100m', 'pressure_0m']
m = Sup3rGan(bench.CFG, os.path.join(bench.CFGDIR, 'test_disc_st_same.json'),
             precision='bf16')
m.set_model_params(lr_features=feats, hr_out_features=feats[:2], s_enhance=5,
                   t_enhance=12)
Sup3rGan.seed(0)
m.init_weights((1, 22, 22, 52, 4), (1, 110, 110, 624, 2))
gen = m._gen
dev, L = gen.dev, _lib.lib()
ph = gen.plan((batch, 22, 22, 52, 4), training=False)
x = dev.to_device(np.random.default_rng(0).standard_normal(
    (batch, 22, 22, 52, 4)).astype(np.float32))
for _ in range(3):
    y = ph.forward(x)
torch.cuda.synchronize()
n = batch * 100 * 100 * 576 * 2
src = dev.empty((n,))
src.fill_(1.5)
side = torch.cuda.Stream(device=dev.torch_device)
pin = torch.empty((n,), dtype=torch.float32, pin_memory=True)
hp = C.c_void_p()
_lib.check(L.s3_host_alloc(dev.ctx, n * 4, 0, C.byref(hp)), dev.ctx, 'alloc')
harr = np.ctypeslib.as_array((C.c_float * n).from_address(hp.value))
torch.cuda.synchronize()


def deliver(mech):
    if mech == 'torch':
        with torch.cuda.stream(side):
            pin.copy_(src, non_blocking=True)
        side.synchronize()
    elif mech == 'memcpy':
        _lib.check(L.s3_d2h_async(dev.ctx, C.c_void_p(src.data_ptr()), hp,
                                  n * 4, C.c_void_p(side.cuda_stream)),
                   dev.ctx, 'd2h_async')
        side.synchronize()
    elif mech.startswith('k'):
        _lib.check(L.s3_d2h_stream(dev.ctx, C.c_void_p(src.data_ptr()), hp,
                                   n * 4, C.c_void_p(side.cuda_stream),
                                   int(mech[1:])), dev.ctx, 'd2h_stream')
        side.synchronize()
    elif mech == 'sdma':
        t = C.c_uint64()
        _lib.check(L.s3_dma_d2h_begin(dev.ctx, C.c_void_p(src.data_ptr()), hp,
                                      n * 4, C.byref(t)), dev.ctx, 'dma')
        _lib.check(L.s3_dma_wait(dev.ctx, t, 0), dev.ctx, 'dma_wait')


out = {}
for mech in mechs:
    stop = threading.Event()
    count = [0, 0.0]

    def bg():
        t0 = time.perf_counter()
        while not stop.is_set():
            deliver(mech)
            count[0] += 1
        count[1] = time.perf_counter() - t0
    th = None
    if mech != 'alone':
        deliver(mech)
        harr[:4] = 0
        deliver(mech)
        ok = bool(harr[0] == 1.5 and harr[-1] == 1.5) if mech != 'torch' \
            else bool(pin[0] == 1.5 and pin[-1] == 1.5)
        th = threading.Thread(target=bg)
        th.start()
        time.sleep(0.05)
    ph.profile_begin(12)
    t0 = time.perf_counter()
    for _ in range(12):
        y = ph.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12
    _, ms = ph.profile_end()
    if th is not None:
        stop.set()
        th.join()
    trunk = [ms[i] for i, op in enumerate(ph.plan.ops)
             if op.get('cin') == 64 and op.get('cout') == 64
             and ph.plan.tensors[op['out']][1:4] == [22, 22, 624]]
    out[mech] = {'forward_ms': round(dt * 1e3, 3),
                 'head_conv_us': round(ms[0] * 1e3, 1),
                 'trunk_us': round(float(np.mean(trunk)) * 1e3, 1),
                 'tail_us': round(ms[-1] * 1e3, 1)}
    if th is not None:
        out[mech].update(delivered_ok=ok, deliveries=count[0],
                         delivery_gbs=round(count[0] * n * 4 / count[1] / 1e9,
                                            1))
    print(mech, out[mech], flush=True)
print(json.dumps(out))
