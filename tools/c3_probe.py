"""C3 executor probe (GPU): per-op time of the batch-8 plan in isolation, then
the executor's rate by delivery mode.  python tools/c3_probe.py [batches]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from sup3r_amd import ForwardPass, Sup3rGan  # noqa: E402
from sup3r_amd.forward_pass import register_model  # noqa: E402
from sup3r_amd.strategy import ArrayStrategy  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
modes = sys.argv[2].split(',') if len(sys.argv) > 2 else \
    ['iso', 'none', 'data']
batch = 8
feats = ['u_100m', 'v_100m', 'temperature_100m', 'pressure_0m']
m = Sup3rGan(bench.CFG, os.path.join(bench.CFGDIR, 'test_disc_st_same.json'),
             precision='bf16')
m.set_model_params(lr_features=feats, hr_out_features=feats[:2], s_enhance=5,
                   t_enhance=12)
Sup3rGan.seed(0)
m.init_weights((1, 22, 22, 52, 4), (1, 110, 110, 624, 2))
out = {}
if 'iso' in modes:
    gen = m._gen
    ph = gen.plan((batch, 22, 22, 52, 4), training=False)
    x = gen.dev.to_device(np.random.default_rng(0).standard_normal(
        (batch, 22, 22, 52, 4)).astype(np.float32))
    for _ in range(3):
        y = ph.forward(x)
    torch.cuda.synchronize()
    ph.profile_begin(10)
    t0 = time.perf_counter()
    for _ in range(10):
        y = ph.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    _, ms = ph.profile_end()
    rows = []
    for i, op in enumerate(ph.plan.ops):
        info = ph.op_info(i)
        rows.append((i, op.get('cin'), op.get('cout'),
                     ph.plan.tensors[op['out']][1:4], info['fwd'],
                     round(ms[i] * 1e3, 1)))
    trunk = [r[-1] for r in rows if r[1] == 64 and r[2] == 64
             and r[3] == [22, 22, 624]]
    out['iso'] = {'forward_ms': dt * 1e3, 'sum_ops_ms': sum(ms),
                  'trunk_us_mean': float(np.mean(trunk)) if trunk else None,
                  'trunk_tflops_useful': (8 * 22 * 22 * 624 * 27 * 64 * 64 * 2
                                          / (np.mean(trunk) * 1e-6) / 1e12)
                  if trunk else None, 'ops': rows}
    del ph, x, y

register_model('Sup3rGan', {'model_dir': 'probe-c3'}, m)
domain = np.random.default_rng(7).standard_normal(
    (400, 400, 720, 4), dtype=np.float32)
st = ArrayStrategy(domain, {'model_dir': 'probe-c3'}, (20, 20, 48),
                   spatial_pad=1, temporal_pad=2, max_nodes=1, model=m)
fwp = ForwardPass(st, 0)
mine = [int(i) for i in st.node_chunks[0]]


def run(ids, return_data=True):
    n, acc = 0, 0.0
    for chunk, failed, data in ForwardPass.iter_chunks(
            (fwp.get_input_chunk(i) for i in ids), m,
            allowed_const=st.allowed_const, batch=batch,
            return_data=return_data):
        assert not failed
        if data is not None:
            acc += float(data[::17, ::17, ::17].sum())
        n += 1
    return n, acc


for mode in modes:
    if mode == 'iso':
        continue
    # modes: none (files only, no delivery) | data
    rd = mode != 'none'
    run(mine[:2 * batch], rd)
    torch.cuda.synchronize()
    php = m._gen.plan((batch, 22, 22, 52, 4), training=False)
    php.profile_begin(nb)
    prof = None
    if os.environ.get('C3_PROBE_CPROFILE'):
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    n, acc = run(mine[2 * batch:(2 + nb) * batch], rd)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if prof is not None:
        import io
        import pstats
        prof.disable()
        sio = io.StringIO()
        pstats.Stats(prof, stream=sio).sort_stats('cumtime').print_stats(28)
        sys.stderr.write(f'==== {mode}\n' + sio.getvalue())
    _, ms = php.profile_end()
    out[mode] = {'chunks_per_s': n / el, 'ms_per_batch': el / nb * 1e3,
                 'checksum': acc, 'ops_ms_in_situ': float(sum(ms)),
                 'head_conv_us_in_situ': ms[0] * 1e3}
print(json.dumps(out))
