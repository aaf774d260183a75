"""chunks/s of a MultiStepGan of two spatial steps (the arrangement of
examples/sup3rwind/run_configs/wind/config_fwp_spatial.json: 3x then 5x with
topography, 75 x 75 x 38 chunks, temporal_pad 5) through
ForwardPass.iter_chunks: the chain on the device (s3_step_handover between the
steps) vs MultiStepGan.generate (every hand-over through host numpy).
usage: python tools/dbg/fwp_chain_probe.py [batch] [s1,s2,t chunk]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd import ForwardPass, MultiStepGan, Sup3rGan  # noqa: E402
from sup3r_amd.forward_pass import register_model  # noqa: E402
from sup3r_amd.strategy import ArrayStrategy  # noqa: E402

CFG = os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs')
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
chunk = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '75,75,38').split(','))
feats = ['u_10m', 'v_10m']
Sup3rGan.seed(3)
means = {f: np.float32(0.1 * (i + 1)) for i, f in enumerate(feats)}
stds = {f: np.float32(1.5 + i) for i, f in enumerate(feats)}
means['topography'], stds['topography'] = np.float32(300), np.float32(150)
m1 = Sup3rGan(os.path.join(CFG, 'sup3r', 'spatial', 'gen_2x_2f.json'), os.path.join(CFG, 'disc_s_same.json'),
              means=means, stdevs=stds, precision='bf16')
m1.set_model_params(lr_features=feats, hr_out_features=feats, s_enhance=2, t_enhance=1)
m1.init_weights((1, 16, 16, 2), (1, 32, 32, 2))
# step 2: 5x with lo-res topography at the input and hi-res topography mid-network
spec2 = json.load(open(os.path.join(CFG, 'sup3r', 'sup3rcc', 'gen_wind_5x_1x_6f.json')))
for layer in spec2['hidden_layers']:
    if layer.get('filters') == 6:
        layer['filters'] = 2
m2 = Sup3rGan(spec2, os.path.join(CFG, 'disc_s_same.json'), means=means, stdevs=stds, precision='bf16')
m2.set_model_params(lr_features=feats + ['topography'], hr_out_features=feats,
                    hr_exo_features=['topography'], s_enhance=5, t_enhance=1)
m2.init_weights((1, 16, 16, 3), (1, 80, 80, 3))
ms = MultiStepGan([m1, m2])
rng = np.random.default_rng(0)
n1, n2, nt = 2 * chunk[0], 2 * chunk[1], 3 * chunk[2]
domain = rng.standard_normal((n1, n2, nt, 2)).astype(np.float32)
topo_hr = (300 + 150 * rng.standard_normal((n1 * 10, n2 * 10, 1))).astype(np.float32)
topo_mid = topo_hr.reshape(n1 * 2, 5, n2 * 2, 5, 1).mean(axis=(1, 3)).astype(np.float32)
exo = {'topography': {'steps': [
    {'model': 1, 'combine_type': 'input', 'data': topo_mid, 's_enhance': 2, 't_enhance': 1},
    {'model': 1, 'combine_type': 'layer', 'data': topo_hr, 's_enhance': 10, 't_enhance': 1}]}}
register_model('MultiStepGan', {'model_dirs': ['a', 'b']}, ms)
st = ArrayStrategy(domain, {'model_dirs': ['a', 'b']}, chunk, spatial_pad=0, temporal_pad=5,
                   exo_data=exo, model_class='MultiStepGan', max_nodes=1, model=ms)
fwp = ForwardPass(st, 0)
ids = [int(i) for i in st.node_chunks[0]]
print(len(ids), 'chunks of', chunk, 'batch', batch, flush=True)
for name, on in (('device chain', True), ('host chain (MultiStepGan.generate)', False)):
    for rep in range(2):
        t0 = time.perf_counter()
        n = 0
        for c, failed, d in ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids), ms, batch=batch,
                                                    options={'device_chains': on}):
            assert not failed
            n += 1
            last = d
        el = time.perf_counter() - t0
        print(f'{name}: {n} chunks in {el*1e3:.1f} ms = {n/el:.2f} chunks/s, out {last.shape} {last.dtype}', flush=True)
if os.environ.get('PROFILE'):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for c, failed, d in ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids), ms, batch=batch):
        pass
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
