"""chunks/s of a 2-D (spatial) generator through ForwardPass.iter_chunks at the
config_fwp_spatial.json chunk shape (75 x 75 x 38, temporal_pad 5)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sup3r_amd import Sup3rGan, ForwardPass
from sup3r_amd.forward_pass import register_model
from sup3r_amd.strategy import ArrayStrategy

CFG = os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs')
feats = ['u_10m', 'v_10m']
Sup3rGan.seed(3)
means = {f: np.float32(0.1 * (i + 1)) for i, f in enumerate(feats)}
stds = {f: np.float32(1.5 + i) for i, f in enumerate(feats)}
m = Sup3rGan(os.path.join(CFG, 'sup3r', 'spatial', 'gen_2x_2f.json'), os.path.join(CFG, 'disc_s_same.json'),
             means=means, stdevs=stds, precision='bf16')
m.set_model_params(lr_features=feats, hr_out_features=feats, s_enhance=2, t_enhance=1)
m.init_weights((1, 16, 16, 2), (1, 32, 32, 2))
print('is_4d', m.is_4d, 'is_5d', m.is_5d, flush=True)
rng = np.random.default_rng(0)
domain = rng.standard_normal((150, 150, 190, 2)).astype(np.float32)
register_model('Sup3rGan', {'model_dir': 'fwp2d'}, m)
st = ArrayStrategy(domain, {'model_dir': 'fwp2d'}, (75, 75, 38), spatial_pad=0, temporal_pad=5, max_nodes=1, model=m)
fwp = ForwardPass(st, 0)
ids = [int(i) for i in st.node_chunks[0]]
print(len(ids), 'chunks', flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    n = 0
    for c, failed, d in ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids), m, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        assert not failed
        n += 1
        last = d
    el = time.perf_counter() - t0
    print(f'{n} chunks in {el*1e3:.1f} ms = {n/el:.1f} chunks/s, out {last.shape} {last.dtype}', flush=True)
if os.environ.get('PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for c, failed, d in ForwardPass.iter_chunks((fwp.get_input_chunk(i) for i in ids), m, batch=4):
        pass
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
