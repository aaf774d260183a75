"""Saved-network files: this package's own
``sup3r_amd.network.v1`` layout — what ``Network.save`` and
``tools/convert_phygnn_pkl.py`` write — and the ``model_params`` dict layout
of ``phygnn.CustomNetwork.save`` (restated, see ``engine.read_network_file``).
"""
import collections
import importlib.util
import json
import os
import pickle

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')


def _oracle(cfg, shape, seed):
    from oracle.network import Network
    with open(os.path.join(CFG, cfg)) as f:
        spec = json.load(f)
    net = Network(spec)
    net.init_weights(np.zeros(shape, np.float32), seed=seed, bias_scale=0.1)
    return spec['hidden_layers'], net


def _phygnn_layout(hidden, onet):
    """model_params as CustomNetwork.save pickles it: weight_dict maps the
    EXPANDED layer index to that layer's get_weights() (empty for layers
    without variables; a shared SkipConnection appears at both indices)"""
    wd = collections.OrderedDict()
    seen = set()
    for i, layer in enumerate(onet.layers):
        own = hasattr(layer, 'kernel') and id(layer) not in seen
        seen.add(id(layer))
        wd[i] = [np.array(w) for w in layer.weights] if own else []
    return {'hidden_layers': hidden, 'weight_dict': wd, 'name': 'generator',
            'version_record': {'phygnn': '0.0.33'}}


def test_read_both_layouts(tmp_path):
    from sup3r_amd.engine import read_network_file
    hidden, onet = _oracle('test_gen_st_2x_4x_2f.json', (1, 6, 6, 6, 2), 3)
    fp = os.path.join(str(tmp_path), 'model_gen.pkl')
    with open(fp, 'wb') as f:
        pickle.dump(_phygnn_layout(hidden, onet), f)
    h, w, name = read_network_file(fp)
    assert h == hidden and name == 'generator'
    assert len(w) == len(onet.weights)
    for a, b in zip(w, onet.weights):
        np.testing.assert_array_equal(a, b)
    fp1 = os.path.join(str(tmp_path), 'v1.pkl')
    with open(fp1, 'wb') as f:
        pickle.dump({'format': 'sup3r_amd.network.v1', 'name': 'g',
                     'hidden_layers': hidden, 'weights': onet.weights}, f)
    h1, w1, _ = read_network_file(fp1)
    assert h1 == hidden and len(w1) == len(w)
    # neither layout / a pickle that needs foreign classes: TypeError
    fp2 = os.path.join(str(tmp_path), 'other.pkl')
    with open(fp2, 'wb') as f:
        pickle.dump({'something': 1}, f)
    with pytest.raises(TypeError):
        read_network_file(fp2)
    fp3 = os.path.join(str(tmp_path), 'foreign.pkl')
    with open(fp3, 'wb') as f:      # GLOBAL no_such_module.Thing
        f.write(b'cno_such_module\nThing\n.')
    with pytest.raises(TypeError, match='convert_phygnn_pkl'):
        read_network_file(fp3)


def test_converter_output_is_the_v1_layout(tmp_path):
    """tools/convert_phygnn_pkl.py's ``network_blob`` on a stand-in for a
    live ``CustomNetwork`` (``.model_params``, ``.weights`` of variables with
    ``.numpy()``): what it writes is what ``read_network_file`` reads"""
    spec = importlib.util.spec_from_file_location(
        'convert_phygnn_pkl', os.path.join(ROOT, 'tools',
                                           'convert_phygnn_pkl.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hidden, onet = _oracle('test_disc_st_same.json', (1, 8, 8, 16, 2), 4)

    class Var:
        def __init__(self, a):
            self.a = a

        def numpy(self):
            return self.a

    class Live:
        model_params = {'hidden_layers': hidden, 'name': 'discriminator'}
        weights = [Var(w.astype(np.float64)) for w in onet.weights]
    blob = mod.network_blob(Live())
    assert blob['format'] == 'sup3r_amd.network.v1'
    assert all(w.dtype == np.float32 for w in blob['weights'])
    fp = os.path.join(str(tmp_path), 'model_disc.pkl')
    with open(fp, 'wb') as f:
        pickle.dump(blob, f)
    from sup3r_amd.engine import read_network_file
    h, w, name = read_network_file(fp)
    assert h == hidden and name == 'discriminator'
    for a, b in zip(w, onet.weights):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_load_a_reference_style_model_dir(tmp_path):
    """a model directory whose networks are phygnn-layout pickles:
    ``Sup3rGan.load`` -> ``generate`` equals the oracle, a checkpoint written
    back re-loads bit-identically"""
    from oracle.gan import norm_input, un_norm_output
    from sup3r_amd import Sup3rGan
    d = str(tmp_path)
    hg, og = _oracle('test_gen_st_2x_4x_2f.json', (1, 6, 6, 6, 2), 3)
    hd, od = _oracle('test_disc_st_same.json', (1, 12, 12, 24, 2), 4)
    for fn, h, o in (('model_gen.pkl', hg, og), ('model_disc.pkl', hd, od)):
        with open(os.path.join(d, fn), 'wb') as f:
            pickle.dump(_phygnn_layout(h, o), f)
    feats = ['u_10m', 'v_10m']
    params = {'name': 'Sup3rGan', 'loss': 'MeanAbsoluteError',
              'means': {'u_10m': 0.3, 'v_10m': -0.2},
              'stdevs': {'u_10m': 1.5, 'v_10m': 2.0},
              'meta': {'lr_features': feats, 'hr_out_features': feats,
                       's_enhance': 2, 't_enhance': 4},
              'optimizer': {'name': 'Adam', 'learning_rate': 1e-4},
              'optimizer_disc': {'name': 'Adam', 'learning_rate': 1e-4},
              'default_device': None}
    with open(os.path.join(d, 'model_params.json'), 'w') as f:
        json.dump(params, f)
    m = Sup3rGan.load(d)
    x = np.random.default_rng(1).standard_normal((2, 6, 6, 6, 2)).astype(
        np.float32)
    y = m.generate(x)
    ref = un_norm_output(og.forward(norm_input(x, [0.3, -0.2], [1.5, 2.0])
                                    .astype(np.float32)),
                         [0.3, -0.2], [1.5, 2.0])
    assert np.abs(y - ref).max() < 1e-4 * max(1, np.abs(ref).max())
    out = os.path.join(d, 'again')
    m.save(out)
    np.testing.assert_array_equal(Sup3rGan.load(out).generate(x), y)
