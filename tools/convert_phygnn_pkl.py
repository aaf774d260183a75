#!/usr/bin/env python
"""Convert a sup3r model directory written by the TensorFlow reference
(``model_gen.pkl`` / ``model_disc.pkl`` = ``phygnn.CustomNetwork.save``,
sup3r/models/base.py:133-157, abstract.py:96-101) into the layout
``sup3r_amd.Sup3rGan.load`` reads.

Runs WHERE phygnn + tensorflow are installed (not in the MI355X image):

    python tools/convert_phygnn_pkl.py /path/to/sup3r_model_dir out_dir

Each network is loaded with ``phygnn.CustomNetwork.load`` — the authoritative
reader of that pickle — and re-written as a ``sup3r_amd.network.v1`` file:
``{'format', 'name', 'hidden_layers', 'weights'}`` with ``hidden_layers`` the
JSON-style layer list the network was built from and ``weights`` its
variables as float32 numpy arrays in keras order (kernel, bias per layer:
what ``CustomNetwork.weights`` yields).  ``model_params.json`` and
``history.csv`` are copied.  The written files contain numpy arrays and
builtins only, so nothing TF-side is needed to read them.

``--check`` re-reads the written files with plain ``pickle`` and compares
the arrays with the live network's variables bit for bit.
"""
import argparse
import os
import pickle
import shutil
import sys

import numpy as np

FORMAT = 'sup3r_amd.network.v1'


def network_blob(net, name=None):
    """``phygnn.CustomNetwork`` -> the dict ``sup3r_amd.engine.Network.load``
    reads."""
    params = net.model_params
    hidden = params.get('hidden_layers')
    if hidden is None:
        raise KeyError('CustomNetwork.model_params has no "hidden_layers": '
                       f'{sorted(params)}')
    weights = [np.asarray(w.numpy() if hasattr(w, 'numpy') else w,
                          dtype=np.float32) for w in net.weights]
    return {'format': FORMAT, 'name': name or params.get('name'),
            'hidden_layers': hidden, 'weights': weights}


def convert_network(fp_in, fp_out, name=None, check=False):
    from phygnn import CustomNetwork
    net = CustomNetwork.load(fp_in)
    blob = network_blob(net, name)
    with open(fp_out, 'wb') as f:
        pickle.dump(blob, f)
    if check:
        with open(fp_out, 'rb') as f:
            back = pickle.load(f)
        assert back['format'] == FORMAT
        assert len(back['weights']) == len(net.weights)
        for a, w in zip(back['weights'], net.weights):
            np.testing.assert_array_equal(a, np.asarray(w.numpy()))
    return len(blob['weights']), int(sum(w.size for w in blob['weights']))


def convert_dir(model_dir, out_dir, check=False):
    os.makedirs(out_dir, exist_ok=True)
    done = {}
    for fn, name in (('model_gen.pkl', 'generator'),
                     ('model_disc.pkl', 'discriminator')):
        src = os.path.join(model_dir, fn)
        if os.path.exists(src):
            done[fn] = convert_network(src, os.path.join(out_dir, fn), name,
                                       check=check)
    for fn in ('model_params.json', 'history.csv'):
        src = os.path.join(model_dir, fn)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(out_dir, fn))
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('model_dir')
    ap.add_argument('out_dir')
    ap.add_argument('--check', action='store_true')
    args = ap.parse_args(argv)
    try:
        import phygnn  # noqa: F401
    except ImportError:
        sys.exit('phygnn is not installed here: run this on the machine that '
                 'trained the model (or any box with tensorflow + phygnn)')
    for fn, (n, size) in convert_dir(args.model_dir, args.out_dir,
                                     args.check).items():
        print(f'{fn}: {n} arrays, {size} parameters')


if __name__ == '__main__':
    main()
