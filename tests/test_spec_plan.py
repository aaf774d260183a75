"""Host-logic tests (CPU): spec parser + fusion planner of sup3r_amd.spec
against the layer-by-layer oracle, and the reference's shape contracts
(tests/training/test_load_configs.py in the reference)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle.network import Network
from sup3r_amd import spec as S
from tests.plan_interp import run_plan

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')
REF_CFG = '/root/reference/sup3r/configs'


def _load(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


CASES = [
    ('test_gen_st_2x_4x_2f.json', (2, 5, 6, 4, 3), None, (2, 10, 12, 16, 2)),
    ('test_gen_st_3x_4x_2f_topo.json', (1, 4, 5, 4, 2), 'topography',
     (1, 12, 15, 16, 2)),
    ('test_gen_s_2x_2f.json', (3, 7, 6, 2), None, (3, 14, 12, 2)),
    ('test_disc_st_same.json', (2, 12, 12, 16, 2), None, (2, 1)),
    ('test_disc_s_same.json', (2, 20, 20, 2), None, (2, 1)),
    ('test_disc_st_valid.json', (2, 14, 13, 15, 2), None, (2, 1)),
]


@pytest.mark.parametrize('cfg,shape,exo_name,out_shape', CASES)
@pytest.mark.parametrize('fuse', [True, False])
def test_fused_plan_matches_oracle(cfg, shape, exo_name, out_shape, fuse):
    rng = np.random.default_rng(7)
    spec = _load(cfg)
    x = rng.standard_normal(shape)
    exo = None
    if exo_name:
        exo = {exo_name: rng.standard_normal(out_shape[:-1] + (1,))}
    net = Network(spec)
    net.init_weights(x, exo, seed=3, bias_scale=0.1)
    net.cast(np.float64)
    y_ref = net.forward(x, exo)
    assert y_ref.shape == out_shape

    layers = S.parse_layers(spec)
    plan = S.build_plan(layers, shape, fuse=fuse)
    assert plan.out_shape == out_shape
    keras_w = net.weights
    assert [tuple(p['shape']) for p in plan.params] == \
        [w.shape for w in keras_w]
    canon = [S.keras_to_canonical(w, p['layout'])
             for w, p in zip(keras_w, plan.params)]
    for w, c, p in zip(keras_w, canon, plan.params):
        np.testing.assert_array_equal(
            S.canonical_to_keras(c, p['layout']), w.astype(np.float32))
    inputs = {'x': x}
    if exo:
        inputs.update(exo)
    y = run_plan(plan, [np.asarray(w, np.float64) if p['layout'] == S.WL_CONV
                        else S.keras_to_canonical(w, p['layout']).astype(
                            np.float64)
                        for w, p in zip(keras_w, plan.params)], inputs)
    # canonical arrays went through float32 for ConvT layouts only
    np.testing.assert_allclose(y, y_ref, rtol=0, atol=2e-6)
    if fuse:
        n_conv = sum(op['kind'] == S.OP_CONV for op in plan.ops)
        assert not any(op['kind'] in (S.OP_PAD, S.OP_CROP)
                       for op in plan.ops)
        assert n_conv == sum(p['kind'] == 'kernel' and len(p['shape']) > 2
                             for p in plan.params)


def test_c2_plan_is_fully_fused():
    layers = S.parse_layers(_load('gen_5x_12x_2f.json'))
    plan = S.build_plan(layers, (1, 16, 16, 24, 4))
    assert plan.out_shape == (1, 80, 80, 288, 2)
    kinds = [op['kind'] for op in plan.ops]
    assert kinds.count(S.OP_CONV) == 38
    assert kinds.count(S.OP_REPEAT_T) == 3
    assert set(kinds) == {S.OP_CONV, S.OP_REPEAT_T}
    # 17 residual adds are conv epilogues, d2s is a store permutation
    assert sum(op['kind'] == S.OP_CONV and op['res'] >= 0
               for op in plan.ops) == 17
    assert sum(op.get('d2s', 1) == 5 for op in plan.ops) == 1
    assert plan.n_params() == 4226170
    s_enh = int(np.prod([L._spatial_mult for L in layers]))
    t_enh = int(np.prod([L._temporal_mult for L in layers]))
    assert (s_enh, t_enh) == (5, 12)
    assert layers[0].rank == 5


@pytest.mark.skipif(not os.path.isdir(REF_CFG),
                    reason='reference configs not present on this box')
def test_reference_generator_configs_shape_contract():
    """Every generator config the reference ships parses unchanged and obeys
    the shape contract of tests/training/test_load_configs.py:17-133."""
    files = sorted(glob.glob(os.path.join(REF_CFG, '*', 'gen_*.json')))
    assert len(files) >= 16
    for fp in files:
        with open(fp) as f:
            spec = json.load(f)
        layers = S.parse_layers(spec)
        s_enh = int(np.prod([L._spatial_mult for L in layers]))
        t_enh = int(np.prod([L._temporal_mult for L in layers]))
        name = os.path.basename(fp)
        parts = name.replace('.json', '').split('_')
        nums = [p for p in parts if p.endswith('x')]
        nf_in = int([p for p in parts if p.endswith('f')][0][:-1])
        is_5d = layers[0].rank == 5
        if is_5d:
            assert (s_enh, t_enh) == (int(nums[0][:-1]), int(nums[1][:-1])), fp
        else:
            assert s_enh == int(nums[0][:-1]), fp
        for n, s, t in ((1, 5, 4), (4, 7, 6)):
            shape = (n, s, s, t, nf_in) if is_5d else (n, s, s, nf_in)
            plan = S.build_plan(layers, shape)
            out = plan.out_shape
            assert out[0] == n and out[1] == s * s_enh and out[2] == s * s_enh
            if is_5d:
                assert out[3] == t * t_enh
            # parameter table is independent of the input shape
            S.build_plan(layers, shape, param_table=plan.params)


def test_bad_expansion_raises():
    spec = [{'class': 'Conv2D', 'filters': 6, 'kernel_size': 3,
             'padding': 'same'},
            {'class': 'SpatialExpansion', 'spatial_mult': 2}]
    with pytest.raises(RuntimeError):
        S.build_plan(S.parse_layers(spec), (1, 8, 8, 2))
    with pytest.raises(KeyError):
        S.parse_layers([{'class': 'NotALayer'}])
    with pytest.raises(KeyError):
        S.parse_layers([{'repeat': [{'class': 'Flatten'}]}])


STRIDED_T = [
    # strided Conv3DTranspose / Conv2DTranspose (named in north_star; 0
    # occurrences in the reference's configs): with and without cropping of
    # the zero tails, behind a REFLECT pad, anisotropic strides
    ([{'class': 'Conv3DTranspose', 'filters': 5, 'kernel_size': 3,
       'strides': 2}], (2, 3, 4, 3, 2), (2, 7, 9, 7, 5)),
    ([{'class': 'Conv3DTranspose', 'filters': 4, 'kernel_size': 3,
       'strides': [2, 1, 3]},
      {'class': 'Cropping3D', 'cropping': [[1, 2], [0, 1], [2, 0]]},
      {'alpha': 0.2, 'class': 'LeakyReLU'}], (1, 4, 3, 3, 3),
     (1, 6, 4, 7, 4)),
    ([{'class': 'FlexiblePadding', 'mode': 'REFLECT',
       'paddings': [[0, 0], [1, 1], [1, 1], [0, 0]]},
      {'class': 'Conv2DTranspose', 'filters': 3, 'kernel_size': 3,
       'strides': 2, 'activation': 'relu'},
      {'class': 'Cropping2D', 'cropping': 2}], (2, 4, 5, 2), (2, 9, 11, 3)),
]


@pytest.mark.parametrize('spec,shape,out_shape', STRIDED_T)
def test_strided_transpose_lowering(spec, shape, out_shape):
    """zero insertion (S3_OP_DILATE) + the stride-1 flipped-kernel conv ==
    keras ConvNDTranspose(strides=s) as the oracle (pinned against
    torch.conv_transpose in tests/test_oracle_vs_torch.py) computes it"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape)
    net = Network(spec)
    net.init_weights(x, seed=2, bias_scale=0.1)
    net.cast(np.float64)
    y_ref = net.forward(x)
    assert y_ref.shape == out_shape
    plan = S.build_plan(S.parse_layers(spec), shape)
    assert plan.out_shape == out_shape
    assert any(op['kind'] == S.OP_DILATE for op in plan.ops)
    params = [S.keras_to_canonical(w, p['layout']).astype(np.float64)
              for w, p in zip(net.weights, plan.params)]
    y = run_plan(plan, params, {'x': x})
    np.testing.assert_allclose(y, y_ref, rtol=0, atol=2e-6)


def test_strided_transpose_with_kernel_smaller_than_stride_is_refused():
    """keras 'valid' Conv*DTranspose output is in * s + max(k - s, 0); the
    zero-insertion lowering gives (in - 1) * s + k — equal only for k >= s,
    so k < s must not build a (shorter) plan silently"""
    import pytest
    from sup3r_amd import spec as S
    layers = S.parse_layers([{'class': 'Conv2DTranspose', 'filters': 4,
                              'kernel_size': 2, 'strides': 3}])
    with pytest.raises(KeyError, match='kernel_size'):
        S.build_plan(layers, (1, 5, 5, 2))
    ok = S.parse_layers([{'class': 'Conv2DTranspose', 'filters': 4,
                          'kernel_size': 3, 'strides': 2}])
    plan = S.build_plan(ok, (1, 5, 5, 2))
    assert tuple(plan.out_shape) == (1, 11, 11, 4)     # in * s + k - s
