"""Pins the numpy oracle (oracle/) against an independent torch-CPU float64
autograd implementation (tests/torch_ref.py) and against exact-integer cases
for the permutation ops.  CPU-only; this is the oracle's own pin (SURVEY.md
§8c: the reference holds no golden tensors for this path)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import layers as L
from oracle.gan import Adam, mae, mse, rel_bce
from oracle.network import Network
from tests.torch_ref import TorchNet

CFG = os.path.join(os.path.dirname(__file__), '..', 'sup3r_amd', 'configs')


def _load(name):
    with open(os.path.join(CFG, name)) as f:
        return json.load(f)


def _net_pair(spec, x, exo=None, seed=1):
    net = Network(spec)
    net.init_weights(x, exo, seed=seed, bias_scale=0.1)
    tnet = TorchNet(spec, net.weights)
    return net, tnet


CASES = [
    ('test_gen_st_2x_4x_2f.json', (2, 5, 6, 4, 3), None),
    ('test_gen_st_3x_4x_2f_topo.json', (1, 4, 5, 4, 2), 'topography'),
    ('test_gen_s_2x_2f.json', (3, 7, 6, 2), None),
    ('test_disc_st_same.json', (2, 12, 12, 16, 2), None),
    ('test_disc_s_same.json', (2, 20, 20, 2), None),
    ('test_disc_st_valid.json', (2, 14, 13, 15, 2), None),
    ('test_gen_st_convT3d.json', (2, 5, 6, 4, 3), None),    # Conv3DTranspose
]


@pytest.mark.parametrize('cfg,shape,exo_name', CASES)
def test_network_forward_backward_vs_torch(cfg, shape, exo_name):
    rng = np.random.default_rng(42)
    spec = _load(cfg)
    x = rng.standard_normal(shape)
    exo = None
    if exo_name:
        # hi-res exo: (N, s1*3, s2*3, t*4, 1)
        hs = (shape[0], shape[1] * 3, shape[2] * 3, shape[3] * 4, 1)
        exo = {exo_name: rng.standard_normal(hs)}
    net, tnet = _net_pair(spec, x, exo)
    net.cast(np.float64)
    y = net.forward(x, exo)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    texo = None if exo is None else {
        k: torch.tensor(v, dtype=torch.float64) for k, v in exo.items()}
    yt = tnet.forward(xt, texo)
    assert tuple(yt.shape) == y.shape
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=0, atol=1e-11)
    dy = rng.standard_normal(y.shape)
    dx = net.backward(dy)
    yt.backward(torch.tensor(dy))
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=0, atol=1e-10)
    for g, tw in zip(net.grads, tnet.weights):
        np.testing.assert_allclose(g, tw.grad.numpy(), rtol=1e-9, atol=1e-9)


def test_depth_to_space_dcr_exact():
    # TF depth_to_space: out[n, h*b+i, w*b+j, c] = in[n, h, w, (i*b+j)*Co + c]
    for b, co in ((2, 3), (3, 2), (5, 8)):
        n, h, w = 2, 3, 4
        x = np.arange(n * h * w * b * b * co, dtype=np.int64).reshape(
            n, h, w, b * b * co)
        y = L.depth_to_space(x, b)
        assert y.shape == (n, h * b, w * b, co)
        for (hh, ww, i, j, c) in [(0, 0, 0, 0, 0), (1, 2, b - 1, 0, co - 1),
                                  (2, 3, 1, b - 1, 0), (0, 1, b - 1, b - 1, 1)]:
            assert y[1, hh * b + i, ww * b + j, c] == \
                x[1, hh, ww, (i * b + j) * co + c]
        np.testing.assert_array_equal(L.space_to_depth(y, b), x)
        # differs from torch's CRD pixel_shuffle channel order
        xc = torch.tensor(x).permute(0, 3, 1, 2)
        crd = torch.nn.functional.pixel_shuffle(xc, b).permute(0, 2, 3, 1)
        assert not np.array_equal(crd.numpy(), y)


def test_temporal_nearest_exact():
    x = np.arange(2 * 2 * 2 * 3 * 2).reshape(2, 2, 2, 3, 2)
    for m in (2, 3):
        y = L.SpatioTemporalExpansion(temporal_mult=m).forward(x)
        assert y.shape == (2, 2, 2, 3 * m, 2)
        for j in range(3 * m):
            np.testing.assert_array_equal(y[..., j, :], x[..., j // m, :])


def test_reflect_pad_conv_crop_identity():
    """REFLECT pad 3 -> valid conv k3 -> crop 2  ==  REFLECT pad 1 -> valid
    conv; the fusion the HIP kernels rely on."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 5, 6, 7, 3))
    conv = L.ConvND(3, 4, 3)
    conv.build(3, rng, np.float64)
    a = L.Cropping(2, 3).forward(conv.forward(
        L.FlexiblePadding([[0, 0], [3, 3], [3, 3], [3, 3], [0, 0]]).forward(x)))
    b = conv.forward(
        L.FlexiblePadding([[0, 0], [1, 1], [1, 1], [1, 1], [0, 0]]).forward(x))
    np.testing.assert_allclose(a, b, atol=1e-13)


def test_losses_vs_torch():
    rng = np.random.default_rng(3)
    a = rng.standard_normal((3, 4, 5, 2))
    b = rng.standard_normal((3, 4, 5, 2))
    for fn, tfn in ((mae, torch.nn.functional.l1_loss),
                    (mse, torch.nn.functional.mse_loss)):
        ta = torch.tensor(a, requires_grad=True)
        tb = torch.tensor(b, requires_grad=True)
        tl = tfn(ta, tb)
        tl.backward()
        loss, ga, gb = fn(a, b)
        np.testing.assert_allclose(loss, tl.item(), rtol=1e-13)
        np.testing.assert_allclose(ga, ta.grad.numpy(), atol=1e-15)
        np.testing.assert_allclose(gb, tb.grad.numpy(), atol=1e-15)
    dt = rng.standard_normal((15, 1)) * 3
    dg = rng.standard_normal((15, 1)) * 3
    tdt = torch.tensor(dt, requires_grad=True)
    tdg = torch.tensor(dg, requires_grad=True)
    logits = torch.cat([tdt - tdg.mean(), tdg - tdt.mean()], 0)
    labels = torch.cat([torch.ones_like(tdt), torch.zeros_like(tdg)], 0)
    tl = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels)
    tl.backward()
    loss, g_t, g_g = rel_bce(dt, dg)
    np.testing.assert_allclose(loss, tl.item(), rtol=1e-13)
    np.testing.assert_allclose(g_t, tdt.grad.numpy(), atol=1e-15)
    np.testing.assert_allclose(g_g, tdg.grad.numpy(), atol=1e-15)


def test_adam_vs_torch():
    """keras-2.15 Adam update vs torch.optim.Adam.  The two differ only in
    where epsilon enters (keras: alpha*m/(sqrt(v)+eps) with bias correction
    folded into alpha; torch: eps added after the sqrt(v)/sqrt(1-b2^t)), so
    they agree to O(eps) — checked with a loose-enough tolerance and an exact
    closed form for step 1."""
    rng = np.random.default_rng(5)
    w = rng.standard_normal((7, 3))
    gs = [rng.standard_normal((7, 3)) for _ in range(3)]
    wk = w.copy()
    opt = Adam(learning_rate=1e-2)
    tw = torch.tensor(w.copy(), requires_grad=True)
    topt = torch.optim.Adam([tw], lr=1e-2, betas=(0.9, 0.999), eps=1e-7)
    for i, g in enumerate(gs):
        opt.apply_gradients([g], [wk])
        tw.grad = torch.tensor(g)
        topt.step()
        np.testing.assert_allclose(wk, tw.detach().numpy(), atol=2e-5)
        if i == 0:
            # t=1: m=(1-b1)g, v=(1-b2)g^2, alpha=lr*sqrt(1-b2)/(1-b1)
            alpha = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)
            exp = w - alpha * 0.1 * g / (np.sqrt(0.001 * g * g) + 1e-7)
            np.testing.assert_allclose(wk, exp, rtol=1e-12, atol=1e-14)


def test_param_counts_match_survey():
    """A1: parse result pinned by the param counts in SURVEY.md §8a."""
    g = Network(_load('gen_5x_12x_2f.json'))
    g.forward(np.zeros((1, 4, 4, 4, 4), np.float32))
    assert sum(w.size for w in g.weights) == 4226170
    d = Network(_load('disc_st.json'))
    # 80x80x288 production shape is too large to run on CPU here; count
    # params analytically from the spec walk: conv stack + dense on (2,2,15,256)
    sp = np.array([80, 80, 288])
    cin, total = 2, 0
    for w in (32, 64, 128, 256):
        for s in (1, 2):
            total += 27 * cin * w + w
            cin = w
            sp = (sp - 3) // s + 1
    flat = int(np.prod(sp)) * cin
    assert tuple(sp) == (2, 2, 15) and flat == 15360
    total += flat * 2048 + 2048 + 2048 * 1024 + 1024 + 1024 + 1
    assert total == 37072513
