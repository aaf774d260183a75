""""TF-CPU proxy" baseline (TEST / BASELINE INFRASTRUCTURE, not product): the
generator forward executed AS TF EXECUTES IT (REFLECT pad-3 -> valid Conv3D ->
crop-2 un-fused, fp32) through torch-CPU, i.e. oneDNN convolutions — the same
x86 conv backend TensorFlow 2.15 enables by default.  TensorFlow itself is not
installable where this repo is built or run, so this number is
labelled a proxy, never a TF measurement.  Weights come from the numpy oracle
network so both baselines compute the same function."""
import numpy as np


def torch_generator_forward(oracle_net, x, threads=None):
    """Run ``oracle_net`` (oracle.network.Network, built) on x with torch-CPU
    ops.  Returns (y numpy, seconds)."""
    import time

    import torch
    import torch.nn.functional as F

    from . import layers as L
    if threads:
        torch.set_num_threads(int(threads))
    t = torch.from_numpy(np.ascontiguousarray(x)).float()
    nd = t.dim() - 2
    t = t.permute(0, nd + 1, *range(1, nd + 1)).contiguous(
        memory_format=torch.channels_last_3d if nd == 3
        else torch.channels_last)
    packed = {}
    for layer in oracle_net.weight_layers:
        if isinstance(layer, L.ConvND):
            w = torch.from_numpy(layer.kernel).float()
            w = w.permute(nd + 1, nd, *range(nd)).contiguous()
            packed[id(layer)] = (w, torch.from_numpy(layer.bias).float())
    skips = {}
    t0 = time.time()
    with torch.no_grad():
        for layer in oracle_net.layers:
            if isinstance(layer, L.FlexiblePadding):
                flat = []
                for lo, hi in reversed(layer.paddings[1:-1]):
                    flat += [lo, hi]
                t = F.pad(t, flat, mode='reflect')
            elif isinstance(layer, L.ConvND):
                w, b = packed[id(layer)]
                t = (F.conv3d if nd == 3 else F.conv2d)(
                    t, w, b, stride=layer.strides)
            elif isinstance(layer, L.Cropping):
                sl = [slice(None), slice(None)] + [
                    slice(lo, t.shape[2 + d] - hi)
                    for d, (lo, hi) in enumerate(layer.cropping)]
                t = t[tuple(sl)]
            elif isinstance(layer, L.LeakyReLU):
                t = F.leaky_relu(t, layer.alpha)
            elif isinstance(layer, L.SkipConnection):
                if layer.name in skips:
                    t = t + skips.pop(layer.name)
                else:
                    skips[layer.name] = t
            elif isinstance(layer, L.SpatioTemporalExpansion):
                if layer._temporal_mult > 1:
                    t = torch.repeat_interleave(t, layer._temporal_mult, dim=4)
                b = layer._spatial_mult
                if b > 1:
                    n, c, s1, s2, tt = t.shape
                    co = c // (b * b)
                    # DCR: channel = (i*b + j)*co + c'
                    t = t.reshape(n, b, b, co, s1, s2, tt).permute(
                        0, 3, 4, 1, 5, 2, 6).reshape(n, co, s1 * b, s2 * b, tt)
            else:
                raise KeyError(type(layer).__name__)
    dt = time.time() - t0
    y = t.permute(0, *range(2, nd + 2), 1).contiguous().numpy()
    return y, dt
