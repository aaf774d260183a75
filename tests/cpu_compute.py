"""TEST INFRASTRUCTURE: a CPU stand-in for ``sup3r_amd.compute.HipGanCompute``
backed by the numpy oracle, installed as ``Sup3rGan._compute_factory`` by the
``-m "not gpu"`` tests of the HOST logic of training (loss futures, running
windows, train / skip gating, sharded multi-GPU step over gloo).  It is never
imported by the product and computes nothing the GPU tests rely on.
"""
import pickle

import numpy as np

from oracle.gan import Adam, GanOracle
from oracle.network import Network as OracleNet
from sup3r_amd import spec as S
from sup3r_amd.compute import (LossFuture, MAX_TERMS, SLOTS_PER_TERM,
                               details_from_scalars)


class _Dev:
    ctx = None

    def __init__(self):
        self.rank, self.nranks = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.nranks = dist.get_rank(), dist.get_world_size()
        except Exception:
            pass


class _Scal:
    """host array posing as the device loss-scalar buffer"""

    def __init__(self):
        self.a = np.zeros(4 + SLOTS_PER_TERM * MAX_TERMS, np.float64)

    def cpu(self):
        return self

    def numpy(self):
        return self.a

    def numel(self):
        return self.a.size


class _Plan:
    def __init__(self, net, shape):
        p = S.build_plan(net.layers, shape)
        self.out_shape = p.out_shape
        self.input_names = list(p.inputs)
        self.in_shapes = {k: p.tensors[v] for k, v in p.inputs.items()}
        self._net = net

    def forward(self, x, exo=None):
        return self._net.oracle.forward(np.asarray(x, np.float32), exo or None)


class CpuNetwork:
    def __init__(self, hidden_layers, name, dev):
        self.name, self.dev = name, dev
        self.hidden_layers = hidden_layers
        self.layers = S.parse_layers(hidden_layers)
        self.oracle = OracleNet(hidden_layers)
        self.built = False
        self.param_table = None
        self.grads_acc = None
        self.adam = None
        self._from_file = False
        self._pending = None

    def build(self, in_shape, seed=None):
        if self.built:
            return
        self.param_table = S.build_plan(self.layers, in_shape).params
        self.oracle.init_weights(np.zeros(in_shape, np.float32), seed=seed
                                 if seed is not None else 0)
        self.built = True
        if self._pending is not None:
            self.oracle.set_weights(self._pending)
            self._pending = None

    def plan(self, in_shape, training=False, **_):
        self.build(tuple(in_shape))
        return _Plan(self, tuple(in_shape))

    @property
    def weights(self):
        if not self.built:
            return list(self._pending or [])
        return [np.array(w) for w in self.oracle.weights]

    def set_weights(self, arrays):
        if not self.built:
            self._pending = [np.asarray(a, np.float32) for a in arrays]
        else:
            self.oracle.set_weights(arrays)

    def mean_abs(self, which, i):
        if self.adam is None or self.adam.m is None:
            return 0.0                  # untouched slots are zero
        slot = self.adam.m if which == 2 else self.adam.v
        return float(np.abs(slot[i]).mean())

    def save(self, fp):
        with open(fp, 'wb') as f:
            pickle.dump({'format': 'sup3r_amd.network.v1', 'name': self.name,
                         'hidden_layers': self.hidden_layers,
                         'weights': self.weights}, f)


class CpuGanCompute:
    """Same surface as HipGanCompute; arithmetic = oracle.gan.GanOracle."""

    supports_defer = True

    def __init__(self, gen_layers, disc_layers, device=None, precision=None):
        self.dev = device or _Dev()
        self.gen = CpuNetwork(gen_layers, 'generator', self.dev)
        self.disc = None if disc_layers is None else CpuNetwork(
            disc_layers, 'discriminator', self.dev)
        self.log = []            # (what, which) trace for the tests

    def new_scalars(self):
        return _Scal()

    def add_scalars(self, dst, src):
        dst.a += src.a

    def tf_generate(self, low_res, hi_res_exo=None, training=False):
        self.gen.build(tuple(np.shape(low_res)))
        return self.gen.oracle.forward(np.asarray(low_res, np.float32),
                                       hi_res_exo or None)

    def tf_discriminate(self, hi_res, training=False, slot=0):
        self.disc.build(tuple(np.shape(hi_res)))
        return self.disc.oracle.forward(np.asarray(hi_res, np.float32))

    def loss_and_grads(self, low_res, hi_res_true, loss_terms,
                       weight_gen_advers=0.001, train_gen=True,
                       train_disc=False, compute_disc=False, exo_names=(),
                       backward=True, hi_res_gen=None, mask=None,
                       accumulate_wgrad=False, scal=None, defer=False,
                       overlap_bucket=None):
        assert hi_res_gen is None and mask is None
        spec = {n: {} for n, *_ in loss_terms}
        spec['term_weights'] = [t[2] for t in loss_terms]
        orc = GanOracle(self.gen.oracle, self.disc.oracle, loss=spec)
        lr = np.asarray(low_res, np.float32)
        hr = np.asarray(hi_res_true, np.float32)
        _, det, grads = orc.loss_and_grads(
            lr, hr, weight_gen_advers, train_gen=train_gen,
            train_disc=train_disc, compute_disc=compute_disc,
            exo_names=exo_names)
        if scal is None:
            scal = _Scal()
        a = scal.a
        a[:] = 0.0                   # the device loss kernels WRITE their slots
        if 'loss_disc' in det:
            a[0] += det['loss_disc']
        if train_gen:
            a[1] += det['loss_gen_advers']
            from oracle.gan import camel_to_underscore
            for i, (name, *_r) in enumerate(loss_terms):
                a[4 + SLOTS_PER_TERM * i] += det[camel_to_underscore(name)]
        if backward and grads is not None:
            net = self.gen if train_gen else self.disc
            grads = [np.array(g, np.float64) for g in grads]
            if accumulate_wgrad and net.grads_acc is not None:
                net.grads_acc = [x + g for x, g in zip(net.grads_acc, grads)]
            else:
                net.grads_acc = grads
        coefs = [[1.0]] * len(loss_terms) if train_gen else None
        with_disc = compute_disc or train_disc

        def recipe(vals):
            return details_from_scalars(vals, loss_terms, coefs, with_disc,
                                        train_gen, True, weight_gen_advers)
        fut = LossFuture(scal, recipe)
        if defer:
            return None, fut, None
        d = fut.resolve()
        key = 'loss_gen' if train_gen else 'loss_disc'
        return d.get(key), d, None

    def _net(self, which):
        return self.gen if which == 'gen' else self.disc

    def allreduce_grads(self, which):
        import torch
        import torch.distributed as dist
        self.log.append(('allreduce', which))
        net = self._net(which)
        out = []
        for g in net.grads_acc:
            t = torch.from_numpy(np.ascontiguousarray(g))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            out.append(t.numpy())
        net.grads_acc = out

    def allreduce_scalars(self, scal):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(scal.a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def broadcast_state(self, root=0):
        import torch
        import torch.distributed as dist
        self.log.append(('broadcast', root))
        for net in (self.gen, self.disc):
            ws = []
            for w in net.oracle.weights:
                t = torch.from_numpy(np.ascontiguousarray(w))
                dist.broadcast(t, src=root)
                ws.append(t.numpy())
            net.oracle.set_weights(ws)

    def apply(self, which, optimizer):
        net = self._net(which)
        cfg = optimizer.get_config()
        if net.adam is None:
            net.adam = Adam(cfg['learning_rate'], cfg['beta_1'], cfg['beta_2'],
                            cfg['epsilon'])
        net.adam.learning_rate = cfg['learning_rate']
        optimizer.iterations += 1
        self.log.append(('adam', which))
        net.adam.apply_gradients(
            [g.astype(np.float32) for g in net.grads_acc], net.oracle.weights)
