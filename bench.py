#!/usr/bin/env python
"""Headline benchmark: BASELINE.json config C2 — SpatioTemporal 5x/12x Sup3rGan
generator forward, lo-res chunks (B,16,16,24,4) -> hi-res (B,80,80,288,2), on
N MI355X (one process per GPU, chunks sharded data-parallel, no data-path
collective: ``scaling = weak``).

A "step" is one pass of the hot path (``Sup3rGan._tf_generate``'s layer loop,
sup3r/models/abstract.py:1131-1173, here one s3_plan_forward) over one batch
of synthetic chunks already resident in HBM.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Order of the default single-GPU run: the timed headline region, then the other
GPU legs (BF16X3 / fp32 parity modes, >= 5 s of C2 training steps, the PMC
traffic passes), and only then the CPU legs (numpy oracle as the parity
checker, C/OpenMP and torch-CPU baselines).

Other modes (``value`` is then that mode's rate):
  --mode train  one ``Sup3rGan._train_batch`` per step, data-parallel over the
                ranks (global batch --batch split on axis 0, RCCL gradient
                SUM, ``scaling = strong``); --config c2 | c4 | c4toy
  --mode c3     the per-chunk executor over a 400x400x720 domain tiled into
                20x20x48 chunks, chunks sharded over the ranks
  --mode c1     BASELINE configs[0]: the spatial 2x generator (gen_2x_2f, 36
                Conv2DTranspose layers) on (--batch, 10, 10, 2) observations —
                one whole-network launch per forward (kernels_fused2d.hip)
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFGDIR = os.path.join(ROOT, 'sup3r_amd', 'configs')
CFG = os.path.join(CFGDIR, 'gen_5x_12x_2f.json')
LR_SHAPE = (16, 16, 24, 4)
HR_SHAPE = (80, 80, 288, 2)
# algorithmic work, SURVEY.md §8(d) / DESIGN.md: generator forward per sample
GEN_FLOP_PER_SAMPLE = 598.9e9
# dominant kernel: Conv3D 64->64 3x3x3 on (16,16,288): 2*73728*64*1728 FLOP
BODY_CONV_FLOP_PER_SAMPLE = 2.0 * 16 * 16 * 288 * 64 * 27 * 64
# ... and its algorithmic HBM bytes per sample: read the un-padded input once +
# write the output once (bf16 mode keeps the 64-channel trunk in bf16)
BODY_CONV_ELEMS_PER_SAMPLE = 2.0 * 16 * 16 * 288 * 64
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0 / 3}
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md (spec; 6.3 TB/s achievable)
TRAIN_GFLOP = {'c2': 2860.0}      # 4 G + 9 D, SURVEY.md §8d


def _body_ops(ph):
    """indices of the 64 -> 64 convs on the full (16,16,288) grid"""
    return [i for i, op in enumerate(ph.plan.ops)
            if ph.op_is_mfma(i) and op['cout'] == 64
            and ph.plan.tensors[op['out']][1:4] == [16, 16, 288]]


# ------------------------------------------------------------- GPU side legs
def time_mode(spec, weights, dev, x, out, precision, steps, warmup=2):
    """samples/s and body-conv launch time of one arithmetic mode"""
    import torch
    from sup3r_amd.engine import Network
    net = Network(spec, name='generator', device=dev, precision=precision)
    net.set_weights(weights)
    ph = net.plan(tuple(x.shape), training=False)
    for _ in range(warmup):
        ph.forward(x, out=out)
    torch.cuda.synchronize()
    ph.profile_begin(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        ph.forward(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _, ms = ph.profile_end()
    body = _body_ops(ph)
    body_ms = float(np.mean([ms[i] for i in body]))
    kinds = sorted({ph.op_info(i)['fwd'] for i in body})
    del ph, net
    return dt, body_ms, kinds


def train_models(config, precision='bf16'):
    """(model, lr shape, hr shape, description, GFLOP/sample | None)"""
    from sup3r_amd import Sup3rGan
    if config == 'c2':
        m = Sup3rGan(CFG, os.path.join(CFGDIR, 'disc_st.json'),
                     loss='MeanAbsoluteError', precision=precision)
        return m, LR_SHAPE, HR_SHAPE, 'gen_5x_12x_2f + disc_st', 2860.0
    if config == 'c4':
        m = Sup3rGan(os.path.join(CFGDIR, 'gen_3x_4x_2f.json'),
                     os.path.join(CFGDIR, 'disc_st_same.json'),
                     loss='MeanAbsoluteError', precision=precision)
        return m, (16, 16, 24, 2), (48, 48, 96, 2), \
            'gen_3x_4x_2f + disc_st_same (C4 body)', None
    if config == 'c1':
        # tests/training/test_train_gan.py (S): batch 15, lr 5 x 5 -> hr 10 x 10
        m = Sup3rGan(os.path.join(CFGDIR, 'gen_2x_2f.json'),
                     os.path.join(CFGDIR, 'disc_s_same.json'),
                     loss='MeanAbsoluteError', precision=precision)
        return m, (5, 5, 2), (10, 10, 2), \
            'gen_2x_2f + disc_s_same (C1, the reference test shape)', None
    if config == 'c4toy':
        m = Sup3rGan(os.path.join(CFGDIR, 'gen_wind_3x_4x_2f_toy.json'),
                     os.path.join(CFGDIR, 'disc_st_same.json'),
                     loss='MeanAbsoluteError', precision=precision)
        m.set_model_params(hr_exo_features=['topography'])
        return m, (4, 4, 4, 2), (12, 12, 16, 3), \
            'sup3rcc/gen_wind_3x_4x_2f (filters: 1 toy, Sup3rConcat ' \
            'topography) + disc_st_same', None
    raise SystemExit(f'unknown --config {config}')


def spec_gmacs(spec, shape):
    """forward GMAC per SAMPLE of a layer spec at an input shape (convs: output
    positions before depth-to-space x taps x C_in x C_out; dense: in x out)"""
    from sup3r_amd import spec as S
    layers = spec if isinstance(spec, list) and spec and not isinstance(
        spec[0], dict) else S.parse_layers(spec)
    plan = S.build_plan(layers, tuple(shape))
    macs = 0
    for op in plan.ops:
        if 'cin' not in op:
            continue
        if 'k' in op:
            npos = int(np.prod(plan.tensors[op['out']][:4])) \
                // (op.get('d2s', 1) or 1) ** 2
            macs += npos * int(np.prod(op['k'])) * op['cin'] * op['cout']
        else:
            macs += int(plan.tensors[op['out']][0]) * op['cin'] * op['cout']
    return macs / shape[0] / 1e9


def condmom_leg(batch, lr_s, hr_s, min_seconds, max_steps):
    """BASELINE.json config 5: ``Sup3rCondMom(spatiotemporal/gen_3x_4x_2f)`` —
    one ``run_gradient_descent`` (generator forward, masked MSE
    (conditional.py:221-283), reverse pass, Adam step; the loss scalars read
    back every step like the reference's ``_train_epoch``,
    conditional.py:363-489) on a resident synthetic batch, mask of ones"""
    import torch
    from sup3r_amd import Sup3rCondMom
    from sup3r_amd.engine import Device
    dev = Device.get()
    cfg = os.path.join(CFGDIR, 'gen_3x_4x_2f.json')
    m = Sup3rCondMom(cfg, precision='bf16')
    rng = np.random.default_rng(0)
    lr = dev.to_device(rng.standard_normal((batch,) + lr_s).astype(np.float32))
    out = dev.to_device(rng.standard_normal((batch,) + hr_s).astype(np.float32))
    mask = dev.to_device(np.ones((batch,) + hr_s, np.float32))
    m.init_weights(lr.shape, out.shape)

    import types
    batch_ = types.SimpleNamespace(low_res=lr, output=out, mask=mask)

    def step():
        # (what _train_epoch runs per mini-batch; the loss scalars read back)
        return m._train_step(batch_).resolve()
    for _ in range(4):
        det = step()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        det = step()
        n += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if n >= max_steps or el >= min_seconds:
            break
    assert np.isfinite(float(det['loss_gen'])), det
    with open(cfg) as f:
        g = spec_gmacs(json.load(f), (batch,) + lr_s)
    dt = el / n
    res = {'workload': 'Sup3rCondMom.run_gradient_descent (forward, masked MSE, '
                       'reverse pass, Adam), gen_3x_4x_2f, lr '
                       f'{(batch,) + lr_s} -> hr {(batch,) + hr_s}, mask of '
                       'ones, bf16 MFMA operands',
           'ms_per_step': dt * 1e3, 'value': batch / dt, 'unit': 'samples/s',
           'steps': n, 'seconds': el,
           # forward + data gradient + weight gradient of every conv
           'algorithmic_gflop_per_sample': 6.0 * g,
           'tflops': 6.0 * g * batch / dt / 1e3,
           'loss_gen': float(det['loss_gen'])}
    rec = getattr(m, '_recorder', None)
    if rec is not None and rec.replays:
        res['recorded'] = {'replays': rec.replays}
    del m
    torch.cuda.empty_cache()
    return res


def train_leg(config, global_batch, world, rank, min_seconds, max_steps,
              multi_gpu, precision='bf16'):
    """ms per ``Sup3rGan._train_batch`` (generator step + discriminator step)
    on synthetic batches; with ``multi_gpu`` every rank computes its 1 / world
    shard of the global batch and the gradients are SUMMED over RCCL."""
    import torch
    from sup3r_amd.engine import Device
    model, lr_s, hr_s, what, gflop = train_models(config, precision)
    lr_shape, hr_shape = (global_batch,) + lr_s, (global_batch,) + hr_s
    rng = np.random.default_rng(0)        # every rank draws the SAME batch
    dev = Device.get()

    class Batch:                          # resident in HBM, like the headline
        low_res = dev.to_device(
            rng.standard_normal(lr_shape).astype(np.float32))
        high_res = dev.to_device(
            rng.standard_normal(hr_shape).astype(np.float32))
    per = global_batch // world if multi_gpu else global_batch
    model.init_weights((per,) + lr_s, (per,) + hr_s)
    if multi_gpu and world > 1:
        model._join_replicas()
        model._sync_replicas()

    def step():
        return model._train_batch(Batch, True, False, False, True, False,
                                  False, 1e-3, multi_gpu=multi_gpu)
    for _ in range(4):        # (a launch-bound step is recorded on its third call)
        step()
    torch.cuda.synchronize()
    c0 = {k: dev.stat(k) for k in ('buckets', 'allreduces', 'bucket_elems')}
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # (every rank takes the same number of steps: decided by rank 0)
        stop = n >= max_steps or el >= min_seconds
        if multi_gpu and world > 1:
            import torch.distributed as dist
            flag = torch.tensor([int(stop)], device='cuda')
            dist.broadcast(flag, src=0)
            stop = bool(flag.item())
        if stop:
            break
    dt = el / n
    out = {'workload': f'Sup3rGan._train_batch (gen step + disc step), {what}, '
                       f'global batch {global_batch}, lr {lr_shape} -> hr '
                       f'{hr_shape}, ' + {
                           'bf16': 'bf16 MFMA operands',
                           'bf16x3': 'split-bf16 (hi*hi + hi*lo + lo*hi) MFMA '
                                     'operands, fp32 activations',
                           'f32': 'exact fp32'}[precision] +
                       ', MeanAbsoluteError'
                       + (f', batch split over {world} GPUs, RCCL gradient '
                          'SUM' if multi_gpu and world > 1 else ''),
           'ms_per_step': dt * 1e3, 'value': global_batch / dt,
           'unit': 'samples/s', 'steps': n, 'seconds': el}
    rec = getattr(model, '_recorder', None)
    if rec is not None and rec.replays:
        # launch-bound step: recorded once, one hipGraphLaunch per mini-batch
        out['recorded'] = {'replays': rec.replays, 'graph_nodes': max(
            e['rec'].nodes for e in rec._entries.values() if e['rec'])}
    if not gflop and getattr(model, '_disc', None) is not None:
        try:
            # 4 G + 9 D: both steps of _train_batch
            gg = spec_gmacs(model._gen.layers, lr_shape)
            dd = spec_gmacs(model._disc.layers, hr_shape)
            gflop = 2.0 * (4 * gg + 9 * dd)
        except Exception:
            gflop = None
    if gflop:
        out['algorithmic_gflop_per_sample'] = gflop
        out['tflops'] = gflop * global_batch / dt / 1e3
    # what RCCL itself reports + the collectives this rank issued per step: a
    # first run on a multi-GPU node verifies itself (world ranks, > 0 buckets)
    import ctypes as _C
    from sup3r_amd import _lib as _L
    nr, rk = _C.c_int32(0), _C.c_int32(0)
    _L.check(_L.lib().s3_comm_info(dev.ctx, _C.byref(nr), _C.byref(rk)),
             dev.ctx, 's3_comm_info')
    out['comm'] = {
        'rccl_comm_count': int(nr.value), 'rccl_user_rank': int(rk.value),
        'world': world,
        'bucket_allreduces_per_step': (dev.stat('buckets') - c0['buckets']) / n,
        'other_allreduces_per_step':
            (dev.stat('allreduces') - c0['allreduces']) / n,
        'gradient_mb_reduced_per_step':
            (dev.stat('bucket_elems') - c0['bucket_elems']) * 4 / n / 1e6}
    del model
    torch.cuda.empty_cache()
    return out


def c3_leg(batch, steps, warmup, world, rank, entry='strategy'):
    """chunks/s over this rank's share of the C3 chunk list (domain
    400x400x720, chunks 20x20x48 + halo 1 / 2), cropped hi-res chunks
    delivered to the host (checksummed, not stored: the whole output is
    276 GB).  ``entry='strategy'``: the reference's entry — ``ForwardPassChunk``
    structures from a strategy's ``init_chunk`` through
    ``ForwardPass.get_input_chunk`` / ``iter_chunks`` (what ``ForwardPass.run
    (strategy, node_index)`` executes; node = rank).  ``entry='domain'``: the
    in-memory ``run_batched`` over a resident domain."""
    import torch
    from sup3r_amd import ChunkSlicer, ForwardPass, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_100m', 'v_100m', 'temperature_100m', 'pressure_0m']
    m = Sup3rGan(CFG, os.path.join(CFGDIR, 'test_disc_st_same.json'),
                 precision='bf16')
    m.set_model_params(lr_features=feats, hr_out_features=feats[:2],
                       s_enhance=5, t_enhance=12)
    Sup3rGan.seed(0)
    m.init_weights((1, 22, 22, 52, 4), (1, 110, 110, 624, 2))
    rng = np.random.default_rng(7)
    # (an 80-row random block repeated along s0: drawing all 461 M values
    # took longer than the bounded run that visits a few rows of them)
    domain = np.tile(rng.standard_normal((80, 400, 720, 4), dtype=np.float32),
                     (5, 1, 1, 1))
    seen = [0, 0.0]
    extra = {}

    def writer(idx, hr_slice, data):
        seen[0] += 1
        seen[1] += float(data[::17, ::17, ::17].sum())
    if entry == 'strategy':
        register_model('Sup3rGan', {'model_dir': 'bench-c3'}, m)
        st = ArrayStrategy(domain, {'model_dir': 'bench-c3'}, (20, 20, 48),
                           spatial_pad=1, temporal_pad=2, max_nodes=world,
                           model=m)
        assert st.n_chunks == 6000
        fwp = ForwardPass(st, rank)
        mine = [int(i) for i in st.node_chunks[rank]]

        def run(ids):
            n = 0
            for chunk, failed, data in ForwardPass.iter_chunks(
                    (fwp.get_input_chunk(i) for i in ids), m,
                    allowed_const=st.allowed_const, batch=batch):
                assert not failed
                writer(chunk.index, None, data)
                n += 1
            return n
        run(mine[:warmup * batch])
        torch.cuda.synchronize()
        seen[0] = 0
        # HIP events around every op of the timed batches, in situ (beside the
        # SDMA delivery of the previous batch)
        ph = m._gen.plan((batch, 22, 22, 52, 4), training=False)
        if not os.environ.get('BENCH_C3_NO_INSITU'):
            ph.profile_begin(steps)
        t0 = time.perf_counter()
        n = run(mine[warmup * batch:(warmup + steps) * batch])
        torch.cuda.synchronize()
        el0 = time.perf_counter() - t0
        n_prof, ms = (0, []) if os.environ.get('BENCH_C3_NO_INSITU') else ph.profile_end()
        assert n == seen[0] == steps * batch, (n, seen[0], steps * batch)
        trunk = [ms[i] for i, op in enumerate(ph.plan.ops)
                 if n_prof and op.get('cin') == 64 and op.get('cout') == 64
                 and ph.plan.tensors[op['out']][1:4] == [22, 22, 624]]
        if trunk and n_prof > 0:
            t_ms = float(np.mean(trunk))
            gflop = batch * 22 * 22 * 624 * 27 * 64 * 64 * 2 / 1e9
            extra = {
                'forwards_profiled': n_prof,
                'all_ops_ms_per_batch': float(sum(ms)),
                'head_conv_4_64_us': ms[0] * 1e3,
                'trunk_conv_launches': len(trunk),
                'trunk_conv_avg_us': t_ms * 1e3,
                'trunk_conv_useful_tflops': gflop / t_ms,
                'trunk_conv_frac_of_peak': gflop / t_ms / PEAK_TFLOPS['bf16'],
                'note': 'useful = the 22 x 22 x 624 positions of a chunk (11 '
                        'half rows; columns 8 + 8 + a strip of 6-column tiles '
                        'in a second launch: no edge tile)'}
        return n, el0, extra
    else:
        slicer = ChunkSlicer((400, 400), 720, 5, 12, (20, 20, 48),
                             spatial_pad=1, temporal_pad=2)
        assert slicer.n_chunks == 6000
        fwp = ForwardPass(m, slicer, rank=rank, nranks=world, shard='block')
        resident = fwp.upload_domain(domain)
        fwp.run_batched(resident, writer=writer, batch=batch,
                        max_chunks=warmup * batch)
        torch.cuda.synchronize()
        seen[0] = 0
        t0 = time.perf_counter()
        n = fwp.run_batched(resident, writer=writer, batch=batch,
                            max_chunks=steps * batch)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert n == seen[0] == steps * batch, (n, seen[0], steps * batch)
    return n, el, extra


# ---------------------------------------------------- power / clock evidence
def fwd2d_leg(dev, steps=20, warmup=3, batch=48, seed=42):
    """the production shape of the reference's SPATIAL step
    (examples/sup3rwind/run_configs/wind/config_fwp_spatial.json: chunks of
    75 x 75 x 38 + temporal_pad 5; a 2-D model sees the 48 time steps as its
    batch axis, /root/reference/sup3r/pipeline/forward_pass.py:274-337) through
    the reference's own spatial/gen_2x_2f.json: 33 x Conv2DTranspose 64 -> 64 +
    64 -> 256 depth-to-space 2 on the weights-stationary Conv2D kernel
    (kernels_conv2d_ws.hip), head / output conv on the logical-axes tile kernel."""
    from sup3r_amd import spec as S
    from sup3r_amd.engine import Network
    with open(os.path.join(CFGDIR, 'sup3r', 'spatial', 'gen_2x_2f.json')) as f:
        spec = json.load(f)
    shape = (batch, 75, 75, 2)
    net = Network(spec, name='generator2d', device=dev, precision='bf16')
    net.build(shape, seed=0)
    ph = net.plan(shape, training=False)
    x = dev.to_device(np.random.default_rng(seed).standard_normal(
        shape).astype(np.float32))
    out = dev.empty(tuple(ph.out_shape))
    for _ in range(warmup):
        ph.forward(x, out=out)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ph.forward(x, out=out)
    dev.sync()
    el = time.perf_counter() - t0
    # per-op HIP-event times in a pass of their own: 36 event records are 10 %
    # of a 1.35 ms forward (they are noise in the 15.8 ms headline)
    ph.profile_begin(steps)
    for _ in range(steps):
        ph.forward(x, out=out)
    dev.sync()
    _, ms = ph.profile_end()
    sel = [ph.op_info(i)['fwd'] if op['kind'] == S.OP_CONV else None
           for i, op in enumerate(ph.plan.ops)]
    trunk = [i for i, op in enumerate(ph.plan.ops)
             if sel[i] == 'conv2d_ws' and op['cout'] == 64]
    npos = batch * 75 * 75
    flop_conv = 2.0 * npos * 9 * 64 * 64
    macs = sum(int(np.prod(ph.plan.tensors[op['out']][:4]))
               // (op.get('d2s', 1) or 1) ** 2 * op['cin'] * op['cout'] * 9
               for op in ph.plan.ops if op['kind'] == S.OP_CONV)
    t_ms = float(np.mean([ms[i] for i in trunk])) if trunk else float('nan')
    bytes_conv = npos * 64 * 2 * 2          # bf16 cells in + out
    res = {
        'value': batch * steps / el, 'unit': 'samples/s (time steps)',
        'ms_per_step': el / steps * 1e3, 'steps': steps, 'warmup': warmup,
        'workload': f'sup3r/configs/spatial/gen_2x_2f.json forward, lo-res '
                    f'({batch},75,75,2) -> hi-res ({batch},150,150,2), the '
                    'config_fwp_spatial.json chunk (75 x 75 x 38 + temporal_pad '
                    '5), bf16 trunk / fp32 I/O, inputs resident in HBM',
        'gflop_per_sample': 2.0 * macs / batch / 1e9,
        'whole_path_tflops': 2.0 * macs * steps / el / 1e12,
        'kernels': {k: sel.count(k) for k in sorted(set(k for k in sel if k))},
        'roofline': {
            'kernel': 'conv2d_ws_pp_kernel (Conv2D / Conv2DTranspose 64->64 3x3, '
                      'reflect pad fused, weights-stationary persistent, two '
                      'half-workgroups half a period apart)',
            # 288 FLOP per algorithmic byte: the ridge of the bf16 MFMA / HBM
            # rooflines.  Ablations: in steady state the phase
            # that holds all the memory instructions (M) runs at 6.1 TB/s when
            # alone — the copy rate of the chip — and the kernel at 4.5 TB/s of
            # read + write traffic
            'bound': 'hbm', 'unit': 'GB/s',
            'achieved': bytes_conv / (t_ms * 1e-3) / 1e9,
            'peak': PEAK_HBM_GBS,
            'frac': bytes_conv / (t_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            'traffic': None,
            'launches_per_step': len(trunk), 'avg_launch_ms': t_ms,
            'algorithmic_bytes_per_launch': bytes_conv,
            'mfma_tflops': flop_conv / (t_ms * 1e-3) / 1e12,
            'mfma_frac': flop_conv / (t_ms * 1e-3) / 1e12 / PEAK_TFLOPS['bf16'],
            'note': '1200 single-image tiles of 16 x 16 positions on 256 CUs '
                    '(4.7 per workgroup: three T/M periods + the prologue with '
                    'the 72 KB filter image) at this shape, 0.44 of 8 TB/s at '
                    '480 x 75 x 75; counter traffic per launch: '
                    'profiles/r05/pmc_fwd2d.txt'}}
    # what the checker needs (fwd2d_parity, run after EVERY timed leg of the line, next
    # to the CPU baseline: nothing under oracle/ is imported before that)
    pick = [0, batch - 1]
    res['_parity_inputs'] = (spec, [np.array(w) for w in net.weights],
                             x.cpu().numpy()[pick], out.cpu().numpy()[pick], pick)
    del ph, net
    return res


def fwd2d_parity(res):
    """checker use of the oracle: images 0 and N - 1 of the fwd2d leg's timed batch
    through the CPU restatement with the same weights — the number the headline
    prints as cpu_baseline.parity, here for the 2-D path in its multi-tile steady
    state (per op: tests/test_ws_multitile.py)"""
    inputs = res.pop('_parity_inputs', None)
    if inputs is None:
        return
    try:
        from oracle.network import Network as OracleNet
        spec, weights, x_np, y_dev, pick = inputs
        ref = OracleNet(spec)
        ref.init_weights(x_np[:1, :8, :8], seed=0)
        ref.set_weights(weights)
        y_ref = ref.forward(x_np)
        res['parity'] = {
            'bf16_linf': float(np.abs(y_dev - y_ref).max()),
            'scale': float(np.abs(y_ref).max()),
            'sample': f'images {pick[0]} and {pick[1]} of the timed batch vs the fp32 '
                      'numpy oracle (same weights); stated bound of the bf16 '
                      'mode: 3e-2 of the scale'}
    except Exception as e:                   # evidence, never fatal
        res['parity'] = {'error': repr(e)[:200]}


def fwp2d_executor_leg(rank=0, batch=4, reps=3):
    """the same spec through the reference's executor entry
    (ForwardPassStrategy-shaped chunks -> ForwardPass.get_input_chunk ->
    iter_chunks): a (150, 150, 760) lo-res domain in 75 x 75 x 38 chunks with
    temporal_pad 5, normalisation statistics set, cropped (150, 150, 38, 2)
    hi-res chunks delivered to the host — 2-D models run their chunks' time
    steps as the batch (forward_pass.py:274-337)"""
    from sup3r_amd import ForwardPass, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_10m', 'v_10m']
    Sup3rGan.seed(3)
    m = Sup3rGan(os.path.join(CFGDIR, 'sup3r', 'spatial', 'gen_2x_2f.json'),
                 os.path.join(CFGDIR, 'disc_s_same.json'),
                 means={f: np.float32(0.1 * (i + 1)) for i, f in enumerate(feats)},
                 stdevs={f: np.float32(1.5 + i) for i, f in enumerate(feats)},
                 precision='bf16')
    m.set_model_params(lr_features=feats, hr_out_features=feats, s_enhance=2,
                       t_enhance=1)
    m.init_weights((1, 16, 16, 2), (1, 32, 32, 2))
    # (80 chunks = 20 launch sequences per pass: with the 20-chunk domain of round
    # 5 a pass was 5 sequences and a quarter of its time was the fill and drain of
    # the three-deep pipeline — 3.6 ms per pass in the rocprofv3 timeline,
    # profiles/r06/README.md — not the executor's steady state)
    domain = np.random.default_rng(7 + rank).standard_normal(
        (150, 150, 760, 2), dtype=np.float32)
    register_model('Sup3rGan', {'model_dir': 'bench-fwp2d'}, m)
    st = ArrayStrategy(domain, {'model_dir': 'bench-fwp2d'}, (75, 75, 38),
                       spatial_pad=0, temporal_pad=5, max_nodes=1, model=m)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]
    best = None
    for _ in range(reps + 1):          # (first pass: plans, pinned rings)
        t0 = time.perf_counter()
        n = 0
        for c, failed, d in ForwardPass.iter_chunks(
                (fwp.get_input_chunk(i) for i in ids), m, batch=batch):
            assert not failed and d.shape == (150, 150, 38, 2)
            n += 1
        el = time.perf_counter() - t0
        best = el if best is None or el < best else best
    return {'value': n / best, 'unit': 'chunks/s', 'chunks': n,
            'chunks_per_launch_sequence': batch,
            'px_per_sec': n / best * 150 * 150 * 38,
            'workload': 'config_fwp_spatial.json shape: 75 x 75 x 38 chunks + '
                        'temporal_pad 5 of a (150, 150, 760, 2) domain through '
                        'ForwardPass.get_input_chunk -> iter_chunks, cropped '
                        '(150, 150, 38, 2) fp32 chunks delivered to the host'}


def fwp2d_chain_leg(rank=0, batch=2, reps=2):
    """the arrangement of examples/sup3rwind/run_configs/wind/
    config_fwp_spatial.json: ``model_class: MultiStepGan`` of two spatial
    steps with topography (here gen_2x_2f then the gen_wind_5x_1x_6f body with
    lo-res topography at its input and hi-res topography through its
    mid-network Sup3rConcat: 10x), 75 x 75 x 38 chunks + temporal_pad 5,
    through ForwardPass.get_input_chunk -> iter_chunks.  Both steps, the
    hand-over between them (s3_step_handover) and the exo fields stay on the
    device; ``host_chain`` = the same chunks through MultiStepGan.generate
    (every hand-over through host numpy, the reference's arrangement)"""
    import json

    from sup3r_amd import ForwardPass, MultiStepGan, Sup3rGan
    from sup3r_amd.forward_pass import register_model
    from sup3r_amd.strategy import ArrayStrategy
    feats = ['u_10m', 'v_10m']
    Sup3rGan.seed(5)
    means = {f: np.float32(0.1 * (i + 1)) for i, f in enumerate(feats)}
    stds = {f: np.float32(1.5 + i) for i, f in enumerate(feats)}
    means['topography'], stds['topography'] = np.float32(300), np.float32(150)
    m1 = Sup3rGan(os.path.join(CFGDIR, 'sup3r', 'spatial', 'gen_2x_2f.json'),
                  os.path.join(CFGDIR, 'disc_s_same.json'), means=means,
                  stdevs=stds, precision='bf16')
    m1.set_model_params(lr_features=feats, hr_out_features=feats, s_enhance=2,
                        t_enhance=1)
    m1.init_weights((1, 16, 16, 2), (1, 32, 32, 2))
    with open(os.path.join(CFGDIR, 'sup3r', 'sup3rcc',
                           'gen_wind_5x_1x_6f.json')) as f:
        spec2 = json.load(f)
    for layer in spec2['hidden_layers']:
        if layer.get('filters') == 6:
            layer['filters'] = 2
    m2 = Sup3rGan(spec2, os.path.join(CFGDIR, 'disc_s_same.json'), means=means,
                  stdevs=stds, precision='bf16')
    m2.set_model_params(lr_features=feats + ['topography'],
                        hr_out_features=feats, hr_exo_features=['topography'],
                        s_enhance=5, t_enhance=1)
    m2.init_weights((1, 16, 16, 3), (1, 80, 80, 3))
    ms = MultiStepGan([m1, m2])
    rng = np.random.default_rng(11 + rank)
    domain = rng.standard_normal((150, 150, 76, 2)).astype(np.float32)
    topo_hr = (300 + 150 * rng.standard_normal((1500, 1500, 1))).astype(
        np.float32)
    topo_mid = topo_hr.reshape(300, 5, 300, 5, 1).mean(axis=(1, 3)).astype(
        np.float32)
    exo = {'topography': {'steps': [
        {'model': 1, 'combine_type': 'input', 'data': topo_mid,
         's_enhance': 2, 't_enhance': 1},
        {'model': 1, 'combine_type': 'layer', 'data': topo_hr,
         's_enhance': 10, 't_enhance': 1}]}}
    kw = {'model_dirs': ['bench-chain-a', 'bench-chain-b']}
    register_model('MultiStepGan', kw, ms)
    st = ArrayStrategy(domain, kw, (75, 75, 38), spatial_pad=0,
                       temporal_pad=5, exo_data=exo,
                       model_class='MultiStepGan', max_nodes=1, model=ms)
    fwp = ForwardPass(st, 0)
    ids = [int(i) for i in st.node_chunks[0]]

    def run(n_rep, **options):
        best = None
        for _ in range(n_rep):
            t0 = time.perf_counter()
            n = 0
            for c, failed, d in ForwardPass.iter_chunks(
                    (fwp.get_input_chunk(i) for i in ids), ms, batch=batch,
                    options=options):
                assert not failed and d.shape == (750, 750, 38, 2)
                n += 1
            el = time.perf_counter() - t0
            best = el if best is None or el < best else best
        return n, best
    assert ForwardPass._device_path(ms, fwp.get_input_chunk(ids[0]))
    n, best = run(reps + 1)          # (first pass: plans, pinned rings)
    _, host = run(1, device_chains=False)
    ForwardPass.release_delivery_buffers()
    return {'value': n / best, 'unit': 'chunks/s', 'chunks': n,
            'chunks_per_launch_sequence': batch,
            'px_per_sec': n / best * 750 * 750 * 38,
            'host_chain': {'value': n / host, 'unit': 'chunks/s',
                           'what': 'MultiStepGan.generate chunk by chunk: '
                                   'hand-overs and exo fields through host '
                                   'numpy'},
            'workload': 'MultiStepGan [spatial/gen_2x_2f, sup3rcc/'
                        'gen_wind_5x_1x_6f body (2 features) + topography]: '
                        '75 x 75 x 38 chunks + temporal_pad 5 of a (150, 150, '
                        '76, 2) domain -> (750, 750, 38, 2) fp32 chunks '
                        '(171 MB each) delivered to the host'}


def _smi_poll(stop, out):
    import re as _re
    while not stop.is_set():
        try:
            t = subprocess.run(['rocm-smi', '--showclocks', '--showpower'],
                               capture_output=True, text=True, timeout=5).stdout
            m = _re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t)
            p_ = _re.search(r'Power \(W\): ([\d.]+)', t)
            if m and p_:
                out.append((int(m.group(1)), float(p_.group(1))))
        except Exception:
            return
        time.sleep(0.1)


def power_leg(spec, weights, dev, x, out, seconds=2.0):
    """The same forward on the benchmark's random operands and on all-zero
    operands (zero weights, zero input: the identical instruction stream with
    no toggling in the MFMA datapath), rocm-smi polled alongside.  A kernel
    bound by its schedule takes the same time on both; one bound by the
    board's power cap speeds up with the shader clock."""
    import threading

    import torch
    from sup3r_amd.engine import Network
    res = {}
    for what in ('random', 'zeros'):
        net = Network(spec, name='generator', device=dev, precision='bf16')
        net.set_weights(weights if what == 'random'
                        else [np.zeros_like(w) for w in weights])
        ph = net.plan(tuple(x.shape), training=False)
        xd = x if what == 'random' else torch.zeros_like(x)
        for _ in range(3):
            ph.forward(xd, out=out)
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=_smi_poll, args=(stop, samples))
        th.start()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                ph.forward(xd, out=out)
            torch.cuda.synchronize()
            n += 10
        el = time.perf_counter() - t0
        stop.set()
        th.join()
        busy = [s_ for s_ in samples if s_[1] > 600] or samples
        res[what] = {'samples_per_s': int(x.shape[0]) * n / el,
                     'sclk_MHz': (float(np.mean([s_[0] for s_ in busy]))
                                  if busy else None),
                     'power_W': (float(np.mean([s_[1] for s_ in busy]))
                                 if busy else None)}
        del ph
        net.clear_plans()
    res['zeros_over_random'] = (res['zeros']['samples_per_s'] /
                                res['random']['samples_per_s'])
    res['note'] = ('same kernels, same launches; only the operand values '
                   'differ.  zeros_over_random > 1 with a higher sclk and a '
                   'lower power = the random-operand run sits at the power '
                   'cap (1400 W), not at a scheduling limit: frac x '
                   'zeros_over_random is what the schedule delivers at the '
                   'unthrottled clock')
    return res


# ------------------------------------------------------- HBM traffic (PMC)
def inner_pmc(args):
    """child process under ``rocprofv3 --pmc``: a few forwards, nothing else"""
    import torch
    from sup3r_amd.engine import Device, Network
    with open(CFG) as f:
        spec = json.load(f)
    dev = Device.get(0)
    net = Network(spec, name='generator', device=dev, precision='bf16')
    shape = (args.batch,) + LR_SHAPE
    net.build(shape, seed=0)
    ph = net.plan(shape, training=False)
    x = dev.to_device(np.random.default_rng(42).standard_normal(shape)
                      .astype(np.float32))
    out = dev.empty((args.batch,) + HR_SHAPE)
    for _ in range(3):
        ph.forward(x, out=out)
    torch.cuda.synchronize()


def hbm_class_ops(ph, ms):
    """the HBM-bound ops of the forward (the C_in <= 8 convs
    and the index ops): algorithmic bytes = read the input once + write the
    output once in their storage dtypes, over the op's HIP-event time"""
    out = []
    for i, op in enumerate(ph.plan.ops):
        info = ph.op_info(i)
        conv = 'cout' in op
        if conv and info['fwd'] in ('mfma_tile', 'mfma_persist'):
            continue
        if not conv and info.get('in_rep'):
            # a temporal repeat read through its consumers' halo index: no
            # launch, no bytes (plan.cpp, the repeat-fusion pass)
            continue
        nb = 0
        for tid in (op['in0'], op['out']):
            n = int(np.prod(ph.plan.tensors[tid]))
            nb += n * (2 if ph.tensor_is_bf16(tid) else 4)
        if ms[i] <= 0:
            continue
        gbs = nb / (ms[i] * 1e-3) / 1e9
        out.append({'op': i, 'what': (f'conv {op["cin"]}->{op["cout"]} '
                                      f'({info["fwd"]})' if conv
                                      else f'index op kind {op["kind"]}'),
                    'out_shape': ph.plan.tensors[op['out']],
                    'algorithmic_bytes': nb, 'ms': ms[i],
                    'achieved_GBps': gbs, 'frac_of_8TBps': gbs / PEAK_HBM_GBS})
    return out


def measure_traffic(batch, timeout=240):
    """HBM bytes per launch of the dominant kernel from the PMC counters, as
    MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    ``rocprofv3 --pmc`` passes of this script's forward (no tracing domain
    combined with --pmc), both in KB, FETCH_SIZE doubled (gfx950 tallies the
    128-B requests of a wide coalesced stream at 64 B).  Only the 33 body-conv
    dispatches of each forward are averaged (positions 2..34 of the 38
    conv3_mfma_persist_kernel<4, *> dispatches per forward at this batch: two head
    convs first, three 64-channel passes of the 64 -> 200 conv last)."""
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return None, 'rocprofv3 not found'
    vals = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='s3pmc_')
        cmd = [rocprof, '--pmc', counter, '--output-format', 'csv', '-d', d,
               '--', sys.executable, os.path.abspath(__file__), '--inner-pmc',
               '--batch', str(batch)]
        try:
            subprocess.run(cmd, cwd='/tmp', timeout=timeout, check=True,
                           capture_output=True,
                           env=dict(os.environ, TMPDIR='/tmp'))
        except Exception as e:
            shutil.rmtree(d, ignore_errors=True)
            return None, f'rocprofv3 --pmc {counter} failed: {e!r}'[:300]
        rows, tail = [], []
        for fp in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            with open(fp) as f:
                for r in csv.DictReader(f):
                    if r.get('Counter_Name') != counter:
                        continue
                    name = r.get('Kernel_Name', '')
                    if 'conv3_mfma_persist_kernel<4' in name:
                        rows.append(r)
                    elif 'conv_tail_' in name:
                        tail.append(float(r['Counter_Value']))
        shutil.rmtree(d, ignore_errors=True)
        if not rows or len(rows) % 38:
            return None, (f'{len(rows)} dispatches of the persistent kernel '
                          f'in the {counter} pass (expected a multiple of 38)')
        rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
        body = [float(r['Counter_Value']) for i, r in enumerate(rows)
                if 2 <= i % 38 <= 34]
        vals[counter] = float(np.mean(body))
        if tail:
            vals['tail_' + counter] = float(np.mean(tail))
    total = (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0
    if 'tail_FETCH_SIZE' in vals and 'tail_WRITE_SIZE' in vals:
        measure_traffic.tail = (2.0 * vals['tail_FETCH_SIZE']
                                + vals['tail_WRITE_SIZE']) * 1024.0
    return total, (f'PMC, this run: FETCH_SIZE {vals["FETCH_SIZE"]:.0f} KB x 2 '
                   f'(gfx950 correction) + WRITE_SIZE {vals["WRITE_SIZE"]:.0f} '
                   'KB per body-conv launch, separate rocprofv3 --pmc passes')


# ------------------------------------------------------------------ CPU legs
def cpu_legs(spec, dev, seconds_budget=30.0):
    """CPU baselines on the host cores + the parity of every device mode, one
    C2 chunk (1,16,16,24,4) -> (1,80,80,288,2), the op sequence AS TF EXECUTES
    IT (un-fused REFLECT pad-3 / valid conv / crop-2, NDHWC fp32, 474 GMAC):

    * the numpy oracle (BLAS GEMM per tap) — here the CHECKER: every device
      mode's output on this chunk is compared with it (``parity``);
    * baseline (i), "TF-CPU proxy": torch-CPU / oneDNN, the x86 conv backend
      TF 2.15 uses (oracle/torch_proxy.py);
    * baseline (ii): the C + OpenMP restatement of the conv (oracle/conv_ref.c)
      under the same numpy layer loop, compiled for this host.
    TensorFlow is not installable here, so none of these is a TF measurement.
    ``value`` is the faster of (i) and (ii)."""
    import torch
    from oracle import c_ref
    from oracle.network import Network as OracleNet
    from oracle.torch_proxy import torch_generator_forward
    from sup3r_amd.engine import Network
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1,) + LR_SHAPE).astype(np.float32)
    net = OracleNet(spec)
    net.init_weights(x[:, :6, :6, :6], seed=0)     # lazy build, tiny input
    t0 = time.time()
    y_np = net.forward(x)
    t_numpy = time.time() - t0
    # ---- parity of the device modes against the oracle (checker use)
    parity = {'sample': 'one C2 chunk, oracle weights (glorot seed 0)',
              'scale': float(np.abs(y_np).max())}
    xd = dev.to_device(np.repeat(x, 8, axis=0))
    for prec in ('bf16', 'bf16x3', 'f32'):
        hnet = Network(spec, name='generator', device=dev, precision=prec)
        hnet.set_weights(net.weights)
        nb = 8 if prec == 'bf16' else 1       # 8: the persistent kernel
        y = hnet.plan((nb,) + LR_SHAPE, training=False).forward(
            xd[:nb].contiguous()).cpu().numpy()
        parity[f'{prec}_linf'] = float(np.abs(y[0] - y_np[0]).max())
        del hnet
    # ---- (ii) C + OpenMP
    lib, how = c_ref.load(native=True)
    threads_c = int(lib.s3ref_threads())
    # BASELINE.md §3 / SURVEY.md §8d protocol: 3 warm-up + 5 timed, median
    for _ in range(3):
        y_c, _, _ = c_ref.forward(net, x, lib=lib)
    assert np.abs(y_c - y_np).max() < 1e-3
    times_c = []
    while len(times_c) < 5:
        _, dt, _ = c_ref.forward(net, x, lib=lib)
        times_c.append(dt)
    # ---- (i) torch / oneDNN
    threads_t = torch.get_num_threads()
    for _ in range(3):
        y_t, _ = torch_generator_forward(net, x)
    assert np.abs(y_t - y_np).max() < 1e-3
    times_t = []
    while len(times_t) < 5:
        _, dt = torch_generator_forward(net, x)
        times_t.append(dt)
    med_c, med_t = float(np.median(times_c)), float(np.median(times_t))
    best = min(med_c, med_t)
    return {
        'value': 1.0 / best, 'unit': 'samples/s',
        'cores': threads_c if med_c <= med_t else int(threads_t),
        'kind': 'port',
        'sample': 'one C2 chunk (1,16,16,24,4)->(1,80,80,288,2), as-TF-'
                  'executes op sequence (474 GMAC/sample); 3 warm-up + median '
                  f'of {len(times_c)} (C/OpenMP, {how}, {threads_c} threads: '
                  f'{med_c:.2f} s/sample) and of {len(times_t)} (torch-CPU/'
                  f'oneDNN "TF-CPU proxy", {threads_t} threads: {med_t:.2f} '
                  f's/sample); numpy oracle: {t_numpy:.1f} s/sample',
        'c_openmp_s_per_sample': med_c, 'torch_onednn_s_per_sample': med_t,
        'numpy_oracle_s_per_sample': t_numpy,
    }, parity


def child_leg(leg_args, world, k, limit):
    """One more bench mode on the same GPUs in a fresh group of processes
    (every rank spawns its own child with its RANK / LOCAL_RANK and a
    rendezvous port of its own).  Returns rank 0's JSON line, or what went
    wrong — a crash or a hang in there never reaches the caller."""
    env = dict(os.environ)
    env['MASTER_PORT'] = str(int(env.get('MASTER_PORT', '29500')) + 101
                             + 7 * k)
    for var in ('TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT',
                'TORCHELASTIC_MAX_RESTARTS', 'TORCHELASTIC_USE_AGENT_STORE'):
        env.pop(var, None)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world)] \
        + list(leg_args)
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True,
                           timeout=limit)
    except subprocess.TimeoutExpired:
        return {'error': f'timed out after {limit} s', 'cmd': leg_args}
    except Exception as e:
        return {'error': repr(e)[:300], 'cmd': leg_args}
    for line in reversed(p.stdout.strip().splitlines()):
        if line.startswith('{'):
            try:
                return json.loads(line)
            except Exception:
                break
    if int(env.get('RANK', '0')) != 0 and p.returncode == 0:
        return {}
    return {'error': f'exit code {p.returncode}',
            'stderr_tail': p.stderr[-600:], 'cmd': leg_args}


# ----------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default='infer',
                    choices=['infer', 'train', 'c3', 'c1', 'fwd2d'])
    ap.add_argument('--config', default='c2', choices=['c2', 'c4', 'c4toy', 'c1', 'c5', 'c5small'],
                    help='--mode train: which GAN')
    ap.add_argument('--batch', type=int, default=None,
                    help='infer: lo-res chunks per GPU per step (default 32; '
                         'throughput vs batch on one MI355X: 4: 1600, 8: 1890, '
                         '16: 1910, 32: 1970, 64: 2000 samples/s); train: '
                         'GLOBAL batch (default 8 x GPUs for c2, 32 for c4); '
                         'c3: chunks per launch sequence (default 8; one MI355X: 2: '
                         '283, 4: 327, 8: 358 chunks/s)')
    ap.add_argument('--precision', default='bf16',
                    choices=['bf16', 'f32', 'bf16x3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-mode', action='store_true',
                    help='skip the BF16X3 / fp32 parity-mode measurements')
    ap.add_argument('--no-train', action='store_true',
                    help='skip the extra training-step measurement')
    ap.add_argument('--no-traffic', action='store_true',
                    help='skip the rocprofv3 --pmc passes (roofline.traffic '
                         'is then null)')
    ap.add_argument('--no-executor', action='store_true',
                    help='fwd2d: skip the ForwardPass executor / chain legs '
                         '(profiling runs: every dispatch is the timed forward)')
    ap.add_argument('--c3-entry', default='strategy',
                    choices=['strategy', 'domain'],
                    help="c3: 'strategy' = ForwardPassChunk structures through "
                         "the reference's entry (get_input_chunk / "
                         "iter_chunks); 'domain' = run_batched over a "
                         'resident in-memory domain')
    ap.add_argument('--train-seconds', type=float, default=5.0)
    ap.add_argument('--dump-ops', default=None,
                    help='write per-op mean ms of the timed region here')
    ap.add_argument('--inner-pmc', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--force-legs', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.inner_pmc:
        args.batch = args.batch or 32
        return inner_pmc(args)

    import torch
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda',
                                                               local_rank))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        tt = torch.tensor([seconds], device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    base = {'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'higher_is_better': True, 'vs_baseline': None, 'data': 'synthetic'}
    if world > 1:
        # self-verification of a multi-GPU line: the process group's size and
        # the distinct devices its ranks sit on (an all-gather over RCCL)
        ids = [None] * world
        dist.all_gather_object(ids, (rank, local_rank,
                                     torch.cuda.get_device_properties(
                                         local_rank).name))
        base['ranks'] = {'dist_world_size': dist.get_world_size(),
                         'backend': dist.get_backend(),
                         'devices': sorted({(r, lr) for r, lr, _ in ids}),
                         'gpu': ids[0][2]}

    if args.mode == 'train':
        gb = args.batch or (8 * world if args.config == 'c2' else
                            15 if args.config == 'c1' else 32)
        if gb % world:
            raise SystemExit(f'global batch {gb} does not divide over {world}')
        barrier()
        if args.config in ('c5', 'c5small'):
            if world > 1:
                raise SystemExit('--config c5 is a 1-GPU leg')
            gb = args.batch or (8 if args.config == 'c5' else 4)
            out = condmom_leg(gb, *(((16, 16, 24, 2), (48, 48, 96, 2))
                                    if args.config == 'c5' else
                                    ((4, 4, 4, 2), (12, 12, 16, 2))),
                              1e9, args.steps)
        else:
            out = train_leg(args.config, gb, world, rank, 1e9, args.steps,
                            multi_gpu=True, precision=args.precision)
        barrier()
        if rank == 0:
            print(json.dumps(dict(
                base, metric='samples/sec, Sup3rGan._train_batch (generator '
                             'step + discriminator step)',
                value=out['value'], unit='samples/s',
                ms_per_step=out['ms_per_step'], scaling='strong',
                dtype=args.precision, steps=out['steps'],
                config={'workload': out['workload'], 'global_batch': gb,
                        'parallelism': f'batch split x{world}, RCCL all-'
                                       'reduce (SUM) of the flat gradient '
                                       'buffer per step'},
                train=out)))
        return

    if args.mode == 'c1':
        from sup3r_amd.engine import Device, Network
        with open(os.path.join(CFGDIR, 'gen_2x_2f.json')) as f:
            spec1 = json.load(f)
        dev = Device.get(local_rank)
        B = args.batch or 256
        shape = (B, 10, 10, 2)
        net = Network(spec1, name='generator', device=dev, precision='bf16')
        net.build(shape, seed=0)
        ph = net.plan(shape, training=False)
        fused = ph.op_info(0)['fwd'] == 'fused2d'
        x = dev.to_device(np.random.default_rng(42 + rank).standard_normal(
            shape).astype(np.float32))
        out = dev.empty((B, 20, 20, 2))
        for _ in range(max(args.warmup, 3)):
            ph.forward(x, out=out)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ph.forward(x, out=out)
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(out).all().item()
        if rank == 0:
            print(json.dumps(dict(
                base, metric='samples/sec (observations), generator forward, '
                             'spatial 2x GAN (C1)',
                value=world * B * args.steps / el, unit='samples/s',
                ms_per_step=el / args.steps * 1e3, scaling='weak',
                dtype='bf16',
                config={'workload': f'C1: gen_2x_2f forward, lo-res ({B},10,10,2)'
                                    f' -> hi-res ({B},20,20,2) per GPU per '
                                    'step, 0.274 GFLOP per observation',
                        'kernel': 'fused2d_kernel (whole network, one launch)'
                        if fused else 'op-by-op launches',
                        'parallelism': f'observations sharded x{world}'})))
        return

    if args.mode == 'fwd2d':
        from sup3r_amd.engine import Device
        dev = Device.get(local_rank)
        barrier()
        out = fwd2d_leg(dev, args.steps, max(args.warmup, 3),
                        args.batch or 48, seed=42 + rank)
        barrier()
        if not args.no_executor:
            try:
                out['executor'] = fwp2d_executor_leg(rank)
            except Exception as e:          # a leg, never the line
                out['executor'] = {'error': repr(e)[:300]}
            try:
                out['chain'] = fwp2d_chain_leg(rank)
            except Exception as e:
                out['chain'] = {'error': repr(e)[:300]}
        ms_ = max_over_ranks(out['ms_per_step'])
        if rank == 0 and not args.no_cpu_baseline:
            fwd2d_parity(out)
        else:
            out.pop('_parity_inputs', None)
        if rank == 0:
            B2 = args.batch or 48
            print(json.dumps(dict(
                base, metric='samples/sec (time steps), 2-D generator forward, '
                             'spatial 2x GAN at the config_fwp_spatial.json '
                             'chunk shape',
                value=world * B2 / (ms_ * 1e-3), unit='samples/s',
                ms_per_step=ms_, scaling='weak', dtype='bf16',
                config={'workload': out['workload'],
                        'parallelism': f'chunks sharded x{world}, no collective'},
                roofline=out['roofline'], fwd2d=out)))
        return

    if args.mode == 'c3':
        b = args.batch or 16     # (16 chunks per launch sequence: + 3 % over 8)
        barrier()
        # (>= 5 untimed batches: the ring of pinned 368 MB delivery buffers is
        # 4 deep, each first allocation costs 12 - 25 ms of hipHostMalloc)
        n, el, extra = c3_leg(b, args.steps, max(args.warmup, 5), world, rank,
                              entry=args.c3_entry)
        barrier()
        el = max_over_ranks(el)
        if rank == 0:
            print(json.dumps(dict(
                base, metric='chunks/sec, ForwardPass over a 400x400x720 '
                             'domain tiled into 20x20x48 chunks, 5x/12x ST-GAN',
                value=world * n / el, unit='chunks/s',
                ms_per_step=el / args.steps * 1e3, scaling='weak',
                dtype='bf16', in_situ=extra,
                px_per_sec=world * n / el * 100 * 100 * 576,
                config={'workload': 'C3: gen_5x_12x_2f through ' + (
                    'ForwardPassChunk structures (ArrayStrategy.init_chunk '
                    '-> ForwardPass.get_input_chunk -> iter_chunks, the '
                    "reference's run(strategy, node) path)"
                    if args.c3_entry == 'strategy' else
                    'ForwardPass.run_batched (resident domain)')
                    + f', {b} chunks (22,22,52,4) per '
                    'launch sequence, cropped (100,100,576,2) '
                    'chunks delivered to the host',
                        'parallelism': f'chunk list sharded x{world}, no '
                                       'collective'})))
        return

    # ------------------------------------------------ headline: C2 inference
    from sup3r_amd.engine import Device, Network
    with open(CFG) as f:
        spec = json.load(f)
    dev = Device.get(local_rank)
    net = Network(spec, name='generator', device=dev,
                  precision=args.precision)
    B = args.batch or 32
    shape = (B,) + LR_SHAPE
    net.build(shape, seed=0)                    # glorot-uniform, zero bias
    ph = net.plan(shape, training=False)
    rng = np.random.default_rng(42 + rank)
    x = dev.to_device(rng.standard_normal(shape).astype(np.float32))
    out = dev.empty((B,) + HR_SHAPE)
    for _ in range(args.warmup):
        ph.forward(x, out=out)
    barrier()
    ph.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ph.forward(x, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    n_prof, ms = ph.profile_end()
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(out).all().item()

    # Multi-GPU run: the data-parallel training step (RCCL gradient SUM) and
    # the chunk-sharded C3 executor are measured as well, each in its OWN
    # group of child processes (same ranks, another rendezvous port) after the
    # headline is complete — a collective that hangs or a runtime that aborts
    # there costs that leg, never the headline line.
    result_holder = {}

    ms_per_step = elapsed / args.steps * 1e3
    samples_per_s = world * B * args.steps / elapsed
    body = _body_ops(ph)
    body_ms = float(np.mean([ms[i] for i in body])) if body else float('nan')
    flop = BODY_CONV_FLOP_PER_SAMPLE * B
    achieved = flop / (body_ms * 1e-3) / 1e12
    esize = 2 if args.precision == 'bf16' else 4
    body_bytes = BODY_CONV_ELEMS_PER_SAMPLE * esize * B
    # (17 of the 33 body convs also read a SkipConnection residual — a
    # compulsory third stream of the fused op, half the in + out bytes)
    n_res = sum(1 for i in body if ph.plan.ops[i].get('res', -1) >= 0)
    body_bytes_res = body_bytes * (1.0 + 0.5 * n_res / max(1, len(body)))
    peak = PEAK_TFLOPS[args.precision]
    conv_ms = sum(ms[i] for i, op in enumerate(ph.plan.ops) if 'cout' in op)
    kclass = ph.op_kernel_class(body[0]) if body else 0
    result = dict(
        base, metric='samples/sec (lo-res chunks), generator forward, '
                     '5x/12x ST-GAN',
        value=samples_per_s, unit='samples/s', ms_per_step=ms_per_step,
        scaling='weak', dtype=args.precision,
        px_per_sec=samples_per_s * float(np.prod(HR_SHAPE[:3])),
        gflop_per_sample=GEN_FLOP_PER_SAMPLE / 1e9,
        whole_path_tflops=samples_per_s / world * GEN_FLOP_PER_SAMPLE / 1e12,
        config={
            'workload': 'C2: gen_5x_12x_2f generator forward, lo-res '
                        f'({B},16,16,24,4) -> hi-res ({B},80,80,288,2) per GPU '
                        'per step, inputs resident in HBM, random-init weights',
            'batch_per_gpu': B, 'precision': args.precision,
            'activations': ('bf16 trunk / fp32 I/O, NDHWC'
                            if args.precision == 'bf16' else 'fp32 NDHWC'),
            'parallelism': f'chunk-sharded x{world}, no collective'},
        roofline={
            'kernel': ('conv3_mfma_persist_kernel' if kclass == 2
                       else 'conv3_mfma_kernel') +
                      ' (Conv3D 64->64 k3, reflect-pad fused)',
            'bound': 'mfma', 'achieved': achieved, 'peak': peak,
            'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': None,
            'algorithmic_bytes_per_launch': body_bytes,
            'algorithmic_bytes_incl_residual_reads': body_bytes_res,
            'launches_per_step': len(body), 'avg_launch_ms': body_ms,
            'hbm_algorithmic_GBps': body_bytes / (body_ms * 1e-3) / 1e9,
            'hbm_frac': body_bytes / (body_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            'conv_ms_per_step': conv_ms, 'all_ops_ms_per_step': sum(ms),
            'forwards_profiled': n_prof,
            'hbm_bound_ops': hbm_class_ops(ph, ms),
            'peak_note': 'dense bf16 MFMA peak (MI355X_MICROARCH.md); the '
                         "guide's best plain-HIP GEMM sustains 1330-1470 "
                         'TFLOP/s on random operands (the chip is power-'
                         'limited under dense MFMA)'})
    if args.dump_ops:
        with open(args.dump_ops, 'w') as f:
            for i, op in enumerate(ph.plan.ops):
                f.write('{:3d} kind={} cin={} cout={} out={} mfma={} ms={:.4f}\n'
                        .format(i, op['kind'], op.get('cin'), op.get('cout'),
                                'x'.join(str(v) for v in
                                         ph.plan.tensors[op['out']]),
                                int(ph.op_is_mfma(i)), ms[i]))
    result_holder['line'] = result
    if world > 1 or args.force_legs:
        if args.precision == 'bf16' and not args.no_train:
            legs = (('train', ['--mode', 'train', '--config', 'c2', '--steps',
                               '20'], 300),
                    ('c3', ['--mode', 'c3', '--steps', '3', '--warmup', '1'],
                     300),
                    ('train_c4', ['--mode', 'train', '--config', 'c4',
                                  '--steps', '20'], 240))
            for k, (name, leg_args, limit) in enumerate(legs):
                sub = child_leg(leg_args, world, k, limit)
                if rank == 0:
                    result[name] = sub
        if rank == 0:
            print(json.dumps(result), flush=True)
        return

    single = args.precision == 'bf16'
    weights = net.weights
    if single and not args.no_parity_mode:
        # the same workload in the two modes that own the L-inf < 1e-3 claim
        # (tests/test_parity_r02.py): BF16X3 (split-bf16 MFMA) and exact fp32
        modes = {}
        for prec, steps in (('bf16x3', 5), ('f32', 3)):
            dt, b_ms, kinds = time_mode(spec, weights, dev, x, out, prec, steps)
            a = flop / (b_ms * 1e-3) / 1e12
            modes[prec] = {
                'value': B / dt, 'unit': 'samples/s', 'ms_per_step': dt * 1e3,
                'body_conv_ms': b_ms, 'kernels': kinds,
                'achieved_tflops_fp32_equivalent': a}
        result['parity_mode'] = dict(
            modes['bf16x3'], dtype='bf16x3',
            kernel='conv3_mfma_kernel<BF16X3> (hi*hi + hi*lo + lo*hi on '
                   'v_mfma_f32_16x16x32_bf16, fp32 activations)',
            mfma_frac=3 * modes['bf16x3']['achieved_tflops_fp32_equivalent']
            / PEAK_TFLOPS['bf16'],
            tolerance='L-inf < 1e-3 vs the fp32 oracle at full C2 size '
                      '(tests/test_parity_r02.py; measured on this run: '
                      'cpu_baseline.parity)',
            f32=dict(modes['f32'],
                     kernel='conv3_mfma_kernel (v_mfma_f32_16x16x4_f32, exact '
                            'fp32)',
                     frac=modes['f32']['achieved_tflops_fp32_equivalent']
                     / PEAK_TFLOPS['f32']),
            speedup_vs_f32=modes['bf16x3']['value'] / modes['f32']['value'])
    if single and not args.no_parity_mode:
        try:
            result['roofline']['power'] = power_leg(spec, weights, dev, x, out)
        except Exception as e:              # evidence, never fatal
            result['roofline']['power'] = {'error': repr(e)[:200]}
    del ph, net
    torch.cuda.empty_cache()
    if single and not args.no_train:
        # the other half of the metric: full
        # Sup3rGan._train_batch steps of the C2 GAN, batch 8 — reported beside
        # the headline, not part of `value`
        result['train'] = train_leg('c2', 8, 1, 0, args.train_seconds, 400,
                                    multi_gpu=False)
        if not args.no_parity_mode:
            # the training mode whose gradients meet the fp32 tolerance
            # (tests/test_parity_r03.py: 1.2e-5 / 3.7e-5 vs the fp32 oracle)
            x3 = train_leg('c2', 8, 1, 0, min(args.train_seconds, 3.0), 100,
                           multi_gpu=False, precision='bf16x3')
            result['train']['parity_mode'] = dict(
                dtype='bf16x3', ms_per_step=x3['ms_per_step'],
                value=x3['value'], unit=x3['unit'], steps=x3['steps'],
                tolerance='gradients of both steps <= 1e-4 of the fp32 '
                          'oracle under the device masks')
    if single and not args.no_train:
        # the path ForwardPassStrategy drives (BASELINE.json config 3, one
        # rank's share): ForwardPassChunk structures through iter_chunks, the
        # cropped hi-res chunks delivered to the host
        try:
            # (a rank's share of C3 is 47 launch sequences of 16 chunks: 32
            # timed ones carry about that job's share of pipeline fill / drain)
            n3, el3, extra3 = c3_leg(16, 32, 5, 1, 0)
            result['c3'] = dict(
                value=n3 / el3, unit='chunks/s', ms_per_step=el3 / 32 * 1e3,
                steps=32, warmup=5, chunks_per_step=16,
                px_per_sec=n3 / el3 * 100 * 100 * 576,
                gflop_per_chunk=1872.0,
                whole_path_tflops=n3 / el3 * 1.872,
                workload='C3: 400x400x720 domain in 20x20x48 chunks + halo '
                         '(22,22,52,4), 16 per launch sequence, through '
                         'ForwardPass.get_input_chunk -> iter_chunks; cropped '
                         '(100,100,576,2) fp32 chunks delivered to pinned host '
                         'memory by SDMA under the next batch',
                in_situ=extra3)
        except Exception as e:              # a leg, never the headline
            result['c3'] = {'error': repr(e)[:300]}
        torch.cuda.empty_cache()
        # the production shape of the reference's 2-D (spatial) steps
        try:
            result['fwd2d'] = fwd2d_leg(dev)
            result['fwd2d']['executor'] = fwp2d_executor_leg()
        except Exception as e:
            result.setdefault('fwd2d', {})['error'] = repr(e)[:300]
        torch.cuda.empty_cache()
        try:
            result['fwd2d']['chain'] = fwp2d_chain_leg()
        except Exception as e:
            result.setdefault('fwd2d', {})['chain'] = {'error': repr(e)[:300]}
        torch.cuda.empty_cache()
        # the reference's own training test shape (BASELINE.json config 1,
        # tests/training/test_train_gan.py:45-114): a launch-bound mini-batch
        try:
            result['train_c1'] = train_leg('c1', 15, 1, 0, 1.0, 400,
                                           multi_gpu=False)
        except Exception as e:
            result['train_c1'] = {'error': repr(e)[:300]}
        torch.cuda.empty_cache()
        # BASELINE.json config 4, ONE GPU's share of the batch of 32 (4 samples;
        # the data-parallel line itself needs --gpus 8): the gen_3x_4x_2f body +
        # discriminator at lr (4,16,16,24,2), and the reference's filters: 1 toy
        # with topography at lr (4,4,4,4,2); `comm` shows world 1
        try:
            result['train_c4'] = train_leg('c4', 4, 1, 0, 2.0, 200,
                                           multi_gpu=False)
            result['train_c4']['toy'] = train_leg('c4toy', 4, 1, 0, 1.0, 400,
                                                  multi_gpu=False)
        except Exception as e:
            result.setdefault('train_c4', {})['error'] = repr(e)[:300]
        torch.cuda.empty_cache()
        # BASELINE.json config 5: Sup3rCondMom over the same conv stack, at
        # BASELINE.md's shape and at a production-like one
        try:
            result['train_c5'] = condmom_leg(8, (16, 16, 24, 2), (48, 48, 96, 2),
                                             2.0, 200)
            result['train_c5']['baseline_shape'] = condmom_leg(
                4, (4, 4, 4, 2), (12, 12, 16, 2), 1.0, 400)
        except Exception as e:
            result.setdefault('train_c5', {})['error'] = repr(e)[:300]
        torch.cuda.empty_cache()
    if single and not args.no_traffic:
        traffic, note = measure_traffic(B)
        result['roofline']['traffic'] = traffic
        result['roofline']['traffic_source'] = note
        if traffic:
            result['roofline']['traffic_over_algorithmic'] = \
                traffic / body_bytes
            result['roofline']['traffic_over_algorithmic_incl_residual'] = \
                traffic / body_bytes_res
        tail = getattr(measure_traffic, 'tail', None)
        for o in result['roofline']['hbm_bound_ops']:
            if tail and 'tail_mfma' in o['what']:
                o['traffic_pmc_bytes'] = tail
                o['traffic_over_algorithmic'] = tail / o['algorithmic_bytes']
    if isinstance(result.get('fwd2d'), dict):
        if args.no_cpu_baseline:
            result['fwd2d'].pop('_parity_inputs', None)
        else:
            fwd2d_parity(result['fwd2d'])
    if not args.no_cpu_baseline:
        cpu, parity = cpu_legs(spec, dev)
        cpu['parity'] = parity
        result['cpu_baseline'] = cpu
        result['speedup_vs_cpu_baseline'] = samples_per_s / cpu['value']
    print(json.dumps(result))


if __name__ == '__main__':
    main()
