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
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = os.path.join(ROOT, 'sup3r_amd', 'configs', 'gen_5x_12x_2f.json')
LR_SHAPE = (16, 16, 24, 4)
HR_SHAPE = (80, 80, 288, 2)
# algorithmic work, SURVEY.md §8(d) / DESIGN.md: generator forward per sample
GEN_FLOP_PER_SAMPLE = 598.9e9
# dominant kernel: Conv3D 64->64 3x3x3 on (16,16,288): 2*73728*64*1728 FLOP
BODY_CONV_FLOP_PER_SAMPLE = 2.0 * 16 * 16 * 288 * 64 * 27 * 64
# ... and its algorithmic HBM bytes per sample: read the un-padded input once +
# write the output once (bf16 mode keeps the 64-channel trunk in bf16)
BODY_CONV_ELEMS_PER_SAMPLE = 2.0 * 16 * 16 * 288 * 64
TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'r01', 'traffic.json')
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}   # MI355X_MICROARCH.md (dense)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(spec, seconds_budget=30.0):
    """CPU baseline on the host cores, one C2 chunk per run, the op sequence AS
    TF EXECUTES IT (un-fused REFLECT pad-3 / valid conv / crop-2, NDHWC fp32):

    * primary ("TF-CPU proxy"): oracle/torch_proxy.py — the oracle network's
      weights run through torch-CPU, i.e. oneDNN convolutions, the x86 conv
      backend TF 2.15 uses; checked against the numpy oracle on this sample;
    * also reported: the numpy oracle itself (BLAS GEMM per tap), 1 run.
    TensorFlow is not installable here, so neither is a TF measurement."""
    import torch
    from oracle.network import Network as OracleNet
    from oracle.torch_proxy import torch_generator_forward
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1,) + LR_SHAPE).astype(np.float32)
    net = OracleNet(spec)
    t0 = time.time()
    net.init_weights(x, seed=0)       # includes one forward (lazy build)
    t_numpy = time.time() - t0
    y_np = net.forward(x) if t_numpy < 8 else None
    threads = torch.get_num_threads()
    y_t, _ = torch_generator_forward(net, x)          # warm-up
    if y_np is not None:
        assert np.abs(y_t - y_np).max() < 1e-3
    n, el = 0, 0.0
    while n < 5 and el < seconds_budget - 10:
        _, dt = torch_generator_forward(net, x)
        n += 1
        el += dt
    return {'value': n / el, 'unit': 'samples/s', 'cores': int(threads),
            'kind': 'port',
            'sample': f'{n} x one C2 chunk (1,16,16,24,4)->(1,80,80,288,2), '
                      'as-TF-executes op sequence (474 GMAC/sample) via '
                      f'torch-CPU/oneDNN ("TF-CPU proxy"), {el / n:.2f} '
                      f's/sample on {threads} threads; numpy oracle (BLAS per '
                      f'tap, incl. lazy build): {t_numpy:.1f} s/sample',
            'numpy_oracle_s_per_sample': t_numpy}


def train_step_rate(batch=8, iters=3):
    """ms per ``Sup3rGan._train_batch`` of the C2 generator + production
    discriminator on synthetic batches (4 G + 9 D = 2 860 GFLOP / sample)"""
    import torch
    from sup3r_amd import Sup3rGan
    cfg = os.path.dirname(CFG)
    model = Sup3rGan(CFG, os.path.join(cfg, 'disc_st.json'),
                     loss='MeanAbsoluteError', precision='bf16')
    lr_shape, hr_shape = (batch,) + LR_SHAPE, (batch,) + HR_SHAPE
    rng = np.random.default_rng(0)

    class Batch:
        low_res = rng.standard_normal(lr_shape).astype(np.float32)
        high_res = rng.standard_normal(hr_shape).astype(np.float32)
    model.init_weights(lr_shape, hr_shape)

    def step():
        return model._train_batch(Batch, True, False, False, True, False,
                                  False, 1e-3)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    del model
    torch.cuda.empty_cache()
    return {'workload': 'C2 Sup3rGan._train_batch (gen step + disc step), '
                        f'batch {batch}, gen_5x_12x_2f + disc_st, bf16 MFMA '
                        'operands, MeanAbsoluteError',
            'ms_per_step': dt * 1e3, 'value': batch / dt, 'unit': 'samples/s',
            'algorithmic_gflop_per_sample': 2860.0,
            'tflops': 2860.0 * batch / dt / 1e3, 'steps': iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32,
                    help='lo-res chunks per GPU per step (throughput vs batch '
                         'on one MI355X: 4: 1600, 8: 1890, 16: 1910, 32: 1970, '
                         '64: 2000 samples/s)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-mode', action='store_true',
                    help='skip the extra fp32 parity-mode measurement')
    ap.add_argument('--no-train', action='store_true',
                    help='skip the extra training-step measurement')
    ap.add_argument('--dump-ops', default=None,
                    help='write per-op mean ms of the timed region here')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda',
                                                               local_rank))

    from sup3r_amd.engine import Device, Network
    with open(CFG) as f:
        spec = json.load(f)
    dev = Device.get(local_rank)
    net = Network(spec, name='generator', device=dev,
                  precision=args.precision)
    B = args.batch
    shape = (B,) + LR_SHAPE
    net.build(shape, seed=0)                    # glorot-uniform, zero bias
    ph = net.plan(shape, training=False)
    rng = np.random.default_rng(42 + rank)
    x = dev.to_device(rng.standard_normal(shape).astype(np.float32))
    out = dev.empty((B,) + HR_SHAPE)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    if world > 1:
        tt = torch.tensor([elapsed], device=dev.torch_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all().item()

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    samples_per_s = world * B * args.steps / elapsed
    # dominant kernel = the 64->64 body convs on the full (16,16,288) grid
    body = [i for i, op in enumerate(ph.plan.ops)
            if ph.op_is_mfma(i) and op['cout'] == 64
            and ph.plan.tensors[op['out']][1:4] == [16, 16, 288]]
    body_ms = float(np.mean([ms[i] for i in body])) if body else float('nan')
    flop = BODY_CONV_FLOP_PER_SAMPLE * B
    achieved = flop / (body_ms * 1e-3) / 1e12
    esize = 2 if args.precision == 'bf16' else 4
    body_bytes = BODY_CONV_ELEMS_PER_SAMPLE * esize * B
    # measured HBM traffic of this kernel (PMC, separate rocprofv3 passes of
    # the same command; committed under profiles/), scaled to this batch
    traffic = None
    if args.precision == 'bf16' and os.path.exists(TRAFFIC_JSON):
        with open(TRAFFIC_JSON) as f:
            tj = json.load(f)
        traffic = tj['traffic_bytes_per_launch'] * B / tj['batch']
    peak = PEAK_TFLOPS[args.precision]
    conv_ms = sum(ms[i] for i, op in enumerate(ph.plan.ops) if 'cout' in op)
    result = {
        'metric': 'samples/sec (lo-res chunks), generator forward, '
                  '5x/12x ST-GAN',
        'value': samples_per_s, 'unit': 'samples/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'px_per_sec': samples_per_s * float(np.prod(HR_SHAPE[:3])),
        'gflop_per_sample': GEN_FLOP_PER_SAMPLE / 1e9,
        'whole_path_tflops': samples_per_s / world * GEN_FLOP_PER_SAMPLE / 1e12,
        'config': {
            'workload': 'C2: gen_5x_12x_2f generator forward, lo-res '
                        f'({B},16,16,24,4) -> hi-res ({B},80,80,288,2) per GPU '
                        'per step, inputs resident in HBM, random-init weights',
            'batch_per_gpu': B, 'precision': args.precision,
            'activations': ('bf16 trunk / fp32 I/O, NDHWC'
                            if args.precision == 'bf16' else 'fp32 NDHWC'),
            'parallelism': f'chunk-sharded x{world}, no collective'},
        'roofline': {
            'kernel': ('conv3_mfma_persist_kernel'
                       if body and ph.op_kernel_class(body[0]) == 2
                       else 'conv3_mfma_kernel') +
                      ' (Conv3D 64->64 k3, reflect-pad fused)',
            'bound': 'mfma', 'achieved': achieved, 'peak': peak,
            'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic,
            'algorithmic_bytes_per_launch': body_bytes,
            'launches_per_step': len(body), 'avg_launch_ms': body_ms,
            'hbm_algorithmic_GBps': body_bytes / (body_ms * 1e-3) / 1e9,
            'hbm_frac': body_bytes / (body_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            'conv_ms_per_step': conv_ms, 'all_ops_ms_per_step': sum(ms),
            'forwards_profiled': n_prof},
    }
    if args.dump_ops:
        with open(args.dump_ops, 'w') as f:
            for i, op in enumerate(ph.plan.ops):
                f.write('{:3d} kind={} cin={} cout={} out={} mfma={} ms={:.4f}\n'
                        .format(i, op['kind'], op.get('cin'), op.get('cout'),
                                'x'.join(str(v) for v in
                                         ph.plan.tensors[op['out']]),
                                int(ph.op_is_mfma(i)), ms[i]))
    if world == 1 and args.precision == 'bf16' and not args.no_parity_mode:
        # the same workload in the exact-fp32 parity mode (the mode that owns
        # the L-inf < 1e-3 claim of tests/test_hip_parity.py), untimed by the
        # contract, reported beside the headline
        net32 = Network(spec, name='generator', device=dev, precision='f32')
        net32.set_weights(net.weights)
        ph32 = net32.plan(shape, training=False)
        for _ in range(2):
            ph32.forward(x, out=out)
        torch.cuda.synchronize()
        ph32.profile_begin(5)
        t1 = time.perf_counter()
        for _ in range(5):
            ph32.forward(x, out=out)
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t1) / 5
        _, ms32 = ph32.profile_end()
        body32 = [i for i, op in enumerate(ph32.plan.ops)
                  if ph32.op_is_mfma(i) and op['cout'] == 64
                  and ph32.plan.tensors[op['out']][1:4] == [16, 16, 288]]
        b32 = float(np.mean([ms32[i] for i in body32]))
        a32 = flop / (b32 * 1e-3) / 1e12
        result['parity_mode'] = {
            'dtype': 'f32', 'value': B / dt32, 'unit': 'samples/s',
            'ms_per_step': dt32 * 1e3,
            'kernel': 'conv3_mfma_kernel (v_mfma_f32_16x16x4_f32, exact fp32)',
            'achieved': a32, 'peak': PEAK_TFLOPS['f32'], 'unit_roofline':
            'TFLOP/s', 'frac': a32 / PEAK_TFLOPS['f32'],
            'tolerance': 'L-inf < 1e-3 vs the oracle at full C2 size '
                         '(tests/test_hip_parity.py); bf16 mode: 3e-2 rel.'}
        del ph32, net32
    if world == 1 and args.precision == 'bf16' and not args.no_train:
        # the other half of the metric: one full
        # Sup3rGan._train_batch (generator step + discriminator step) of the
        # C2 GAN, batch 8 — reported beside the headline, not part of `value`
        result['train'] = train_step_rate()
    if world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(spec)
        result['speedup_vs_cpu_baseline'] = \
            samples_per_s / result['cpu_baseline']['value']
    print(json.dumps(result))


if __name__ == '__main__':
    main()
