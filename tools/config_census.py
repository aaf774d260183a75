"""Per-config census of the shipped spec surface on the device.

For every generator under ``sup3r_amd/configs/sup3r/`` (the reference's 16)
plus the C2 generator, at ONE production-like low-res shape per family (from
the reference's ``examples/**/config_fwp_*.json`` chunk shapes + pads):

* which forward kernel every conv selected (``s3_plan_op_info``), with the
  per-op HIP-event time (``s3_plan_profile_begin/end``),
* forward samples/s, useful TFLOP/s (2 * MACs of the convs / time),
* the training plan's wgrad / dgrad kernel selection and fwd+bwd time.

Writes a markdown table to ``gpurun_out/config_census.md`` (copied to
``profiles/rNN/`` by hand).  Usage: python tools/config_census.py [--quick]
"""
import argparse
import collections
import glob
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402

CFG = os.path.join(ROOT, 'sup3r_amd', 'configs')

# (config, low-res shape, origin of the shape)
SHAPES = [
    ('sup3r/spatial/gen_2x_1f.json', (48, 75, 75, 1),
     'sup3rwind config_fwp_spatial: chunk [75,75,38] + temporal_pad 5, time -> batch'),
    ('sup3r/spatial/gen_2x_2f.json', (48, 75, 75, 2), 'same'),
    ('sup3r/spatial/gen_10x_2f.json', (48, 20, 20, 2), '200x200 output tiles'),
    ('sup3r/spatiotemporal/gen_2x_2x_2f.json', (4, 20, 20, 24, 2), 'C3-like chunk'),
    ('sup3r/spatiotemporal/gen_2x_12x_14f.json', (2, 20, 20, 12, 14), 'C3-like chunk'),
    ('sup3r/spatiotemporal/gen_3x_4x_1f.json', (4, 20, 20, 24, 1), 'C3-like chunk'),
    ('sup3r/spatiotemporal/gen_3x_4x_2f.json', (4, 20, 20, 24, 2), 'C3-like chunk'),
    ('sup3r/spatiotemporal/gen_3x_4x_10f.json', (4, 20, 20, 24, 10), 'C3-like chunk'),
    ('sup3r/spatiotemporal/gen_3x_4x_14f.json', (4, 20, 20, 24, 14), 'C3-like chunk'),
    ('sup3r/spatiotemporal/gen_4x_24x_3f.json', (2, 16, 16, 8, 3), 'C3-like chunk'),
    ('sup3r/sup3rcc/gen_solar_1x_8x_1f.json', (8, 54, 54, 3, 3),
     'sup3rcc solar step1: chunk [52,52,1] + pad 1'),
    ('sup3r/sup3rcc/gen_trh_1x_24x_2f.json', (1, 54, 54, 10, 4),
     'sup3rcc nearsurf step1: chunk [52,52,37] + pad 1 (t cut to 10)'),
    ('sup3r/sup3rcc/gen_wind_1x_24x_6f.json', (1, 54, 54, 10, 6), 'same'),
    ('sup3r/sup3rcc/gen_solar_5x_1x_1f.json', (60, 16, 16, 3),
     'sup3rcc step2: chunk [10,10,48] + pad 3 / 6, time -> batch'),
    ('sup3r/sup3rcc/gen_wind_5x_1x_6f.json', (60, 16, 16, 7), 'same'),
    ('sup3r/sup3rcc/gen_wind_3x_4x_2f.json', (4, 20, 20, 24, 2), 'toy (filters: 1)'),
    ('gen_5x_12x_2f.json', (32, 16, 16, 24, 4), 'BASELINE C2, batch 32'),
]


def conv_macs(plan):
    total = 0
    per_op = {}
    for i, op in enumerate(plan.ops):
        if op['kind'] == S.OP_CONV:
            osh = plan.tensors[op['out']]
            b = op.get('d2s', 1) or 1
            npos = osh[0] * (osh[1] // b) * (osh[2] // b) * osh[3]
            m = npos * op['cin'] * op['cout'] * int(np.prod(op['k']))
        elif op['kind'] == S.OP_DENSE:
            m = plan.tensors[op['out']][0] * op['cin'] * op['cout']
        else:
            m = 0
        per_op[i] = m
        total += m
    return total, per_op


def exo_inputs(ph, dev, rng):
    out = {}
    for name, sh in ph.in_shapes.items():
        if name != 'x':
            out[name] = dev.to_device(
                rng.standard_normal(tuple(sh)).astype(np.float32))
    return out


def time_fwd(ph, x, exo, iters):
    dev = ph.dev
    for _ in range(2):
        ph.forward(x, exo)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        ph.forward(x, exo)
    dev.sync()
    return (time.perf_counter() - t0) / iters * 1e3


def census_one(rel, shape, prec, iters, train):
    with open(os.path.join(CFG, rel)) as f:
        spec = json.load(f)
    rng = np.random.default_rng(0)
    net = Network(spec, precision=prec)
    net.build(shape, seed=1)
    dev = net.dev
    ph = net.plan(shape, training=False)
    macs, per_op = conv_macs(ph.plan)
    x = dev.to_device(rng.standard_normal(shape).astype(np.float32))
    exo = exo_inputs(ph, dev, rng)
    ms = time_fwd(ph, x, exo, iters)
    ph.profile_begin(iters)
    for _ in range(iters):
        ph.forward(x, exo)
    dev.sync()
    _, op_ms = ph.profile_end()
    rows = []
    agg = collections.OrderedDict()
    for i, op in enumerate(ph.plan.ops):
        if op['kind'] not in (S.OP_CONV, S.OP_DENSE):
            key = ('op', {S.OP_REPEAT_T: 'repeat_t', S.OP_D2S: 'd2s',
                          S.OP_ACT: 'act', S.OP_ADD: 'add',
                          S.OP_CONCAT: 'concat', S.OP_VIEW: 'view',
                          S.OP_ROLL_T: 'roll_t'}.get(op['kind'],
                                                     str(op['kind'])), '')
        else:
            info = ph.op_info(i)
            k = op.get('k', [1, 1, 1])
            nd = 3 if k[2] > 1 else 2
            key = (f"conv{nd}d {op['cin']}->{op['cout']}"
                   + (f" d2s{op['d2s']}" if op.get('d2s', 1) > 1 else '')
                   if op['kind'] == S.OP_CONV else
                   f"dense {op['cin']}->{op['cout']}", info['fwd'], '')
        a = agg.setdefault(key, [0, 0.0, 0])
        a[0] += 1
        a[1] += op_ms[i]
        a[2] += per_op[i]
    for (what, kern, _), (cnt, t, m) in agg.items():
        rows.append(dict(what=what, kernel=kern, n=cnt, ms=t,
                         tf=(2 * m / (t * 1e-3) / 1e12) if t > 0 and m else 0.0))
    res = dict(config=rel, shape=shape, out=tuple(ph.out_shape), prec=prec,
               gflop_per_sample=2 * macs / shape[0] / 1e9, ms=ms,
               samples_s=shape[0] / (ms * 1e-3),
               tf=2 * macs / (ms * 1e-3) / 1e12, rows=rows)
    del ph
    net.clear_plans()
    if train:
        n = max(1, shape[0] // 4)
        tshape = (n,) + tuple(shape[1:])
        pht = net.plan(tshape, training=True)
        xt = dev.to_device(rng.standard_normal(tshape).astype(np.float32))
        exot = exo_inputs(pht, dev, rng)
        dy = dev.to_device(rng.standard_normal(
            tuple(pht.out_shape)).astype(np.float32))
        for _ in range(2):
            pht.forward(xt, exot)
            pht.backward(dy, need_dx=False)
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(max(2, iters // 2)):
            pht.forward(xt, exot)
            pht.backward(dy, need_dx=False)
        dev.sync()
        tms = (time.perf_counter() - t0) / max(2, iters // 2) * 1e3
        sel = collections.Counter()
        for i, op in enumerate(pht.plan.ops):
            if op['kind'] == S.OP_CONV:
                info = pht.op_info(i)
                sel[(f"{op['cin']}->{op['cout']}", info['fwd'], info['wgrad'],
                     info['dgrad'])] += 1
        res['train'] = dict(shape=tshape, ms=tms,
                            tf=3 * 2 * macs * n / shape[0] / (tms * 1e-3) / 1e12,
                            sel=sel)
        del pht
        net.clear_plans()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--only', default=None)
    ap.add_argument('--precisions', default='bf16,bf16x3')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out',
                                                  'config_census.md'))
    a = ap.parse_args()
    iters = 3 if a.quick else 8
    lines = ['# Config census: every shipped generator spec on the HIP path',
             '',
             'One production-like low-res shape per family; `TF/s` = 2 x conv '
             'MACs / time (useful work, bf16 dense peak 2 500).  Kernel names '
             'are `s3_plan_op_info` forward selections: `mfma_persist` / '
             '`mfma_tile` = LDS-halo dense MFMA, `tail_mfma` = few-C_out MFMA, '
             '`gconv*` = gather (implicit-GEMM) MFMA, `halo32` / `fused2d` / '
             '`direct` = VALU kernels.', '']
    summary = ['| config | low-res shape | output | precision | GFLOP/sample | '
               'ms/batch | samples/s | TF/s | train fwd+bwd ms (batch) | TF/s |',
               '|---|---|---|---|---|---|---|---|---|---|']
    detail = []
    for rel, shape, origin in SHAPES:
        if a.only and not re.search(a.only, rel):
            continue
        for prec in a.precisions.split(','):
            try:
                r = census_one(rel, shape, prec, iters,
                               train=(prec == 'bf16'))
            except Exception as e:   # a spec that does not run is a finding
                summary.append(f'| {rel} | {shape} | FAILED: '
                               f'{type(e).__name__}: {str(e)[:120]} | {prec} '
                               '| | | | | | |')
                print('FAILED', rel, prec, repr(e), flush=True)
                continue
            tr = r.get('train')
            summary.append(
                f"| {rel.replace('sup3r/', '')} | {shape} | {r['out']} | "
                f"{prec} | {r['gflop_per_sample']:.1f} | {r['ms']:.2f} | "
                f"{r['samples_s']:.1f} | {r['tf']:.0f} | "
                + (f"{tr['ms']:.2f} ({tr['shape'][0]}) | {tr['tf']:.0f} |"
                   if tr else '| |'))
            print(summary[-1], flush=True)
            detail.append(f"### {rel} {shape} {prec} — {origin}")
            detail.append('')
            detail.append('| op | forward kernel | count | ms | TF/s |')
            detail.append('|---|---|---|---|---|')
            for row in r['rows']:
                detail.append(f"| {row['what']} | {row['kernel']} | {row['n']} "
                              f"| {row['ms']:.3f} | {row['tf']:.0f} |")
            if tr:
                detail.append('')
                detail.append('training plan selection (conv: fwd / wgrad / '
                              'dgrad x count): ' + '; '.join(
                                  f"{k[0]}: {k[1]} / {k[2]} / {k[3]} x{v}"
                                  for k, v in tr['sel'].items()))
            detail.append('')
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        f.write('\n'.join(lines + summary + [''] + detail) + '\n')
    print('wrote', a.out)


if __name__ == '__main__':
    main()
