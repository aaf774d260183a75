"""per-op HIP-event times of ONE inference plan: kernel, dtypes, ms, TF/s
usage: python tools/dbg/op_profile.py <config under sup3r_amd/configs> <n,h,w[,t],c> [precision]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sup3r_amd import spec as S  # noqa: E402
from sup3r_amd.engine import Network  # noqa: E402

rel, shape = sys.argv[1], tuple(int(v) for v in sys.argv[2].split(','))
prec = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
spec = json.load(open(os.path.join(ROOT, 'sup3r_amd', 'configs', rel)))
rng = np.random.default_rng(0)
net = Network(spec, precision=prec)
net.build(shape, seed=1)
dev = net.dev
ph = net.plan(shape, training=False)
x = dev.to_device(rng.standard_normal(shape).astype(np.float32))
exo = {k: dev.to_device(rng.standard_normal(tuple(sh)).astype(np.float32))
       for k, sh in ph.in_shapes.items() if k != 'x'}
for _ in range(2):
    ph.forward(x, exo)
dev.sync()
iters = 3
ph.profile_begin(iters)
for _ in range(iters):
    ph.forward(x, exo)
dev.sync()
tot, op_ms = ph.profile_end()
names = {S.OP_CONV: 'conv', S.OP_DENSE: 'dense', S.OP_REPEAT_T: 'repeat_t', S.OP_D2S: 'd2s',
         S.OP_ACT: 'act', S.OP_ADD: 'add', S.OP_CONCAT: 'concat', S.OP_VIEW: 'view',
         S.OP_ROLL_T: 'roll_t'}
print('total %.3f ms' % sum(op_ms))
for i, op in enumerate(ph.plan.ops):
    k = names.get(op['kind'], str(op['kind']))
    osh = ph.plan.tensors[op['out']]
    d = lambda t: '16' if ph.tensor_is_bf16(t) else '32'
    extra = ''
    if op['kind'] == S.OP_CONV:
        info = ph.op_info(i)
        b = op.get('d2s', 1) or 1
        npos = osh[0] * (osh[1] // b) * (osh[2] // b) * osh[3]
        fl = 2.0 * npos * op['cin'] * op['cout'] * int(np.prod(op['k']))
        extra = '%d->%d%s %s res=%s %6.0f TF/s' % (
            op['cin'], op['cout'], (' d2s%d' % b) if b > 1 else '', info['fwd'],
            ('-' if op.get('res', -1) < 0 else d(op['res'])), fl / (op_ms[i] * 1e-3) / 1e12 if op_ms[i] > 0 else 0)
    print('%3d %-8s in%s out%s %-28s %9.3f ms  %s' % (i, k, d(op['in0']) if 'in0' in op else '--', d(op['out']),
                                                    str(tuple(osh)), op_ms[i], extra))
