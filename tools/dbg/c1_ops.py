import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from sup3r_amd import Sup3rGan
CFG = os.path.join(os.path.dirname(__file__), '..', '..', 'sup3r_amd', 'configs')
LR, HR = (15, 5, 5, 2), (15, 10, 10, 2)
Sup3rGan.seed(1)
m = Sup3rGan(os.path.join(CFG, 'gen_2x_2f.json'), os.path.join(CFG, 'disc_s_same.json'),
             loss='MeanAbsoluteError', precision='bf16')
m.init_weights(LR, HR)
for name, net, shape in (('gen', m._compute.gen, LR), ('disc', m._compute.disc, HR)):
    ph = net.plan(shape, training=True)
    c = collections.Counter()
    for i, op in enumerate(ph.plan.ops):
        info = ph.op_info(i)
        if info['kind'] == 1:
            key = (op.get('cin'), op.get('cout'), tuple(ph.plan.tensors[op['out']]), info['fwd'], info['wgrad'], info['dgrad'])
            c[key] += 1
    print(name)
    for k, v in c.items():
        print('  ', v, k)
