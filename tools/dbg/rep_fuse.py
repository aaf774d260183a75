"""debug: kernel selection of the repeat-fusion test network"""
import numpy as np
from sup3r_amd.configs.author_configs import pcc
from tests.test_hip_parity import _hip_net, _oracle_net
from sup3r_amd.engine import Device
spec = pcc(3, 64) + pcc(3, 64) + \
    [{'class': 'SpatioTemporalExpansion', 'temporal_mult': 2,
      'temporal_method': 'nearest'}] + pcc(3, 64) + \
    [{'class': 'SpatioTemporalExpansion', 'temporal_mult': 3,
      'temporal_method': 'nearest'},
     {'class': 'SkipConnection', 'name': 'a'},
     {'class': 'SkipConnection', 'name': 'b'}] + \
    pcc(3, 64) + pcc(3, 64, act=False) + \
    [{'class': 'SkipConnection', 'name': 'b'}] + \
    pcc(3, 64, act=False) + \
    [{'class': 'SkipConnection', 'name': 'a'}] + pcc(3, 2, act=False)
shape = (9, 18, 20, 13, 4)
x = np.zeros(shape, np.float32)
ref = _oracle_net(spec, x[:1], None)
Device.get().set_option('TRACE', 1)
net = _hip_net(spec, ref.weights, precision='bf16')
ph = net.plan(shape, training=False)
for i in range(len(ph.plan.ops)):
    print(i, ph.op_info(i))
