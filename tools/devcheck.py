import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
t = time.time()
from sup3r_amd.engine import Device
d = Device.get()
print('device ok', time.time() - t, flush=True)
