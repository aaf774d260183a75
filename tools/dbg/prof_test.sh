cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r4o_prof.txt 2>&1
import cProfile, pstats, sys, io, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
print('numpy', np.__version__); 
try:
    import threadpoolctl; print(threadpoolctl.threadpool_info())
except Exception as e: print(e)
print('cpus', os.cpu_count())
from tests import test_parity_r03 as T
pr = cProfile.Profile(); t=time.time(); pr.enable()
T.test_c2_generator_bf16_backward_batch8_under_device_masks()
pr.disable(); print('total', time.time()-t)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30); print(s.getvalue())
PY
python -m pytest tests/test_parity_r02.py -x -q -k "persistent_kernel or wide_conv or sliding or first_disc or c4_ or stride2 or valid_conv" 2>&1 | tail -5
