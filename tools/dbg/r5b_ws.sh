#!/bin/bash
# one gpurun call: ws kernel parity tests + same-box A/B + fwd2d bench
mkdir -p gpurun_out/r5b
timeout 600 python -m pytest tests/test_mfma_gen.py tests/test_forward_pass_gpu.py tests/test_ref_surface.py -m gpu -x -q > gpurun_out/r5b/tests_ws.log 2>&1
tail -5 gpurun_out/r5b/tests_ws.log
timeout 300 bash tools/dbg/ws_ab.sh tools/ab/ws_head.so sup3r_amd/lib/libsup3r_hip.so 2>&1 | tee gpurun_out/r5b/ws_ab.log
timeout 300 python tools/dbg/ws_scaling.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5b/ws_scaling.log
cat gpurun_out/r5b/ws_scaling.log
timeout 600 python bench.py --mode fwd2d > gpurun_out/r5b/fwd2d.json 2> gpurun_out/r5b/fwd2d.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5b/fwd2d.json').read().strip().split('\n')[-1])
print('fwd2d', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['fwd2d'].items() if k in('executor','chain')})
P
