#!/bin/bash
# persist-kernel change: parity tests that pin it + same-box A/B of the headline
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_parity_r02.py tests/test_hip_parity.py tests/test_poison_gpu.py -m gpu -x -q -n 1 2>&1 | tail -4 | tee gpurun_out/r5c/tests.log
timeout 600 bash tools/ab.sh tools/ab/base.so sup3r_amd/lib/libsup3r_hip.so 2>&1 | tee gpurun_out/r5c/ab.log
