#!/bin/bash
# In-call A/B of library builds (GPU boxes differ by ~10 % in sustained clocks,
# so variants are only comparable inside one gpurun call).
# Usage: bash tools/ab.sh tools/ab/base.so sup3r_amd/lib/libsup3r_hip.so [...]
for rep in 1 2; do
for v in "$@"; do
  echo -n "$v  "
  SUP3R_AMD_LIB=$PWD/$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-train 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['avg_launch_ms'],4), 'ms/launch')"
done; done
