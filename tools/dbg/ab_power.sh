#!/bin/bash
# same-box A/B of library builds on the headline incl. the all-zero-operand (unthrottled clock) leg
for rep in 1; do
for v in "$@"; do
  echo -n "$v  "
  SUP3R_AMD_LIB=$PWD/$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); p=d['roofline'].get('power',{})
print(round(d['value'],1), 'samples/s', round(d['roofline']['avg_launch_ms'],4), 'ms/launch', {k:(round(v['samples_per_s'],1), round(v['sclk_MHz']), round(v['power_W'])) for k,v in p.items() if isinstance(v,dict)})"
done; done
