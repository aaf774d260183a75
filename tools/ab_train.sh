# in-call A/B of one environment switch on the C2 training step
# usage: bash tools/ab_train.sh SUP3R_AMD_NO_XYZ=1
for rep in 1 2; do
for v in "A=1" "$1"; do
  echo -n "[$v] "
  env $v python bench.py --mode train --config c2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['train']['ms_per_step'],3))"
done; done
