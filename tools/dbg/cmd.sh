cd /root/repo
for i in 1 2; do
for v in 0 1; do
if [ $v = 1 ]; then export SUP3R_AMD_NO_DIRECT_OUTPUT=1; else unset SUP3R_AMD_NO_DIRECT_OUTPUT; fi
python bench.py --mode infer --steps 30 --no-cpu-baseline --no-parity-mode --no-train --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('infer nodirect=$v', round(d['value'],1), round(d['ms_per_step'],3))"
python bench.py --mode c3 --batch 16 --steps 32 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3    nodirect=$v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
