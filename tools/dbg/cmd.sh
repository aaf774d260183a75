cd /root/repo
python -m pytest tests/test_parity_r04.py tests/test_hip_parity.py tests/test_parity_r02.py -m gpu -x -q -k "persist or strip or repeat or full_size or c2_generator_forward or trunk" 2>&1 | grep -E "passed|failed|Error" | head
for i in 1 2; do
for v in 0 1; do
if [ $v = 1 ]; then export SUP3R_AMD_NO_RES_TOUCH=1; else unset SUP3R_AMD_NO_RES_TOUCH; fi
python bench.py --mode infer --steps 30 --no-cpu-baseline --no-parity-mode --no-train --no-traffic --dump-ops gpurun_out/ops_touch$v.txt 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('infer notouch=$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done; done
grep -E "^ (2[0-3]) " gpurun_out/ops_touch0.txt; grep -E "^ (2[0-3]) " gpurun_out/ops_touch1.txt
