cd /root/repo
python -m pytest tests/test_forward_pass_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
for v in 0 1; do
if [ $v = 1 ]; then export SUP3R_AMD_NO_TAIL_WINDOW=1; else unset SUP3R_AMD_NO_TAIL_WINDOW; fi
python bench.py --mode c3 --batch 16 --steps 32 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3    nowindow=$v', round(d['value'],1), round(d['ms_per_step'],3), d['in_situ']['all_ops_ms_per_batch'])"
done
