cd /root/repo
python -m pytest tests/test_parity_r04.py -m gpu -x -q -k "tail_conv" 2>&1 | grep -E "passed|failed|Error|assert" | head -20
python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train --no-traffic --dump-ops gpurun_out/r4ah_x3ops.txt 2>/dev/null | tail -1 > gpurun_out/r4ah_x3.json
cat gpurun_out/r4ah_x3.json | cut -c1-300
grep -n "tail\|small" gpurun_out/r4ah_x3ops.txt | head
tail -4 gpurun_out/r4ah_x3ops.txt
python bench.py --mode train --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
