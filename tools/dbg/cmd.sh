cd /root/repo
python -m pytest tests -m gpu -x -q -k "tail" 2>&1 | grep -E "passed|failed"
python bench.py --mode train --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-420
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/tp -o t -- python /root/repo/bench.py --mode train --steps 6 --no-cpu-baseline > /dev/null 2>&1
grep -E "tail_slide|wgrad_tail" /root/repo/gpurun_out/tp/t_kernel_stats.csv | cut -c1-200
rm -rf /root/repo/gpurun_out/tp
