#!/bin/bash
# Round-6 measurement set (GPU box): bash tools/round6_end.sh gpurun_out/r06e [skip_tests]
OUT=${1:-gpurun_out/r06e}
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd "$ROOTD"
if [ -z "$2" ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed" > "$OUT/gputests.txt"
fi
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
bash tools/pmc_bench.sh "$OUT/pmc_bench" > "$OUT/pmc_bench.log" 2>&1
bash tools/pmc_bench.sh "$OUT/pmc_fwd2d" --mode fwd2d --no-executor > "$OUT/pmc_fwd2d.log" 2>&1
# the new kernels: BF16X3 2-D trunk (+ ablations), the taps-as-columns output conv
( python tools/dbg/x3_2d_probe.py; for d in 2 4 8 32 41 57; do echo "MFMA_DBG=$d"; DBG=$d ORACLE=0 python tools/dbg/x3_2d_probe.py | grep "^ws_x3:"; done;
  echo "480 images"; ORACLE=0 python tools/dbg/x3_2d_probe.py 480 | grep "^ws_x3:\|^tile" ) > "$OUT/x3_2d.log" 2>&1
( for a in "48 150 150 2" "192 150 150 2" "12 750 750 2" "12 750 750 6" "4 150 150 6"; do python tools/dbg/out_conv_probe.py $a | grep -v amdgpu; done ) > "$OUT/out_conv.log" 2>&1
bash tools/pmc_cmd.sh "$OUT/pmc_out_conv" "conv2d_out_kernel" -- python tools/dbg/out_conv_probe.py 48 150 150 2 > "$OUT/pmc_out_conv.log" 2>&1
# the plane-sweep tail conv and its weight gradient
( python tools/dbg/tail_probe.py 32; python tools/dbg/tail_probe.py 8 ) 2>&1 | grep -v amdgpu > "$OUT/tail_probe.log"
bash tools/pmc_cmd.sh "$OUT/pmc_tail" "conv_tail_sweep" -- python tools/dbg/tail_probe.py 32 > "$OUT/pmc_tail.log" 2>&1
ORACLE=0 bash tools/pmc_cmd.sh "$OUT/pmc_x3" "conv2d_ws_x3" -- python tools/dbg/x3_2d_probe.py > "$OUT/pmc_x3.log" 2>&1
# executor timeline
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$ROOTD/$OUT/tl" -- python $ROOTD/tools/dbg/fwp2d_exec_probe.py 4 2 > "$ROOTD/$OUT/tl.log" 2>&1;
  python $ROOTD/tools/dbg/timeline_gaps.py "$ROOTD/$OUT/tl" 0.5 > "$ROOTD/$OUT/fwp2d_timeline.txt" 2>&1; rm -rf "$ROOTD/$OUT/tl" )
# training steps
C2="--gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4"
bash tools/train_prof.sh "$OUT/train_prof" $C2 --precision bf16 --iters 15 > "$OUT/train_prof.log" 2>&1
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/$OUT/seq" -- python $ROOTD/tools/train_probe.py $C2 --precision bf16 --iters 4 > "$ROOTD/$OUT/seq.log" 2>&1;
  python $ROOTD/tools/dbg/step_sequence.py "$ROOTD/$OUT/seq" 6 > "$ROOTD/$OUT/train_step_sequence.txt" 2>&1; rm -rf "$ROOTD/$OUT/seq" )
bash tools/dbg/kstats.sh 30 -- python $ROOTD/bench.py --mode train --config c4 --batch 4 --steps 60 > "$OUT/train_c4_kernel_stats.txt" 2>&1
bash tools/dbg/kstats.sh 24 -- python $ROOTD/bench.py --mode train --config c4toy --batch 4 --steps 100 >> "$OUT/train_c4_kernel_stats.txt" 2>&1
bash tools/dbg/kstats.sh 30 -- python $ROOTD/bench.py --mode train --config c5 --steps 60 > "$OUT/train_c5_kernel_stats.txt" 2>&1
bash tools/dbg/kstats.sh 24 -- python $ROOTD/bench.py --mode train --config c5small --steps 100 >> "$OUT/train_c5_kernel_stats.txt" 2>&1
bash tools/pmc_cmd.sh "$OUT/pmc_wgrad_tail" "conv_wgrad_tail_sweep" -- python tools/train_probe.py $C2 --precision bf16 --iters 4 > "$OUT/pmc_wgrad_tail.log" 2>&1
python tools/config_census.py --out "$OUT/config_census.md" > "$OUT/census.log" 2>&1
python tools/dbg/c3_ops.py 16 > "$OUT/c3_ops.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
