#!/bin/bash
# kernels of one C1 _train_batch in launch order (GPU box): bash tools/dbg/c1_seq.sh
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gpurun_out/c1seq; cd /tmp; export TMPDIR=/tmp
C1="--gen gen_2x_2f.json --disc disc_s_same.json --lr-shape 15,5,5,2 --precision bf16 --iters 3"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c1seq/t -- python $R/tools/train_probe.py $C1 > $R/gpurun_out/c1seq/run.log 2>&1
python $R/tools/dbg/step_sequence.py $R/gpurun_out/c1seq/t > $R/gpurun_out/c1seq/c1.seq
find $R/gpurun_out/c1seq -name "*.csv" -delete
tail -2 $R/gpurun_out/c1seq/run.log
