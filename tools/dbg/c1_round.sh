#!/bin/bash
# C1 / C4-toy training measurements after the fewpos MFMA kernels (GPU box)
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
OUT=gpurun_out/c1_round; mkdir -p $OUT
for cfg in c1 c4toy; do
  python bench.py --mode train --config $cfg --steps 50 > $OUT/bench_train_$cfg.json 2> $OUT/bench_train_$cfg.err
done
C1="--gen gen_2x_2f.json --disc disc_s_same.json --lr-shape 15,5,5,2"
bash tools/train_prof.sh $OUT/c1_prof $C1 --precision bf16 --iters 20 > $OUT/c1_prof.log 2>&1
bash tools/dbg/c1_seq.sh > /dev/null 2>&1; cp gpurun_out/c1seq/c1.seq $OUT/c1_step_sequence.txt
bash tools/dbg/c1_ab.sh > $OUT/c1_ab.txt 2>&1
SUP3R_AMD_NO_FEWPOS_MFMA=1 python tools/train_probe.py $C1 --precision bf16 --iters 200 2>&1 | tail -1 > $OUT/c1_old_family.txt
find $OUT -name "*.csv" -size +2M -delete
tail -2 $OUT/bench_train_c1.json | cut -c1-400; cat $OUT/c1_ab.txt $OUT/c1_old_family.txt
