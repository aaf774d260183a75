#!/bin/bash
OUT=gpurun_out/r03_final4
mkdir -p $OUT
python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --mode train --config c2 --precision bf16x3 --steps 6 > $OUT/bench_train_c2_bf16x3.json 2>/dev/null
for cfg in c2 c4 c4toy c1; do python bench.py --mode train --config $cfg --steps 20 > $OUT/bench_train_$cfg.json 2>/dev/null; done
python bench.py --mode c3 --steps 20 > $OUT/bench_c3.json 2>/dev/null
C2="--gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4"
bash tools/train_prof.sh $OUT/train_prof $C2 --precision bf16 > $OUT/train_prof.log 2>&1
bash tools/train_prof.sh $OUT/train_prof_x3 $C2 --precision bf16x3 --iters 2 > $OUT/train_prof_x3.log 2>&1
bash tools/pmc_train.sh $OUT/pmc_train $C2 --precision bf16 --iters 2 > $OUT/pmc_train.log 2>&1
bash tools/pmc_bench.sh $OUT/pmc_bench > $OUT/pmc_bench.log 2>&1
find $OUT -name "*.csv" -size +2M -delete
head -5 $OUT/pmc_bench/kernel_stats_summary.txt
