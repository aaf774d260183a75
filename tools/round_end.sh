#!/bin/bash
# The measurements a round's profiles/ directory holds, in one go.
# Usage on the GPU box: bash tools/round_end.sh gpurun_out/r03_final
OUT=${1:-gpurun_out/round_end}
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"
cd "$ROOTD"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --mode c3 --steps 20 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
for cfg in c2 c4 c4toy c1; do
  python bench.py --mode train --config $cfg --steps 20 > "$OUT/bench_train_$cfg.json" 2> "$OUT/bench_train_$cfg.err"
done
python bench.py --mode train --config c2 --precision bf16x3 --steps 6 > "$OUT/bench_train_c2_bf16x3.json" 2> "$OUT/bench_train_c2_bf16x3.err"
C2="--gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4"
bash tools/train_prof.sh "$OUT/train_prof" $C2 --precision bf16 > "$OUT/train_prof.log" 2>&1
bash tools/train_prof.sh "$OUT/train_prof_x3" $C2 --precision bf16x3 --iters 2 > "$OUT/train_prof_x3.log" 2>&1
bash tools/pmc_train.sh "$OUT/pmc_train" $C2 --precision bf16 --iters 2 > "$OUT/pmc_train.log" 2>&1
bash tools/pmc_bench.sh "$OUT/pmc_bench" > "$OUT/pmc_bench.log" 2>&1
# raw counter / trace csv files are large: keep the summaries
find "$OUT" -name "*.csv" -size +2M -delete
ls -la "$OUT"
