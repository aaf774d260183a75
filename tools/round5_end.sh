#!/bin/bash
# Round-5 measurement set (GPU box): bash tools/round5_end.sh gpurun_out/r05e
OUT=${1:-gpurun_out/r05e}
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd "$ROOTD"
python -m pytest tests -m gpu -q --timeout 1200 2>&1 | grep "passed\|failed" > "$OUT/gputests.txt"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python tools/config_census.py --out "$OUT/config_census.md" > "$OUT/census.log" 2>&1
bash tools/pmc_bench.sh "$OUT/pmc_bench" > "$OUT/pmc_bench.log" 2>&1
bash tools/pmc_bench.sh "$OUT/pmc_fwd2d" --mode fwd2d > "$OUT/pmc_fwd2d.log" 2>&1
bash tools/census_prof.sh "$OUT/prof_2d" "spatial/gen_2x_2f" > /dev/null 2>&1
C2="--gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4"
bash tools/train_prof.sh "$OUT/train_prof" $C2 --precision bf16 --iters 15 > "$OUT/train_prof.log" 2>&1
bash tools/train_prof.sh "$OUT/train2d_prof" --gen sup3r/spatial/gen_2x_2f.json --disc sup3r/spatial/disc.json --lr-shape 16,75,75,2 --precision bf16 --iters 6 > "$OUT/train2d_prof.log" 2>&1
bash tools/prof_cmd.sh "$OUT/prof_chain" -- python tools/dbg/fwp_chain_probe.py 2 > /dev/null 2>&1
python tools/dbg/op_profile.py sup3r/sup3rcc/gen_wind_5x_1x_6f.json 96,150,150,7 > "$OUT/op_profile_wind5x_hires.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
