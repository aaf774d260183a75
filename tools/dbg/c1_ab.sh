#!/bin/bash
# C1 _train_batch, eager / recorded, with and without the weight-gradient side stream
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
C1="--gen gen_2x_2f.json --disc disc_s_same.json --lr-shape 15,5,5,2 --precision bf16 --iters 200"
for rep in 1 2; do
for side in 0 1; do
  if [ $side = 1 ]; then export SUP3R_AMD_WGRAD_SIDE_STREAM=1; else unset SUP3R_AMD_WGRAD_SIDE_STREAM; fi
  echo "side=$side eager:    $(python tools/train_probe.py $C1 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\).*/\1/')"
  echo "side=$side recorded: $(python tools/train_probe.py $C1 --capture 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\).*/\1/')"
done; done
