for sh in 16096040 20096040 16096020 8096080 10096080 12096040 20032040 40032020 16064040 24096040; do
  echo "shape $sh"; SUP3R_AMD_TAIL_SWEEP_SHAPE=$sh bash tools/dbg/kernel_time.sh "wgrad_tail_sweep" -- python $PWD/tools/train_probe.py --gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4 --precision bf16 --iters 6 2>&1 | grep "avg"
done
