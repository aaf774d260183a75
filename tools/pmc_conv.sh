#!/bin/bash
# rocprofv3 PMC passes for conv3_mfma_kernel (one pass per counter group; SQ has
# 8 slots, FETCH_SIZE / WRITE_SIZE need their own passes).  Run on the GPU box:
#   bash tools/pmc_conv.sh bf16 gpurun_out/pmc_bf16
PREC=${1:-bf16}; OUT=${2:-gpurun_out/pmc_$PREC}; ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOTD/$OUT/pass$i" -- \
    python "$ROOTD/tools/conv_probe.py" --precision $PREC --iters 5 > "$ROOTD/$OUT/pass$i.log" 2>&1
done
python - "$ROOTD/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-60:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            fo.write(f'   {c:34s} mean/dispatch {sum(v)/len(v):16.1f}  n={len(v)}\n')
print(open(out + '/summary.txt').read())
PY
