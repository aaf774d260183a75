#!/bin/bash
# PMC counters of the training step kernels (own rocprofv3 passes, --pmc with --kernel-trace only)
# Usage on the GPU box: bash tools/pmc_train.sh gpurun_out/pmc_train [train_probe args]
OUT=${1:-gpurun_out/pmc_train}; shift
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOTD/tools/train_probe.py $@"
i=0
for grp in \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS" \
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOTD/$OUT/pass$i" -- $CMD > "$ROOTD/$OUT/pass$i.log" 2>&1
done
python - "$ROOTD/$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:60]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_summary.txt', 'w') as fo:
    for k, d in sorted(agg.items()):
        if not any(s in k for s in ('wgrad_bf16', 'dgrad_c2', 'dgrad_s2', 'wgrad_c2', 'wgrad_tail', 'fewch', 'conv3_mfma', 'gconv', 'halo32', 'halo_s2', 'conv_tail', 'gather_bwd', 'epilogue_bwd', 'fold16', 'dense_', 'loss_content', 'partial_reduce', 'adam')):
            continue
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            fo.write('    %-34s mean %14.1f  n %d\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/pmc_summary.txt').read())
PY
