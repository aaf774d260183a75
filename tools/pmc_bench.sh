#!/bin/bash
# rocprofv3 passes over the headline bench (python bench.py --steps 3):
#   pass 0: --kernel-trace --stats   (per-kernel time, no counters)
#   pass 1..: --pmc groups (own runs; FETCH_SIZE and WRITE_SIZE separately)
# Usage on the GPU box:  bash tools/pmc_bench.sh gpurun_out/pmc_bench [extra bench args]
OUT=${1:-gpurun_out/pmc_bench}; shift
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
# (no parity-mode / power / traffic legs: every dispatch is a forward of the timed workload on its random operands)
CMD="python $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-parity-mode --no-traffic $@"
# (the stats pass at the bench's own default step count: its per-kernel average is
# what roofline.avg_launch_ms of the bench line has to agree with)
STATCMD="python $ROOTD/bench.py --no-cpu-baseline --no-train --no-parity-mode --no-traffic $@"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/$OUT/stats" -- $STATCMD > "$ROOTD/$OUT/stats.log" 2>&1
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES" \
 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOTD/$OUT/pass$i" -- $CMD > "$ROOTD/$OUT/pass$i.log" 2>&1
done
python - "$ROOTD/$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:70]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_summary.txt', 'w') as fo:
    for k, d in sorted(agg.items()):
        if 'at::' in k or 'rocclr' in k: continue
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            fo.write(f'   {c:34s} mean/dispatch {sum(v)/len(v):18.1f}  n={len(v)}\n')
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + '/kernel_stats_summary.txt', 'w') as fo:
        for r in rows[:12]:
            fo.write('{:70s} calls={:5s} avg_ns={:12s} pct={}\n'.format(short(r['Name']), r['Calls'], r['AverageNs'], r['Percentage']))
print(open(out + '/kernel_stats_summary.txt').read())
print(open(out + '/pmc_summary.txt').read())
PY
